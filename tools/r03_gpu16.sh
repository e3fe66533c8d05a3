#!/bin/bash
# round 3, call 16: AUTO's 64x64 plain-launch rule for ragged shapes (relative padding); the rim's isolation table with the final code
set -u
O=gpurun_out/r03n; mkdir -p $O
export TMPDIR=/tmp
PARTS="tests offgrid" bash tools/r03_final.sh
timeout 300 python tools/rim_ab.py --ab 1024,1152,1408,1664,2048 1,2,4 > $O/rim_ab.md 2>&1; cat $O/rim_ab.md
