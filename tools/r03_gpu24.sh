#!/bin/bash
# round 3, call 24: the evidence again with AUTO's whole-round 128x64 rule (N = 4096 now runs that tile)
set -u
export TMPDIR=/tmp
rm -f gpurun_out/r03z/cold_start.txt
PARTS="tests sweeps bench offgrid prof" bash tools/r03_final.sh
