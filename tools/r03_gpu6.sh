#!/bin/bash
set -u
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python tools/i8_ksweep.py 0,8,9 > $O/i8_ksweep.txt 2>&1
timeout 200 python tools/i8_ab.py 0,8,9 > $O/i8_ab.txt 2>&1
cat $O/i8_ksweep.txt; tail -n 4 $O/i8_ab.txt
