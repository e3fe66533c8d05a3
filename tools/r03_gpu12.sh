#!/bin/bash
# round 3, call 12: the rim -- 16-element units, interleaved dispatch, s_setprio; rim alone / tiles alone from the A/B library
set -u
O=gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "rim" > $O/pytest_rim.txt 2>&1; tail -3 $O/pytest_rim.txt
timeout 300 python tools/rim_ab.py --ab 1024,1408 1,2 > $O/rim_ab.md 2>&1; cat $O/rim_ab.md
