#!/bin/bash
# round 3, call 19: the harness sweeps again with the settle-detecting warm-up (REF=threads idles the GPU for seconds per size)
set -u
export TMPDIR=/tmp
PARTS="sweeps" bash tools/r03_final.sh
