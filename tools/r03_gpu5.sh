#!/bin/bash
set -u
O=gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/i8_ld_probe.py 0,8 > $O/i8_ld_probe.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_round3.py -m gpu -q -k "cliff or shard" > $O/pytest_round3.txt 2>&1
timeout 300 python tools/shard_dryrun.py > $O/shard_dryrun.md 2> $O/shard_dryrun.err
cat $O/i8_ld_probe.txt; tail -n 3 $O/pytest_round3.txt; cat $O/shard_dryrun.md | cut -c1-200
