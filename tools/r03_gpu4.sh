#!/bin/bash
# round 3, GPU call 4: graph capture after the per-stream workspaces, the round-3 suite, int8 modes, policy re-check
set -u
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -s --maxfail=20 > $O/pytest_round3.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "graph or stream_k or split_k or large_ragged" > $O/pytest_sk.txt 2>&1
timeout 300 python tools/i8_ab.py 0,7,8 > $O/i8_ab.txt 2>&1
timeout 600 python tools/offgrid_sweep.py --set steps --variants auto,mfma_64x64_dma,rocblas,hipblaslt --out $O/offgrid_steps > $O/offgrid_steps.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -n 6 $O/pytest_round3.txt $O/pytest_sk.txt; tail -n 4 $O/i8_ab.txt
