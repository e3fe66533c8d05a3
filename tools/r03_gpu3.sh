#!/bin/bash
# round 3, GPU call 3: the round-3 tests and the stream-K / DMA subset after the per-stream workspaces and the
# bit-flag hand-over; r02 A/B again; the full off-grid sweep
set -u
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -s --maxfail=20 > $O/pytest_round3.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stream_k or streamk or large_ragged or split_k or graph or dma or fuzz" > $O/pytest_sk.txt 2>&1
timeout 300 python tools/ab_r02.py > $O/ab_r02.txt 2>&1
timeout 900 python tools/offgrid_sweep.py --set all --out $O/offgrid > $O/offgrid.log 2>&1
tail -n 5 $O/pytest_round3.txt $O/pytest_sk.txt; cat $O/ab_r02.txt | cut -c1-120
