#!/bin/bash
# round 3, GPU call 2: round-2 library vs now on the stream-K sizes, the rest of the round-3 tests, int8 A/B
set -u
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/ab_r02.py > $O/ab_r02.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -s --maxfail=20 > $O/pytest_round3.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stream_k or streamk or large_ragged or split_k or graph or dma" > $O/pytest_sk.txt 2>&1
timeout 300 python tools/i8_ab.py 0,6,7 > $O/i8_ab.txt 2>&1
( cd how-to-optimize-gemm_amd/harness && REF=skip KERNEL=auto timeout 120 ./test_MMult.x > ../../$O/harness_auto_refconv.m 2>&1 )
( cd how-to-optimize-gemm_amd/harness && REF=skip KERNEL=auto WARMUP_MS=50 TRIALS=3 timeout 200 ./test_MMult.x > ../../$O/harness_auto_sustained.m 2>&1 )
tail -n 3 $O/pytest_round3.txt $O/pytest_sk.txt; cat $O/ab_r02.txt | cut -c1-200; tail -n 4 $O/i8_ab.txt
