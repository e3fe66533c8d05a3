#!/bin/bash
# round 3, GPU call 1: semantics probe, the whole GPU suite, a quick off-grid sweep, the bench line
set -u
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
tools/probes/lds_dma_align_probe.x > $O/probe_align.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -x -k "round3" > $O/pytest_round3.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -k "not round3" > $O/pytest_rest.txt 2>&1
timeout 400 python tools/offgrid_sweep.py --quick --set all --out $O/offgrid > $O/offgrid.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
( cd how-to-optimize-gemm_amd/harness && REF=skip KERNEL=auto timeout 120 ./test_MMult.x > ../../$O/harness_auto_refconv.m 2>&1 )
( cd how-to-optimize-gemm_amd/harness && REF=skip KERNEL=hipblaslt PINC=512 timeout 120 ./test_MMult.x > ../../$O/harness_hipblaslt.m 2>&1 )
tail -3 $O/pytest_round3.txt $O/pytest_rest.txt; cat $O/probe_align.txt | tail -8
