#!/bin/bash
# round 3, call 14: the VALU rung with its fragments read one k-step ahead
set -u
O=gpurun_out/r03l; mkdir -p $O
export TMPDIR=/tmp
H=how-to-optimize-gemm_amd/harness
timeout 300 python -m pytest tests -m gpu -q -k "valu or VALU or ladder or kernels_agree or smoke" > $O/pytest_valu.txt 2>&1; tail -2 $O/pytest_valu.txt
for nb in 1 2; do
  ( cd $H && MMH_VALU_NBUF=$nb KERNEL=valu REF=skip WARMUP_MS=50 TRIALS=3 PINC=512 timeout 300 ./test_MMult.x ) > $O/output_MMult_hip_valu_nbuf$nb.m 2> $O/valu$nb.err
done
paste <(grep -E "^[0-9]+ " $O/output_MMult_hip_valu_nbuf1.m | awk '{print $1, $2}') <(grep -E "^[0-9]+ " $O/output_MMult_hip_valu_nbuf2.m | awk '{print $2}')
