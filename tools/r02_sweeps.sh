#!/bin/bash
# The reference sweep through the C++ harness in the reference's result format, sustained-clock form
# (WARMUP_MS=50 TRIALS=3) and reference convention (no warm-up), real REF diff column at every point.
set -u
OUT=gpurun_out/r02f
mkdir -p $OUT
H=how-to-optimize-gemm_amd/harness
sweep() {
  local name=$1; shift
  ( cd $H && echo "version = 'MMult_hip_${name}';" > ../../$OUT/output_MMult_hip_${name}.m && \
    env "$@" timeout 900 ./test_MMult.x >> ../../$OUT/output_MMult_hip_${name}.m ) 2> $OUT/sweep_${name}.err
}
sweep auto KERNEL=auto REF=threads WARMUP_MS=50 TRIALS=3
sweep auto_extended KERNEL=auto REF=threads WARMUP_MS=50 TRIALS=3 EXTENDED=1
sweep rocblas KERNEL=rocblas REF=threads WARMUP_MS=50 TRIALS=3
sweep valu KERNEL=valu REF=threads WARMUP_MS=50 TRIALS=3
sweep mfma KERNEL=mfma REF=threads WARMUP_MS=50 TRIALS=3
sweep auto_vs_blas KERNEL=auto REF=blas WARMUP_MS=50 TRIALS=3
paste <(awk 'NF==3 && $1+0>0{print $1, $2, $3}' $OUT/output_MMult_hip_auto.m) <(awk 'NF==3 && $1+0>0{print $2}' $OUT/output_MMult_hip_rocblas.m)
