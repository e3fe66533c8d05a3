#!/bin/bash
# round 3, call 9: N = 2176 reads 120 TF inside the full harness sweep and 137 on its own -- what differs?
set -u
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
H=how-to-optimize-gemm_amd/harness
run() { local tag=$1; shift; ( cd $H && env "$@" WARMUP_MS=50 TRIALS=3 timeout 300 ./test_MMult.x ) 2>&1 | grep -E "^(2048|2176|2304) " | sed "s/^/$tag /"; }
{
run "auto_full_skip" KERNEL=auto REF=skip
run "auto_full_threads" KERNEL=auto REF=threads
run "auto_from2048_skip" KERNEL=auto REF=skip PFIRST=2048 PLAST=2304
run "auto_from1920_skip" KERNEL=auto REF=skip PFIRST=1920 PLAST=2304
run "128x64_full_skip" KERNEL=mfma_128x64_dma REF=skip
run "128x128_full_skip" KERNEL=mfma_128x128_dma REF=skip
run "auto_full_skip_nopin" KERNEL=auto REF=skip MMH_NO_PIN=1
} | tee $O/harness_2176_context.txt
