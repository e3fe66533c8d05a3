#!/bin/bash
# round 3, call 13: N = 2944 reads 132 TF inside the harness sweep and 145 in the off-grid tool -- what differs?
set -u
O=gpurun_out/r03k; mkdir -p $O
export TMPDIR=/tmp
H=how-to-optimize-gemm_amd/harness
run() { local tag=$1; shift; ( cd $H && env "$@" WARMUP_MS=50 TRIALS=3 timeout 300 ./test_MMult.x ) 2>&1 | grep -E "^(2816|2944|3072) " | sed "s/^/$tag /"; }
{
run "alone_skip" KERNEL=auto REF=skip PFIRST=2944 PLAST=2944
run "alone_threads" KERNEL=auto REF=threads PFIRST=2944 PLAST=2944
run "from2816_skip" KERNEL=auto REF=skip PFIRST=2816 PLAST=3072
run "full_skip" KERNEL=auto REF=skip
run "full_skip_nosk_order" KERNEL=auto REF=skip MMH_STREAMK_ORDER=0
run "128x64_alone" KERNEL=mfma_128x64_dma REF=skip PFIRST=2944 PLAST=2944
run "128x128_alone" KERNEL=mfma_128x128_dma REF=skip PFIRST=2944 PLAST=2944
run "64x64_alone" KERNEL=mfma_64x64_dma REF=skip PFIRST=2944 PLAST=2944
run "256_alone" KERNEL=mfma_256x256 REF=skip PFIRST=2944 PLAST=2944
} | tee $O/harness_2944_context.txt
python tools/rim_ab.py 2944 0 2>&1 | tail -2
