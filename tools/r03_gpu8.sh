#!/bin/bash
# round 3, call 8: the harness-vs-sweep protocol gap at N = 2176, the VALU rung after the wait-count fix
set -u
O=gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
H=how-to-optimize-gemm_amd/harness
timeout 300 python -m pytest tests -m gpu -q -k "fuzz or valu or VALU" > $O/pytest_some.txt 2>&1; tail -2 $O/pytest_some.txt
timeout 300 python tools/protocol_probe.py 2176,2304 > $O/protocol_probe.txt 2>&1; cat $O/protocol_probe.txt | cut -c1-330
for kk in auto mfma_128x64_dma mfma_128x128_dma; do
  ( cd $H && KERNEL=$kk REF=skip WARMUP_MS=50 TRIALS=3 PFIRST=2176 PLAST=2304 PINC=128 timeout 120 ./test_MMult.x ) 2>&1 | grep -E "^2[0-9]{3} " | sed "s/^/$kk /"
done | tee $O/harness_2176.txt
( cd $H && KERNEL=valu REF=skip WARMUP_MS=50 TRIALS=3 timeout 300 ./test_MMult.x ) > $O/output_MMult_hip_valu.m 2> $O/valu.err
grep -E "^(1024|1536|2048|3072|4096) " $O/output_MMult_hip_valu.m
TAG=r03g/prof_valu KERNEL=valu PASSES="trace pmc1 pmc2 pmc5" PMC_TIMEOUT=120 bash tools/gpu_profile.sh > $O/prof_valu.log 2>&1
python tools/summarize_profile.py $O/prof_valu "sgemm_valu_kernel" > $O/prof_valu_summary.json 2>> $O/prof_valu.log
cat $O/prof_valu_summary.json | head -80
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
