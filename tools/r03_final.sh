#!/bin/bash
# Round-3 evidence pass (run ON THE GPU BOX from the repo root via gpurun).  Everything lands under gpurun_out/r03z/;
# tools/r03_collect.py copies the summaries into profiles/.
#   PARTS="tests sweeps bench offgrid prof i8 shard"   (default: all)
set -u
OUT=gpurun_out/r03z
mkdir -p $OUT
H=how-to-optimize-gemm_amd/harness
export TMPDIR=/tmp
PARTS=${PARTS:-"tests sweeps bench offgrid prof i8 shard"}
has() { [[ " $PARTS " == *" $1 "* ]]; }
sweep() {   # name, extra env...
  local name=$1; shift
  ( cd $H && echo "version = 'MMult_hip_${name}';" > ../../$OUT/output_MMult_hip_${name}.m && \
    env "$@" timeout 900 ./test_MMult.x >> ../../$OUT/output_MMult_hip_${name}.m ) 2> $OUT/sweep_${name}.err
}
if has tests; then
  ( time timeout 1200 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1
  tail -3 $OUT/pytest_gpu.log
fi
if has sweeps; then
  # the reference's convention first (a cold process: 20 launches, no warm-up), then the sustained forms
  sweep auto_ref_convention KERNEL=auto REF=threads WARMUP=0
  sweep auto KERNEL=auto REF=threads WARMUP_MS=50 TRIALS=3 JSON=../../$OUT/sweep_auto_launches.json
  sweep rocblas KERNEL=rocblas REF=threads WARMUP_MS=50 TRIALS=3
  sweep hipblaslt KERNEL=hipblaslt REF=threads WARMUP_MS=50 TRIALS=3
  sweep valu KERNEL=valu REF=skip WARMUP_MS=50 TRIALS=3
  sweep mfma KERNEL=mfma REF=skip WARMUP_MS=50 TRIALS=3
  sweep auto_vs_blas KERNEL=auto REF=blas WARMUP_MS=50 TRIALS=3
  paste <(awk 'NF==3 && $1+0>0{print $1, $2, $3}' $OUT/output_MMult_hip_auto.m) <(awk 'NF==3 && $1+0>0{print $2}' $OUT/output_MMult_hip_rocblas.m) \
        <(awk 'NF==3 && $1+0>0{print $2}' $OUT/output_MMult_hip_hipblaslt.m) <(awk 'NF==3 && $1+0>0{print $2}' $OUT/output_MMult_hip_auto_ref_convention.m)
fi
if has bench; then
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --ramp-csv $OUT/clock_ramp.csv > $OUT/bench.json 2> $OUT/bench.err
  tail -c 300 $OUT/bench.json; echo
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --ramp 0 --no-extras --no-cpu-baseline > $OUT/bench_noramp.json 2> $OUT/bench_noramp.err
  timeout 300 python bench.py --gpus 1 --force-shard --n 8192 --steps 5 --warmup 2 --sweep > $OUT/bench_forceshard.json 2> $OUT/bench_forceshard.err
  timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_gpus2.out 2> $OUT/bench_gpus2.err; echo "bench --gpus 2 rc=$?" >> $OUT/bench_gpus2.err
  for i in 1 2; do
    for kk in auto mfma_64x64_dma mfma_256x256; do
      timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --kernel $kk 2> /dev/null | \
        python -c "import json,sys; d=json.load(sys.stdin); c=d['cold']; print('$kk', 'sustained', d['value'], 'launch1_ms', c['launch_1_ms'], 'first20', c['reference_convention_20_launches_no_warmup_tflops'], 'launches2to21', c['launches_2_to_21_tflops'], 'within1pct_after', c['launches_until_within_1pct_of_sustained'])" >> $OUT/cold_start.txt
      sleep 2
    done
  done
  cat $OUT/cold_start.txt
fi
if has offgrid; then
  timeout 1200 python tools/offgrid_sweep.py --set all --out $OUT/offgrid > $OUT/offgrid.log 2>&1
  tail -2 $OUT/offgrid.log | cut -c1-200
fi
if has shard; then
  timeout 300 python tools/shard_dryrun.py > $OUT/shard_dryrun.md 2> $OUT/shard_dryrun.err
  ( cd $H && MMH_SHARD_SHARE_DEVICE=1 FLAVOUR=sharded NGPUS=1 KERNEL=auto REF=skip PFIRST=4096 PLAST=16384 PINC=12288 NREPEATS=3 EXTENDED=1 timeout 600 ./test_MMult.x ) > $OUT/harness_sharded_1gpu.txt 2>&1
  cat $OUT/shard_dryrun.md | head -8 | cut -c1-160
fi
if has prof; then
  TAG=r03z/prof4096 KERNEL=auto bash tools/gpu_profile.sh > $OUT/prof4096.log 2>&1
  TAG=r03z/prof4096_64 KERNEL=mfma_64x64_dma PASSES="trace pmc1 pmc3 pmc4" bash tools/gpu_profile.sh > $OUT/prof4096_64.log 2>&1
  TAG=r03z/prof4096_mfma128 KERNEL=mfma bash tools/gpu_profile.sh > $OUT/prof4096_mfma128.log 2>&1
  TAG=r03z/prof4096_256 KERNEL=mfma_256x256 PASSES="trace pmc1 pmc3 pmc4" bash tools/gpu_profile.sh > $OUT/prof4096_256.log 2>&1
  TAG=r03z/prof3584 KERNEL=auto BENCH_ARGS="--n 3584" bash tools/gpu_profile.sh > $OUT/prof3584.log 2>&1
  TAG=r03z/prof1023 KERNEL=auto BENCH_ARGS="--n 1023" PASSES="trace pmc1 pmc3 pmc4" bash tools/gpu_profile.sh > $OUT/prof1023.log 2>&1
  python tools/summarize_profile.py $OUT/prof4096 "sgemm_mfma_dma_kernel" > $OUT/prof4096_summary.json 2>> $OUT/prof4096.log
  python tools/summarize_profile.py $OUT/prof4096_64 "sgemm_mfma_dma_kernel" > $OUT/prof4096_64_summary.json 2>> $OUT/prof4096_64.log
  python tools/summarize_profile.py $OUT/prof4096_mfma128 "sgemm_mfma_kernel" > $OUT/prof4096_mfma128_summary.json 2>> $OUT/prof4096_mfma128.log
  python tools/summarize_profile.py $OUT/prof4096_256 "sgemm_mfma_kernel" > $OUT/prof4096_256_summary.json 2>> $OUT/prof4096_256.log
  python tools/summarize_profile.py $OUT/prof3584 "sgemm_dma_streamk_kernel" > $OUT/prof3584_summary.json 2>> $OUT/prof3584.log
  python tools/summarize_profile.py $OUT/prof1023 "sgemm_mfma_dma_kernel" > $OUT/prof1023_summary.json 2>> $OUT/prof1023.log
  cp $OUT/prof4096/trace/*kernel_stats.csv $OUT/prof4096_kernel_stats.csv 2>/dev/null
  head -c 900 $OUT/prof4096_summary.json; echo
  ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$OUT/vendor_trace -o trace -- python $OLDPWD/tools/vendor_trace.py > $OLDPWD/$OUT/vendor_trace.log 2>&1 )
  python - $OUT > $OUT/vendor_kernels.md <<'PY'
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
f = glob.glob(os.path.join(sys.argv[1], "vendor_trace", "**", "*kernel_trace.csv"), recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"])) if f else []
print("| dispatch order | kernel | grid | workgroup | us |")
print("|---|---|---|---|---|")
last = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if any(s in name for s in ("elementwise", "fill", "distribution", "uniform")):
        continue
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")))
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if key != last:
        print(f"| {r['Dispatch_Id']} | `{name[:110]}` | {key[1]} | {r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))} | {us:.1f} |")
    last = key
PY
  head -20 $OUT/vendor_kernels.md | cut -c1-200
fi
if has i8; then
  timeout 400 python tools/i8_ksweep.py 6,8 > $OUT/i8_ksweep.txt 2>&1
  timeout 200 python tools/i8_ab.py 0,6,8 > $OUT/i8_ab.txt 2>&1
  I8_MODES=6,8 TAG=r03z/i8prof bash tools/i8_profile.sh > $OUT/i8prof.log 2>&1
  TAG=r03z/qprof bash tools/q_profile.sh > $OUT/qprof.log 2>&1
  cat $OUT/i8_ksweep.txt | cut -c1-260; tail -5 $OUT/i8prof.log
fi
# keep what is merged back small: drop the raw per-dispatch CSVs, keep logs + summaries
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
