#!/bin/bash
# Round-6 evidence pass (run ON THE GPU BOX from the repo root via gpurun).  Everything lands under gpurun_out/r06z/;
# tools/r06_collect.py copies the summaries into profiles/.
#   PARTS="tests sweeps nonsquare bench offgrid prof shard misc i8 fuzz"   (default: all; regret and sktl: round 5's, on request)
set -u
OUT=gpurun_out/r06z
mkdir -p $OUT
H=how-to-optimize-gemm_amd/harness
export TMPDIR=/tmp
PARTS=${PARTS:-"tests sweeps nonsquare bench offgrid prof shard misc i8 fuzz"}
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
  sweep valu KERNEL=valu REF=threads WARMUP_MS=50 TRIALS=3 JSON=../../$OUT/sweep_valu_launches.json   # (round 6: a real diff column -- VERDICT r05 item 5)
  sweep mfma KERNEL=mfma REF=skip WARMUP_MS=50 TRIALS=3
  sweep auto_vs_blas KERNEL=auto REF=blas WARMUP_MS=50 TRIALS=3
  # a second sustained pass of the three columns, REF=skip (no host work between the sizes): the stability of the comparison
  sweep auto_refskip KERNEL=auto REF=skip WARMUP_MS=50 TRIALS=3
  sweep rocblas_refskip KERNEL=rocblas REF=skip WARMUP_MS=50 TRIALS=3
  sweep hipblaslt_refskip KERNEL=hipblaslt REF=skip WARMUP_MS=50 TRIALS=3
  paste <(awk 'NF==3 && $1+0>0{print $1, $2, $3}' $OUT/output_MMult_hip_auto.m) <(awk 'NF==3 && $1+0>0{print $2}' $OUT/output_MMult_hip_rocblas.m) \
        <(awk 'NF==3 && $1+0>0{print $2}' $OUT/output_MMult_hip_hipblaslt.m) <(awk 'NF==3 && $1+0>0{print $2}' $OUT/output_MMult_hip_auto_ref_convention.m) \
        <(awk 'NF==3 && $1+0>0{print $2}' $OUT/output_MMult_hip_auto_refskip.m) <(awk 'NF==3 && $1+0>0{print $2}' $OUT/output_MMult_hip_hipblaslt_refskip.m)
fi
if has nonsquare; then
  # the reference's M / N / K / LDA macros (armv7/parameters.h:15-17,38-46): twelve non-square shapes and three padded
  # leading dimensions, one harness process per library, rows numbered 1 .. 15 in the reference's result format
  SHAPES="8192,1024,4096 1024,8192,512 16384,128,4096 300,5000,7000 4096,4096,4100 2049,2049,2049 6000,3000,1000 1000,6000,3000 128,16384,4096 5000,5000,5000 3000,4000,8192 12288,512,2048"
  for kk in auto rocblas hipblaslt; do
    f=$OUT/output_MMult_hip_${kk}_nonsquare.m
    echo "version = 'MMult_hip_${kk}_nonsquare';" > $f
    echo "% rows: $SHAPES ; then 2048^3 with lda=ldb=ldc=2052, 3072^3 with 3073, 4096^3 with 4100 (M N K LDA LDB LDC of the harness)" >> $f
    echo "MY_MMult = [" >> $f
    i=0
    for s in $SHAPES; do
      i=$((i+1)); IFS=, read m n k <<< "$s"
      ( cd $H && env KERNEL=$kk REF=threads WARMUP_MS=50 TRIALS=3 M=$m N=$n K=$k PFIRST=$i PLAST=$i PINC=1 timeout 600 ./test_MMult.x 2>> ../../$OUT/nonsquare_${kk}.err | awk 'NF==3 && $1+0>0' ) >> $f
    done
    for s in 2048,2052 3072,3073 4096,4100; do
      i=$((i+1)); IFS=, read p ld <<< "$s"
      ( cd $H && env KERNEL=$kk REF=threads WARMUP_MS=50 TRIALS=3 M=$p N=$p K=$p LDA=$ld LDB=$ld LDC=$ld PFIRST=$i PLAST=$i PINC=1 timeout 600 ./test_MMult.x 2>> ../../$OUT/nonsquare_${kk}.err | awk 'NF==3 && $1+0>0' ) >> $f
    done
    echo "];" >> $f
  done
  paste <(awk 'NF==3 && $1+0>0{print $1, $2, $3}' $OUT/output_MMult_hip_auto_nonsquare.m) <(awk 'NF==3 && $1+0>0{print $2}' $OUT/output_MMult_hip_rocblas_nonsquare.m) \
        <(awk 'NF==3 && $1+0>0{print $2}' $OUT/output_MMult_hip_hipblaslt_nonsquare.m)
fi
if has bench; then
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --ramp-csv $OUT/clock_ramp.csv > $OUT/bench.json 2> $OUT/bench.err
  tail -c 300 $OUT/bench.json; echo
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --ramp 0 --no-extras --no-cpu-baseline > $OUT/bench_noramp.json 2> $OUT/bench_noramp.err
  timeout 300 python bench.py --gpus 1 --force-shard --n 8192 --steps 5 --warmup 2 --sweep --b-chunks 4 > $OUT/bench_forceshard.json 2> $OUT/bench_forceshard.err
  timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_gpus2.out 2> $OUT/bench_gpus2.err; echo "bench --gpus 2 rc=$?" >> $OUT/bench_gpus2.err
  for i in 1 2; do
    for kk in auto mfma_128x64_dma mfma_64x64_dma5 mfma_256x256; do
      timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --kernel $kk 2> /dev/null | \
        python -c "import json,sys; d=json.load(sys.stdin); c=d['cold']; print('$kk', 'sustained', d['value'], 'launch1_ms', c['launch_1_ms'], 'first20', c['reference_convention_20_launches_no_warmup_tflops'], 'launches2to21', c['launches_2_to_21_tflops'], 'within1pct_after', c['launches_until_within_1pct_of_sustained'])" >> $OUT/cold_start.txt
      sleep 2
    done
  done
  cat $OUT/cold_start.txt
fi
if has offgrid; then
  timeout 2600 python tools/offgrid_sweep.py --vendor-harness --set all --variants "auto,mfma_64x64_dma5,mfma_128x64_dma5,mfma_128x128_dma5,mfma_96x96_dma5,mfma_96x64_dma5,mfma_160x160_dma5,mfma_256x256,mfma_64x64_dma,mfma_128x64_dma,mfma_128x128_dma,rocblas,hipblaslt" \
    --out $OUT/offgrid > $OUT/offgrid.log 2>&1
  tail -2 $OUT/offgrid.log | cut -c1-200
fi
if has shard; then
  timeout 300 python tools/shard_dryrun.py > $OUT/shard_dryrun.md 2> $OUT/shard_dryrun.err
  ( cd $H && MMH_SHARD_SHARE_DEVICE=1 FLAVOUR=sharded NGPUS=1 KERNEL=auto REF=skip PFIRST=4096 PLAST=16384 PINC=12288 NREPEATS=3 EXTENDED=1 timeout 600 ./test_MMult.x ) > $OUT/harness_sharded_1gpu.txt 2>&1
  ( cd $H && MMH_SHARD_FORCE_RCCL=1 FLAVOUR=sharded NGPUS=1 KERNEL=auto REF=skip PFIRST=4096 PLAST=16384 PINC=12288 NREPEATS=3 EXTENDED=1 timeout 600 ./test_MMult.x ) > $OUT/harness_sharded_rccl1.txt 2>&1
  ( cd $H && MMH_SHARD_FORCE_RCCL=1 FLAVOUR=sharded NGPUS=1 KERNEL=auto REF=skip PFIRST=4096 PLAST=16384 PINC=12288 NREPEATS=3 EXTENDED=1 B_CHUNKS=8 timeout 600 ./test_MMult.x ) > $OUT/harness_sharded_rccl1_streamed.txt 2>&1
  cat $OUT/shard_dryrun.md | head -8 | cut -c1-160; tail -4 $OUT/harness_sharded_rccl1.txt | cut -c1-200
fi
if has prof; then
  TAG=r06z/prof4096 KERNEL=auto bash tools/gpu_profile.sh > $OUT/prof4096.log 2>&1
  TAG=r06z/prof2560 KERNEL=auto BENCH_ARGS="--n 2560" PASSES="trace pmc1 pmc3 pmc4" bash tools/gpu_profile.sh > $OUT/prof2560.log 2>&1
  TAG=r06z/prof1152 KERNEL=auto BENCH_ARGS="--n 1152" PASSES="trace pmc1 pmc3 pmc4" bash tools/gpu_profile.sh > $OUT/prof1152.log 2>&1
  TAG=r06z/prof1024 KERNEL=auto BENCH_ARGS="--n 1024" PASSES="trace pmc1 pmc2" bash tools/gpu_profile.sh > $OUT/prof1024.log 2>&1
  TAG=r06z/prof1536 KERNEL=auto BENCH_ARGS="--n 1536" PASSES="trace pmc1 pmc2" bash tools/gpu_profile.sh > $OUT/prof1536.log 2>&1
  TAG=r06z/prof_valu1024 KERNEL=valu BENCH_ARGS="--n 1024" PASSES="trace pmc1" bash tools/gpu_profile.sh > $OUT/prof_valu1024.log 2>&1
  python tools/summarize_profile.py $OUT/prof_valu1024 "sgemm_valu_dma5_kernel" > $OUT/prof_valu1024_summary.json 2>> $OUT/prof_valu1024.log
  TAG=r06z/prof_valu2048 KERNEL=valu BENCH_ARGS="--n 2048" PASSES="trace pmc1" bash tools/gpu_profile.sh > $OUT/prof_valu2048.log 2>&1
  python tools/summarize_profile.py $OUT/prof_valu2048 "sgemm_valu_dma5_kernel" > $OUT/prof_valu2048_summary.json 2>> $OUT/prof_valu2048.log
  TAG=r06z/prof_valu2176 KERNEL=valu BENCH_ARGS="--n 2176" PASSES="trace pmc1" bash tools/gpu_profile.sh > $OUT/prof_valu2176.log 2>&1
  python tools/summarize_profile.py $OUT/prof_valu2176 "sgemm_valu_dma5_streamk_kernel" > $OUT/prof_valu2176_summary.json 2>> $OUT/prof_valu2176.log
  python tools/summarize_profile.py $OUT/prof4096 "sgemm_mfma_dma5_kernel" > $OUT/prof4096_summary.json 2>> $OUT/prof4096.log
  python tools/summarize_profile.py $OUT/prof2560 "sgemm_mfma_dma5_kernel" > $OUT/prof2560_summary.json 2>> $OUT/prof2560.log   # (the 160x160 tile, one whole round, since this round)
  TAG=r06z/prof3584 KERNEL=auto BENCH_ARGS="--n 3584" PASSES="trace pmc1 pmc3 pmc4" bash tools/gpu_profile.sh > $OUT/prof3584.log 2>&1
  python tools/summarize_profile.py $OUT/prof3584 "sgemm_dma5_streamk_kernel" > $OUT/prof3584_summary.json 2>> $OUT/prof3584.log
  python tools/summarize_profile.py $OUT/prof1152 "sgemm_mfma_dma5_kernel" > $OUT/prof1152_summary.json 2>> $OUT/prof1152.log
  python tools/summarize_profile.py $OUT/prof1024 "sgemm_mfma_dma5_kernel" > $OUT/prof1024_summary.json 2>> $OUT/prof1024.log
  python tools/summarize_profile.py $OUT/prof1536 "sgemm_mfma_dma5_kernel" > $OUT/prof1536_summary.json 2>> $OUT/prof1536.log
  cp $OUT/prof4096/trace/*kernel_stats.csv $OUT/prof4096_kernel_stats.csv 2>/dev/null
  head -c 900 $OUT/prof4096_summary.json; echo
fi
if has i8; then   # configs[4]: launch forms and tiles, the tile-boundary timeline, counters (matrix-pipe busy), the named instruction
  timeout 400 python tools/i8_persist_ab.py 0,8,9,6,7 4096x4096x4096,8192x8192x8192,8192x8192x1024,8192x8192x2048,5120x5120x5120,6144x6144x6144,16384x16384x4096,3072x3072x3072,2048x2048x2048 2>&1 | grep -v amdgpu.ids > $OUT/i8_persist_ab.txt
  tail -3 $OUT/i8_persist_ab.txt | cut -c1-300
  timeout 200 python tools/i8_timeline.py 8192x8192x8192 4096x4096x4096 2>&1 | grep -v amdgpu.ids > $OUT/i8_timeline.txt
  TAG=r06z/i8prof I8_MODES=0,6 bash tools/i8_profile.sh > $OUT/i8prof.log 2>&1
  cp $OUT/i8prof/summary.json $OUT/i8prof_summary.json 2>/dev/null; tail -c 1200 $OUT/i8prof.log
  timeout 200 python tools/i8_instr_ab.py > $OUT/i8_instr_ab.md 2> $OUT/i8_instr_ab.err
fi
if has fuzz; then
  ( timeout 600 python tools/fuzz.py 1200 61 2>&1 | tail -4 ) > $OUT/fuzz.txt
  ( timeout 300 python tools/fuzz_i8.py 200 62 2>&1 | tail -2 ) >> $OUT/fuzz.txt
  ( MMH_I8_GRID_CAP=3 timeout 300 python tools/fuzz_i8.py 200 63 2>&1 | tail -2 ) >> $OUT/fuzz.txt
  cat $OUT/fuzz.txt
fi
if has misc; then
  timeout 200 python tools/create_time.py --runs 3 > $OUT/create_time.json 2>&1
  cat $OUT/create_time.json | tr -d '\n ' | cut -c1-400; echo
  timeout 100 python tools/valu_probe.py > $OUT/valu_probe.json 2> /dev/null; cat $OUT/valu_probe.json | tr -d '\n ' | cut -c1-400; echo
fi
if has regret; then   # the held-out shapes once more with the final table in the library: the `auto` column against the best forced candidate
  TAG=r06z STEPS="dataset" DATASETS="heldout" bash tools/gpu_call.sh > $OUT/regret_pass.log 2>&1
  python tools/policy_fit.py --fit profiles/r06_policy_dataset_fit.json --heldout $OUT/dataset_heldout.json 2>/dev/null | grep "^held-out" | cut -c1-600
fi
if has sktl; then     # where a persistent workgroup's time goes, part by part (timeline build)
  TAG=r06z STEPS="sktl" SKTL_SHAPES="2304,2304,2304 2303,2303,2303 1152,1152,1152 1280,1280,1280" bash tools/gpu_call.sh > $OUT/sktl.log 2>&1
  grep -c "part  kind" $OUT/sktl.log
fi
# keep what is merged back small: drop the raw per-dispatch CSVs, keep logs + summaries
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
