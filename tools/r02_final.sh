#!/bin/bash
# Round-2 evidence pass (run ON THE GPU BOX from the repo root via gpurun): the parity suite, the
# reference sweep in the reference's own result format (real REF diff column at all 25 points, reference
# timing convention AND sustained), the bench line + clock-ramp trace, the interleaved sweep against the
# vendor libraries, and the rocprofv3 passes (kernel trace + stats; PMC passes each in their own run).
# Everything lands under gpurun_out/r02f/; tools/r02_collect.py copies the summaries into profiles/.
set -u
OUT=gpurun_out/r02f
rm -rf $OUT; mkdir -p $OUT
H=how-to-optimize-gemm_amd/harness
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
sweep() {   # name, extra env...
  local name=$1; shift
  ( cd $H && echo "version = 'MMult_hip_${name}';" > ../../$OUT/output_MMult_hip_${name}.m && \
    env "$@" timeout 900 ./test_MMult.x >> ../../$OUT/output_MMult_hip_${name}.m ) 2> $OUT/sweep_${name}.err
}
sweep auto_ref_convention KERNEL=auto REF=threads WARMUP=0
sweep auto_splitk KERNEL=auto SPLITK=1 REF=threads WARMUP=30 PLAST=2048
tail -2 $OUT/output_MMult_hip_auto_ref_convention.m
# (the sustained-clock sweeps: tools/r02_sweeps.sh)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --ramp-csv $OUT/clock_ramp.csv > $OUT/bench.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.json; echo
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_gpus2.out 2> $OUT/bench_gpus2.err; echo "bench --gpus 2 rc=$?" >> $OUT/bench_gpus2.err
timeout 300 python bench.py --gpus 1 --force-shard --n 8192 --steps 5 --warmup 2 > $OUT/bench_forceshard.json 2> $OUT/bench_forceshard.err
timeout 900 python tools/smalln_sweep.py --rounds 3 --sizes $(seq -s, 1024 128 4096) --variants auto,rocblas,hipblaslt,valu > $OUT/sweep_vs_vendor.md 2> $OUT/sweep_vs_vendor.err
grep -v "^<" $OUT/sweep_vs_vendor.md | tail -26
for panels in 0 -1; do
  ( cd $H && MMULT_HOST_PANELS=$panels FLAVOUR=host KERNEL=auto REF=skip PFIRST=2048 PLAST=4096 PINC=2048 NREPEATS=5 WARMUP=1 \
    timeout 300 ./test_MMult.x | sed "s/^/panels=$panels /" ) >> $OUT/host_flavour.txt 2>> $OUT/host_flavour.err
done
grep -E "^panels=[-0-9]+ [0-9]" $OUT/host_flavour.txt
python - > $OUT/probes.txt 2>&1 <<'PY'
import how_to_optimize_gemm_amd as H
mm = H.MMult(0)
for _ in range(3):
    print("hbm copy GB/s", round(mm.probe_hbm_copy(1 << 30), 1), "hbm read GB/s", round(mm.probe_hbm_read(1 << 30), 1),
          "mfma f32 TF", round(mm.probe_mfma_f32(), 1))
PY
cat $OUT/probes.txt
hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_patterns.hip -o /tmp/hbm_patterns 2> /dev/null && /tmp/hbm_patterns > $OUT/hbm_patterns.txt 2>&1
# cold start of the two N=4096 candidates, fresh process each, alternating (auto = the 256x256 tile)
for i in 1 2 3; do
  for kk in auto mfma_64x64_dma; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --kernel $kk 2> /dev/null | \
      python -c "import json,sys; d=json.load(sys.stdin); c=d['cold']; print('$kk', 'sustained', d['value'], 'launch1_ms', c['launch_1_ms'], 'first20', c['reference_convention_20_launches_no_warmup_tflops'], 'launches2to21', c['launches_2_to_21_tflops'], 'within1pct_after', c['launches_until_within_1pct_of_sustained'])" >> $OUT/cold_start.txt
    sleep 2
  done
done
cat $OUT/cold_start.txt
# rocprofv3: the default bench command (N=4096, auto -> 256x256 tile), then the LDS-DMA tiles
# (N=3072: 64x64 tile, nine per CU; N=2048: 128x64; N=1024: 64x64, one per CU)
TAG=r02f/prof4096 KERNEL=auto bash tools/gpu_profile.sh > $OUT/prof4096.log 2>&1
TAG=r02f/prof3072 KERNEL=auto BENCH_ARGS="--n 3072" bash tools/gpu_profile.sh > $OUT/prof3072.log 2>&1
TAG=r02f/prof1024 KERNEL=auto BENCH_ARGS="--n 1024" bash tools/gpu_profile.sh > $OUT/prof1024.log 2>&1
TAG=r02f/prof2048 KERNEL=auto BENCH_ARGS="--n 2048" bash tools/gpu_profile.sh > $OUT/prof2048.log 2>&1
python tools/summarize_profile.py $OUT/prof4096 "sgemm_mfma_kernel" > $OUT/prof4096_summary.json 2>> $OUT/prof4096.log
python tools/summarize_profile.py $OUT/prof3072 "sgemm_mfma_dma_kernel" > $OUT/prof3072_summary.json 2>> $OUT/prof3072.log
python tools/summarize_profile.py $OUT/prof1024 "sgemm_mfma_dma_kernel" > $OUT/prof1024_summary.json 2>> $OUT/prof1024.log
python tools/summarize_profile.py $OUT/prof2048 "sgemm_mfma_dma_kernel" > $OUT/prof2048_summary.json 2>> $OUT/prof2048.log
cp $OUT/prof4096/trace/*kernel_stats.csv $OUT/prof4096_kernel_stats.csv 2>/dev/null
head -c 1200 $OUT/prof4096_summary.json; echo
TAG=r02f/qprof bash tools/q_profile.sh > $OUT/qprof.log 2>&1
I8_MODES=5,6 TAG=r02f/i8prof bash tools/i8_profile.sh > $OUT/i8prof.log 2>&1
tail -5 $OUT/i8prof.log
# keep what is merged back small: drop the raw per-dispatch CSVs, keep logs + summaries
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
