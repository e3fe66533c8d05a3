#!/bin/bash
# gpu_call.sh -- the single-purpose gpurun calls of a round, one script (round 3 had 25 copies of this):
#   gpurun --timeout 900 -- 'STEPS="k32tests sweep" bash tools/gpu_call.sh'
# Runs ON THE GPU BOX from the repo root; every step writes under gpurun_out/<TAG>/ (TAG defaults to r04).
set -u
TAG=${TAG:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=${STEPS:-"k32tests sweep"}   # k32tests tests hsweep sweep abl pmc exp1 exp2 ab3 dataset edge2 tl5 oo sktl
has() { [[ " $STEPS " == *" $1 "* ]]; }
if has k32tests; then   # the parity tests of the 32x32x2 LDS-DMA tiles only
  ( time timeout 600 python -m pytest tests -m gpu -x -q -k "${K32_FILTER:-mfma32 or dma5}" ) > $OUT/pytest_k32.log 2>&1
  tail -5 $OUT/pytest_k32.log
fi
if has tests; then
  ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1
  tail -5 $OUT/pytest_gpu.log
fi
if has hsweep; then     # the reference harness over the square sweep: auto (sustained), both vendor libraries, REF=skip (fast)
  H=how-to-optimize-gemm_amd/harness
  for kk in ${HSWEEP_KERNELS:-auto rocblas hipblaslt}; do
    ( cd $H && echo "version = 'MMult_hip_${kk}';" > ../../$OUT/hsweep_${kk}.m && env KERNEL=$kk REF=skip WARMUP_MS=50 TRIALS=3 timeout 600 ./test_MMult.x >> ../../$OUT/hsweep_${kk}.m ) 2> $OUT/hsweep_${kk}.err
  done
  paste <(awk 'NF==3 && $1+0>0{print $1, $2}' $OUT/hsweep_auto.m) <(awk 'NF==3 && $1+0>0{print $2}' $OUT/hsweep_rocblas.m) <(awk 'NF==3 && $1+0>0{print $2}' $OUT/hsweep_hipblaslt.m)
fi
if has sweep; then      # every tile family forced, plain and stream-K, over the reference sweep
  timeout 900 python tools/tile_sweep.py --check ${SWEEP_ARGS:-} --out $OUT/tile_sweep${SWEEP_TAG:-} > $OUT/tile_sweep${SWEEP_TAG:-}.log 2>&1
  tail -30 $OUT/tile_sweep${SWEEP_TAG:-}.log | cut -c1-400
fi
if has abl; then        # timing-only ablations of the K2M loop (tools build)
  for n in ${ABL_SIZES:-4096 1024}; do
    timeout 300 python tools/ab_bench.py --n $n ${ABL_VARIANTS:-mfma_128x64_dma mfma32_128x64_dma 52 53 54 55 mfma_64x64_dma mfma32_64x64_dma 56 57 58 59 mfma32_64x128_dma} > $OUT/abl_$n.txt 2>&1
    cat $OUT/abl_$n.txt | grep -v amdgpu.ids
  done
fi
if has pmc; then        # counters (each group its own run) for a list of kernels at one size
  for kk in ${PMC_KERNELS:-mfma_128x64_dma mfma32_128x64_dma mfma32_64x128_dma}; do
    TAG=$TAG/pmc_$kk KERNEL=$kk PASSES="${PMC_PASSES:-pmc1 pmc2}" BENCH_ARGS="${PMC_BENCH_ARGS:-}" bash tools/gpu_profile.sh > $OUT/pmc_$kk.log 2>&1
    python tools/summarize_profile.py $OUT/pmc_$kk "${PMC_MATCH:-sgemm_}" > $OUT/pmc_$kk.json 2>> $OUT/pmc_$kk.log
    python - $OUT/pmc_$kk.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for p, c in d.get("pmc_mean_per_dispatch", {}).items():
    print(sys.argv[1].split("/")[-1], p, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in c.items()})
PY
  done
  find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
fi
if has exp1; then       # round-4 batch 1: K2W loader count / ring depth / look-ahead, whole-round tiles, thin edge tiles, persistent launches
  V64="mfma_64x64_dma,mfma_64x64_dma5,exp5_64x64_b3l1d2,exp5_64x64_b3l2d3,exp5_64x64_b4l2d2,exp5_64x64_b5l2d3,exp5_64x64_b5l2d6,exp5_64x64_b5l1d2"
  V12864="mfma_128x64_dma,mfma_128x64_dma5,exp5_128x64_b3l1d3,exp5_128x64_b3l2d2,exp5_128x64_b4l1d6"
  V128="mfma_128x128_dma,mfma_128x128_dma5,exp5_128x128_b3l1d3,exp5_128x128_b3l2d2,exp5_128x128_b4l2d6"
  VNEW="mfma_96x96_dma5,exp5_96x96_b3l1d3,exp5_96x96_b4l1d2,mfma_160x96_dma5,mfma_160x160_dma5,exp5_160x160_b3l1d3"
  timeout 300 python tools/tile_sweep.py --ab --check --sizes 1024:1536:128 --variants "auto,$V64,$V12864,mfma_96x96_dma5,exp5_96x96_b3l1d3,exp5_96x96_b4l1d2,rocblas,hipblaslt" \
    --out $OUT/exp1_small > $OUT/exp1_small.log 2>&1; tail -6 $OUT/exp1_small.log | cut -c1-900
  timeout 300 python tools/tile_sweep.py --ab --check --shapes "1664,1664,1664;1792,1792,1792;1920,1920,1920;2048,2048,2048;2176,2176,2176;2304,2304,2304;2432,2432,2432;2560,2560,2560;2688,2688,2688" \
    --variants "auto,mfma_64x64_dma,mfma_64x64_dma5,$V12864,$V128,$VNEW,rocblas,hipblaslt" --out $OUT/exp1_mid > $OUT/exp1_mid.log 2>&1; tail -10 $OUT/exp1_mid.log | cut -c1-900
  timeout 300 python tools/tile_sweep.py --ab --check --shapes "3072,3072,3072;4096,4096,4096;6144,6144,6144" \
    --variants "auto,mfma_128x64_dma,mfma_128x64_dma/p1,mfma_128x64_dma5,mfma_128x64_dma5/p1,exp5_128x64_b3l1d3,mfma_64x64_dma,mfma_64x64_dma/p1,mfma_64x64_dma5,mfma_64x64_dma5/p1,mfma_96x96_dma5,mfma_160x160_dma5,mfma_256x256,rocblas,hipblaslt" \
    --out $OUT/exp1_big > $OUT/exp1_big.log 2>&1; tail -4 $OUT/exp1_big.log | cut -c1-900
  timeout 300 python tools/tile_sweep.py --ab --check --shapes "1025,1025,1025;1040,1040,1040;1281,1281,1281;1409,1409,1409;2049,2049,2049;1023,1023,1023;1100,1100,1100;1537,1537,1537;2561,2561,2561" \
    --variants "auto,mfma_64x64_dma,mfma_64x64_dma5,mfma_128x64_dma,mfma_128x64_dma5,mfma_128x128_dma5,mfma_96x96_dma5,mfma_160x160_dma5,rocblas,hipblaslt" --out $OUT/exp1_edge > $OUT/exp1_edge.log 2>&1; tail -10 $OUT/exp1_edge.log | cut -c1-700
  for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    d=$OUT/exp1_pmc_$(echo $grp | cut -d' ' -f1)
    ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OLDPWD/$d -o pmc -- python $OLDPWD/tools/pmc_launch.py --ab --n 4096 \
        --variants "mfma_128x64_dma,mfma_128x64_dma/p1,mfma_128x64_dma5,mfma_128x64_dma5/p1,mfma_64x64_dma,mfma_64x64_dma5/p1,mfma_256x256" ) > $d.log 2>&1
  done
  python tools/pmc_by_kernel.py $OUT/exp1_pmc_FETCH_SIZE $OUT/exp1_pmc_WRITE_SIZE > $OUT/exp1_pmc.json 2>> $OUT/exp1_pmc.err; cat $OUT/exp1_pmc.json | head -120
  find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
fi
if has exp2; then       # round-4 batch 2: four loaders, thin tiles last, raster group height / padded rows (fabric traffic), create time
  V64="mfma_64x64_dma,mfma_64x64_dma5,exp5_64x64_l1d2,exp5_64x64_l4d2,exp5_64x64_l2d3,exp5_64x64_l4d3"
  V12864="mfma_128x64_dma,mfma_128x64_dma5,exp5_128x64_l1d2,exp5_128x64_l4d2,exp5_128x64_l2d3,exp5_128x64_l4d3"
  V128="mfma_128x128_dma,mfma_128x128_dma5,exp5_128x128_l1d2,exp5_128x128_l4d2,exp5_128x128_l2d3,exp5_128x128_l4d3"
  V96="mfma_96x96_dma5,exp5_96x96_l2d2,exp5_96x96_l4d2,exp5_96x96_l2d3"
  timeout 300 python tools/tile_sweep.py --ab --check --sizes 1024:1536:128 --variants "auto,$V64,$V12864,$V96,rocblas,hipblaslt" \
    --out $OUT/exp2_small > $OUT/exp2_small.log 2>&1; tail -6 $OUT/exp2_small.log | cut -c1-900
  timeout 300 python tools/tile_sweep.py --ab --check --shapes "1664,1664,1664;1792,1792,1792;1920,1920,1920;2048,2048,2048;2176,2176,2176;2304,2304,2304;2432,2432,2432;2560,2560,2560;2688,2688,2688;3200,3200,3200;3584,3584,3584;4096,4096,4096" \
    --variants "auto,mfma_64x64_dma,mfma_64x64_dma5,$V12864,$V128,rocblas,hipblaslt" --out $OUT/exp2_mid > $OUT/exp2_mid.log 2>&1; tail -13 $OUT/exp2_mid.log | cut -c1-900
  timeout 300 python tools/tile_sweep.py --ab --check --shapes "1025,1025,1025;1040,1040,1040;1032,1032,1032;1281,1281,1281;1409,1409,1409;1537,1537,1537;2049,2049,2049;2561,2561,2561;4097,4097,4097;1023,1023,1023" \
    --variants "auto,mfma_64x64_dma,mfma_64x64_dma/sk0,mfma_64x64_dma5,mfma_64x64_dma5/sk0,mfma_128x64_dma5,mfma_128x64_dma5/sk0,mfma_128x128_dma5,mfma_96x96_dma5,rocblas,hipblaslt" --out $OUT/exp2_edge > $OUT/exp2_edge.log 2>&1; tail -11 $OUT/exp2_edge.log | cut -c1-700
  for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    d=$OUT/exp2_pmc_$(echo $grp | cut -d' ' -f1)
    ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OLDPWD/$d -o pmc -- python $OLDPWD/tools/pmc_launch.py --ab --n 4096 --warm 30 --reps 6 \
        --variants "mfma_128x64_dma5,mfma_128x64_dma5/g1,mfma_128x64_dma5/g2,mfma_128x64_dma5/g4,mfma_128x64_dma5/g16,mfma_128x64_dma5/g32" ) > $d.log 2>&1
    d=$OUT/exp2_pmcpad_$(echo $grp | cut -d' ' -f1)
    ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OLDPWD/$d -o pmc -- python $OLDPWD/tools/pmc_launch.py --ab --n 4096 --pad 32 --warm 30 --reps 6 \
        --variants "mfma_128x64_dma5,mfma_128x64_dma5/g4,mfma_128x64_dma5/g16" ) > $d.log 2>&1
  done
  python tools/pmc_by_kernel.py $OUT/exp2_pmc_FETCH_SIZE $OUT/exp2_pmc_WRITE_SIZE --group 36 > $OUT/exp2_pmc.json 2>> $OUT/exp2_pmc.err
  python tools/pmc_by_kernel.py $OUT/exp2_pmcpad_FETCH_SIZE $OUT/exp2_pmcpad_WRITE_SIZE --group 36 > $OUT/exp2_pmcpad.json 2>> $OUT/exp2_pmc.err
  python - $OUT/exp2_pmc.json $OUT/exp2_pmcpad.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    for k, v in json.load(open(f)).items():
        if v.get("_n_FETCH_SIZE", 0) >= 3:
            print(f.split("/")[-1], k[-60:], "us", v.get("_us_FETCH_SIZE"), "fetch MB", round(v.get("fetch_bytes", 0) / 1e6, 1), "l2hit", v.get("l2_hit"))
PY
  find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
  timeout 200 python tools/create_time.py --runs 2 > $OUT/create_time.json 2>&1; cat $OUT/create_time.json | tr -d '\n' | cut -c1-600; echo
fi
if has dataset; then    # every (tile family, launch form) forced over the fit / held-out shape lists: what MMH_KERNEL_AUTO's table is fitted on
  # (round 5: the K2L tiles forced both ways too -- candidates of the table since this round -- and the 96x64 tile)
  DV="auto,mfma_64x64_dma5/sk0,mfma_64x64_dma5/sk2,mfma_128x64_dma5/sk0,mfma_128x64_dma5/sk2,mfma_128x128_dma5/sk0,mfma_128x128_dma5/sk2,mfma_96x96_dma5,mfma_96x64_dma5,mfma_160x160_dma5,mfma_256x256/sk0,mfma_256x256/sk2,mfma_64x64_dma/sk0,mfma_64x64_dma/sk2,mfma_128x64_dma/sk0,mfma_128x64_dma/sk2,mfma_128x128_dma/sk0,mfma_128x128_dma/sk2"
  for which in ${DATASETS:-fit heldout}; do
    timeout 900 python tools/tile_sweep.py --shape-file tools/policy_shapes_$which.txt --variants "$DV" --rounds 2 --reps 10 --warm-ms 10 \
      --out $OUT/dataset_$which > $OUT/dataset_$which.log 2>&1
    tail -2 $OUT/dataset_$which.log | cut -c1-300
  done
fi
if has ab3; then        # deferred publish and raster group height, A/B in one process (tools build)
  timeout 300 python tools/tile_sweep.py --ab --check --rounds 5 --shapes "2176,2176,2176;2304,2304,2304;2432,2432,2432;2560,2560,2560;2816,2816,2816;3200,3200,3200;3584,3584,3584;3968,3968,3968" \
    --variants "mfma_128x128_dma5/sk2,mfma_128x128_dma5/sk2/nd,mfma_128x128_dma5/sk2/g8,mfma_128x128_dma5/sk2/nd/g8,mfma_128x64_dma5/sk2,mfma_128x64_dma5/sk2/g8,mfma_128x64_dma5/sk2/nd/g8" \
    --out $OUT/ab3_mid > $OUT/ab3_mid.log 2>&1; tail -9 $OUT/ab3_mid.log | cut -c1-500
  timeout 300 python tools/tile_sweep.py --ab --check --rounds 5 --shapes "1152,1152,1152;1280,1280,1280;1664,1664,1664;1792,1792,1792;1920,1920,1920;2048,2048,2048;4096,4096,4096;3072,3072,3072" \
    --variants "mfma_64x64_dma5/sk2,mfma_64x64_dma5/sk2/nd,mfma_128x64_dma5/sk2,mfma_128x64_dma5/sk2/nd,mfma_128x64_dma5/sk2/g8,mfma_128x64_dma5/sk0,mfma_128x64_dma5/sk0/g8,mfma_128x128_dma5/sk0,mfma_128x128_dma5/sk0/g8,mfma_64x64_dma5/sk0" \
    --out $OUT/ab3_small > $OUT/ab3_small.log 2>&1; tail -9 $OUT/ab3_small.log | cut -c1-600
fi
if has edge2; then      # thin tiles: plain against persistent launches of the K2W tiles one element past a tile boundary
  timeout 300 python tools/tile_sweep.py --check --shapes "${EDGE_SHAPES:-1025,1025,1025;1040,1040,1040;1281,1281,1281;1409,1409,1409;1537,1537,1537;2049,2049,2049;2561,2561,2561;1024,1024,1024;1023,1023,1023;1100,1100,1100}" \
    --variants "auto,mfma_64x64_dma,mfma_64x64_dma/sk0,mfma_64x64_dma5,mfma_64x64_dma5/sk0,mfma_128x64_dma5,mfma_128x64_dma5/sk0,mfma_96x96_dma5,rocblas,hipblaslt" --out $OUT/edge2 > $OUT/edge2.log 2>&1; tail -11 $OUT/edge2.log | cut -c1-600
fi
if has tl5; then        # per-workgroup timeline of plain K2W launches (thin tiles last)
  timeout 600 python tools/dma5_timeline.py --kernel ${TL_KERNEL:-mfma_64x64_dma5} --shape ${TL_SHAPES:-1025,1025,1025 1024,1024,1024 1040,1040,1040} > $OUT/tl5_${TL_KERNEL:-mfma_64x64_dma5}.txt 2>&1
  cat $OUT/tl5_${TL_KERNEL:-mfma_64x64_dma5}.txt | grep -v amdgpu.ids
fi
if has oo; then         # prepared at the end of round 4, never run: whole-tile K2W stream-K launches bounded by their OWN instantiation's
                        # residency (three 64x64 / two 128x64 workgroups per CU instead of two / one) -- tools build, option 103
  timeout 500 python tools/tile_sweep.py --ab --check --rounds 3 --sizes ${OO_SIZES:-1152:4096:128} \
    --variants "auto,mfma_128x64_dma5/sk2,mfma_128x64_dma5/sk2/oo,mfma_64x64_dma5/sk2,mfma_64x64_dma5/sk2/oo,mfma_128x128_dma5/sk2,hipblaslt" \
    --out $OUT/own_occ > $OUT/own_occ.log 2>&1; grep "^{" $OUT/own_occ.log | cut -c1-330
fi
if has l2sk; then       # round 5: L2 behaviour of the chained K2W stream-K launches (phase order on / off, raster group height), per size
  LV=${L2SK_VARIANTS:-"mfma_128x128_dma5/sk2,mfma_128x128_dma5/sk2/no,mfma_128x128_dma5/sk2/g1,mfma_128x128_dma5/sk2/g2,mfma_128x128_dma5/sk2/g8,mfma_128x128_dma5/sk2/g16"}
  NV=$(echo $LV | tr ',' '\n' | wc -l)
  for n in ${L2SK_SIZES:-2560 2304}; do
    for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
      d=$OUT/l2sk_${n}_$(echo $grp | cut -d' ' -f1)
      ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OLDPWD/$d -o pmc -- python $OLDPWD/tools/pmc_launch.py --ab --n $n --warm 30 --reps 6 \
          --variants "$LV" ) > $d.log 2>&1
    done
    python tools/pmc_by_kernel.py $OUT/l2sk_${n}_FETCH_SIZE $OUT/l2sk_${n}_WRITE_SIZE --group 36 > $OUT/l2sk_$n.json 2>> $OUT/l2sk.err
    grep -- "->" $OUT/l2sk_${n}_FETCH_SIZE.log | cut -c1-260 > $OUT/l2sk_${n}_launched.txt
    python - $OUT/l2sk_$n.json <<'PY'
import json, sys
for k, v in json.load(open(sys.argv[1])).items():
    if v.get("_n_FETCH_SIZE", 0) >= 3:
        print(sys.argv[1].split("/")[-1], k[-70:], "us", v.get("_us_FETCH_SIZE"), "fetch MB", round(v.get("fetch_bytes", 0) / 1e6, 1), "write MB", round(v.get("write_bytes", 0) / 1e6, 1), "l2hit", v.get("l2_hit"))
PY
  done
  find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
fi
if has om; then         # round 5: phase-ordered stream-K tables below 1.8 tiles per workgroup (option 104), and the 96x64 tile (tools build)
  timeout 500 python tools/tile_sweep.py --ab --check --rounds 3 --sizes ${OM_SIZES:-1152:3072:128} \
    --variants "auto,mfma_128x128_dma5/sk2,mfma_128x128_dma5/sk2/om10,mfma_128x64_dma5/sk2,mfma_128x64_dma5/sk2/om10,mfma_64x64_dma5/sk2,mfma_64x64_dma5/sk2/om10,exp5_96x64_l4/sk0,exp5_96x64_l2/sk0,exp5_96x64_l4/sk2,exp5_64x96_l2,hipblaslt" \
    --out $OUT/om > $OUT/om.log 2>&1; grep "^{" $OUT/om.log | cut -c1-420
fi
if has k1w; then        # round 5: K1W (the vector-ALU rung with loader waves) against K1, variants of ring depth / loaders / A read width / tile
  timeout 600 python tools/tile_sweep.py --ab --check --rounds 3 --sizes ${K1W_SIZES:-1024:4096:256} \
    --variants "valu/k1old,valu,valu_128x128/k1old,valu_128x128,valu_64x64/k1old,valu_64x64,k1w_128x128_b3l2a4,k1w_128x128_b3l4a4,k1w_128x128_b2l4a2,k1w_64x64_a2,k1w_128x64,k1w_64x128" \
    --out $OUT/k1w > $OUT/k1w.log 2>&1; grep "^{" $OUT/k1w.log | cut -c1-520
fi
if has sktl; then       # per-workgroup, per-part timeline of stream-K K2W launches
  for kern in ${SKTL_KERNELS:-mfma_128x128_dma5 mfma_64x64_dma5}; do
    timeout 600 python tools/sk_timeline.py --kernel $kern --shape ${SKTL_SHAPES:-2304,2304,2304 1152,1152,1152} > $OUT/sktl_$kern.txt 2>&1
    grep -v amdgpu.ids $OUT/sktl_$kern.txt
  done
fi
du -sh $OUT
