#!/bin/bash
# Round-2 GPU pass 4: DMA tiles under stream-K (parity + sweep), new HBM probes, chunk-pattern quant passes.
set -u
OUT=gpurun_out/r02
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu4.log 2>&1
tail -4 $OUT/pytest_gpu4.log
timeout 600 python tools/smalln_sweep.py --rounds 5 --variants auto,mfma_64x64_dma,mfma_128x64_dma,mfma_128x128_dma,rocblas,hipblaslt > $OUT/smalln_dma_sk.md 2> $OUT/smalln_dma_sk.err
cat $OUT/smalln_dma_sk.md | grep -v "^<"
timeout 600 python tools/smalln_sweep.py --rounds 3 --sizes 2048,2176,2304,2432,2560,2688,2816,2944,3072,3328,3584,3840,4096 --variants auto,mfma,mfma_128x64_dma,mfma_128x128_dma,mfma_256x256,rocblas,hipblaslt > $OUT/midn.md 2> $OUT/midn.err
cat $OUT/midn.md | grep -v "^<"
cat > /tmp/quant_trace.py <<'PY'
import torch
import how_to_optimize_gemm_amd as H
mm = H.MMult(0)
for n in (4096, 8192):
    x = torch.rand((n, n), device='cuda') * 2 - 1
    y = torch.rand((n, n), device='cuda') * 2 - 1
    o = torch.empty((n, n), device='cuda')
    for _ in range(20):
        mm.quantize_sym_s8(x)
    for _ in range(20):
        mm.qgemm(x, y, out=o)
torch.cuda.synchronize()
PY
REPO=$PWD
( cd /tmp && PYTHONPATH=$REPO timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qprof -o q -- python /tmp/quant_trace.py > /tmp/qprof.log 2>&1 )
python - <<'PY' > gpurun_out/r02/quant_kernels.txt 2>&1
import csv, glob, collections
f = glob.glob('/tmp/qprof/**/*kernel_trace.csv', recursive=True)
agg = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    key = (r['Kernel_Name'][:60], r['Grid_Size_X'])
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    v.sort()
    print(f"{v[len(v)//2]:9.1f} us x{len(v):3d}  grid={k[1]}  {k[0]}")
PY
cat gpurun_out/r02/quant_kernels.txt
timeout 300 python tools/misc_bench.py quant > $OUT/quant2.txt 2> $OUT/quant2.err
python - > $OUT/probes2.txt 2>&1 <<'PY'
import how_to_optimize_gemm_amd as H
mm = H.MMult(0)
for _ in range(3):
    print("hbm copy GB/s", round(mm.probe_hbm_copy(1 << 30), 1), "hbm read GB/s", round(mm.probe_hbm_read(1 << 30), 1))
PY
cat $OUT/probes2.txt $OUT/quant2.txt | tail -8
