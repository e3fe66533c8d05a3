#!/bin/bash
# Round-2 GPU pass 3: the LDS-DMA small-tile kernel (parity + speed), vendor kernel names, HBM patterns.
set -u
OUT=gpurun_out/r02
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dma or golden or seeded or accumulate or empty" ) > $OUT/pytest_dma.log 2>&1
tail -4 $OUT/pytest_dma.log
timeout 600 python tools/smalln_sweep.py --rounds 5 --variants auto,mfma_64x64,mfma_64x64_dma,mfma_64x64_dma4,mfma_128x64,mfma_128x64_dma,rocblas,hipblaslt > $OUT/smalln_dma.md 2> $OUT/smalln_dma.err
cat $OUT/smalln_dma.md
timeout 300 python tools/smalln_sweep.py --sizes 4096 --rounds 3 --variants mfma_64x64,mfma_64x64_dma,mfma_64x64_dma4,mfma_128x64,mfma_128x64_dma >> $OUT/smalln_dma.md 2>> $OUT/smalln_dma.err
tail -2 $OUT/smalln_dma.md
cat > /tmp/vendor_small.py <<'PY'
import torch, sys
import how_to_optimize_gemm_amd as H
mm = H.MMult(0)
torch.backends.cuda.matmul.allow_tf32 = False
for n in (1024, 1152, 1408, 1792, 2048):
    a = torch.rand((n, n), device='cuda'); b = torch.rand((n, n), device='cuda'); c = torch.empty((n, n), device='cuda')
    for _ in range(30):
        mm.matmul_rocblas(a, b, out=c)
    for _ in range(30):
        torch.mm(a, b, out=c)
torch.cuda.synchronize()
PY
REPO=$PWD
( cd /tmp && PYTHONPATH=$REPO timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/vendor_prof -o vendor -- python /tmp/vendor_small.py > /tmp/vendor_prof.log 2>&1 )
python - <<'PY' > gpurun_out/r02/vendor_kernels.txt 2>&1
import csv, glob, collections
f = glob.glob('/tmp/vendor_prof/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
print(list(rows[0].keys()))
agg = collections.OrderedDict()
for r in rows:
    key = (r['Kernel_Name'][:200], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')), r.get('LDS_Block_Size', ''), r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''))
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    v.sort()
    print(f"{v[len(v)//2]:9.1f} us x{len(v):3d}  grid={k[1]} wg={k[2]} lds={k[3]} vgpr={k[4]} agpr={k[5]}  {k[0]}")
PY
cat gpurun_out/r02/vendor_kernels.txt | cut -c1-330
hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_patterns.hip -o /tmp/hbm_patterns 2> /dev/null && /tmp/hbm_patterns > $OUT/hbm_patterns.txt 2>&1
cat $OUT/hbm_patterns.txt
