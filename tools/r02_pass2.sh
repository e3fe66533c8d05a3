#!/bin/bash
# Round-2 GPU pass 2: diagnostics for the small-N regime (64x64 tile ablations, what the vendor library
# launches at those sizes) and an HBM probe sweep.
set -u
OUT=gpurun_out/r02
mkdir -p $OUT
export TMPDIR=/tmp
for n in 1024 4096; do
  echo "== ablations of the 64x64 configuration, N=$n (41 no gload, 42 +no ldswrite, 43 +no barrier, 44 mfma only)" >> $OUT/abl64.txt
  timeout 300 python tools/ab_bench.py --n $n --rounds 5 --reps 20 mfma_64x64 41 42 43 44 mfma_128x64 37 38 39 40 >> $OUT/abl64.txt 2>> $OUT/abl64.err
done
cat $OUT/abl64.txt
# which kernels do rocBLAS / hipBLASLt launch at small N, and how long do they take
cat > /tmp/vendor_small.py <<'PY'
import torch, sys
sys.path.insert(0, '.')
import how_to_optimize_gemm_amd as H
mm = H.MMult(0)
torch.backends.cuda.matmul.allow_tf32 = False
for n in (1024, 1152, 1408, 1792, 2048):
    a = torch.rand((n, n), device='cuda'); b = torch.rand((n, n), device='cuda'); c = torch.empty((n, n), device='cuda')
    for _ in range(30):
        mm.matmul_rocblas(a, b, out=c)
    for _ in range(30):
        torch.mm(a, b, out=c)
    mm.set_kernel('auto')
    for _ in range(30):
        mm.matmul(a, b, out=c)
torch.cuda.synchronize()
PY
REPO=$PWD
( cd /tmp && PYTHONPATH=$REPO rocprofv3 --kernel-trace --stats -d /tmp/vendor_prof -o vendor -- python /tmp/vendor_small.py > /tmp/vendor_prof.log 2>&1 )
tail -3 /tmp/vendor_prof.log
python - <<'PY' > gpurun_out/r02/vendor_kernels.txt 2>&1
import csv, glob, collections
f = glob.glob('/tmp/vendor_prof/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
print(list(rows[0].keys()))
agg = collections.OrderedDict()
for r in rows:
    key = (r['Kernel_Name'][:160], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')), r.get('LDS_Block_Size', ''), r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''))
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    v.sort()
    print(f"{v[len(v)//2]:9.1f} us x{len(v):3d}  grid={k[1]} wg={k[2]} lds={k[3]} vgpr={k[4]} agpr={k[5]}  {k[0]}")
PY
cat gpurun_out/r02/vendor_kernels.txt
python - > $OUT/hbm_sweep.txt 2>&1 <<'PY'
import ctypes, torch
import how_to_optimize_gemm_amd as H
mm = H.MMult(0)
print("probe copy/read GB/s:", [round(mm.probe_hbm_copy(1 << 30)) for _ in range(3)], [round(mm.probe_hbm_read(1 << 30)) for _ in range(3)])
# torch / runtime references on the same box
x = torch.empty(1 << 28, device='cuda'); y = torch.empty_like(x)
for name, fn, nbytes in (("torch copy_", lambda: y.copy_(x), 2 * x.numel() * 4), ("torch sum", lambda: x.sum(), x.numel() * 4),
                         ("torch fill_", lambda: y.fill_(1.0), x.numel() * 4)):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9:.0f} GB/s")
PY
cat $OUT/hbm_sweep.txt
