#!/bin/bash
# Round-2 GPU pass 5: full parity suite, 256x128 DMA tile and int8 128x256 tile experiments.
set -u
OUT=gpurun_out/r02
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu5.log 2>&1
tail -4 $OUT/pytest_gpu5.log
timeout 600 python tools/smalln_sweep.py --rounds 3 --sizes 2048,2560,3072,3584,4096,5120,6144,8192 --variants auto,mfma_128x128_dma,mfma_256x128_dma,mfma_256x256,hipblaslt > $OUT/bign.md 2> $OUT/bign.err
cat $OUT/bign.md | grep -v "^<"
python - > $OUT/panel.txt 2>&1 <<'PY'
import torch, statistics
import how_to_optimize_gemm_amd as H
mm = H.MMult(0)
stream = torch.cuda.current_stream().cuda_stream
m, n, k = 2048, 16384, 16384
a = torch.rand((m, k), device='cuda') * 2 - 1
b = torch.rand((k, n), device='cuda') * 2 - 1
c = torch.empty((m, n), device='cuda')
for kern in ("auto", "mfma_256x128_dma", "mfma_128x128_dma", "auto", "mfma_256x128_dma"):
    mm.set_kernel(kern)
    ms = mm.time_sgemm(m, n, k, a.data_ptr(), k, b.data_ptr(), n, c.data_ptr(), n, warmup=5, reps=10, stream=stream)
    print(f"config-4 panel 2048x16384x16384 {kern}: {2.0 * m * n * k / (ms * 1e-3) / 1e12:.1f} TFLOP/s  ({H.last_launch()})")
PY
cat $OUT/panel.txt
I8_MODES=0,6,7,5 I8_SIZES=2048,4096,8192 timeout 300 python tools/misc_bench.py i8 > $OUT/i8_modes.txt 2> $OUT/i8_modes.err
grep "int8 N" $OUT/i8_modes.txt
