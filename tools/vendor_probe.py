import sys, os
sys.path.insert(0, os.getcwd())
import torch, how_to_optimize_gemm_amd as H
mm = H.MMult(0, "auto")
s = torch.cuda.current_stream().cuda_stream
for lib in ("hipblaslt", "rocblas"):
    for p in (2560, 3584, 4096, 1024):
        a = torch.rand((p, p), device="cuda") * 2 - 1; b = torch.rand((p, p), device="cuda") * 2 - 1; c = torch.empty((p, p), device="cuda")
        ms = mm.time_comparator(lib, p, p, p, a.data_ptr(), p, b.data_ptr(), p, c.data_ptr(), p, warmup=3, reps=10, stream=s)
        ms = min(mm.time_comparator(lib, p, p, p, a.data_ptr(), p, b.data_ptr(), p, c.data_ptr(), p, warmup=max(3, int(50 / ms)), reps=20, stream=s) for _ in range(3))
        ref = (a.double() @ b.double()).float()
        print(lib, p, round(2.0 * p ** 3 / ms / 1e9, 1), "TF  max err", float((c - ref).abs().max()))
for l in open("/proc/self/maps"):
    if "blaslt" in l or "rocblas" in l:
        print(l.split()[-1]); 
