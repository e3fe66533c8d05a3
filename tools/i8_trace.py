import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import how_to_optimize_gemm_amd as H
mm = H.MMult(0)
for n in (4096, 8192):
    a = torch.randint(-127, 128, (n, n), device="cuda", dtype=torch.int8)
    b = torch.randint(-127, 128, (n, n), device="cuda", dtype=torch.int8)
    c = torch.empty((n, n), device="cuda", dtype=torch.int32)
    for mode in (0, 1):
        mm.set_igemm_mode(mode)
        for _ in range(30):
            mm.igemm_s8(a, b, out=c)
torch.cuda.synchronize()
