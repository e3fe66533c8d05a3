"""Launch loop for rocprofv3 (tools/i8_profile.sh): every int8 kernel configuration named in
I8_MODES (MMH_OPT_IGEMM_MODE values) at N = 4096 and 8192, I8_REPS launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import how_to_optimize_gemm_amd as H
mm = H.MMult(0)
modes = [int(x) for x in os.environ.get("I8_MODES", "3,4").split(",")]
reps = int(os.environ.get("I8_REPS", "30"))
for n in (4096, 8192):
    a = torch.randint(-127, 128, (n, n), device="cuda", dtype=torch.int8)
    b = torch.randint(-127, 128, (n, n), device="cuda", dtype=torch.int8)
    c = torch.empty((n, n), device="cuda", dtype=torch.int32)
    for mode in modes:
        mm.set_igemm_mode(mode)
        for _ in range(reps):
            mm.igemm_s8(a, b, out=c)
torch.cuda.synchronize()
