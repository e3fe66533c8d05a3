"""Launch loop for rocprofv3 (tools/q_profile.sh): the quantised GEMM pipeline (abs-max, quantise,
int8 GEMM with the dequantisation in its epilogue) at N = 4096, Q_REPS calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import how_to_optimize_gemm_amd as H
mm = H.MMult(0)
n = int(os.environ.get("Q_N", "4096"))
x = torch.rand((n, n), device="cuda") * 2 - 1
y = torch.rand((n, n), device="cuda") * 2 - 1
o = torch.empty((n, n), device="cuda")
for _ in range(int(os.environ.get("Q_REPS", "40"))):
    mm.qgemm(x, y, out=o)
torch.cuda.synchronize()
