"""Launch loop for rocprofv3 --kernel-trace (tools/r03_final.sh): which kernels stand behind each comparator column.
At every size the three MY_MMult implementations -- ours (auto), mmh_sgemm_rocblas, mmh_sgemm_hipblaslt -- run three
launches each, separated by a marker launch of a distinct tiny shape so that the trace can be cut per (size, column)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import how_to_optimize_gemm_amd as H
mm = H.MMult(0, "auto")
s = torch.cuda.current_stream().cuda_stream
for n in (1024, 2048, 3072, 4096):
    a = torch.rand((n, n), device="cuda")
    b = torch.rand((n, n), device="cuda")
    c = torch.empty((n, n), device="cuda")
    mm.time_sgemm(n, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, warmup=0, reps=3, stream=s)
    mm.time_comparator("rocblas", n, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, warmup=0, reps=3, stream=s)
    mm.time_comparator("hipblaslt", n, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, warmup=0, reps=3, stream=s)
torch.cuda.synchronize()
