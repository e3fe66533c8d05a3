#!/usr/bin/env python3
"""protocol_probe.py -- why a size can read differently under the harness's timing protocol (bursts of 8 calls + a device
synchronise for WARMUP_MS, then 3 x 20 calls between events, operands from hipMalloc) and the sweep tools' (40 untimed
+ 20 timed calls back to back from C, operands from torch's allocator).  Prints TFLOP/s and stream-K delegation counts
for each (size, kernel, protocol, allocator)."""
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


def raw_copy_of(t):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), t.numel() * 4) == 0
    assert hip.hipMemcpy(p, C.c_void_p(t.data_ptr()), t.numel() * 4, 3) == 0
    return p.value


def main():
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2176, 2304]
    kernels = sys.argv[2].split(",") if len(sys.argv) > 2 else ["auto", "mfma_128x64_dma", "mfma_128x128_dma"]
    for n in sizes:
        a = torch.rand((n, n), device="cuda") * 2 - 1
        b = torch.rand((n, n), device="cuda") * 2 - 1
        c = torch.empty((n, n), device="cuda")
        ptrs = {"torch": (a.data_ptr(), b.data_ptr(), c.data_ptr())}
        ptrs["hipMalloc"] = (raw_copy_of(a), raw_copy_of(b), raw_copy_of(c))
        for kern in kernels:
            mm = H.MMult(0, kern)
            for alloc, (pa, pb, pc) in ptrs.items():
                tf = lambda ms: round(2.0 * n ** 3 / (ms * 1e-3) / 1e12, 1)
                mm.set_option(H.OPT_STREAMK_DELEGATIONS, 0)
                back = sorted(tf(mm.time_sgemm(n, n, n, pa, n, pb, n, pc, n, warmup=40, reps=20)) for _ in range(3))[1]
                d_back = mm.get_option(H.OPT_STREAMK_DELEGATIONS)
                res = {}
                for name, sync in (("bursts_with_sync", True), ("bursts_no_sync", False)):
                    mm.set_option(H.OPT_STREAMK_DELEGATIONS, 0)
                    t_end = time.time() + 0.05
                    while time.time() < t_end:
                        for _ in range(8):
                            mm.sgemm(n, n, n, pa, n, pb, n, pc, n)
                        if sync:
                            torch.cuda.synchronize()
                    trials = sorted(tf(mm.time_sgemm(n, n, n, pa, n, pb, n, pc, n, warmup=0, reps=20)) for _ in range(3))
                    res[name] = trials
                    res[name + "_delegations"] = mm.get_option(H.OPT_STREAMK_DELEGATIONS)
                print(json.dumps({"n": n, "kernel": kern, "alloc": alloc, "back_to_back_40+20": back, "delegations": d_back,
                                  **res, "launched": H.last_launch()[:60]}), flush=True)
            del mm
        for p in ptrs["hipMalloc"]:
            hip.hipFree(C.c_void_p(p))


if __name__ == "__main__":
    main()
