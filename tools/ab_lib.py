#!/usr/bin/env python3
"""ab_lib.py -- the current product library against ANOTHER BUILD of it (an earlier libmmult_hip.so kept beside it), same
process, same box, interleaved bursts through the C ABI both know: what a change to kernel code that cannot be switched
at run time (entry code, instruction order) is worth.  Bits of the two builds are compared too.

    cp how-to-optimize-gemm_amd/libmmult_hip.so how-to-optimize-gemm_amd/libmmult_hip_prev.so   # before the change
    ... edit, rebuild ...
    python tools/ab_lib.py --other how-to-optimize-gemm_amd/libmmult_hip_prev.so --kernels auto,mfma_64x64_dma5 --sizes 1024,1152,1280

Needs a GPU."""
import argparse
import ctypes as C
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402  (device memory only)
import how_to_optimize_gemm_amd as H  # noqa: E402

vp, fp = C.c_void_p, C.POINTER(C.c_float)


def load(path):
    L = C.CDLL(path)
    gemm = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int]
    L.mmh_create.argtypes = [C.POINTER(vp), C.c_int]
    L.mmh_set_kernel.argtypes = [vp, C.c_int]
    L.mmh_kernel_id.argtypes = [C.c_char_p]
    L.mmh_sgemm.argtypes = gemm + [C.c_int, vp]
    L.mmh_time_sgemm.argtypes = gemm + [C.c_int, C.c_int, vp, fp]
    L.mmh_last_launch.restype = C.c_char_p
    h = vp()
    assert L.mmh_create(C.byref(h), 0) == 0
    return L, h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--other", required=True)
    ap.add_argument("--kernels", default="auto")
    ap.add_argument("--sizes", default="1024,1152,1280,1408,1536,2048,4096")
    ap.add_argument("--shapes", default="", help="m,n,k;m,n,k;... instead of square sizes")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--warm-ms", type=float, default=20.0)
    args = ap.parse_args()
    libs = {"new": load(H.LIB_PATH), "old": load(os.path.abspath(args.other))}
    stream = torch.cuda.current_stream().cuda_stream
    for kern in args.kernels.split(","):
        for which, (L, h) in libs.items():
            kid = L.mmh_kernel_id(kern.encode())
            assert kid >= 0 and L.mmh_set_kernel(h, kid) == 0, (which, kern)
        shapes = ([tuple(int(x) for x in t.split(",")) for t in args.shapes.split(";") if t] if args.shapes
                  else [(int(x),) * 3 for x in args.sizes.split(",")])
        for (m, n, k) in shapes:
            a = torch.rand((m, k), device="cuda") * 2 - 1
            b = torch.rand((k, n), device="cuda") * 2 - 1
            c = {w: torch.empty((m, n), device="cuda") for w in libs}
            res = {w: [] for w in libs}
            launched, warm = {}, {}
            for w, (L, h) in libs.items():
                assert L.mmh_sgemm(h, m, n, k, a.data_ptr(), k, b.data_ptr(), n, c[w].data_ptr(), n, 0, stream) == 0
                launched[w] = L.mmh_last_launch().decode()
                ms = C.c_float(0)
                assert L.mmh_time_sgemm(h, m, n, k, a.data_ptr(), k, b.data_ptr(), n, c[w].data_ptr(), n, 3, 5, stream, C.byref(ms)) == 0
                warm[w] = max(3, int(args.warm_ms / max(ms.value, 1e-3)))
            torch.cuda.synchronize()
            same = bool(torch.equal(c["new"], c["old"]))
            for rnd in range(args.rounds):
                for w in (("old", "new") if rnd % 2 else ("new", "old")):
                    L, h = libs[w]
                    ms = C.c_float(0)
                    assert L.mmh_time_sgemm(h, m, n, k, a.data_ptr(), k, b.data_ptr(), n, c[w].data_ptr(), n, warm[w], args.reps, stream,
                                            C.byref(ms)) == 0
                    res[w].append(2.0 * m * n * k / (ms.value * 1e-3) / 1e12)
            med = {w: sorted(v)[len(v) // 2] for w, v in res.items()}
            print(json.dumps({"kernel": kern, "shape": [m, n, k], "old_tf": round(med["old"], 2), "new_tf": round(med["new"], 2),
                              "new_over_old": round(med["new"] / med["old"], 4), "bit_equal": same,
                              "launched": launched["new"][:60]}), flush=True)


if __name__ == "__main__":
    main()
