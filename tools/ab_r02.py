#!/usr/bin/env python3
"""ab_r02.py -- round-2 library vs the current one, same process, same box, interleaved: MMH_KERNEL_AUTO through the C
ABI both know (mmh_create / mmh_set_kernel / mmh_time_sgemm), on the reference sweep's stream-K sizes.  Needs
how-to-optimize-gemm_amd/libmmult_hip_r02.so (built from the round-2 tree by hand; not part of the product)."""
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
    L.mmh_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.mmh_time_sgemm.argtypes = gemm + [C.c_int, C.c_int, vp, fp]
    L.mmh_last_launch.restype = C.c_char_p
    h = vp()
    assert L.mmh_create(C.byref(h), 0) == 0
    return L, h


def main():
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1152, 1536, 2176, 2304, 2816, 2944, 3328, 3584, 3968, 4352]
    new_l, new_h = H.lib(), None
    mm = H.MMult(0, "auto")
    old_l, old_h = load(os.path.join(REPO, "how-to-optimize-gemm_amd", "libmmult_hip_r02.so"))
    old_l.mmh_set_kernel(old_h, 0)
    stream = torch.cuda.current_stream().cuda_stream
    rows = []
    for n in sizes:
        a = torch.rand((n, n), device="cuda") * 2 - 1
        b = torch.rand((n, n), device="cuda") * 2 - 1
        c = torch.empty((n, n), device="cuda")
        res = {"old": [], "new": []}
        launched = {}
        for rnd in range(4):
            for which in ("old", "new"):
                ms = C.c_float(0)
                if which == "old":
                    rc = old_l.mmh_time_sgemm(old_h, n, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, 40, 30, stream, C.byref(ms))
                    assert rc == 0
                    launched[which] = old_l.mmh_last_launch().decode()
                    v = ms.value
                else:
                    v = mm.time_sgemm(n, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, warmup=40, reps=30, stream=stream)
                    launched[which] = H.last_launch()
                if rnd:
                    res[which].append(2.0 * n ** 3 / (v * 1e-3) / 1e12)
        row = {"n": n, "old_tf": round(sorted(res["old"])[1], 1), "new_tf": round(sorted(res["new"])[1], 1),
               "delegations": mm.get_option(H.OPT_STREAMK_DELEGATIONS), "old": launched["old"][:70], "new": launched["new"][:90]}
        mm.set_option(H.OPT_STREAMK_DELEGATIONS, 0)
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
