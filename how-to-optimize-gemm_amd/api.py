"""Python host side over the C ABI of libmmult_hip.so (include/mmult_hip.h).

Mirrors the reference's operator interface for the hot path -- the names,
argument order and semantics of `MY_MMult(m, n, k, a, lda, b, ldb, c, ldc)`
(host flavour: armv7/test_MMult.c:8,76 / aarch64/test_MMult.cpp:17,113,
C += A*B on host buffers; device flavour: cuda/test_MMult.cpp:13-14,102,
C = A*B on device pointers, asynchronous) -- so parity tests read like the
reference's own harness.  PyTorch appears only as a source of device memory,
streams and torch.distributed; every FLOP is issued by the HIP library.

There is NO CPU fallback: if the library or a gfx950 device is missing, calls
raise MMultError.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libmmult_hip.so")
AB_LIB_PATH = os.path.join(PKG_DIR, "libmmult_hip_ab.so")   # tools-only build (see use_ab_library)

# status codes / kernel ids (include/mmult_hip.h)
OK, ERR_INVALID_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_ALLOC, ERR_COMM = 0, -1, -2, -3, -4, -5, -6
KERNEL_AUTO, KERNEL_VALU, KERNEL_MFMA, KERNEL_MFMA_256, KERNEL_NAIVE, KERNEL_MFMA_SIMPLE, KERNEL_MFMA_PIPE = 0, 1, 2, 3, 4, 5, 6
OPT_STREAMK, OPT_STREAMK_TIMEOUTS, OPT_IGEMM_MODE = 1, 2, 3
OPT_SPLITK, OPT_HOST_PANELS, OPT_STREAMK_SPIN_LIMIT, OPT_FAULT_INJECT, OPT_STREAMK_ORDER, OPT_DMA_EDGE, OPT_STREAMK_DELEGATIONS = 4, 5, 6, 7, 8, 9, 10
OPT_RIM = 11
OPT_STREAMK_CHAIN = 12
OPT_PERSIST = 13
OPT_RIM5 = 14
KERNELS = {"auto": KERNEL_AUTO, "valu": KERNEL_VALU, "mfma": KERNEL_MFMA,
           "mfma256": KERNEL_MFMA_256, "naive": KERNEL_NAIVE, "mfma_simple": KERNEL_MFMA_SIMPLE,
           "mfma_pipe": KERNEL_MFMA_PIPE, "mfma_tiles": 10, "mfma_128x64": 8, "mfma_64x64": 11, "mfma_256x256": 12,
           "valu_128x128": 13, "valu_64x64": 14, "valu_128x64": 9, "mfma_splitk": 15, "mfma_splitk_128x64": 20,
           "mfma_64x64_dma": 25, "mfma_128x64_dma": 27, "mfma_128x128_dma": 28,
           "mfma_64x64_dma5": 29, "mfma_128x64_dma5": 30, "mfma_128x128_dma5": 31,
           "mfma_96x96_dma5": 7, "mfma_96x64_dma5": 26, "mfma_160x160_dma5": 100,
           }
# (tools build only, libmmult_hip_ab.so: the 32x32x2 tiles mfma32_* / mfma32b_* (ids 48-51, 60-62), exp5_*, the rim --
# their names resolve through the library's own table, mmh_kernel_id)
# kernels that keep the one-chain-per-element contract (bit-identical results); the split-K ids do not
CHAIN_KERNELS = [k for k in KERNELS if "splitk" not in k]

# every symbol include/mmult_hip.h declares (tests assert the .so exports them all)
EXPORTS = [
    "mmh_strerror", "mmh_last_error", "mmh_last_launch", "mmh_version", "mmh_is_ab_build", "mmh_device_count",
    "mmh_device_info",
    "mmh_create", "mmh_destroy", "mmh_warm", "mmh_reserve_stream", "mmh_set_kernel", "mmh_get_kernel", "mmh_kernel_name", "mmh_kernel_id",
    "mmh_set_option", "mmh_get_option",
    "mmh_sgemm", "mmh_sgemm_host", "mmh_sgemm_host_timed", "mmh_igemm_s8", "mmh_quantize_sym_s8", "mmh_qgemm_f32",
    "mmh_sgemm_rocblas", "mmh_sgemm_hipblaslt", "mmh_shard_rows",
    "mmh_shard_create", "mmh_shard_destroy", "mmh_shard_set_kernel", "mmh_shard_info", "mmh_shard_sgemm", "mmh_shard_sgemm_streamed", "mmh_shard_chunks", "mmh_shard_pin",
    "mmh_shard_unpin",
    "mmh_rccl_version",
    "mmh_sgemm_sharded", "mmh_time_sgemm", "mmh_time_comparator", "mmh_trace_sgemm", "mmh_probe_mfma_f32", "mmh_probe_valu_f32", "mmh_probe_mfma_i8",
    "mmh_probe_mfma_i8_sustained", "mmh_probe_hbm_copy", "mmh_probe_hbm_read", "mmh_probe_lds_read", "mmh_streamk_plan", "mmh_auto_plan",
]


class MMultError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: status {status} ({detail})")


_lib: Optional[C.CDLL] = None
_lib_path = LIB_PATH


def use_ab_library(build: bool = True) -> str:
    """tools/ only: load libmmult_hip_ab.so (the product kernels plus the scheduling A/B variants, the families that
    measured slower -- the 32x32x2 tiles, the rim -- and the timing-only ablation builds, whose results are WRONG)
    instead of the product library.  Must be called before the first lib(); builds the library on demand
    (build=False: only if it is already there -- raises otherwise)."""
    global _lib_path
    if _lib is not None and _lib_path != AB_LIB_PATH:
        raise MMultError(ERR_INVALID_ARG, "use_ab_library", "the product library is already loaded")
    if build:
        from . import build as _build
        _build.build_ab_library()
    elif not os.path.exists(AB_LIB_PATH):
        raise MMultError(ERR_UNSUPPORTED, "use_ab_library", "libmmult_hip_ab.so has not been built")
    _lib_path = AB_LIB_PATH
    return AB_LIB_PATH


def streamk_plan(tiles: int, nk: int, grid: int):
    """(order, place) of a phase-ordered stream-K launch as numpy int32 arrays (host arithmetic only)."""
    order = np.zeros(grid, dtype=np.int32)
    place = np.zeros(tiles, dtype=np.int32)
    _check(lib().mmh_streamk_plan(tiles, nk, grid, order.ctypes.data_as(C.POINTER(C.c_int)),
                                  place.ctypes.data_as(C.POINTER(C.c_int))), "mmh_streamk_plan")
    return order, place


def auto_plan(m: int, n: int, k: int, lda: int = 0, ldb: int = 0, ldc: int = 0, base_align: int = 16, cu_count: int = 256):
    """What MMH_KERNEL_AUTO would run for a shape (host arithmetic only, no device): (short kernel name, tiles,
    stream-K grid) -- grid 0 = one workgroup per tile, -1 = needs the device's occupancy query."""
    kern, grid, tiles = C.c_int(), C.c_int(), C.c_long()
    _check(lib().mmh_auto_plan(m, n, k, lda or k, ldb or n, ldc or n, base_align, cu_count, C.byref(kern), C.byref(tiles),
                               C.byref(grid)), "mmh_auto_plan")
    names = {v: name for name, v in KERNELS.items() if name != "mfma256"}
    return names.get(kern.value, str(kern.value)), tiles.value, grid.value


def use_timeline_library() -> str:
    """tools/dma_timeline.py only: the A/B library built with per-workgroup timeline stamps."""
    global _lib_path
    if _lib is not None:
        raise MMultError(ERR_INVALID_ARG, "use_timeline_library", "a library is already loaded")
    from . import build as _build
    _lib_path = _build.build_timeline_library()
    return _lib_path


def _share_hip_runtime_with_torch() -> None:
    """One process must hold ONE HIP runtime.  PyTorch's ROCm wheels bundle their
    own libamdhip64.so.7; libmmult_hip.so needs the same SONAME.  If this
    library pulled in /opt/rocm's copy first, a later `import torch` would load a
    second runtime and find "No HIP GPUs".  So when torch is installed (it need
    not be imported), its copy is loaded first and ours binds to it.  C/C++
    callers without torch in the process simply use /opt/rocm's runtime."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> C.CDLL:
    """Load libmmult_hip.so (built in-tree by build.py).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(_lib_path):
        raise MMultError(ERR_UNSUPPORTED, "load",
                         f"{_lib_path} is missing -- run __graft_entry__.build(); "
                         "there is no CPU fallback")
    L = C.CDLL(_lib_path)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    L.mmh_strerror.argtypes = [C.c_int]
    L.mmh_strerror.restype = C.c_char_p
    L.mmh_last_error.restype = C.c_char_p
    L.mmh_last_launch.restype = C.c_char_p
    L.mmh_version.restype = C.c_int
    L.mmh_is_ab_build.restype = C.c_int
    L.mmh_device_count.argtypes = [ip]
    L.mmh_device_info.argtypes = [C.c_int, C.c_char_p, ip, ip]
    L.mmh_create.argtypes = [C.POINTER(vp), C.c_int]
    L.mmh_destroy.argtypes = [vp]
    L.mmh_warm.argtypes = [vp]
    L.mmh_reserve_stream.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    L.mmh_kernel_id.argtypes = [C.c_char_p]
    L.mmh_set_kernel.argtypes = [vp, C.c_int]
    L.mmh_get_kernel.argtypes = [vp, ip]
    L.mmh_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.mmh_get_option.argtypes = [vp, C.c_int, ip]
    L.mmh_kernel_name.argtypes = [C.c_int]
    L.mmh_kernel_name.restype = C.c_char_p
    gemm = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int]
    L.mmh_sgemm.argtypes = gemm + [C.c_int, vp]
    L.mmh_sgemm_host.argtypes = gemm + [C.c_int]
    L.mmh_sgemm_host_timed.argtypes = gemm + [C.c_int, C.POINTER(C.c_float)]
    L.mmh_igemm_s8.argtypes = gemm + [C.c_int, vp]
    L.mmh_sgemm_rocblas.argtypes = gemm + [vp]
    L.mmh_sgemm_hipblaslt.argtypes = gemm + [vp]
    L.mmh_qgemm_f32.argtypes = gemm + [vp]
    L.mmh_quantize_sym_s8.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp]
    L.mmh_shard_rows.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip]
    L.mmh_sgemm_sharded.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int,
                                    vp, C.c_int, C.c_int, fp]
    L.mmh_shard_create.argtypes = [C.POINTER(vp), C.c_int, ip]
    L.mmh_shard_destroy.argtypes = [vp]
    L.mmh_shard_set_kernel.argtypes = [vp, C.c_int]
    L.mmh_shard_info.argtypes = [vp, ip, ip]
    L.mmh_shard_sgemm.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, fp]
    L.mmh_shard_sgemm_streamed.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, fp]
    L.mmh_shard_chunks.argtypes = [C.c_int, C.c_int, ip]
    L.mmh_shard_pin.argtypes = [vp, vp, C.c_size_t]
    L.mmh_shard_unpin.argtypes = [vp, vp]
    L.mmh_rccl_version.argtypes = [ip]
    L.mmh_time_sgemm.argtypes = gemm + [C.c_int, C.c_int, vp, fp]
    L.mmh_time_comparator.argtypes = [vp, C.c_int] + gemm[1:] + [C.c_int, C.c_int, vp, fp]
    L.mmh_trace_sgemm.argtypes = gemm + [C.c_int, vp, fp]
    L.mmh_probe_hbm_read.argtypes = [vp, C.c_size_t, fp]
    L.mmh_probe_lds_read.argtypes = [vp, C.c_int, fp]
    L.mmh_streamk_plan.argtypes = [C.c_long, C.c_int, C.c_int, ip, ip]
    L.mmh_auto_plan.argtypes = [C.c_int] * 8 + [ip, C.POINTER(C.c_long), ip]
    L.mmh_probe_mfma_f32.argtypes = [vp, fp]
    L.mmh_probe_valu_f32.argtypes = [vp, C.c_int, C.c_int, fp]
    L.mmh_probe_mfma_i8.argtypes = [vp, fp]
    L.mmh_probe_mfma_i8_sustained.argtypes = [vp, C.c_int, C.c_float, fp]
    L.mmh_probe_hbm_copy.argtypes = [vp, C.c_size_t, fp]
    _lib = L
    return L


def _check(status: int, where: str) -> None:
    if status != OK:
        L = lib()
        detail = L.mmh_strerror(status).decode()
        last = L.mmh_last_error().decode()
        raise MMultError(status, where, f"{detail}; {last}" if last else detail)


def device_count() -> int:
    n = C.c_int(0)
    _check(lib().mmh_device_count(C.byref(n)), "mmh_device_count")
    return n.value


def rccl_version() -> int:
    """ncclGetVersion's code as libmmult_hip.so sees RCCL (dlopen); raises MMultError(ERR_UNSUPPORTED)
    when librccl or an entry point the shard needs is missing.  Needs no GPU."""
    v = C.c_int(0)
    _check(lib().mmh_rccl_version(C.byref(v)), "mmh_rccl_version")
    return v.value


def shard_chunks(k: int, b_chunks: int) -> list[int]:
    """K boundaries of mmh_shard_sgemm_streamed's broadcast chunks (host arithmetic, no GPU): [0, ..., k]."""
    k0 = (C.c_int * 65)()
    c = lib().mmh_shard_chunks(int(k), int(b_chunks), k0)
    if c < 0:
        raise MMultError(c, "mmh_shard_chunks")
    return [k0[i] for i in range(c + 1)]


def shard_rows(m: int, nranks: int, rank: int) -> tuple[int, int]:
    """(row0, rows) of the C/A row panel owned by `rank` (pure host arithmetic)."""
    r0, nr = C.c_int(0), C.c_int(0)
    _check(lib().mmh_shard_rows(m, nranks, rank, C.byref(r0), C.byref(nr)), "mmh_shard_rows")
    return r0.value, nr.value


def last_launch() -> str:
    """Which kernel configuration the last sgemm call on this thread launched."""
    return lib().mmh_last_launch().decode()


def kernel_name(kernel: int) -> Optional[str]:
    s = lib().mmh_kernel_name(kernel)
    return s.decode() if s else None


def _kernel_id(kernel) -> int:
    if isinstance(kernel, str):
        if kernel in KERNELS:
            return KERNELS[kernel]
        kid = lib().mmh_kernel_id(kernel.encode())      # the library's own table (A/B ids of the tools build too)
        if kid < 0:
            raise KeyError(kernel)
        return kid
    return int(kernel)


def _np_ptr(x: np.ndarray) -> int:
    return x.ctypes.data


class MMult:
    """One handle = one device + the selected kernel variant (the reference's
    cublasHandle_t lifetime, cuda/test_MMult.cpp:43-44,142)."""

    def __init__(self, device: int = 0, kernel="auto"):
        self._h = C.c_void_p(None)
        _check(lib().mmh_create(C.byref(self._h), device), "mmh_create")
        self.device = device
        self.set_kernel(kernel)

    def close(self) -> None:
        if self._h:
            lib().mmh_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def warm(self) -> None:
        """Everything a first launch would pay for (code objects, LDS opt-ins, residency queries, stream-K
        workspaces) -- mmh_create does it already unless MMH_LAZY=1; idempotent."""
        _check(lib().mmh_warm(self._h), "mmh_warm")

    # -- configuration ------------------------------------------------------
    def set_kernel(self, kernel) -> None:
        _check(lib().mmh_set_kernel(self._h, _kernel_id(kernel)), "mmh_set_kernel")

    def get_kernel(self) -> int:
        k = C.c_int(0)
        _check(lib().mmh_get_kernel(self._h, C.byref(k)), "mmh_get_kernel")
        return k.value

    def reserve_stream(self, stream: int, m: int, n: int, k: int) -> None:
        """Give `stream` (a raw hipStream_t) a stream-K workspace set of its own, large enough for any launch of an
        m x n x k problem: what a launch that is to be CAPTURED on that stream needs (include/mmult_hip.h, hipGraphs)."""
        _check(lib().mmh_reserve_stream(self._h, stream, m, n, k), "mmh_reserve_stream")

    def set_streamk(self, on) -> None:
        """False / 0 never, True / 1 when a round would be > 7 % empty (default), 2 whenever ragged."""
        _check(lib().mmh_set_option(self._h, OPT_STREAMK, int(on)), "mmh_set_option")

    def set_igemm_mode(self, mode: int) -> None:
        """0 B read in place (default; packed B for unaligned operands): the ping-pong 256x256 kernel from one tile
        per CU up, 128x128 tiles below; 1 in-kernel transpose; 2 correctness-first kernel; 3 / 4 packed-B + LDS-DMA
        with 128x128 / 256x256 tiles; 5 / 6 the lockstep in-place kernel likewise; 7 / 8 the ping-pong kernel with
        16 / 32 MFMAs per phase.  (10..13, timing-only ablations with wrong results, exist only in
        the tools build libmmult_hip_ab.so; the product library rejects them.)"""
        _check(lib().mmh_set_option(self._h, OPT_IGEMM_MODE, int(mode)), "mmh_set_option")

    def streamk_timeouts(self) -> int:
        """Synchronises; the handle's sticky error count: stream-K / split-K hand-off waits that timed
        out since it was last cleared (must be 0; while it is not, every call on the handle raises)."""
        v = C.c_int(0)
        _check(lib().mmh_get_option(self._h, OPT_STREAMK_TIMEOUTS, C.byref(v)), "mmh_get_option")
        return v.value

    def clear_error(self) -> None:
        """Synchronises and clears the sticky error."""
        _check(lib().mmh_set_option(self._h, OPT_STREAMK_TIMEOUTS, 0), "mmh_set_option")

    def set_option(self, option: int, value: int) -> None:
        _check(lib().mmh_set_option(self._h, int(option), int(value)), "mmh_set_option")

    def get_option(self, option: int) -> int:
        v = C.c_int(0)
        _check(lib().mmh_get_option(self._h, int(option), C.byref(v)), "mmh_get_option")
        return v.value

    def set_splitk(self, parts: int) -> None:
        """OPT-IN split-K for MMH_KERNEL_AUTO: 0 off (default), 1 as many parts as fill the chip,
        2..16 that many.  Deterministic, inside the harness tolerance, NOT the chain's bits."""
        self.set_option(OPT_SPLITK, parts)

    def set_host_panels(self, panels: int) -> None:
        """Row panels of the host flavour's copy/compute pipeline: -1 automatic, 0/1 plain, 2..16."""
        self.set_option(OPT_HOST_PANELS, panels)

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cu, mhz = C.c_int(0), C.c_int(0)
        _check(lib().mmh_device_info(self.device, name, C.byref(cu), C.byref(mhz)), "mmh_device_info")
        return {"name": name.value.decode(), "cu_count": cu.value, "clock_mhz": mhz.value}

    # -- the hot path, raw-pointer form (device flavour of MY_MMult) ---------
    def sgemm(self, m, n, k, dA: int, lda, dB: int, ldb, dC: int, ldc, accumulate=False,
              stream: int = 0) -> None:
        _check(lib().mmh_sgemm(self._h, m, n, k, dA, lda, dB, ldb, dC, ldc, int(bool(accumulate)),
                               stream), "mmh_sgemm")

    def MY_MMult_device(self, m, n, k, d_A: int, lda, d_B: int, ldb, d_C: int, ldc, stream: int = 0):
        """cuda/test_MMult.cpp:102 -- MY_MMult(handle, m, n, k, d_A, k, d_B, n, d_C, n):
        C = A*B on device pointers, asynchronous."""
        self.sgemm(m, n, k, d_A, lda, d_B, ldb, d_C, ldc, False, stream)

    # -- host flavour ----------------------------------------------------------
    def MY_MMult(self, m, n, k, a: np.ndarray, lda, b: np.ndarray, ldb, c: np.ndarray, ldc) -> None:
        """armv7/MMult0.c:9-24 semantics on host buffers: C = A*B + C (the caller
        pre-zeroes C, armv7/test_MMult.c:57,71).  a, b, c are flat or 2-D fp32
        numpy buffers addressed row-major with the given leading dimensions."""
        for x in (a, b, c):
            if x.dtype != np.float32 or not x.flags["C_CONTIGUOUS"]:
                raise MMultError(ERR_INVALID_ARG, "MY_MMult", "need C-contiguous float32 buffers")
        if k > 0 and m > 0 and (a.size < (m - 1) * lda + k or b.size < (k - 1) * ldb + n):
            raise MMultError(ERR_INVALID_ARG, "MY_MMult", "input buffer smaller than (rows-1)*ld+cols")
        if m > 0 and n > 0 and c.size < (m - 1) * ldc + n:
            raise MMultError(ERR_INVALID_ARG, "MY_MMult", "C buffer smaller than (m-1)*ldc+n")
        _check(lib().mmh_sgemm_host(self._h, m, n, k, _np_ptr(a), lda, _np_ptr(b), ldb, _np_ptr(c),
                                    ldc, 1), "mmh_sgemm_host")

    def MY_MMult_ms(self, m, n, k, a: np.ndarray, b: np.ndarray, c: np.ndarray) -> float:
        """vulkan/test_MMult.cpp:10,55 semantics: dense row-major host buffers (lda = k, ldb = n,
        ldc = n), C = A*B, returns the device time of the GEMM in milliseconds."""
        for x in (a, b, c):
            if x.dtype != np.float32 or not x.flags["C_CONTIGUOUS"]:
                raise MMultError(ERR_INVALID_ARG, "MY_MMult_ms", "need C-contiguous float32 buffers")
        if a.size < m * k or b.size < k * n or c.size < m * n:
            raise MMultError(ERR_INVALID_ARG, "MY_MMult_ms", "buffer smaller than rows*cols")
        ms = C.c_float(0.0)
        _check(lib().mmh_sgemm_host_timed(self._h, m, n, k, _np_ptr(a), max(k, 1), _np_ptr(b), max(n, 1),
                                          _np_ptr(c), max(n, 1), 0, C.byref(ms)), "mmh_sgemm_host_timed")
        return float(ms.value)

    def sgemm_host(self, a: np.ndarray, b: np.ndarray, c: Optional[np.ndarray] = None,
                   accumulate: bool = False) -> np.ndarray:
        m, k = a.shape
        k2, n = b.shape
        if k != k2:
            raise MMultError(ERR_INVALID_ARG, "sgemm_host", "inner dimensions differ")
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        if c is None:
            if accumulate:
                raise MMultError(ERR_INVALID_ARG, "sgemm_host", "accumulate needs c=")
            c = np.zeros((m, n), dtype=np.float32)
        elif (not isinstance(c, np.ndarray) or c.dtype != np.float32 or not c.flags["C_CONTIGUOUS"]
              or not c.flags["WRITEABLE"] or c.shape != (m, n)):
            raise MMultError(ERR_INVALID_ARG, "sgemm_host", f"c must be a writable C-contiguous float32 ({m},{n}) array")
        _check(lib().mmh_sgemm_host(self._h, m, n, k, _np_ptr(a), max(k, 1), _np_ptr(b), max(n, 1),
                                    _np_ptr(c), max(n, 1), int(accumulate)), "mmh_sgemm_host")
        return c

    # -- torch device-tensor glue (device memory + streams only) -------------
    def _tensor_args(self, t, rows, cols, what, dtype):
        """(data_ptr, leading dimension) of a (rows, cols) row-major window, after checking what the
        kernels take on trust: dtype, device (the handle's), unit column stride, and a row stride
        that does not fold rows onto each other (an expanded / overlapping view has stride(0) < cols)."""
        if not t.is_cuda:
            raise MMultError(ERR_INVALID_ARG, what, "tensor is not on a GPU (no CPU fallback)")
        if t.dtype != dtype:
            raise MMultError(ERR_INVALID_ARG, what, f"need dtype {dtype}, got {t.dtype}")
        if t.device.index != self.device:
            raise MMultError(ERR_INVALID_ARG, what,
                             f"tensor lives on cuda:{t.device.index}, the handle on cuda:{self.device}")
        if t.dim() != 2 or t.shape[0] != rows or t.shape[1] != cols or (cols > 1 and t.stride(1) != 1):
            raise MMultError(ERR_INVALID_ARG, what, f"need a ({rows},{cols}) row-major 2-D tensor")
        if rows > 1 and t.stride(0) < max(cols, 1):
            raise MMultError(ERR_INVALID_ARG, what,
                             f"row stride {t.stride(0)} < {cols} columns: rows overlap (expanded view?)")
        ld = t.stride(0) if rows > 1 else max(cols, 1)
        return t.data_ptr(), max(ld, 1)

    def matmul(self, a, b, out=None, accumulate: bool = False):
        """C = A @ B (+ C) for fp32 CUDA tensors, on torch's current stream."""
        import torch
        if a.dtype != torch.float32 or b.dtype != torch.float32:
            raise MMultError(ERR_INVALID_ARG, "matmul", "fp32 only")
        m, k = a.shape
        k2, n = b.shape
        if k != k2:
            raise MMultError(ERR_INVALID_ARG, "matmul", "inner dimensions differ")
        if out is None:
            if accumulate:
                raise MMultError(ERR_INVALID_ARG, "matmul", "accumulate needs out=")
            out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        pa, lda = self._tensor_args(a, m, k, "matmul(A)", torch.float32)
        pb, ldb = self._tensor_args(b, k, n, "matmul(B)", torch.float32)
        pc, ldc = self._tensor_args(out, m, n, "matmul(C)", torch.float32)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        self.sgemm(m, n, k, pa, lda, pb, ldb, pc, ldc, accumulate, stream)
        return out

    def igemm_s8(self, a, b, out=None, accumulate: bool = False):
        """int8 x int8 -> int32 for CUDA tensors (inputs expected in [-127,127])."""
        import torch
        if a.dtype != torch.int8 or b.dtype != torch.int8:
            raise MMultError(ERR_INVALID_ARG, "igemm_s8", "int8 inputs only")
        m, k = a.shape
        k2, n = b.shape
        if k != k2:
            raise MMultError(ERR_INVALID_ARG, "igemm_s8", "inner dimensions differ")
        if out is None:
            if accumulate:
                raise MMultError(ERR_INVALID_ARG, "igemm_s8", "accumulate needs out=")
            out = torch.empty((m, n), dtype=torch.int32, device=a.device)
        pa, lda = self._tensor_args(a, m, k, "igemm_s8(A)", torch.int8)
        pb, ldb = self._tensor_args(b, k, n, "igemm_s8(B)", torch.int8)
        pc, ldc = self._tensor_args(out, m, n, "igemm_s8(C)", torch.int32)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        _check(lib().mmh_igemm_s8(self._h, m, n, k, pa, lda, pb, ldb, pc, ldc, int(accumulate), stream),
               "mmh_igemm_s8")
        return out

    def quantize_sym_s8(self, x):
        """fp32 CUDA tensor -> (int8 tensor in [-127,127], scale as a 1-element CUDA tensor)."""
        import torch
        if x.dim() != 2:
            raise MMultError(ERR_INVALID_ARG, "quantize(X)", "need a 2-D tensor")
        rows, cols = x.shape
        px, ldx = self._tensor_args(x, rows, cols, "quantize(X)", torch.float32)
        q = torch.empty((rows, cols), dtype=torch.int8, device=x.device)
        scale = torch.empty(1, dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _check(lib().mmh_quantize_sym_s8(self._h, rows, cols, px, ldx, q.data_ptr(), max(cols, 1),
                                         scale.data_ptr(), stream), "mmh_quantize_sym_s8")
        return q, scale

    def qgemm(self, a, b, out=None):
        """C_f32 = dequantise(quantise(A) @ quantise(B)): chgemm-style symmetric int8 GEMM."""
        import torch
        m, k = a.shape
        k2, n = b.shape
        if k != k2:
            raise MMultError(ERR_INVALID_ARG, "qgemm", "inner dimensions differ")
        if out is None:
            out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        pa, lda = self._tensor_args(a, m, k, "qgemm(A)", torch.float32)
        pb, ldb = self._tensor_args(b, k, n, "qgemm(B)", torch.float32)
        pc, ldc = self._tensor_args(out, m, n, "qgemm(C)", torch.float32)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        _check(lib().mmh_qgemm_f32(self._h, m, n, k, pa, lda, pb, ldb, pc, ldc, stream), "mmh_qgemm_f32")
        return out

    def matmul_rocblas(self, a, b, out=None):
        """Vendor comparator (cuda/MMult_cuBLAS_1.cpp:11-19)."""
        import torch
        m, k = a.shape
        k2, n = b.shape
        if k != k2:
            raise MMultError(ERR_INVALID_ARG, "matmul_rocblas", "inner dimensions differ")
        if out is None:
            out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        pa, lda = self._tensor_args(a, m, k, "rocblas(A)", torch.float32)
        pb, ldb = self._tensor_args(b, k, n, "rocblas(B)", torch.float32)
        pc, ldc = self._tensor_args(out, m, n, "rocblas(C)", torch.float32)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        _check(lib().mmh_sgemm_rocblas(self._h, m, n, k, pa, lda, pb, ldb, pc, ldc, stream),
               "mmh_sgemm_rocblas")
        return out

    def matmul_hipblaslt(self, a, b, out=None):
        """The second vendor comparator (cuda/MMult_cuBLAS_2.cpp:11-26): hipBLASLt, fp32 compute."""
        import torch
        m, k = a.shape
        k2, n = b.shape
        if k != k2:
            raise MMultError(ERR_INVALID_ARG, "matmul_hipblaslt", "inner dimensions differ")
        if out is None:
            out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        pa, lda = self._tensor_args(a, m, k, "hipblaslt(A)", torch.float32)
        pb, ldb = self._tensor_args(b, k, n, "hipblaslt(B)", torch.float32)
        pc, ldc = self._tensor_args(out, m, n, "hipblaslt(C)", torch.float32)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        _check(lib().mmh_sgemm_hipblaslt(self._h, m, n, k, pa, lda, pb, ldb, pc, ldc, stream),
               "mmh_sgemm_hipblaslt")
        return out

    # -- measurement -------------------------------------------------------------
    def time_sgemm(self, m, n, k, dA, lda, dB, ldb, dC, ldc, warmup=1, reps=20, stream: int = 0) -> float:
        """Mean ms per call: one hipEvent pair around `reps` back-to-back launches on
        `stream` (the reference's convention, cuda/test_MMult.cpp:98-114)."""
        ms = C.c_float(0)
        _check(lib().mmh_time_sgemm(self._h, m, n, k, dA, lda, dB, ldb, dC, ldc, warmup, reps, stream,
                                    C.byref(ms)), "mmh_time_sgemm")
        return ms.value

    def time_comparator(self, which: str, m, n, k, dA, lda, dB, ldb, dC, ldc, warmup=1, reps=20, stream: int = 0) -> float:
        """time_sgemm for a vendor comparator ("rocblas" / "hipblaslt"): calls issued from C, one event pair."""
        ms = C.c_float(0)
        _check(lib().mmh_time_comparator(self._h, {"rocblas": 1, "hipblaslt": 2}[which], m, n, k, dA, lda, dB, ldb, dC, ldc,
                                         warmup, reps, stream, C.byref(ms)), "mmh_time_comparator")
        return ms.value

    def trace_sgemm(self, m, n, k, dA, lda, dB, ldb, dC, ldc, count=400, stream: int = 0):
        """Per-launch ms of `count` back-to-back launches (one hipEvent pair each): the clock ramp."""
        buf = (C.c_float * count)()
        _check(lib().mmh_trace_sgemm(self._h, m, n, k, dA, lda, dB, ldb, dC, ldc, count, stream, buf),
               "mmh_trace_sgemm")
        return list(buf)

    def probe_mfma_f32(self) -> float:
        v = C.c_float(0)
        _check(lib().mmh_probe_mfma_f32(self._h, C.byref(v)), "mmh_probe_mfma_f32")
        return v.value

    def probe_valu_f32(self, packed: bool = True, waves_per_simd: int = 2) -> float:
        """TFLOP/s of a v_pk_fma_f32-only (packed) or v_fma_f32-only loop: the vector-ALU rung's measured roof."""
        v = C.c_float(0)
        _check(lib().mmh_probe_valu_f32(self._h, int(bool(packed)), int(waves_per_simd), C.byref(v)), "mmh_probe_valu_f32")
        return v.value

    def probe_mfma_i8(self) -> float:
        v = C.c_float(0)
        _check(lib().mmh_probe_mfma_i8(self._h, C.byref(v)), "mmh_probe_mfma_i8")
        return v.value

    def probe_mfma_i8_sustained(self, random_operands: bool = True, min_ms: float = 50.0) -> float:
        """int8 MFMA-only rate of the LAST 2.3 ms launch after `min_ms` of back-to-back launches,
        with per-MFMA pseudo-random operands (what the power manager sustains on real data) or
        with constant ones."""
        v = C.c_float(0)
        _check(lib().mmh_probe_mfma_i8_sustained(self._h, int(bool(random_operands)), float(min_ms), C.byref(v)),
               "mmh_probe_mfma_i8_sustained")
        return v.value

    def probe_hbm_copy(self, nbytes: int = 1 << 30) -> float:
        v = C.c_float(0)
        _check(lib().mmh_probe_hbm_copy(self._h, nbytes, C.byref(v)), "mmh_probe_hbm_copy")
        return v.value

    def probe_hbm_read(self, nbytes: int = 1 << 30) -> float:
        v = C.c_float(0)
        _check(lib().mmh_probe_hbm_read(self._h, nbytes, C.byref(v)), "mmh_probe_hbm_read")
        return v.value

    def probe_lds_read(self, width: int = 16) -> float:
        """LDS fragment-read rate summed over the chip, GB/s (width 16 / 8 / 4 bytes per lane, -8 = the
        transposing ds_read_b64_tr_b8)."""
        v = C.c_float(0.0)
        _check(lib().mmh_probe_lds_read(self._h, int(width), C.byref(v)), "mmh_probe_lds_read")
        return float(v.value)


class ShardedMMult:
    """Single-process row-panel shard over `ngpus` devices with everything persistent (one RCCL
    communicator, per-device streams, product handles and buffers): mmh_shard_create/_sgemm/_destroy.
    Raises MMultError(ERR_NO_DEVICE) when fewer devices are visible -- never fewer ranks silently."""

    def __init__(self, ngpus: int, devices=None, kernel="auto"):
        self._h = C.c_void_p(None)
        dev = None
        if devices is not None:
            if len(devices) != ngpus:
                raise MMultError(ERR_INVALID_ARG, "ShardedMMult", "len(devices) != ngpus")
            dev = (C.c_int * ngpus)(*devices)
        _check(lib().mmh_shard_create(C.byref(self._h), ngpus, dev), "mmh_shard_create")
        _check(lib().mmh_shard_set_kernel(self._h, _kernel_id(kernel)), "mmh_shard_set_kernel")

    def close(self) -> None:
        if self._h:
            lib().mmh_shard_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def info(self) -> dict:
        n, r = C.c_int(0), C.c_int(0)
        _check(lib().mmh_shard_info(self._h, C.byref(n), C.byref(r)), "mmh_shard_info")
        return {"ngpus": n.value, "rccl_ranks": r.value}

    def pin(self, x: np.ndarray) -> None:
        """Page-lock a host array that will be passed to sgemm() repeatedly (unpin before it is freed)."""
        _check(lib().mmh_shard_pin(self._h, _np_ptr(x), x.nbytes), "mmh_shard_pin")

    def unpin(self, x: np.ndarray) -> None:
        _check(lib().mmh_shard_unpin(self._h, _np_ptr(x)), "mmh_shard_unpin")

    def sgemm(self, a: np.ndarray, b: np.ndarray, c: Optional[np.ndarray] = None, gemm_reps: int = 1, b_chunks: int = 1):
        """C = A @ B on host arrays.  Returns (C, {"h2d","bcast","gemm","d2h"} ms; gemm is per rep); with b_chunks > 1
        (mmh_shard_sgemm_streamed: B broadcast in K-chunks under the GEMMs that consume them) also "overlapped",
        "chunks", "first_pass", "wall"."""
        m, k = a.shape
        k2, n = b.shape
        if k != k2:
            raise MMultError(ERR_INVALID_ARG, "ShardedMMult.sgemm", "inner dimensions differ")
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        if c is None:
            c = np.zeros((m, n), dtype=np.float32)
        elif c.dtype != np.float32 or not c.flags["C_CONTIGUOUS"] or c.shape != (m, n):
            raise MMultError(ERR_INVALID_ARG, "ShardedMMult.sgemm", "c must be C-contiguous float32 (m,n)")
        if b_chunks > 1:
            t = (C.c_float * 8)()
            _check(lib().mmh_shard_sgemm_streamed(self._h, m, n, k, _np_ptr(a), max(k, 1), _np_ptr(b), max(n, 1), _np_ptr(c),
                                                  max(n, 1), gemm_reps, b_chunks, t), "mmh_shard_sgemm_streamed")
            return c, dict(zip(("h2d", "bcast", "gemm", "d2h", "overlapped", "chunks", "first_pass", "wall"), list(t)))
        t = (C.c_float * 4)()
        _check(lib().mmh_shard_sgemm(self._h, m, n, k, _np_ptr(a), max(k, 1), _np_ptr(b), max(n, 1), _np_ptr(c),
                                     max(n, 1), gemm_reps, t), "mmh_shard_sgemm")
        return c, dict(zip(("h2d", "bcast", "gemm", "d2h"), list(t)))


def sgemm_sharded(ngpus: int, a: np.ndarray, b: np.ndarray, kernel="mfma"):
    """Single-process multi-device row-panel shard (mmh_sgemm_sharded).
    Returns (C, {"h2d","bcast","gemm","d2h"} ms)."""
    m, k = a.shape
    _, n = b.shape
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    c = np.zeros((m, n), dtype=np.float32)
    t = (C.c_float * 4)()
    _check(lib().mmh_sgemm_sharded(ngpus, m, n, k, _np_ptr(a), max(k, 1), _np_ptr(b), max(n, 1),
                                   _np_ptr(c), max(n, 1), _kernel_id(kernel), t), "mmh_sgemm_sharded")
    return c, dict(zip(("h2d", "bcast", "gemm", "d2h"), list(t)))


__all__ = ["MMult", "ShardedMMult", "MMultError", "lib", "use_ab_library", "device_count", "rccl_version", "shard_rows", "shard_chunks",
           "kernel_name", "last_launch", "use_timeline_library", "streamk_plan", "auto_plan", "sgemm_sharded", "KERNELS", "CHAIN_KERNELS", "AB_LIB_PATH",
           "OPT_SPLITK", "OPT_HOST_PANELS", "OPT_STREAMK_SPIN_LIMIT", "OPT_FAULT_INJECT", "OPT_STREAMK_ORDER", "OPT_DMA_EDGE", "OPT_STREAMK_DELEGATIONS", "OPT_RIM", "OPT_STREAMK_CHAIN", "OPT_PERSIST", "OPT_RIM5", "KERNEL_AUTO", "KERNEL_VALU", "KERNEL_MFMA", "KERNEL_MFMA_256", "KERNEL_NAIVE", "KERNEL_MFMA_SIMPLE", "KERNEL_MFMA_PIPE",
           "EXPORTS", "LIB_PATH", "OPT_STREAMK", "OPT_STREAMK_TIMEOUTS", "OPT_IGEMM_MODE", "OK", "ERR_INVALID_ARG", "ERR_HIP", "ERR_NO_DEVICE",
           "ERR_UNSUPPORTED", "ERR_ALLOC", "ERR_COMM"]
