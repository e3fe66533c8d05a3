"""ctypes front-end for the CPU oracle (oracle/oracle_mmult.c) and, where it
has been built, for the reference's own compiled objects under oracle/_ref/.

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never from the shipped package.

Parity status: pinned (see oracle_mmult.c header and tests/test_oracle.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
REFERENCE_ROOT = "/root/reference"

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(with_ref: bool | None = None) -> None:
    """Compile liboracle.so; compile oracle/_ref from the reference's own
    sources when /root/reference is present (build container only)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if with_ref is None:
        with_ref = os.path.isdir(REFERENCE_ROOT)
    if with_ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(with_ref=False)
        L = C.CDLL(path)
        mm = [C.c_int, C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int]
        for name in ("orc_ref_mmult", "orc_ref_mmult_fma"):
            getattr(L, name).argtypes = mm
            getattr(L, name).restype = None
        for name in ("orc_ref_mmult_fast", "orc_ref_mmult_fma_fast"):
            getattr(L, name).argtypes = mm + [C.c_int]
            getattr(L, name).restype = None
        L.orc_ref_mmult_f64.argtypes = [C.c_int, C.c_int, C.c_int, _f32p, C.c_int, _f32p,
                                        C.c_int, _f64p, C.c_int, C.c_int]
        L.orc_ref_mmult_f64.restype = None
        L.orc_compare_matrices.argtypes = [C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.c_int,
                                           C.POINTER(C.c_int)]
        L.orc_compare_matrices.restype = C.c_float
        L.orc_random_matrix.argtypes = [C.c_int, C.c_int, _f32p, C.c_int]
        L.orc_random_matrix.restype = None
        L.orc_pattern_matrix.argtypes = [C.c_int, C.c_int, _f32p, C.c_int, C.c_int]
        L.orc_pattern_matrix.restype = None
        L.orc_srand48.argtypes = [C.c_long]
        L.orc_srand48.restype = None
        L.orc_copy_matrix.argtypes = [C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.c_int]
        L.orc_copy_matrix.restype = None
        L.orc_dclock.restype = C.c_double
        L.orc_ref_igemm_s8.argtypes = [C.c_int, C.c_int, C.c_int, _i8p, C.c_int, _i8p, C.c_int,
                                       _i32p, C.c_int, C.c_int]
        L.orc_ref_igemm_s8.restype = None
        L.orc_quantize_sym_s8.argtypes = [C.c_size_t, _f32p, _i8p]
        L.orc_quantize_sym_s8.restype = C.c_float
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


# ---------------------------------------------------------------- helpers --
def _ld(x: np.ndarray) -> int:
    assert x.ndim == 2 and x.strides[1] == x.itemsize
    return x.strides[0] // x.itemsize


def _base(x: np.ndarray) -> np.ndarray:
    """The contiguous parent buffer of a row-strided 2-D view (for ndpointer)."""
    if x.flags["C_CONTIGUOUS"]:
        return x
    b = x.base
    assert b is not None and b.flags["C_CONTIGUOUS"], "need a view of a contiguous buffer"
    assert x.ctypes.data == b.ctypes.data, "view must start at the parent's origin"
    return b


def ref_mmult(a, b, c=None, *, fma=False, fast=True, nthreads=0):
    """C = A*B + C with the reference's i/j/p summation order.
    a: (m,k) b: (k,n) fp32, row-strided views allowed (ld = row stride)."""
    m, k = a.shape
    k2, n = b.shape
    assert k == k2
    if c is None:
        c = np.zeros((m, n), dtype=np.float32)
    L = lib()
    name = "orc_ref_mmult" + ("_fma" if fma else "") + ("_fast" if fast else "")
    args = [m, n, k, _base(a), _ld(a), _base(b), _ld(b), _base(c), _ld(c)]
    if fast:
        args.append(nthreads)
    getattr(L, name)(*args)
    return c


def ref_mmult_f64(a, b, nthreads=0):
    m, k = a.shape
    _, n = b.shape
    c = np.zeros((m, n), dtype=np.float64)
    lib().orc_ref_mmult_f64(m, n, k, _base(a), _ld(a), _base(b), _ld(b), c, n, nthreads)
    return c


def compare_matrices(a, b):
    m, n = a.shape
    bad = (C.c_int * 2)()
    d = lib().orc_compare_matrices(m, n, _base(a), _ld(a), _base(b), _ld(b), bad)
    return float(d), (bad[0], bad[1])


def random_matrix(m, n, lda=None, *, seed=None, pattern=None):
    """Restates cuda/random_matrix.cpp: column-major fill a[j*lda+i] of an
    (m x n) logical matrix, returns the flat buffer of n*lda floats.
    pattern: None -> drand48 uniform [-1,1); 3 / 2 -> (j-i)%pattern; 0 -> ones."""
    lda = m if lda is None else lda
    # the reference calls this with lda < m for non-square `cold`
    # (cuda/test_MMult.cpp:79); size the buffer for what the loop touches.
    buf = np.zeros((n - 1) * lda + max(m, lda), dtype=np.float32)
    L = lib()
    if seed is not None:
        L.orc_srand48(seed)
    if pattern is None:
        L.orc_random_matrix(m, n, buf, lda)
    else:
        L.orc_pattern_matrix(m, n, buf, lda, pattern)
    return buf


def harness_inputs(m, n, k, *, seed=None, pattern=None):
    """Reproduces cuda/test_MMult.cpp:77-81: random_matrix(m,k,a,m);
    random_matrix(k,n,b,k); random_matrix(m,n,cold,n) (burned), and returns
    (A as the (m,k) row-major array the kernels then read with lda=k,
     B as (k,n) row-major with ldb=n)."""
    a = random_matrix(m, k, lda=m, seed=seed, pattern=pattern)
    b = random_matrix(k, n, lda=k, pattern=pattern)
    random_matrix(m, n, lda=n, pattern=pattern)  # `cold`, overwritten with 0 by the harness
    return a[:m * k].reshape(m, k), b[:k * n].reshape(k, n)


def ref_igemm_s8(a, b, c=None, nthreads=0):
    m, k = a.shape
    _, n = b.shape
    if c is None:
        c = np.zeros((m, n), dtype=np.int32)
    lib().orc_ref_igemm_s8(m, n, k, _base(a), _ld(a), _base(b), _ld(b), _base(c), _ld(c), nthreads)
    return c


def quantize_sym_s8(x):
    q = np.empty(x.shape, dtype=np.int8)
    s = lib().orc_quantize_sym_s8(x.size, np.ascontiguousarray(x).reshape(-1), q.reshape(-1))
    return q, float(s)


# --------------------------------------------- the compiled reference (_ref) --
def have_ref() -> bool:
    return all(os.path.exists(os.path.join(REF_DIR, f)) for f in
               ("libref_armv7.so", "libref_armv7_fma.so", "libref_cuda_utils.so"))


_reflibs = {}


def reflib(name: str) -> C.CDLL:
    """name in {"armv7", "armv7_fma", "cuda_utils"}."""
    if name not in _reflibs:
        L = C.CDLL(os.path.join(REF_DIR, f"libref_{name}.so"))
        mm = [C.c_int, C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int]
        if name.startswith("armv7"):
            L.REF_MMult.argtypes = mm
            L.REF_MMult.restype = None
            L.MY_MMult.argtypes = mm
            L.MY_MMult.restype = None
            L.compare_matrices.argtypes = [C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.c_int]
            L.compare_matrices.restype = C.c_float
            L.random_matrix.argtypes = [C.c_int, C.c_int, _f32p, C.c_int]
            L.copy_matrix.argtypes = [C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.c_int]
            L.dclock.restype = C.c_double
        else:
            L.compare = getattr(L, "_Z16compare_matricesiiPfiS_i")
            L.compare.argtypes = [C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.c_int]
            L.compare.restype = C.c_float
            L.random = getattr(L, "_Z13random_matrixiiPfi")
            L.random.argtypes = [C.c_int, C.c_int, _f32p, C.c_int]
            L.random.restype = None
            L.copy = getattr(L, "_Z11copy_matrixiiPfiS_i")
            L.copy.argtypes = [C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.c_int]
            L.ref6 = getattr(L, "_Z9REF_MMultiiiPfS_S_")
            L.ref6.argtypes = [C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p]
            L.ref6.restype = None
            L.dclock = getattr(L, "_Z6dclockv")
            L.dclock.restype = C.c_double
        _reflibs[name] = L
    return _reflibs[name]


def reference_ref_mmult(a, b, c=None, *, fma=False):
    """The reference's own armv7/REF_MMult.c object (serial i/j/p loop)."""
    m, k = a.shape
    _, n = b.shape
    if c is None:
        c = np.zeros((m, n), dtype=np.float32)
    L = reflib("armv7_fma" if fma else "armv7")
    L.REF_MMult(m, n, k, _base(a), _ld(a), _base(b), _ld(b), _base(c), _ld(c))
    return c


def max_threads() -> int:
    return int(lib().orc_max_threads())
