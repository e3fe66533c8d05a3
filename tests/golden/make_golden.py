#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE.

Run in the build container (needs /root/reference):
    make -C oracle ref && python tests/golden/make_golden.py

For every case the inputs are produced by the reference's input generator
(cuda/random_matrix.cpp compiled into oracle/_ref/libref_cuda_utils.so, after
srand48(seed); or the (j-i)%3 / (j-i)%2 / all-ones known-answer patterns of
cuda/random_matrix.cpp:13-14, armv7/random_matrix.c:15,
aarch64/random_matrix.cpp:16) and the expected C by the reference's oracle
(armv7/REF_MMult.c compiled unmodified: libref_armv7.so = `-O2`, unfused;
libref_armv7_fma.so = `-O2 -mfma -ffp-contract=fast`, the aarch64 `-march=native`
behaviour).  Nothing from oracle/oracle_mmult.c is involved, so the fixtures
pin BOTH our restatement and the GPU kernels to the reference itself.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import oracle as O  # noqa: E402  (only for loading the _ref objects)

# name, m, n, k, lda, ldb, ldc, seed, pattern
CASES = [
    ("armv7_first_40", 40, 40, 40, 40, 40, 40, 20260101, None),     # armv7/parameters.h PFIRST
    ("aarch64_first_48", 48, 48, 48, 48, 48, 48, 20260102, None),   # aarch64/parameters.h PFIRST
    ("ragged_96x80x72", 96, 80, 72, 72, 80, 80, 20260103, None),
    ("square_128", 128, 128, 128, 128, 128, 128, 20260104, None),
    ("padded_ld_200x136x264", 200, 136, 264, 272, 144, 160, 20260105, None),
    ("one_tile_k_tail_128x128x50", 128, 128, 50, 50, 128, 128, 20260106, None),
    ("tiny_1x1x1", 1, 1, 1, 1, 1, 1, 20260107, None),
    ("row_vector_1x130x33", 1, 130, 33, 33, 130, 130, 20260108, None),
    ("pattern_mod3_64", 64, 64, 64, 64, 64, 64, None, 3),            # cuda/random_matrix.cpp:13-14
    ("pattern_mod2_72", 72, 72, 72, 72, 72, 72, None, 2),            # armv7/random_matrix.c:15
    ("ones_48", 48, 48, 48, 48, 48, 48, None, 0),                    # aarch64/random_matrix.cpp:16
    ("accumulate_64", 64, 64, 64, 64, 64, 64, 20260109, None),       # C pre-loaded, C = A*B + C
]


def ref_random(lib_cuda, rows, cols, ld, seed):
    """Row-major (rows x cols) logical matrix with leading dim ld, values drawn
    by the reference generator in ITS order (it fills a[j*lda+i] column-major
    for an (m x n) call; the harness then reads the buffer row-major --
    cuda/test_MMult.cpp:77-79 -- so we call it as the harness does on a dense
    buffer and then embed into the padded ld)."""
    dense = np.zeros(rows * cols, dtype=np.float32)
    lib_cuda.random(rows, cols, dense, rows)
    out = np.zeros((rows, ld), dtype=np.float32)
    out[:, :cols] = dense.reshape(rows, cols)
    return out


def pattern(rows, cols, ld, mod):
    # the known-answer generators, restated with numpy integer ops:
    # A(i,j) = (j-i) % mod with C remainder semantics; stored column-major
    # a[j*rows+i] and then read row-major, like ref_random above.
    i = np.arange(rows)[:, None]
    j = np.arange(cols)[None, :]
    vals = np.ones((rows, cols)) if mod == 0 else np.fmod(j - i, mod)
    dense = np.zeros(rows * cols, dtype=np.float32)
    dense[(j * rows + i).ravel()] = vals.astype(np.float32).ravel()
    out = np.zeros((rows, ld), dtype=np.float32)
    out[:, :cols] = dense.reshape(rows, cols)
    return out


def main():
    assert O.have_ref(), "run `make -C oracle ref` first"
    cu = O.reflib("cuda_utils")
    std, fma = O.reflib("armv7"), O.reflib("armv7_fma")
    libc = C.CDLL(None)
    libc.srand48.argtypes = [C.c_long]
    for name, m, n, k, lda, ldb, ldc, seed, pat in CASES:
        if pat is None:
            libc.srand48(seed)
            a = ref_random(cu, m, k, lda, seed)
            b = ref_random(cu, k, n, ldb, seed)
        else:
            a, b = pattern(m, k, lda, pat), pattern(k, n, ldb, pat)
        c0 = np.zeros((m, ldc), dtype=np.float32)
        if name.startswith("accumulate"):
            libc.srand48(seed + 1)
            c0 = ref_random(cu, m, n, ldc, seed + 1)
        c_std, c_fma = c0.copy(), c0.copy()
        std.REF_MMult(m, n, k, a, lda, b, ldb, c_std, ldc)
        fma.REF_MMult(m, n, k, a, lda, b, ldb, c_fma, ldc)
        # the reference's own MY_MMult (armv7/MMult0.c) and compare_matrices
        c_my = c0.copy()
        std.MY_MMult(m, n, k, a, lda, b, ldb, c_my, ldc)
        diff_self = std.compare_matrices(m, n, c_my, ldc, c_std, ldc)
        diff_fma = cu.compare(m, n, c_fma, ldc, c_std, ldc)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), m=m, n=n, k=k, lda=lda, ldb=ldb,
                            ldc=ldc, a=a, b=b, c0=c0, c_ref=c_std, c_ref_fma=c_fma,
                            diff_mmult0_vs_ref=np.float32(diff_self),
                            diff_fma_vs_ref=np.float32(diff_fma))
        print(f"{name}: diff(MMult0,REF)={diff_self:g} diff(fma,REF)={diff_fma:g}")

    # output format golden: the reference's own armv7 driver + MMult0
    exe = os.path.join(O.REF_DIR, "test_MMult_armv7.x")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600).stdout
    lines = out.splitlines()
    rows = [ln.split() for ln in lines[1:-1]]
    with open(os.path.join(HERE, "armv7_driver_output.txt"), "w") as f:
        f.write("# stdout of the reference's armv7/test_MMult.c + MMult0.c, gflops column masked\n")
        f.write(lines[0] + "\n")
        for r in rows:
            f.write(f"{r[0]} <gflops> {r[2]} \n")
        f.write(lines[-1] + "\n")
    print("driver rows:", len(rows))


if __name__ == "__main__":
    main()
