"""CPU tests that PIN the oracle (oracle/oracle_mmult.c) to the reference:
against the committed golden fixtures (produced by the reference's own
compiled sources, tests/golden/make_golden.py), against those compiled
objects directly when oracle/_ref exists, and against the reference's
known-answer input patterns."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden


def _views(g):
    a = g["a"][:, :g["k"]]
    b = g["b"][:, :g["n"]]
    return a, b


@pytest.mark.parametrize("name", golden_cases())
@pytest.mark.parametrize("fast", [False, True])
def test_restatement_matches_reference_golden(oracle, name, fast):
    g = load_golden(name)
    a, b = _views(g)
    for fma, key in ((False, "c_ref"), (True, "c_ref_fma")):
        c = g["c0"].copy()
        cv = c[:, :g["n"]]
        oracle.ref_mmult(a, b, cv, fma=fma, fast=fast)
        assert np.array_equal(c, g[key]), f"{name}: oracle (fma={fma}, fast={fast}) != reference"


@pytest.mark.parametrize("name", golden_cases())
def test_compare_matrices_matches_reference_golden(oracle, name):
    g = load_golden(name)
    n = g["n"]
    d, bad = oracle.compare_matrices(g["c_ref_fma"][:, :n], g["c_ref"][:, :n])
    assert np.float32(d) == g["diff_fma_vs_ref"]
    assert bad == (-1, -1)
    # the reference's own MY_MMult (MMult0) reproduced its REF exactly
    assert g["diff_mmult0_vs_ref"] == 0.0


def test_known_answer_patterns(oracle):
    # all-ones inputs: every C(i,j) == k exactly (aarch64/random_matrix.cpp:16)
    g = load_golden("ones_48")
    assert np.all(g["c_ref"][:, :48] == 48.0)
    # integer patterns: fp32 sums are exact, fused == unfused == int64 matmul
    for name in ("pattern_mod3_64", "pattern_mod2_72"):
        g = load_golden(name)
        a, b = _views(g)
        exact = (a.astype(np.int64) @ b.astype(np.int64)).astype(np.float32)
        assert np.array_equal(g["c_ref"][:, :g["n"]], exact)
        assert np.array_equal(g["c_ref_fma"][:, :g["n"]], exact)
        assert np.array_equal(oracle.ref_mmult(a, b, fma=True), exact)


def test_pattern_generator_matches_fixture(oracle):
    for name, mod in (("pattern_mod3_64", 3), ("pattern_mod2_72", 2), ("ones_48", 0)):
        g = load_golden(name)
        m, k = g["m"], g["k"]
        buf = oracle.random_matrix(m, k, lda=m, pattern=mod)
        assert np.array_equal(buf.reshape(m, k), g["a"][:, :k])
    # C remainder semantics: (j-i)%3 is negative below the diagonal
    buf = oracle.random_matrix(4, 4, pattern=3)
    assert buf.min() == -2.0 and buf.max() == 2.0


def test_compare_matrices_reports_first_bad(oracle):
    a = np.zeros((5, 7), dtype=np.float32)
    b = a.copy()
    b[2, 3] = 0.4
    b[3, 1] = -0.75
    b[4, 6] = 3.0
    d, bad = oracle.compare_matrices(a, b)
    assert d == 3.0 and bad == (3, 1)


def test_fast_forms_are_bit_identical_and_thread_invariant(oracle):
    a, b = oracle.harness_inputs(70, 90, 110, seed=424242)
    for fma in (False, True):
        lit = oracle.ref_mmult(a, b, fma=fma, fast=False)
        for nt in (1, 3, 8):
            assert np.array_equal(lit, oracle.ref_mmult(a, b, fma=fma, fast=True, nthreads=nt))
    # fused and unfused differ (else the distinction would be vacuous) but only by rounding
    c0, c1 = oracle.ref_mmult(a, b, fma=False), oracle.ref_mmult(a, b, fma=True)
    assert not np.array_equal(c0, c1)
    assert np.abs(c0 - c1).max() < 1e-4
    c64 = oracle.ref_mmult_f64(a, b)
    assert np.abs(c1 - c64).max() < 5e-5


def test_seeded_inputs_reproducible_and_in_range(oracle):
    a1, b1 = oracle.harness_inputs(33, 17, 29, seed=7)
    a2, b2 = oracle.harness_inputs(33, 17, 29, seed=7)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2)
    assert a1.min() >= -1.0 and a1.max() < 1.0
    assert a1.shape == (33, 29) and b1.shape == (29, 17)


def test_against_compiled_reference_objects(oracle):
    """Direct differential test against the reference's own object code."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    cu = oracle.reflib("cuda_utils")
    # input generator: same drand48 stream, same column-major placement
    oracle.lib().orc_srand48(99)
    ref = np.zeros(60 * 44, dtype=np.float32)
    cu.random(60, 44, ref, 60)
    ours = oracle.random_matrix(60, 44, seed=99)
    assert np.array_equal(ref, ours)
    # REF_MMult, 9-arg with padded leading dimensions, fused and unfused builds
    a = np.zeros((37, 64), dtype=np.float32)
    b = np.zeros((53, 48), dtype=np.float32)
    rng = np.random.default_rng(5)
    a[:, :53] = rng.uniform(-1, 1, (37, 53)).astype(np.float32)
    b[:, :41] = rng.uniform(-1, 1, (53, 41)).astype(np.float32)
    for fma in (False, True):
        want = oracle.reference_ref_mmult(a[:, :53], b[:, :41], fma=fma)
        got = oracle.ref_mmult(a[:, :53], b[:, :41], fma=fma, fast=True)
        assert np.array_equal(want, got)
    # 6-arg aarch64/REF_MMult.cpp == 9-arg with ld = row length
    a6 = np.ascontiguousarray(a[:, :53])
    b6 = np.ascontiguousarray(b[:, :41])
    c6 = np.zeros((37, 41), dtype=np.float32)
    cu.ref6(37, 41, 53, a6, b6, c6)
    assert np.array_equal(c6, oracle.ref_mmult(a6, b6, fma=False))
    # compare_matrices
    x = rng.uniform(-1, 1, (20, 30)).astype(np.float32)
    # differences below 0.5: above it the reference's compare_matrices also prints the element
    # (cuda/compare_matrices.cpp:18-24), straight to the process's stdout
    y = (x + rng.uniform(-0.3, 0.3, (20, 30))).astype(np.float32)
    assert np.float32(cu.compare(20, 30, x, 30, y, 30)) == np.float32(oracle.compare_matrices(x, y)[0])


def test_int8_oracle(oracle):
    rng = np.random.default_rng(11)
    a = rng.integers(-127, 128, (45, 77), dtype=np.int8)
    b = rng.integers(-127, 128, (77, 51), dtype=np.int8)
    want = a.astype(np.int64) @ b.astype(np.int64)
    got = oracle.ref_igemm_s8(a, b)
    assert np.array_equal(got, want.astype(np.int32))
    # symmetric quantisation never produces -128 and hits +-127 at the extreme
    x = rng.uniform(-3, 3, (16, 16)).astype(np.float32)
    q, s = oracle.quantize_sym_s8(x)
    assert q.min() >= -127 and abs(q).max() == 127
    assert np.abs(q / s - x).max() <= 0.5 / s + 1e-6


def test_driver_output_format_fixture():
    """The reference driver's stdout contract (cuda/test_MMult.cpp:41,128,144;
    parsed by cuda/plot.py:5-28): header, 'p gflops diff ' rows, '];'."""
    import os
    from conftest import GOLDEN_DIR
    lines = open(os.path.join(GOLDEN_DIR, "armv7_driver_output.txt")).read().splitlines()[1:]
    assert lines[0] == "MY_MMult = [" and lines[-1] == "];"
    ps = [int(ln.split()[0]) for ln in lines[1:-1]]
    assert ps == list(range(40, 701, 40))                     # armv7/parameters.h:5-7
    assert all(float(ln.split()[2]) == 0.0 for ln in lines[1:-1])
