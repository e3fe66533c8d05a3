"""The families that live in the TOOLS build only (libmmult_hip_ab.so, -DMMH_AB_BUILD) -- kept for their measurements,
not shipped: the rim (round 3), the 32x32x2 LDS-DMA tiles and the one-loader / 160-wide forms of K2W (round 4), the
16-MFMA-per-phase int8 ping-pong.  Their results are still held to the oracle's bits here, in a process of their own
that loads the tools library -- when that library has been built (`python -c "import how_to_optimize_gemm_amd as H;
H.build.build_ab_library()"`, minutes); the product library's side -- it rejects every one of these ids and switches --
is tests/test_abi.py and the last test below."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB_LIB = os.path.join(REPO, "how-to-optimize-gemm_amd", "libmmult_hip_ab.so")
needs_ab = pytest.mark.skipif(not os.path.exists(AB_LIB), reason="libmmult_hip_ab.so (the tools build) has not been built")


def _run(script, timeout=900):
    r = subprocess.run([sys.executable, "-c", script, REPO], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and "tools-build ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


_HEAD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch
import how_to_optimize_gemm_amd as H
from oracle import oracle
H.use_ab_library(build=False)
assert H.lib().mmh_is_ab_build() == 1
def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()
"""

_RIM = _HEAD + r"""
RIM_SHAPES = [(1025, 1025, 1025), (1024, 1027, 300), (1031, 1024, 77), (1032, 1032, 64), (2049, 2049, 129),
              (1288, 1025, 511), (3073, 1026, 96), (257, 4097, 200), (1025, 1281, 257), (1153, 1025, 190)]
mm = H.MMult(0, "auto")
assert mm.get_option(H.OPT_RIM) == 0
rims = 0
for i, (m, n, k) in enumerate(RIM_SHAPES):
    a, b = oracle.harness_inputs(m, n, k, seed=m + 7 * n + k)
    lda, ldb, ldc = k + (i % 3), n + (i % 2) * 3, n + ((i + 1) % 2) * 5
    abuf = torch.full((m * lda + 9,), float("nan"), device="cuda")
    bbuf = torch.full((k * ldb + 9,), float("nan"), device="cuda")
    cbuf = torch.full((m * ldc + 9,), float("nan"), device="cuda")
    off = i % 2
    av = abuf[off:off + m * lda].view(m, lda)
    bv = bbuf[off:off + k * ldb].view(k, ldb)
    cv = cbuf[off:off + m * ldc].view(m, ldc)
    av[:, :k] = dev(a)
    bv[:, :n] = dev(b)
    c0 = np.random.default_rng(i).uniform(-1, 1, (m, n)).astype(np.float32)
    for accumulate in (False, True):
        want = oracle.ref_mmult(a, b, c0.copy() if accumulate else None, fma=True)
        for rim in (8, 0):
            mm.set_option(H.OPT_RIM, rim)
            cv[:, :n] = dev(c0)
            mm.sgemm(m, n, k, av.data_ptr(), lda, bv.data_ptr(), ldb, cv.data_ptr(), ldc, accumulate,
                     torch.cuda.current_stream().cuda_stream)
            launched = H.last_launch()
            if rim == 0:
                assert "rim" not in launched, launched
            rims += "on the rim" in launched
            got = cv[:, :n].cpu().numpy()
            assert np.array_equal(got, want), (m, n, k, accumulate, launched)
            if ldc > n:
                assert torch.isnan(cv[:, n:]).all(), (m, n, k)
            assert torch.isnan(cbuf[:off]).all() and torch.isnan(cbuf[off + m * ldc:]).all()
assert rims >= 8, rims          # the small shapes trim onto a one-round plain launch of the 64x64 tile
mm.set_option(H.OPT_RIM, 0)
try:
    mm.set_option(H.OPT_RIM, 17)
    raise SystemExit("MMH_OPT_RIM = 17 accepted")
except H.MMultError:
    pass
# the int8 rungs that left the product in round 6 (tools/ab/igemm_s8_k3.hpp): K3 (mode 1), the packed-B kernel (3 / 4)
rng = np.random.default_rng(7)
for mode in (1, 3, 4):
    mm.set_igemm_mode(mode)
    for (m, n, k) in [(256, 256, 128), (512, 768, 640), (257, 255, 129), (100, 92, 72), (1280, 1024, 384)]:
        qa = rng.integers(-127, 128, (m, k), dtype=np.int8)
        qb = rng.integers(-127, 128, (k, n), dtype=np.int8)
        assert np.array_equal(mm.igemm_s8(dev(qa), dev(qb)).cpu().numpy(), oracle.ref_igemm_s8(qa, qb)), (mode, m, n, k)
mm.set_igemm_mode(0)
mm.close()
print("tools-build ok")
"""

_TILES = _HEAD + r"""
mm = H.MMult(0, "auto")
shapes = [(256, 256, 64), (384, 512, 192), (1024, 1024, 1024), (1152, 1152, 1152), (300, 259, 101), (1025, 1023, 257), (2304, 2304, 512)]
for name in ("mfma32_64x64_dma", "mfma32_128x64_dma", "mfma32_64x128_dma", "mfma32_128x128_dma", "mfma32b_128x64_dma",
             "mfma32b_64x128_dma", "mfma32b_128x128_dma", "exp5_64x64_l1d2", "exp5_128x64_l1d2", "exp5_128x128_l1d2",
             "exp5_160x96_l1d2", "exp5_160x160_l1d2",
             # round 6: the fragment reads as a block in front of the k-step's MFMAs (what rounds 4-6 shipped; RS = 0)
             "exp5_160x160_rs0", "exp5_128x128_rs0", "exp5_128x64_rs0", "exp5_64x64_rs0", "exp5_96x96_rs0"):
    mm.set_kernel(name)
    for sk in (1, 2):
        mm.set_streamk(sk)
        for (m, n, k) in shapes:
            a, b = oracle.harness_inputs(m, n, k, seed=m + n + k)
            got = mm.matmul(dev(a), dev(b)).cpu().numpy()
            assert np.array_equal(got, oracle.ref_mmult(a, b, fma=True)), (name, sk, m, n, k, H.last_launch())
mm.set_streamk(1)
mm.close()
print("tools-build ok")
"""

_TILES = _HEAD + r"""
mm = H.MMult(0, "auto")
shapes = [(256, 256, 64), (384, 512, 192), (1024, 1024, 1024), (1152, 1152, 1152), (300, 259, 101), (1025, 1023, 257), (2304, 2304, 512)]
for name in ("mfma32_64x64_dma", "mfma32_128x64_dma", "mfma32_64x128_dma", "mfma32_128x128_dma", "mfma32b_128x64_dma",
             "mfma32b_64x128_dma", "mfma32b_128x128_dma", "exp5_64x64_l1d2", "exp5_128x64_l1d2", "exp5_128x128_l1d2",
             "exp5_160x96_l1d2", "exp5_160x160_l1d2",
             # round 6: the fragment reads as a block in front of the k-step's MFMAs (what rounds 4-6 shipped; RS = 0)
             "exp5_160x160_rs0", "exp5_128x128_rs0", "exp5_128x64_rs0", "exp5_64x64_rs0", "exp5_96x96_rs0"):
    mm.set_kernel(name)
    for sk in (1, 2):
        mm.set_streamk(sk)
        for (m, n, k) in shapes:
            a, b = oracle.harness_inputs(m, n, k, seed=m + n + k)
            got = mm.matmul(dev(a), dev(b)).cpu().numpy()
            assert np.array_equal(got, oracle.ref_mmult(a, b, fma=True)), (name, sk, m, n, k, H.last_launch())
mm.set_streamk(1)
# the int8 ping-pong kernel with 16 MFMAs per phase (MMH_OPT_IGEMM_MODE = 7)
rng = np.random.default_rng(7)
mm.set_igemm_mode(7)
for (m, n, k) in [(256, 256, 128), (512, 768, 640), (257, 255, 129), (1280, 1024, 384)]:
    qa = rng.integers(-127, 128, (m, k), dtype=np.int8)
    qb = rng.integers(-127, 128, (k, n), dtype=np.int8)
    assert np.array_equal(mm.igemm_s8(dev(qa), dev(qb)).cpu().numpy(), oracle.ref_igemm_s8(qa, qb)), (m, n, k)
mm.set_igemm_mode(0)
mm.close()
print("tools-build ok")
"""


_RIM5 = _HEAD + r"""
RIM5_SHAPES = [(1025, 1025, 1025), (1024, 1025, 300), (1025, 1024, 77), (1025, 1090, 64), (2049, 2049, 129), (1030, 1025, 511),
               (65, 65, 40), (129, 64, 33), (64, 129, 31), (3073, 1025, 96), (257, 4097, 200), (1025, 1281, 257)]
mm = H.MMult(0, "mfma_64x64_dma5")
assert mm.get_option(H.OPT_RIM5) == 0
rims = 0
for i, (m, n, k) in enumerate(RIM5_SHAPES):
    a, b = oracle.harness_inputs(m, n, k, seed=m + 7 * n + k)
    lda, ldb, ldc = k + (i % 3), n + (i % 2) * 3, n + ((i + 1) % 2) * 5
    abuf = torch.full((m * lda + 9,), float("nan"), device="cuda")
    bbuf = torch.full(((k + 3) * ldb + 9,), float("nan"), device="cuda")
    cbuf = torch.full((m * ldc + 9,), float("nan"), device="cuda")
    off = i % 2
    av = abuf[off:off + m * lda].view(m, lda)
    bv = bbuf[off:off + k * ldb].view(k, ldb)
    cv = cbuf[off:off + m * ldc].view(m, ldc)
    av[:, :k] = dev(a)
    bv[:, :n] = dev(b)
    c0 = np.random.default_rng(i).uniform(-1, 1, (m, n)).astype(np.float32)
    for accumulate in (False, True):
        want = oracle.ref_mmult(a, b, c0.copy() if accumulate else None, fma=True)
        for rim in (1, 0):
            mm.set_option(H.OPT_RIM5, rim)
            mm.set_streamk(0)
            cv[:, :n] = dev(c0)
            mm.sgemm(m, n, k, av.data_ptr(), lda, bv.data_ptr(), ldb, cv.data_ptr(), ldc, accumulate,
                     torch.cuda.current_stream().cuda_stream)
            launched = H.last_launch()
            assert ("rim wave" in launched) == (rim == 1), launched
            rims += "rim wave" in launched
            got = cv[:, :n].cpu().numpy()
            assert np.array_equal(got, want), (m, n, k, accumulate, rim, launched)
            if ldc > n:
                assert torch.isnan(cv[:, n:]).all(), (m, n, k)
            assert torch.isnan(cbuf[:off]).all() and torch.isnan(cbuf[off + m * ldc:]).all()
assert rims == 2 * len(RIM5_SHAPES), rims
mm.close()
print("tools-build ok")
"""


@pytest.mark.gpu
@needs_ab
def test_the_fused_rim_keeps_the_tiles_bits():
    """MMH_OPT_RIM5 (sgemm_dma5.hpp, rim_wave; round 4, tools build): m and / or n ONE past a multiple of 64 run the
    64x64 K2W tiles of the TRIMMED shape; the last tile row / column's workgroups compute the rim in an extra wave on the
    vector ALU out of the K-slices in LDS.  Measured 2.2x slower per edge tile than a whole tile (the f32 MFMA runs on the
    vector ALU's FMA lanes) -- not shipped; the bits are the oracle's all the same."""
    _run(_RIM5)


@pytest.mark.gpu
@needs_ab
def test_the_rim_runs_on_the_vector_alu_with_the_tiles_bits():
    """MMH_OPT_RIM (sgemm_dma.hpp, "the rim"; round 3, measured slower than the edge tiles it replaces): shapes a few
    elements past a multiple of 64 run as the K2L tiles of the trimmed shape + extra vector-ALU workgroups for the strips
    beyond it, in one launch.  Every element -- tile or rim -- is the oracle's fused chain over ascending k: bit-equal to
    the oracle and to the same handle with the rim off, overwrite and accumulate, odd leading dimensions, NaN padding."""
    _run(_RIM)


@pytest.mark.gpu
@needs_ab
def test_the_tile_families_that_lost_keep_the_chains_bits():
    """The 32x32x2 tiles (K2M), K2W with ONE loader wave, the 160-wide whole-round tiles, int8 mode 7: measured,
    documented in profiles/r04_notes.md, not shipped -- and still the oracle's bits, plain and under forced stream-K."""
    _run(_TILES)


@pytest.mark.gpu
def test_the_product_library_refuses_the_tools_builds_switches(mm):
    import how_to_optimize_gemm_amd as H
    assert H.lib().mmh_is_ab_build() == 0
    for name in ("mfma32_64x64_dma", "mfma32b_128x128_dma", "exp5_64x64_l1d2", "exp5_160x160_l1d2"):
        with pytest.raises((KeyError, H.MMultError)):
            mm.set_kernel(name)
    for kid in (48, 49, 50, 51, 60, 61, 62, 64, 68, 72, 79, 80):
        with pytest.raises(H.MMultError):
            mm.set_kernel(kid)
    with pytest.raises(H.MMultError):
        mm.set_option(H.OPT_RIM, 8)
    mm.set_option(H.OPT_RIM, 0)
    with pytest.raises(H.MMultError):
        mm.set_option(H.OPT_RIM5, 1)
    mm.set_option(H.OPT_RIM5, 0)
    for mode in (1, 3, 4, 10):    # (K3 / packed-B rungs and the timing-only ablations: tools build)
        with pytest.raises(H.MMultError):
            mm.set_igemm_mode(mode)
    mm.set_igemm_mode(0)
    mm.set_kernel("auto")
