"""Round-6 GPU tests: the vector-ALU rung (BASELINE.json configs[1]) under K2W's persistent stream-K body, and the
policy that picks its tile AND launch form.  (The int8 tests of the round -- the persistent ping-pong kernel, the
config-named 16x16x32 instruction, the coalesced C stores -- sit with the other int8 tests in test_gpu_parity.py.)
All call through the C ABI (api.py is ctypes)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("kernel,bm,bn", [("valu_128x128", 128, 128), ("valu_128x64", 128, 64), ("valu_64x64", 64, 64)])
def test_k1w_under_stream_k_is_the_same_chain(mm, oracle, kernel, bm, bn):
    """K1Wp (csrc/sgemm_valu_dma5.hpp, Dma5ValuConsumer): a ragged tile count runs as ONE persistent launch whose
    workgroups each take tiles x K-slices / grid of the work -- a head (partial tile published write-through), whole
    tiles, a tail that resumes the previous range's head.  Partial sums travel as fp32 and every element's chain
    continues where the head stopped: the oracle's fused chain, bit for bit (the reference's non-tensor rungs,
    cuda/MMult_cuda_3.cu:10-53 ...).  Slice counts on every phase of the two- and three-deep rings, overwrite and
    accumulate, padded leading dimensions."""
    import torch
    import how_to_optimize_gemm_amd as H
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    side = int(np.ceil(np.sqrt(1.13 * cus)))                 # ~1.13 tiles per CU: 17 x 17 on 256 CUs
    mm.set_kernel(kernel)
    mm.set_streamk(2)
    try:
        for (tm, tn, k) in [(side, side, 32), (side, side, 96), (side, side + 1, 160), (side + 2, side, 224), (2 * side - 3, side, 64),
                            (side, side, 1024)]:
            m, n = tm * bm, tn * bn
            a, b = oracle.harness_inputs(m, n, k, seed=3 * tm + 5 * tn + k)
            got = mm.matmul(dev(a), dev(b)).cpu().numpy()
            launched = H.last_launch()
            if (tm * tn) % cus:
                assert "sgemm_valu_dma5_streamk_kernel" in launched and "persistent" in launched, (m, n, k, launched)
            want = oracle.ref_mmult(a, b, fma=True)
            assert np.array_equal(got, want), (kernel, m, n, k, float(np.abs(got - want).max()), launched)
            c0 = np.random.default_rng(k).uniform(-1, 1, (m, n)).astype(np.float32)
            out = dev(c0)
            mm.matmul(dev(a), dev(b), out=out, accumulate=True)
            assert np.array_equal(out.cpu().numpy(), oracle.ref_mmult(a, b, c0.copy(), fma=True)), (kernel, m, n, k)
            for rep in range(2):                                  # back to back: every hand-over word was put back to zero
                assert np.array_equal(mm.matmul(dev(a), dev(b)).cpu().numpy(), want), (kernel, m, n, k, rep)
        # padded leading dimensions, NaN in the padding and around C
        m, n, k = side * bm, side * bn, 128
        a, b = oracle.harness_inputs(m, n, k, seed=77)
        abuf = torch.full((m, k + 8), float("nan"), device="cuda")
        bbuf = torch.full((k, n + 4), float("nan"), device="cuda")
        cbuf = torch.full((m, n + 12), float("nan"), device="cuda")
        abuf[:, :k] = dev(a)
        bbuf[:, :n] = dev(b)
        mm.matmul(abuf[:, :k], bbuf[:, :n], out=cbuf[:, :n])
        assert "sgemm_valu_dma5_streamk_kernel" in H.last_launch(), H.last_launch()
        assert np.array_equal(cbuf[:, :n].cpu().numpy(), oracle.ref_mmult(a, b, fma=True))
        assert bool(torch.isnan(cbuf[:, n:]).all())
        # MMH_OPT_STREAMK = 0: the plain launch of the same tile, the same bits
        mm.set_streamk(0)
        mm.matmul(abuf[:, :k], bbuf[:, :n], out=cbuf[:, :n])
        assert "streamk" not in H.last_launch(), H.last_launch()
        assert np.array_equal(cbuf[:, :n].cpu().numpy(), oracle.ref_mmult(a, b, fma=True))
    finally:
        mm.set_streamk(1)


def test_the_valu_policy_picks_tile_and_launch_form(mm, oracle):
    """MMH_KERNEL_VALU (k1_pick_tile, launch_valu.hip): whole rounds stay plain launches of the largest tile that fills them;
    a ragged count of 128x128 tiles above one per CU runs persistent (N = 2176: 289 tiles for 256 CUs ran 56 TFLOP/s as a
    plain launch, 79 under stream-K); shapes that are not whole tiles stay on K1's guarded kernel."""
    import torch
    import how_to_optimize_gemm_amd as H
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the expectations below are the 256-CU chip's")
    mm.set_kernel("valu")
    for (n, k), want in [((2048, 64), "sgemm_valu_dma5_kernel<128,128>"), ((2176, 64), "sgemm_valu_dma5_streamk_kernel<128,128>"),
                         ((2560, 32), "sgemm_valu_dma5_streamk_kernel<128,128>"), ((1024, 96), "sgemm_valu_dma5_kernel<64,64>"),
                         ((1152, 64), "sgemm_valu_dma5_streamk_kernel<64,64>"), ((1100, 40), "sgemm_valu_kernel<")]:
        a, b = oracle.harness_inputs(n, n, k, seed=n + k)
        got = mm.matmul(dev(a), dev(b)).cpu().numpy()
        assert want in H.last_launch(), (n, k, H.last_launch())
        assert np.array_equal(got, oracle.ref_mmult(a, b, fma=True)), (n, k)
    mm.set_kernel("auto")
