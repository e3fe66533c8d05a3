"""Registers, spills and scratch of the kernels in the built product library, read from its embedded code objects on the
CPU (tools/kernel_resources.py): a hot kernel that starts to spill, or outgrows the register budget its co-residency
needs, is a slow number on the GPU box and nothing else -- round 4's VALU rung went to 512 registers and 139 spilled
ones on the way to its final form.  (The reference has no analogue: nvcc's -Xptxas -v output is not checked anywhere.)"""
import os
import re
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
LIB = os.path.join(REPO, "how-to-optimize-gemm_amd", "libmmult_hip.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="libmmult_hip.so has not been built")


def _rows():
    import kernel_resources as K
    return K.resources(LIB)


def _alloc(vgpr):
    return (vgpr + 7) // 8 * 8          # gfx950 allocates vector registers in blocks of eight


def _workgroups_per_cu(vgpr, threads):
    """By registers alone: 512 per SIMD lane, at most 8 waves per SIMD, a workgroup's waves dealt over the four SIMDs."""
    per_simd = min(8, 512 // max(_alloc(vgpr), 1))
    return (4 * per_simd) // (threads // 64)


def test_every_code_object_is_readable_and_gfx950_only():
    rows = _rows()
    assert len(rows) >= 80, len(rows)
    names = {r["kernel"].split("<")[0] for r in rows}
    for want in ("sgemm_mfma_dma5_kernel", "sgemm_dma5_streamk_kernel", "sgemm_mfma_kernel", "sgemm_valu_kernel", "sgemm_naive_kernel"):
        assert want in names, want


def test_no_fp32_gemm_kernel_spills_except_the_known_256x256_forms():
    """The guarded and the stream-K instantiations of the 256x256 register-staged tile live at the 256-register cap of a
    512-thread workgroup and spill a few dozen registers around their epilogues (rounds 2-3, measured: not in the loop);
    everything else -- K1, K2, K2L, K2W in every instantiation -- holds its state in registers."""
    known = re.compile(r"sgemm_mfma(_streamk)?_kernel<256,256,")
    for r in _rows():
        if not r["kernel"].startswith("sgemm_"):
            continue
        if known.match(r["kernel"]):
            assert r["vgpr_spill"] <= 200 and r["scratch"] <= 320, r
            continue
        if r["kernel"].startswith("sgemm_valu_dma5_kernel<128,128,"):
            # K1W's 128x128 tile is held to 168 registers (three waves per SIMD: two workgroups of six waves per CU) and
            # parks four of them around its prologue / C store, outside the K loop (round 5)
            assert r["vgpr_spill"] <= 8 and r["scratch"] <= 32 and r["sgpr_spill"] == 0, r
            continue
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0, r
        # scalars parked in a vector register's lanes (a v_readlane to get one back): none in any one-workgroup-per-tile
        # kernel; the persistent stream-K bodies carry a range's bookkeeping beside a segment's -- 2 to 49 as the round
        # ends (outside the K loop; a lever nobody has pulled yet)
        # (round 6: the vector-ALU consumer under the same body parks 78 on its 128x128 tile -- the thread tile's address
        # constants are scalars too)
        cap = 96 if r["kernel"].startswith("sgemm_valu_dma5_streamk_kernel") else 64 if "streamk" in r["kernel"] else 0
        assert r["sgpr_spill"] <= cap, r


def test_the_k2w_tiles_fit_the_co_residency_their_launches_count_on():
    """Plain launches: three 64x64 workgroups per CU, two 128x64 / 96x96, one 128x128 (csrc/policy_table.inc's w).
    Stream-K grids are bounded, for every launch of a tile, by its GUARDED CHAINED instantiation (csrc/launch_dma5.hip):
    116 registers on the 64x64 tile = TWO workgroups per CU where three rings fit, 165 on the 128x64 tile = ONE where two
    fit (mmh_auto_plan reports grids accordingly).  The whole-tile instantiations (77 / 117 registers) would run three and
    two: bounding a whole-tile launch by its own instantiation is a lever the round found on the CPU, at its end, and
    left for the next one."""
    rows = {r["kernel"]: r for r in _rows()}
    plain = {"64,64,32,2,2,3": 3, "128,64,32,4,2,3": 2, "128,128,32,4,4,3": 1, "96,96,32,3,3,3": 2, "96,64,32,3,2,3": 2,
             "160,160,32,5,5,3": 1}
    seen = 0
    for name, r in rows.items():
        m = re.match(r"sgemm_mfma_dma5_kernel<(\d+,\d+,32,\d,\d,3),", name)
        if m:
            assert _workgroups_per_cu(r["vgpr"], r["threads"]) >= plain[m.group(1)], r
            seen += 1
    assert seen == 12, seen
    guarded = {"64,64,32,2,2,3": 2, "128,64,32,4,2,3": 1, "128,128,32,4,4,3": 1}
    whole = {"64,64,32,2,2,3": 3, "128,64,32,4,2,3": 2, "128,128,32,4,4,3": 1}
    for tile in guarded:
        # (the trailing 1: RS, the fragment reads spread behind the k-step's first MFMAs -- round 6)
        g = rows[f"sgemm_dma5_streamk_kernel<{tile},true,true,{'2,2' if tile.startswith('64') else '4,2'},1>"]
        w = rows[f"sgemm_dma5_streamk_kernel<{tile},false,true,{'2,2' if tile.startswith('64') else '4,2'},1>"]
        assert _workgroups_per_cu(g["vgpr"], g["threads"]) == guarded[tile], g
        assert _workgroups_per_cu(w["vgpr"], w["threads"]) >= whole[tile], w
    # ... and what mmh_auto_plan reports for a persistent launch is a grid the launcher can have
    import how_to_optimize_gemm_amd as H
    for (m, n, k) in [(3329, 3329, 3329), (2303, 2303, 2303), (1664, 1664, 1664), (3000, 3000, 3000), (1792, 1792, 1792), (5000, 3000, 777)]:
        name, tiles, grid = H.auto_plan(m, n, k)
        if grid > 0 and name in ("mfma_64x64_dma5", "mfma_128x64_dma5", "mfma_128x128_dma5"):
            assert grid // 256 <= {"mfma_64x64_dma5": 2, "mfma_128x64_dma5": 1, "mfma_128x128_dma5": 1}[name], (m, n, k, name, grid)


def test_the_valu_rung_keeps_its_waves():
    """K1: three waves per SIMD for the 64x64 tile (at most 168 registers; 132-142 with four k-steps of look-ahead), two
    for the 128x128 tile (at most 256), no accumulator registers and no spills in any look-ahead instantiation."""
    n = 0
    for r in _rows():
        m = re.match(r"sgemm_valu_kernel<(\d+),(\d+),", r["kernel"])
        if not m:
            continue
        n += 1
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0 and r["agpr"] == 0, r
        assert _alloc(r["vgpr"]) <= (168 if m.group(1) == "64" else 256), r
    assert n == 4, n      # (round 5: one look-ahead per tile, guarded and not; the A/B instantiations left with K1W's arrival)


def test_k1w_fits_the_co_residency_its_launches_count_on():
    """K1W (csrc/sgemm_valu_dma5.hpp): four FMA waves + two loader waves per workgroup.  The 128x128 tile (64 KiB ring) and
    the 128x64 tile (72 KiB) run TWO workgroups per CU -- 12 waves, three per SIMD: at most 168 registers; the 64x64 tile
    (48 KiB) three -- 18 waves, five per SIMD on two of them: at most 96."""
    want = {"128,128": 2, "128,64": 2, "64,64": 3}
    seen = set()
    for r in _rows():
        m = re.match(r"sgemm_valu_dma5_kernel<(\d+,\d+),", r["kernel"])
        if not m:
            continue
        seen.add(m.group(1))
        assert r["agpr"] == 0 and r["threads"] == 384, r
        assert _workgroups_per_cu(r["vgpr"], r["threads"]) >= want[m.group(1)], r
    assert seen == set(want), seen


def test_the_cost_tables_stream_k_residency_is_one_the_binary_allows():
    """MMH_KERNEL_AUTO priced stream-K launches from the plain co-residency w (three 64x64 workgroups per CU)
    while launch_streamk bounded the grid by the guarded chained instantiation's registers (two): `tiles % (w' CUs)` meant
    different things in the two places.  Round 5: policy_table.inc carries `skw` per family, the plan hands it to the
    launcher (GemmArgs::sk_w, an upper bound there), and this test holds it to what the binary's registers and the LDS
    ring allow for the instantiation that bounds the grid."""
    text = open(os.path.join(REPO, "how-to-optimize-gemm_amd", "csrc", "policy_table.inc")).read()
    fams = re.findall(r"\{(MMH_KERNEL_\w+), (\d+), (\d+), (\d+), (\d+), (\d+),", text)
    assert len(fams) == 10, fams
    rows = {r["kernel"]: r for r in _rows()}
    bound = {   # the instantiation launch_*.hip passes as `occ_kern`, and its LDS request in KiB
        "MMH_KERNEL_MFMA_64X64_DMA5": ("sgemm_dma5_streamk_kernel<64,64,32,2,2,3,true,true,2,2,1>", 48),
        "MMH_KERNEL_MFMA_128X64_DMA5": ("sgemm_dma5_streamk_kernel<128,64,32,4,2,3,true,true,4,2,1>", 72),
        "MMH_KERNEL_MFMA_128X128_DMA5": ("sgemm_dma5_streamk_kernel<128,128,32,4,4,3,true,true,4,2,1>", 96),
        "MMH_KERNEL_MFMA_64X64_DMA": ("sgemm_dma_streamk_kernel<64,64,32,2,2,3,true>", 48),
        "MMH_KERNEL_MFMA_128X64_DMA": ("sgemm_dma_streamk_kernel<128,64,32,4,2,3,true>", 72),
        "MMH_KERNEL_MFMA_128X128_DMA": ("sgemm_dma_streamk_kernel<128,128,32,4,4,3,true>", 96),
        "MMH_KERNEL_MFMA_256X256": ("sgemm_mfma_streamk_kernel<256,256,true,4,8,32>", 128),
    }
    for macro, bm, bn, w, has_sk, skw in fams:
        w, has_sk, skw = int(w), int(has_sk), int(skw)
        if not has_sk:
            assert skw == 0, (macro, skw)
            continue
        name, lds_kib = bound[macro]
        r = rows[name]
        fits = min(_workgroups_per_cu(r["vgpr"], r["threads"]), 160 // lds_kib)
        assert 1 <= skw <= min(w, fits), (macro, skw, w, fits, r)
