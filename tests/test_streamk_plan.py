"""Host logic of the phase-ordered stream-K launch (mmh_streamk_plan = the tables sk_tables_for uploads),
checked without a GPU against a Python restatement of streamk_body's range arithmetic
(how-to-optimize-gemm_amd/csrc/sgemm_mfma.hpp): whatever range a chip position takes and whatever tile a
slot holds, every (tile, K-slice) unit must be computed exactly once, each tile's parts must tile [0, nk)
in ascending order (the chain), and the tables must really sort by phase."""
import random

import numpy as np
import pytest

import how_to_optimize_gemm_amd as H


def _library_loads():
    """mmh_streamk_plan is host arithmetic, but it lives in libmmult_hip.so (a hipcc build that needs the HIP
    runtime to LOAD): on a machine without the library or without libamdhip64 these tests skip, not error."""
    try:
        H.lib()
        return True
    except (OSError, H.MMultError):
        return False


pytestmark = pytest.mark.skipif(not _library_loads(), reason="libmmult_hip.so (or the HIP runtime it links) is not loadable here")


def ranges(tiles, nk, grid):
    total = tiles * nk
    return [total * r // grid for r in range(grid + 1)]


def segments_of(q, S, nk):
    """What workgroup `q` computes, in its execution order: (slot, kb, ke) -- streamk_body's run() calls."""
    u0, u1 = S[q], S[q + 1]
    if u1 <= u0:
        return []
    t_first, k_first = divmod(u0, nk)
    t_last = (u1 - 1) // nk
    k_last_end = u1 - t_last * nk
    if t_first == t_last:
        return [(t_first, k_first, k_last_end)]
    first_partial, last_partial = k_first != 0, k_last_end != nk
    out = []
    if last_partial:
        out.append((t_last, 0, k_last_end))                      # 1. head of the last tile
    for t in range(t_first + (1 if first_partial else 0), t_last - (1 if last_partial else 0) + 1):
        out.append((t, 0, nk))                                   # 2. whole tiles
    if first_partial:
        out.append((t_first, k_first, nk))                       # 3. rest of the first tile
    return out


SHAPES = [(1058, 92, 512), (2312, 8, 512), (1600, 16, 768), (552, 12, 256), (961, 7, 512), (784, 3, 256),
          (512, 64, 256), (4097, 5, 768), (300, 128, 256), (1000, 1, 256)]


@pytest.mark.parametrize("tiles,nk,grid", SHAPES)
def test_tables_are_bijections_and_every_unit_is_computed_once_in_chain_order(tiles, nk, grid):
    order, place = H.streamk_plan(tiles, nk, grid)
    assert sorted(order.tolist()) == list(range(grid))
    assert sorted(place.tolist()) == list(range(tiles))
    S = ranges(tiles, nk, grid)
    parts = {}                                                   # actual tile -> [(kb, ke, range)]
    for rho in range(grid):
        q = int(order[rho])
        for (slot, kb, ke) in segments_of(q, S, nk):
            assert 0 <= kb < ke <= nk
            parts.setdefault(int(place[slot]), []).append((kb, ke, q))
    assert sorted(parts) == list(range(tiles))
    for t, segs in parts.items():
        segs.sort()
        assert segs[0][0] == 0 and segs[-1][1] == nk
        for (a, b) in zip(segs, segs[1:]):
            assert a[1] == b[0]                                  # contiguous, ascending: one chain
            assert b[2] == a[2] + 1                              # continued by the NEXT range (its slot q - 1 / q)


@pytest.mark.parametrize("tiles,nk,grid", SHAPES)
def test_order_is_by_phase_and_levels_are_dealt_in_that_order(tiles, nk, grid):
    order, place = H.streamk_plan(tiles, nk, grid)
    S = ranges(tiles, nk, grid)
    phase = [S[int(q) + 1] % nk for q in order]                  # head length of the range at each chip position
    assert phase == sorted(phase)
    first = [-(-S[r] // nk) for r in range(grid)] + [tiles]      # first slot each range owns
    have0 = [int(q) for q in order if first[int(q) + 1] > first[int(q)]]
    # level 0 (the first slot of every range that owns one) occupies tiles 0 .. len-1 in phase order
    assert [int(place[first[q]]) for q in have0] == list(range(len(have0)))


def test_random_shapes():
    rng = random.Random(5)
    for _ in range(60):
        grid = rng.choice([8, 64, 256, 512, 768])
        tiles = rng.randint(grid, 6 * grid)
        nk = rng.randint(1, 140)
        order, place = H.streamk_plan(tiles, nk, grid)
        assert sorted(order.tolist()) == list(range(grid)) and sorted(place.tolist()) == list(range(tiles))
        S = ranges(tiles, nk, grid)
        seen = np.zeros((tiles, nk), dtype=np.int32)
        for rho in range(grid):
            for (slot, kb, ke) in segments_of(int(order[rho]), S, nk):
                seen[int(place[slot]), kb:ke] += 1
        assert (seen == 1).all()


def test_invalid_arguments():
    with pytest.raises(H.MMultError):
        H.streamk_plan(100, 8, 256)        # fewer tiles than workgroups: not a stream-K launch
    with pytest.raises(H.MMultError):
        H.streamk_plan(512, 0, 256)
