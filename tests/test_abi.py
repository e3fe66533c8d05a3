"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every
symbol include/mmult_hip.h declares, and its host-side logic (shard plan,
error strings, argument validation before any device work) behaves.  No
compute is attempted here."""
import ctypes
import os
import re

import pytest

import how_to_optimize_gemm_amd as H
from conftest import REPO


def declared_symbols():
    text = open(os.path.join(REPO, "include", "mmult_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = H.lib()
    syms = declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/mmult_hip.h but not exported"
    assert set(syms) == set(H.EXPORTS)


def test_library_is_in_tree_and_has_gfx950_code_object():
    assert os.path.dirname(H.LIB_PATH) == os.path.join(REPO, "how-to-optimize-gemm_amd")
    blob = open(H.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"sgemm_mfma_kernel" in blob


def test_strerror_and_names():
    L = H.lib()
    assert L.mmh_strerror(0) == b"success"
    for code in range(-6, 0):
        assert L.mmh_strerror(code) not in (b"", b"unknown status")
    assert L.mmh_strerror(-99) == b"unknown status"
    assert H.kernel_name(H.KERNEL_MFMA) == "MMult_hip_mfma"
    assert H.kernel_name(H.KERNEL_VALU) == "MMult_hip_valu"
    assert H.kernel_name(77) is None
    assert L.mmh_version() >= 100


@pytest.mark.parametrize("m", [0, 1, 100, 128, 300, 1000, 4096, 16384, 16389])
@pytest.mark.parametrize("nranks", [1, 2, 3, 4, 8])
def test_shard_rows_partition(m, nranks):
    panels = [H.shard_rows(m, nranks, r) for r in range(nranks)]
    # contiguous, disjoint, cover [0, m)
    pos = 0
    for r0, rows in panels:
        assert rows >= 0
        if rows:
            assert r0 == pos
            pos += rows
    assert pos == m
    # every internal boundary is 128-row aligned
    for r0, rows in panels:
        if rows and r0 + rows != m:
            assert (r0 + rows) % 128 == 0
    # balance: whole tiles differ by at most one between ranks
    tiles = [rows // 128 for _, rows in panels]
    assert max(tiles) - min(tiles) <= 1
    if m == 16384 and nranks == 8:
        assert panels == [(2048 * r, 2048) for r in range(8)]   # BASELINE.json config 4


def test_shard_rows_rejects_bad_arguments():
    L = H.lib()
    r0, nr = ctypes.c_int(), ctypes.c_int()
    assert L.mmh_shard_rows(-1, 2, 0, ctypes.byref(r0), ctypes.byref(nr)) == H.ERR_INVALID_ARG
    assert L.mmh_shard_rows(10, 0, 0, ctypes.byref(r0), ctypes.byref(nr)) == H.ERR_INVALID_ARG
    assert L.mmh_shard_rows(10, 2, 2, ctypes.byref(r0), ctypes.byref(nr)) == H.ERR_INVALID_ARG
    assert L.mmh_shard_rows(10, 2, 0, None, ctypes.byref(nr)) == H.ERR_INVALID_ARG


def test_null_handle_is_an_error_not_a_crash():
    L = H.lib()
    assert L.mmh_sgemm(None, 1, 1, 1, None, 1, None, 1, None, 1, 0, None) == H.ERR_INVALID_ARG
    assert L.mmh_sgemm_host(None, 1, 1, 1, None, 1, None, 1, None, 1, 0) == H.ERR_INVALID_ARG
    assert L.mmh_sgemm_host_timed(None, 1, 1, 1, None, 1, None, 1, None, 1, 0, None) == H.ERR_INVALID_ARG
    assert L.mmh_destroy(None) == H.OK
    assert L.mmh_set_kernel(None, 2) == H.ERR_INVALID_ARG


def test_options_and_kernel_ids_validate_without_a_device():
    L = H.lib()
    v = ctypes.c_int(0)
    assert L.mmh_set_option(None, H.OPT_STREAMK, 1) == H.ERR_INVALID_ARG
    assert L.mmh_get_option(None, H.OPT_STREAMK, ctypes.byref(v)) == H.ERR_INVALID_ARG
    for name, kid in H.KERNELS.items():
        assert H.kernel_name(kid) is not None, name
    # the header's kernel ids and the Python mirror agree
    text = open(os.path.join(REPO, "include", "mmult_hip.h")).read()
    ids = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"#define MMH_KERNEL_(\w+) (\d+)", text)}
    for name, kid in ids.items():
        assert H.KERNELS[{"mfma_256": "mfma256"}.get(name, name)] == kid, name


def test_rccl_is_loadable_and_every_entry_point_the_shard_needs_resolves():
    """mmh_rccl_version drives the library's own RCCL loader (dlopen + dlsym of ncclGetVersion,
    ncclCommInitAll, ncclCommDestroy, ncclCommCount, ncclGroupStart/End, ncclBroadcast) without a GPU:
    MMH_ERR_UNSUPPORTED here would mean the multi-GPU shard cannot run on this image at all."""
    v = H.rccl_version()
    assert v >= 20000, v                       # NCCL-style version code, e.g. 22707
    L = H.lib()
    assert L.mmh_rccl_version(None) == H.ERR_INVALID_ARG


def test_shard_handle_argument_plumbing_without_a_device():
    """mmh_shard_create validates before it touches a device: bad counts, NULL out-pointer, and --
    on a box with fewer devices than asked for -- MMH_ERR_NO_DEVICE, never a smaller shard."""
    L = H.lib()
    h = ctypes.c_void_p()
    assert L.mmh_shard_create(None, 1, None) == H.ERR_INVALID_ARG
    assert L.mmh_shard_create(ctypes.byref(h), 0, None) == H.ERR_INVALID_ARG
    assert L.mmh_shard_create(ctypes.byref(h), 65, None) == H.ERR_INVALID_ARG
    n = H.device_count()
    assert L.mmh_shard_create(ctypes.byref(h), n + 1, None) == H.ERR_NO_DEVICE
    assert not h.value
    assert b"fewer visible devices" in L.mmh_last_error()
    assert L.mmh_shard_destroy(None) == H.OK
    assert L.mmh_shard_set_kernel(None, 0) == H.ERR_INVALID_ARG
    assert L.mmh_shard_info(None, None, None) == H.ERR_INVALID_ARG
    assert L.mmh_shard_sgemm(None, 1, 1, 1, None, 1, None, 1, None, 1, 1, None) == H.ERR_INVALID_ARG
    with pytest.raises(H.MMultError) as e:
        H.ShardedMMult(n + 1)
    assert e.value.status == H.ERR_NO_DEVICE
    # the one-shot form goes through the same handle path
    import numpy as np
    with pytest.raises(H.MMultError) as e:
        H.sgemm_sharded(n + 1, np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32))
    assert e.value.status == H.ERR_NO_DEVICE


def test_product_library_carries_no_ablation_builds():
    """The timing-only ablation kernels (wrong results) and scheduling A/B variants live in the
    tools-only libmmult_hip_ab.so: the product library neither names nor contains them."""
    L = H.lib()
    assert L.mmh_is_ab_build() == 0
    tools_only = (list(range(16, 20)) + list(range(21, 25)) + list(range(32, 48)) +     # scheduling A/Bs, ablations, 8-wave forms
                  list(range(48, 64)) + list(range(64, 96)))                              # 32x32x2 tiles (round 4), exp5_*, 160-wide tiles
    for kid in tools_only:
        assert H.kernel_name(kid) is None, kid
    blob = open(H.LIB_PATH, "rb").read()
    assert b"ablate" not in blob and b"cadence_" not in blob
    # round 4: the rim, the 32x32x2 tiles and the one-loader K2W forms left the product library too
    assert b"dma_rim_kernel" not in blob and b"sgemm_dma32" not in blob and b"exp5_" not in blob
    # every id the Python mirror offers is a product kernel
    assert sorted(H.KERNELS.values()) == sorted(set(H.KERNELS.values()))
    assert not set(tools_only) & set(H.KERNELS.values())


def test_no_device_fails_loudly_without_fallback():
    """On a box without a gfx950 GPU the product path must refuse, not
    compute on the CPU."""
    if H.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(H.MMultError) as e:
        H.MMult(0)
    assert e.value.status == H.ERR_NO_DEVICE
    import numpy as np
    with pytest.raises(H.MMultError):
        H.sgemm_sharded(1, np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: no file of the shipped package, the C
    ABI or the harness may reference it."""
    pkg = os.path.join(REPO, "how-to-optimize-gemm_amd")
    offenders = []
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".c")) or f in ("makefile", "Makefile"):
                p = os.path.join(root, f)
                text = open(p, errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|oracle/|liboracle|orc_", text):
                    offenders.append(p)
    assert offenders == []


def test_the_kernel_headers_carry_no_tools_build_preprocessor_switches():
    """the product's kernel headers hold what ships.  The tile families that were measured and lost
    live under tools/ab/ (compiled by build_ab_library() only); where an A/B switch rides in a kernel argument's spare
    bits the header tests `kAbBuild` with `if constexpr` -- no `#ifdef MMH_AB_BUILD` inside any csrc/*.hpp but the one
    in ab_build.hpp that defines the constant (its own header since round 6, included by every header that tests it:
    ADVICE r05)."""
    import glob
    csrc = os.path.join(REPO, "how-to-optimize-gemm_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*.hpp"))):
        text = open(path).read()
        n = sum(1 for line in text.splitlines() if line.lstrip().startswith("#if") and "MMH_AB_BUILD" in line)
        assert n == (1 if path.endswith("ab_build.hpp") else 0), (path, n)
        if "kAbBuild" in text and not path.endswith("ab_build.hpp"):
            assert '#include "ab_build.hpp"' in text or '#include "sgemm_tile.hpp"' in text or '#include "igemm_s8.hpp"' in text, path
    for name in ("sgemm_dma32.hpp", "launch_dma32.hip", "sgemm_dma_rim.hpp", "sgemm_dma5_rim.hpp"):
        assert os.path.exists(os.path.join(REPO, "tools", "ab", name)), name
        assert not os.path.exists(os.path.join(csrc, name)), name


def test_streamed_broadcast_chunks_are_whole_k_blocks_that_cover_k():
    """mmh_shard_chunks: the K boundaries mmh_shard_sgemm_streamed cuts B's broadcast at (host arithmetic).  Every chunk is a
    run of whole 128-deep K blocks (a chunk's A columns then start 512 bytes into a row: the unchunked launch's alignment
    class), the chunks cover [0, k) in order, none is empty, and there are min(b_chunks, ceil(k / 128), 64) of them -- one
    for b_chunks <= 1 or k <= 128 (and for k = 0: the GEMM then only clears C)."""
    for k in (0, 1, 96, 128, 129, 300, 1024, 4096, 16384, 16385, 100000):
        for b in (0, 1, 2, 3, 7, 8, 64, 65, 1000):
            k0 = H.shard_chunks(k, b)
            c = len(k0) - 1
            blocks = (k + 127) // 128
            assert c == max(1, min(b, blocks, 64)), (k, b, k0)
            assert k0[0] == 0 and k0[-1] == k, (k, b, k0)
            assert all(x % 128 == 0 for x in k0[:-1]), (k, b, k0)
            if k > 0:
                assert all(k0[i] < k0[i + 1] for i in range(c)), (k, b, k0)
                sizes = [k0[i + 1] - k0[i] for i in range(c)]
                assert max(sizes) - min(sizes[:-1] or sizes) <= 128 or c == 1, (k, b, sizes)      # near-equal runs
    with pytest.raises(H.MMultError):
        H.shard_chunks(-1, 4)
