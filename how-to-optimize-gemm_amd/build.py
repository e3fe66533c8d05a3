"""Build recipe for libmmult_hip.so (hipcc, gfx950 only) and the C++ harness.

`python -m how_to_optimize_gemm_amd.build` or __graft_entry__.build() runs it.
The library is built IN-TREE next to this file so that it travels with the
repository snapshot to the GPU box; nothing is installed into site-packages.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB = os.path.join(PKG_DIR, "libmmult_hip.so")
AB_LIB = os.path.join(PKG_DIR, "libmmult_hip_ab.so")   # tools-only build: A/B variants + timing-only ablations
HARNESS = os.path.join(PKG_DIR, "harness")
ARCH = "gfx950"


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X backend cannot be built")
    return exe


def _stale(target: str, srcs: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def library_sources() -> list[str]:
    out = [os.path.join(REPO, "include", "mmult_hip.h")]
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp", ".inc", ".map")):
            out.append(os.path.join(CSRC, f))
    return out


AB_SRC = os.path.join(REPO, "tools", "ab")   # tile families that were built, measured and lost: tools build only


def translation_units(ab: bool = False) -> list[str]:
    tus = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    if ab and os.path.isdir(AB_SRC):
        tus += sorted(os.path.join(AB_SRC, f) for f in os.listdir(AB_SRC) if f.endswith(".hip"))
    return tus


def _build_shared(target: str, defines: list[str], objdir: str, force: bool, verbose: bool) -> str:
    """hipcc -c every translation unit of csrc/ (in parallel: the kernel-heavy ones take a minute each),
    then link them into `target`.  Objects live under build/<objdir>/ (git-ignored) and are reused while
    neither their .hip nor any header has changed."""
    from concurrent.futures import ThreadPoolExecutor
    ab = "-DMMH_AB_BUILD" in defines
    headers = [s for s in library_sources() if s.endswith((".hpp", ".h", ".inc"))]
    if ab and os.path.isdir(AB_SRC):
        headers += [os.path.join(AB_SRC, f) for f in sorted(os.listdir(AB_SRC)) if f.endswith((".hpp", ".inc"))]
    odir = os.path.join(PKG_DIR, "build", objdir)
    os.makedirs(odir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-I" + CSRC] + defines
    if ab:
        flags.append("-I" + AB_SRC)
    jobs = []
    for tu in translation_units(ab):
        obj = os.path.join(odir, os.path.basename(tu)[:-4] + ".o")
        if force or _stale(obj, [tu] + headers):
            jobs.append([hipcc()] + flags + ["-c", tu, "-o", obj])
    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
            list(pool.map(run, jobs))
    objs = [os.path.join(odir, os.path.basename(tu)[:-4] + ".o") for tu in translation_units(ab)]
    if force or jobs or _stale(target, objs + [os.path.join(CSRC, "exports.map")]):
        run([hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs +
            ["-o", target, "-ldl", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")])
    return target


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> how-to-optimize-gemm_amd/libmmult_hip.so"""
    return _build_shared(LIB, [], "product", force, verbose)


def build_timeline_library(force: bool = False) -> str:
    """-DMMH_AB_BUILD -DMMH_DMA_TIMELINE -> libmmult_hip_tl.so: the A/B library whose plain LDS-DMA kernels
    also write per-workgroup wall-clock stamps (tools/dma_timeline.py only; the stamps perturb the
    schedule of the big tiles, so no other tool measures with this build)."""
    return _build_shared(os.path.join(PKG_DIR, "libmmult_hip_tl.so"), ["-DMMH_AB_BUILD", "-DMMH_DMA_TIMELINE"], "timeline",
                         force, False)


def build_ab_library(force: bool = False, verbose: bool = False) -> str:
    """The same sources with -DMMH_AB_BUILD -> libmmult_hip_ab.so: the product kernels PLUS the
    scheduling A/B variants and the timing-only ablation builds (wrong results).  Loaded by
    tools/ab_bench.py and tools/misc_bench.py only; never by the package, the harness or the tests
    of the product path."""
    return _build_shared(AB_LIB, ["-DMMH_AB_BUILD"], "ab", force, verbose)


def build_harness(force: bool = False) -> str:
    """The C++ host side (test_MMult.x and friends) via its makefile."""
    exe = os.path.join(HARNESS, "test_MMult.x")
    args = ["make", "-s", "-C", HARNESS]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args + ["all"])
    return exe


def build_all(force: bool = False) -> None:
    build_library(force)
    if os.path.isdir(HARNESS):
        build_harness(force)


if __name__ == "__main__":
    build_all(force=True)
    print(LIB)
