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
        if f.endswith((".hip", ".hpp")):
            out.append(os.path.join(CSRC, f))
    return out


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> how-to-optimize-gemm_amd/libmmult_hip.so"""
    srcs = library_sources()
    if force or _stale(LIB, srcs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
               "-Wno-unused-result", os.path.join(CSRC, "mmult_hip.hip"), "-o", LIB, "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_timeline_library(force: bool = False) -> str:
    """-DMMH_AB_BUILD -DMMH_DMA_TIMELINE -> libmmult_hip_tl.so: the A/B library whose plain LDS-DMA kernels
    also write per-workgroup wall-clock stamps (tools/dma_timeline.py only; the stamps perturb the
    schedule of the big tiles, so no other tool measures with this build)."""
    target = os.path.join(PKG_DIR, "libmmult_hip_tl.so")
    if force or _stale(target, library_sources()):
        subprocess.check_call([hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC", "-DMMH_AB_BUILD",
                               "-DMMH_DMA_TIMELINE", "-Wno-unused-result", os.path.join(CSRC, "mmult_hip.hip"), "-o",
                               target, "-ldl"])
    return target


def build_ab_library(force: bool = False, verbose: bool = False) -> str:
    """The same sources with -DMMH_AB_BUILD -> libmmult_hip_ab.so: the product kernels PLUS the
    scheduling A/B variants and the timing-only ablation builds (wrong results).  Loaded by
    tools/ab_bench.py and tools/misc_bench.py only; never by the package, the harness or the tests
    of the product path."""
    srcs = library_sources()
    if force or _stale(AB_LIB, srcs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC", "-DMMH_AB_BUILD",
               "-Wno-unused-result", os.path.join(CSRC, "mmult_hip.hip"), "-o", AB_LIB, "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return AB_LIB


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
