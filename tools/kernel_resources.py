#!/usr/bin/env python3
"""kernel_resources.py -- registers, spills, scratch and LDS of every kernel in a built library, read from the code
objects embedded in the .so (ELF notes, NT_AMDGPU_METADATA, msgpack): no GPU, no llvm tools.

    python tools/kernel_resources.py [how-to-optimize-gemm_amd/libmmult_hip.so] [--match sgemm]

What it is for: a kernel that starts to spill (round 4: the VALU rung's unrolled K-slice went to 512 registers and 139
spilled ones until every k-step was pinned in place) or outgrows the register budget its co-residency needs shows up
on the CPU (tests/test_kernel_resources.py), not as a slow number on the GPU box."""
import argparse
import os
import struct
import sys

import msgpack

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EM_AMDGPU = 224
NT_AMDGPU_METADATA = 32


def code_objects(blob):
    """Offsets of the ELF64 code objects for the GPU inside a host shared library's fat binary."""
    out, i = [], blob.find(b"\x7fELF", 1)
    while i != -1:
        if i + 20 < len(blob) and struct.unpack_from("<H", blob, i + 18)[0] == EM_AMDGPU:
            out.append(i)
        i = blob.find(b"\x7fELF", i + 1)
    return out


def kernels_of(elf):
    e_shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    e_shentsize, e_shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    found = []
    for s in range(e_shnum):
        _, sh_type, _, _, sh_offset, sh_size = struct.unpack_from("<IIQQQQ", elf, e_shoff + s * e_shentsize)
        if sh_type != 7:      # SHT_NOTE
            continue
        p = sh_offset
        while p < sh_offset + sh_size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12 + ((namesz + 3) & ~3)
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == NT_AMDGPU_METADATA:
                found += msgpack.unpackb(desc, raw=False, strict_map_key=False).get("amdhsa.kernels", [])
    return found


def demangled_head(name):
    """mmh::kernel<template args as they stand in the mangled name> -- enough to tell instantiations apart."""
    s = name
    if s.startswith("_ZN3mmh"):
        s = s[len("_ZN3mmh"):]
        n = ""
        while s and s[0].isdigit():
            n += s[0]
            s = s[1:]
        ident, rest = s[:int(n)], s[int(n):]
        args = []
        if rest.startswith("I"):
            body = rest[1:].split("EEEv")[0] if "EEEv" in rest else rest[1:]
            for tok in body.split("E"):
                if tok.startswith("Li"):
                    args.append(tok[2:])
                elif tok.startswith("Lb"):
                    args.append("true" if tok[2:] == "1" else "false")
        return f"{ident}<{','.join(args)}>" if args else ident
    return s


def resources(path):
    blob = open(path, "rb").read()
    rows = []
    for off in code_objects(blob):
        for k in kernels_of(blob[off:]):
            rows.append({
                "name": k[".name"], "kernel": demangled_head(k[".name"]),
                "vgpr": k.get(".vgpr_count", 0), "agpr": k.get(".agpr_count", 0), "sgpr": k.get(".sgpr_count", 0),
                "vgpr_spill": k.get(".vgpr_spill_count", 0), "sgpr_spill": k.get(".sgpr_spill_count", 0),
                "scratch": k.get(".private_segment_fixed_size", 0), "lds_static": k.get(".group_segment_fixed_size", 0),
                "threads": k.get(".max_flat_workgroup_size", 0),
            })
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("library", nargs="?", default=os.path.join(REPO, "how-to-optimize-gemm_amd", "libmmult_hip.so"))
    ap.add_argument("--match", default="")
    args = ap.parse_args()
    rows = [r for r in resources(args.library) if args.match in r["kernel"]]
    print(f"{len(rows)} kernels in {os.path.relpath(args.library, REPO)}")
    print("| kernel | threads | VGPR | AGPR | spilled VGPR | scratch bytes | static LDS |")
    print("|---|---|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: r["kernel"]):
        print(f"| `{r['kernel']}` | {r['threads']} | {r['vgpr']} | {r['agpr']} | {r['vgpr_spill']} | {r['scratch']} | {r['lds_static']} |")


if __name__ == "__main__":
    sys.exit(main())
