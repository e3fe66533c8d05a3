#!/bin/bash
# What the K1 (VALU rung) kernels' inner loops are made of: compile launch_valu.hip's device side, count
# instruction classes between the loop's back-edge target and its branch.  Runs without a GPU (hipcc cross-compiles).
#   bash tools/valu_isa.sh > profiles/r03_valu_isa.txt
set -eu
cd "$(dirname "$0")/.."
SRC=${SRC:-how-to-optimize-gemm_amd/csrc/launch_valu.hip}
T=$(mktemp -d)
# the same flags build.py compiles the translation unit with, device side only
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result --cuda-device-only -S $SRC -o $T/dev.s
python3 - $T/dev.s <<'PY'
import re, sys
text = open(sys.argv[1]).read()
for m in re.finditer(r"^(_Z\w*sgemm_valu\w*):[^\n]*\n(.*?)^\s*s_endpgm", text, re.S | re.M):
    name, f = m.group(1), m.group(2)
    lines = [l.split(";")[0].strip() for l in f.split("\n")]
    lines = [l for l in lines if l]
    labels = {l[:-1]: i for i, l in enumerate(lines) if re.match(r"\.LBB\w+:$", l)}
    best = None            # innermost hot loop = the backward branch whose body holds the most FMAs
    for i, l in enumerate(lines):
        mm = re.match(r"s_c?branch\w* (\.LBB\w+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            body = lines[labels[mm.group(1)]:i + 1]
            if any(re.match(r"\.LBB\w+:$", b) for b in body[1:]):
                continue    # not innermost
            nf = sum(1 for b in body if "fma" in b)
            if best is None or nf > best[0]:
                best = (nf, body)
    if best is None:
        continue
    body = best[1]
    count = lambda pat: sum(1 for b in body if re.match(pat, b))
    waits = [b.split(None, 1)[1] for b in body if b.startswith("s_waitcnt")]
    vg = re.search(re.escape(name) + r"\.num_vgpr, (\d+)", text)
    sc = re.search(re.escape(name) + r"\.private_seg_size, (\d+)", text)
    print(name)
    print(f"  innermost loop: {len(body)} instructions per trip")
    print(f"    v_pk_fma_f32 {count(r'v_pk_fma_f32')}   v_fma_f32|v_fmac_f32 {count(r'v_fma(c)?_f32')}   ds_read* {count(r'ds_read')}   "
          f"ds_write* {count(r'ds_write')}   global|buffer loads {count(r'(global|buffer)_load')}   s_barrier {count(r's_barrier')}")
    print(f"    s_waitcnt in the loop: {waits}")
    print(f"  num_vgpr {vg.group(1) if vg else '?'}   scratch bytes {sc.group(1) if sc else '?'}")
PY
rm -rf $T
