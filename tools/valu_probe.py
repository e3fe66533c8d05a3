#!/usr/bin/env python3
"""valu_probe.py -- the vector ALU's fp32 FMA rate as this box sustains it (mmh_probe_valu_f32): v_pk_fma_f32 and
v_fma_f32, one to four waves per SIMD, beside the MFMA probe.  The denominator of the K1 / K1W rung.  Needs a GPU."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import how_to_optimize_gemm_amd as H

mm = H.MMult(0, "auto")
out = {"mfma_f32_tflops": round(mm.probe_mfma_f32(), 1)}
for packed in (True, False):
    for w in (1, 2, 3, 4):
        vals = sorted(mm.probe_valu_f32(packed, w) for _ in range(3))
        out[f"{'v_pk_fma_f32' if packed else 'v_fma_f32'}_{w}_waves_per_simd_tflops"] = round(vals[1], 1)
mm.close()
print(json.dumps(out, indent=1))
