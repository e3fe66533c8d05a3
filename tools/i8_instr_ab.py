#!/usr/bin/env python3
"""i8_instr_ab.py -- v_mfma_i32_16x16x32_i8 (the instruction BASELINE.json configs[4] names) against
v_mfma_i32_16x16x64_i8 (what the int8 kernels issue) and v_mfma_i32_32x32x32_i8: the matrix pipe's sustained
rate on each, MFMA-only loops, constant and pseudo-random operands (tools/probes/mfma_i8_shapes.hip, compiled here
with hipcc), next to the int8 GEMM's end-to-end rates at 4096^3 and 8192^3.  Prints profiles/r04_i8_instr_ab.md.
The int8 path is PARITY UNPINNED (the reference holds no int8 code, README.md:71-85 is prose).  Needs a GPU."""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    exe = "/tmp/mfma_i8_shapes"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", os.path.join(REPO, "tools", "probes", "mfma_i8_shapes.hip"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300).stdout
    rates = {}
    for line in out.splitlines():
        m = re.match(r"(v_mfma_\w+), (\w+)\s+operands:\s+([\d.]+) TOPS", line)
        if m:
            rates[(m.group(1), m.group(2))] = float(m.group(3))
    import torch
    import how_to_optimize_gemm_amd as H
    mm = H.MMult(0, "auto")
    gemm = {}
    for n in (4096, 8192):
        g = torch.Generator(device="cuda").manual_seed(7)
        qa = torch.randint(-127, 128, (n, n), device="cuda", dtype=torch.int8, generator=g)
        qb = torch.randint(-127, 128, (n, n), device="cuda", dtype=torch.int8, generator=g)
        qc = torch.empty((n, n), device="cuda", dtype=torch.int32)
        for _ in range(60 if n == 8192 else 300):
            mm.igemm_s8(qa, qb, out=qc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40 if n == 8192 else 200
        e0.record()
        for _ in range(reps):
            mm.igemm_s8(qa, qb, out=qc)
        e1.record()
        torch.cuda.synchronize()
        gemm[n] = 2.0 * n ** 3 / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12
    mm.close()
    print("# int8: the instruction BASELINE configs[4] names against the one the kernels issue (parity unpinned)\n")
    print("MFMA-only loops, 512 workgroups x 4 waves x 8 independent accumulators, ~100 ms back to back (the power-managed")
    print("state); TOPS = 2 x MACs.  `tools/i8_instr_ab.py`, one MI355X.\n")
    print("| instruction | K per instruction | operand bytes / lane (A + B) | constant operands | random operands |")
    print("|---|---|---|---|---|")
    for name, kk, ob in (("v_mfma_i32_16x16x32_i8", 32, 16), ("v_mfma_i32_16x16x64_i8", 64, 32), ("v_mfma_i32_32x32x32_i8", 32, 32)):
        print(f"| `{name}` | {kk} | {ob} | {rates.get((name, 'constant'), float('nan')):.0f} TOPS | {rates.get((name, 'random'), float('nan')):.0f} TOPS |")
    r64, r32 = rates.get(("v_mfma_i32_16x16x64_i8", "random"), 0.0), rates.get(("v_mfma_i32_16x16x32_i8", "random"), 0.0)
    print(f"\nThe named instruction sustains {r32:.0f} TOPS on random operands, the double-rate form {r64:.0f}: a kernel on")
    print("`16x16x32` has the pipe of a bf16 kernel -- its ceiling is below what the shipped int8 GEMM already measures end to end:\n")
    print("| int8 GEMM (MMH_KERNEL_AUTO, `igemm_s8_pp_kernel` on `16x16x64`) | TOPS | of the 16x16x64 random-operand rate | of the 16x16x32 random-operand rate |")
    print("|---|---|---|---|")
    for n, v in gemm.items():
        print(f"| {n}^3 | {v * 1:.0f} | {v / r64:.2f} | {v / r32:.2f} |" if r64 and r32 else f"| {n}^3 | {v:.0f} | | |")
    print("\n(`extras.int8_8192_tops` and `extras.int8_frac_of_measured_pipe` in the bench line are these figures taken in the bench run.)")


if __name__ == "__main__":
    main()
