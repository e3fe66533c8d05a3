#!/usr/bin/env python3
"""bench.py -- the driver's measurement contract for the MY_MMult hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one MY_MMult call (device flavour, C = A*B) on synthetic inputs
already resident in HBM (the reference excludes H2D/D2H from its timed region
too: cuda/test_MMult.cpp:85-98,121).

  N = 1  workload = BASELINE.json configs[2]: fp32 N=4096 square SGEMM on the
         MFMA kernel (MMH_KERNEL_AUTO's cost table picks the 128x64 K2W tile -- LDS-DMA
         by loader waves -- at this size: 2048 tiles = four whole rounds of two
         workgroups per CU, a plain launch) -- the
         configuration the headline metric ("% of MI355X fp32 MFMA peak at
         N=4096") is quoted on.
  N > 1  workload = configs[3]: fp32 N=16384, C row panels sharded over the N
         ranks (mmh_shard_rows), B replicated by one RCCL broadcast from rank 0
         BEFORE the timed region (it is data placement, the multi-GPU analogue
         of the H2D copy; its time is reported as `bcast_ms`, and the
         broadcast-inclusive rate as `value_incl_bcast`).  Total work is fixed
         -> "scaling": "strong".
         `python bench.py --gpus N` WITHOUT a launcher starts the N ranks itself
         (re-executes under torch.distributed.run on 127.0.0.1) and FAILS when
         fewer than N devices are visible -- it never falls back to fewer ranks.

roofline.traffic, roofline.mfma_busy_frac and roofline.hbm_gbps are measured by the run itself (N=1, unless --no-extras /
--no-live-traffic): three rocprofv3 passes -- FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE, each in
its own run -- over a child process that launches the timed kernel on the same shape, while the host times REF_MMult;
the committed pass of profiles/pmc_traffic.json rides along as traffic_committed_pass.  extras.sweep_gflops carries the
metric's whole sweep (cuda/parameters.h:5-7: 25 sizes) for `auto`, rocBLAS, hipBLASLt and the VALU rung, and the literal
configs[2] tile (mfma_tiles_4096, mfma_128x128_dma5_4096); extras.int8_roofline is configs[4]'s own roofline object.

value = GFLOPS = 2*m*n*k*1e-9 / t  (cuda/test_MMult.cpp:116-118), whole job.
Rank 0 prints ONE JSON line.

Cold vs sustained.  The reference times NREPEATS = 20 launches with no warm-up
(cuda/test_MMult.cpp:98-118).  An idle MI355X starts a launch train below its
sustained clock, so that convention and the contract's (W warm-ups, K timed
steps at steady state) give different numbers; both are reported:
  * the run opens with a per-launch trace of the process's first launches (`ramp_launches` of them: up to 400,
    about half a second; --ramp 0 switches it off, --ramp N bounds it) (mmh_trace_sgemm, one hipEvent pair
    each) -> `cold`: launch #1 (nothing but a launch since mmh_create warms the handle), the mean of launches
    1..20 (= the reference convention) and 2..21, and where the ramp ends;
  * then W warm-up steps and K timed steps: `warmup` = W exactly, `ramp_launches` = the traced launches in
    front of them, `untimed_launches` = their sum.  With --ramp 0 the W warm-ups are the only untimed
    launches of the process.
  --sweep adds the reference's square sweep (cuda/parameters.h:5-7: 1024 .. 4096 step 128, plus 8192 and
  16384) under the same sharding as the headline workload -> `extras.sweep_gflops_sharded`: the north star's
  "GFLOPS on the square-N sweep at 1/2/4/8 GPUs".
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz
METRIC = "GFLOPS vs N (square SGEMM sweep); % of MI355X fp32 MFMA peak at N=4096"
RAMP = 400                        # per-launch traced launches that open the run (the clock ramp), at most
RAMP_SECONDS = 0.5                # ... and about this long (a 16384-row panel takes tens of ms per launch)
# what the N = 1 line carries beyond the contract's keys (tests/test_bench_line_schema.py pins these)
SWEEP_SIZES = tuple(range(1024, 4097, 128))                      # cuda/parameters.h:5-7
SWEEP_KERNELS_ALL_SIZES = ("auto", "rocblas", "hipblaslt", "valu")
SWEEP_KEYS_AT_4096 = ("mfma_tiles_4096", "mfma_128x128_dma5_4096")   # configs[2]'s literal 128x128 tile
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "mfma_busy_frac", "hbm_gbps",
                 "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch")
INT8_ROOFLINE_KEYS = ("bound", "achieved", "peak_spec", "peak_measured_random", "frac", "kernel", "kernel_ms", "achieved_8192",
                      "tops_4096_on_v_mfma_i32_16x16x32_i8", "16x16x32_bit_equal_to_16x16x64")
SETTLE_MS = 150.0                 # sharded runs: untimed launches in front of the W warm-ups of BOTH timed regions (ranks, single-GPU reference)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)       # NREPEATS, cuda/parameters.h:24
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--n", type=int, default=0, help="override the square size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sweep / probes extras")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed PMC pass instead of two rocprofv3 runs now")
    ap.add_argument("--force-shard", action="store_true",
                    help="run the multi-GPU code path (process group, broadcast, all_reduce) even with one rank")
    ap.add_argument("--ramp-csv", default="", help="write the per-launch clock-ramp trace to this CSV")
    ap.add_argument("--ramp", type=int, default=RAMP, help="per-launch traced launches that open the run (0: none)")
    ap.add_argument("--b-chunks", type=int, default=8,
                    help="N > 1: K-chunks the streamed broadcast of B travels in (gemm_with_streamed_b)")
    ap.add_argument("--no-single-gpu-reference", action="store_true",
                    help="N > 1: skip rank 0's run of the WHOLE problem on one GPU (the denominator of scaling_efficiency)")
    ap.add_argument("--sweep", action="store_true",
                    help="also run the reference's square sweep under this run's sharding (extras.sweep_gflops_sharded)")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves.
    Fails loudly -- non-zero exit, a message on stderr, no JSON line -- when fewer than N devices
    are visible; the 1-GPU workload is never substituted for the N-GPU one."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} asked for, {have} HIP device(s) visible: refusing to run "
                         f"(the multi-GPU workload is never replaced by a smaller one)\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline(n: int) -> dict:
    """REF_MMult timed on the host beside the GPU (rank 0, N=1 only).  Uses the
    reference's OWN object code (oracle/_ref/libref_armv7.so = armv7/REF_MMult.c,
    gcc -O2, 1 core) when it travelled with the repo, else our C restatement of
    the same loop.  Bounded sample: the first ROWS rows of the N^3 problem
    (m=ROWS, n=k=N) -- identical per-row work to the full problem."""
    import numpy as np
    from oracle import oracle as O
    rows = 128 if n >= 4096 else min(n, 256)        # ~15 s of the serial triple loop at N=4096
    a, b = O.harness_inputs(rows, n, n, seed=2026)
    c = np.zeros((rows, n), dtype=np.float32)
    if O.have_ref():
        kind, fn = "reference", O.reflib("armv7").REF_MMult
        dclock = O.reflib("armv7").dclock
    else:
        kind, fn = "port", O.lib().orc_ref_mmult
        dclock = O.lib().orc_dclock
    t = dclock()
    fn(rows, n, n, a, n, b, n, c, n)
    dt = dclock() - t                                   # dclock(): cuda/dclock.cpp:8-22
    flops = 2.0 * rows * n * n
    # the thread-parallel, bit-identical port (what the parity tests use), all cores
    cores = O.max_threads()
    a2, b2 = O.harness_inputs(min(n, 1024), n, n, seed=2027)
    t0 = time.perf_counter()
    O.ref_mmult(a2, b2, fma=False, fast=True)
    dt2 = time.perf_counter() - t0
    out = {"value": round(flops * 1e-9 / dt, 3), "unit": "GFLOPS", "cores": 1, "kind": kind,
           "sample": f"REF_MMult triple loop ({'the reference object code, armv7/REF_MMult.c gcc -O2' if kind == 'reference' else 'the C restatement'}), "
                     f"first {rows} rows of the {n}^3 problem (m={rows}, n=k={n}), {dt:.1f} s; parallel_port beside it is the "
                     f"RESTATED loop (i-p-j, row-parallel, bit-identical), not the reference object",
           "parallel_port": {"value": round(2.0 * a2.shape[0] * n * n * 1e-9 / dt2, 2),
                             "unit": "GFLOPS", "cores": cores,
                             "sample": f"i-p-j row-parallel restatement, m={a2.shape[0]}, n=k={n}"}}
    # the cuda directory's oracle is a host BLAS (cuda/REF_MMult.cpp:9-13): numpy's sgemm beside it
    try:
        a3, b3 = O.harness_inputs(n, n, n, seed=2028) if n <= 4096 else (a2, b2)
        t0 = time.perf_counter()
        a3 @ b3
        dt3 = time.perf_counter() - t0
        out["host_blas"] = {"value": round(2.0 * a3.shape[0] * n * n * 1e-9 / dt3, 1), "unit": "GFLOPS", "cores": cores,
                            "sample": f"numpy (OpenBLAS) sgemm, m={a3.shape[0]}, n=k={n} -- the analogue of "
                                      f"cuda/REF_MMult.cpp's cblas_sgemm"}
    except Exception:
        pass
    return out


def pmc_traffic(n: int):
    """HBM bytes per launch from the committed PMC pass (profiles/), if any."""
    p = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
        return d.get(str(n), {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
N_SIMDS = 1024                     # 256 CUs x 4 SIMDs: the denominator of the matrix pipe's busy fraction
N_XCDS = 8                         # GRBM_GUI_ACTIVE is summed over the XCDs


def live_counters(n: int, kernel: str):
    """Counters of the timed kernel measured NOW: three rocprofv3 passes (FETCH_SIZE, WRITE_SIZE, and
    SQ_VALU_MFMA_BUSY_CYCLES with GRBM_GUI_ACTIVE -- each pass its own run, kernel trace only beside it:
    MI355X_MICROARCH.md, HBM / rocprofv3) over a child process that launches the same kernel on the same shape eight
    times.  Returns ({"traffic": bytes, "mfma_busy_frac": f, "effective_clock_ghz": g, "kernel_us_under_counters": t},
    description) with None for what could not be read; never raises."""
    import csv
    import glob
    import shutil
    import tempfile
    out = {"traffic": None, "mfma_busy_frac": None, "effective_clock_ghz": None, "kernel_us_under_counters": None}
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return out, "rocprofv3 not found"
    child = (f"import sys; sys.path.insert(0, {REPO!r}); import torch, how_to_optimize_gemm_amd as H; "
             f"mm = H.MMult(0, {kernel!r}); a = torch.rand(({n}, {n}), device='cuda') * 2 - 1; "
             f"b = torch.rand(({n}, {n}), device='cuda') * 2 - 1; c = torch.empty(({n}, {n}), device='cuda'); "
             f"[mm.matmul(a, b, out=c) for _ in range(8)]; torch.cuda.synchronize()")
    val, why = {}, []
    for ctrs in PMC_PASSES:
        d = tempfile.mkdtemp(prefix="mmh_pmc_", dir="/tmp")
        try:
            # its own process group, so that a profiler that stops responding is killed WITH the child it started
            p = subprocess.Popen([exe, "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "-o", "pmc",
                                  "--", sys.executable, "-c", child], cwd="/tmp",
                                 env={**os.environ, "TMPDIR": "/tmp", "MMH_LAZY": "1"},   # no warm-up launches in the trace
                                 stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=90)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(p.pid, signal.SIGKILL)
                p.wait()
                why.append(f"rocprofv3 --pmc {' '.join(ctrs)} did not finish within 90 s")
                continue
            per = {c: {} for c in ctrs}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    name, c = r.get("Kernel_Name", ""), r.get("Counter_Name")
                    if "sgemm_" in name and "naive" not in name and c in per:
                        per[c][r["Dispatch_Id"]] = per[c].get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            for c in ctrs:
                vals = [v for _, v in sorted(per[c].items(), key=lambda kv: int(kv[0]))][2:]   # skip the cold ones
                if vals:
                    val[c] = sum(vals) / len(vals)
                else:
                    why.append(f"no {c} rows for the kernel in rocprofv3's output")
            if "GRBM_GUI_ACTIVE" in ctrs:
                durs = []
                for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                    for r in csv.DictReader(open(f)):
                        if "sgemm_" in r.get("Kernel_Name", "") and "naive" not in r.get("Kernel_Name", ""):
                            durs.append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3))
                durs = [u for _, u in sorted(durs)][2:]
                if durs:
                    val["_us"] = sum(durs) / len(durs)
        except Exception as e:   # a profiler that is missing, refused or slow must not cost the run its line
            why.append(f"{type(e).__name__}: {e}"[:200])
        finally:
            shutil.rmtree(d, ignore_errors=True)
    how = []
    if "FETCH_SIZE" in val and "WRITE_SIZE" in val:
        out["traffic"] = int(round(val["FETCH_SIZE"] * 1024 * 2 + val["WRITE_SIZE"] * 1024))
        how.append(f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes run by this invocation (each its own run, mean of 6 "
                   f"launches of the timed kernel): FETCH_SIZE {val['FETCH_SIZE']:.0f} KiB x 2 (gfx950, 16 B/lane reads) + "
                   f"WRITE_SIZE {val['WRITE_SIZE']:.0f} KiB")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in val and val.get("GRBM_GUI_ACTIVE"):
        cycles = val["GRBM_GUI_ACTIVE"] / N_XCDS
        out["mfma_busy_frac"] = round(val["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMDS / cycles, 4)
        how.append(f"third pass: SQ_VALU_MFMA_BUSY_CYCLES {val['SQ_VALU_MFMA_BUSY_CYCLES']:.4g} / ({N_SIMDS} SIMDs x "
                   f"GRBM_GUI_ACTIVE {val['GRBM_GUI_ACTIVE']:.4g} / {N_XCDS} XCDs)")
        if val.get("_us"):
            out["kernel_us_under_counters"] = round(val["_us"], 2)
            out["effective_clock_ghz"] = round(cycles / (val["_us"] * 1e-6) / 1e9, 3)
    return out, "; ".join(how + why)


def sweep_summary(sweep: dict) -> dict:
    have = [p for p in SWEEP_SIZES if f"auto_{p}" in sweep]
    vend = lambda p: max(sweep.get(f"rocblas_{p}", 0.0), sweep.get(f"hipblaslt_{p}", 0.0))
    ahead = [p for p in have if vend(p) > 0 and sweep[f"auto_{p}"] >= vend(p)]
    return {"sizes": len(have),
            "auto_min_pct_of_peak": round(min(sweep[f"auto_{p}"] for p in have) / (PEAK_FP32_MFMA_TFLOPS * 10), 2) if have else None,
            "auto_sizes_at_or_above_92_pct": len([p for p in have if sweep[f"auto_{p}"] >= 0.92 * PEAK_FP32_MFMA_TFLOPS * 1e3]),
            "auto_ahead_of_both_vendor_libraries_at": len(ahead),
            "behind_at": [p for p in have if vend(p) > 0 and p not in ahead],
            "auto_over_best_vendor_min": round(min(sweep[f"auto_{p}"] / vend(p) for p in have if vend(p) > 0), 4) if any(vend(p) > 0 for p in have) else None}


def vendor_sweep_via_harness(kern: str, sizes) -> dict:
    """rocBLAS / hipBLASLt over the square sweep through the reference-shaped C++ harness (harness/test_MMult.x, KERNEL=<lib>
    REF=skip WARMUP_MS=50 TRIALS=3): a process WITHOUT torch, so that `dlopen("libhipblaslt.so")` finds the image's ROCm 7.2
    library.  Inside this Python process the same call resolves to the copy bundled in torch's wheel (ROCm 7.0), which is
    15-20 % slower on the stream-K sizes (2560: 122.6 against 148.4 TFLOP/s) -- rounds 1-5's bench lines quoted that copy.
    Returns {p: GFLOPS} (empty when the harness is not there or fails)."""
    exe = os.path.join(REPO, "how-to-optimize-gemm_amd", "harness", "test_MMult.x")
    if not os.path.exists(exe):
        return {}
    env = {**os.environ, "KERNEL": kern, "REF": "skip", "WARMUP_MS": "50", "TRIALS": "3",
           "PFIRST": str(min(sizes)), "PLAST": str(max(sizes)), "PINC": "128"}
    try:
        r = subprocess.run([exe], cwd=os.path.dirname(exe), env=env, capture_output=True, text=True, timeout=120)
    except Exception:
        return {}
    out = {}
    for line in r.stdout.splitlines():
        f = line.split()
        if len(f) == 3 and f[0].isdigit():
            try:
                out[int(f[0])] = round(float(f[1]), 1)
            except ValueError:
                pass
    return out if r.returncode == 0 else {}


def int8_roofline(mm, torch, dev, H) -> dict:
    """BASELINE.json configs[4] (parity unpinned: the reference holds no int8 code): mmh_igemm_s8 at 4096^3 and 8192^3,
    end to end and sustained, beside the spec peak and what the matrix pipe sustains on random operands in this run."""
    out = {}
    gq = torch.Generator(device=dev).manual_seed(7)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = {}
    for p, warm, reps in ((4096, 300, 200), (8192, 40, 25)):     # ~20 ms of launches first: the power manager's sustained state
        qa = torch.randint(-127, 128, (p, p), device=dev, dtype=torch.int8, generator=gq)
        qb = torch.randint(-127, 128, (p, p), device=dev, dtype=torch.int8, generator=gq)
        qc = torch.empty((p, p), device=dev, dtype=torch.int32)
        best = None
        for _ in range(2):
            for _ in range(warm):
                mm.igemm_s8(qa, qb, out=qc)
            e0.record()
            for _ in range(reps):
                mm.igemm_s8(qa, qb, out=qc)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            best = ms if best is None else min(best, ms)
        times[p] = best
        if p == 4096:
            out["kernel"] = ("igemm_s8_pp_kernel<256x256>: v_mfma_i32_16x16x64_i8, wave tile 128x64, two wave groups in ping-pong, "
                             "128-deep slices x 2 ring buffers by LDS-DMA (B in place, ds_read_b64_tr_b8), C through a per-wave LDS "
                             "transposer; 256 workgroups of 512 threads")
            # the config-named instruction (mode 7: v_mfma_i32_16x16x32_i8), same bits, beside it
            want = qc.clone()
            mm.set_igemm_mode(7)
            try:
                for _ in range(warm // 2):
                    mm.igemm_s8(qa, qb, out=qc)
                e0.record()
                for _ in range(reps // 2):
                    mm.igemm_s8(qa, qb, out=qc)
                e1.record()
                torch.cuda.synchronize()
                out["tops_4096_on_v_mfma_i32_16x16x32_i8"] = round(2.0 * p ** 3 / (e0.elapsed_time(e1) / (reps // 2) * 1e-3) / 1e12, 1)
                out["16x16x32_bit_equal_to_16x16x64"] = bool(torch.equal(qc, want))
            finally:
                mm.set_igemm_mode(0)
        del qa, qb, qc
    constant = mm.probe_mfma_i8_sustained(False, 50.0)
    random_ = mm.probe_mfma_i8_sustained(True, 50.0)
    tops = {p: 2.0 * p ** 3 / (ms * 1e-3) / 1e12 for p, ms in times.items()}
    out.update({
        "bound": "mfma", "unit": "TOP/s", "achieved": round(tops[4096], 1), "peak_spec": 5030.0,
        "frac": round(tops[4096] / 5030.0, 4), "kernel_ms": round(times[4096], 5),
        "peak_measured_random": round(random_, 1), "peak_measured_constant": round(constant, 1),
        "frac_of_measured_random": round(tops[4096] / random_, 4) if random_ else None,
        "achieved_8192": round(tops[8192], 1), "kernel_ms_8192": round(times[8192], 5),
        "frac_8192": round(tops[8192] / 5030.0, 4),
        "frac_of_measured_random_8192": round(tops[8192] / random_, 4) if random_ else None,
        "algorithmic_ops_per_launch": 2.0 * 4096 ** 3, "algorithmic_bytes_per_launch": 2.0 * 4096 ** 2 + 4.0 * 4096 ** 2,
        "what": "mmh_igemm_s8 (int8 x int8 -> int32, MMH_OPT_IGEMM_MODE 0) end to end, sustained; peak_spec = 5.03 POP/s (the "
                "double-rate instructions at 2.4 GHz), peak_measured_* = an MFMA-only loop of v_mfma_i32_16x16x64_i8 on random / "
                "constant operands at the power-managed clock; parity unpinned (no int8 code in the reference)"})
    return out


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    # Rank 0 must print ONE JSON line on stdout, but RCCL writes its version banner to the
    # process's stdout at communicator creation: park fd 1 on stderr until the line is ready.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import how_to_optimize_gemm_amd as H

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to run a different job")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank}, only {torch.cuda.device_count()} device(s) "
                         f"visible: refusing to share a device between ranks")
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.force_shard
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    mm = H.MMult(local_rank, args.kernel)
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream(dev).cuda_stream

    if not sharded:
        n = args.n or 4096
        m = n
        row0, rows = 0, n
        workload = f"sgemm fp32 square N={n}, 1xMI355X, kernel={args.kernel} (BASELINE configs[2])"
        parallelism, scaling = "single", "strong"      # total work is fixed for every --gpus N (see N > 1 below)
    else:
        n = args.n or 16384
        m = n
        from how_to_optimize_gemm_amd.shard import RowPanelShard
        sh = RowPanelShard(m, n, n, rank, world)
        row0, rows = sh.row0, sh.rows
        workload = (f"sgemm fp32 square N={n}, C row panels over {world} GPUs, one RCCL broadcast of B "
                    f"(BASELINE configs[3])")
        parallelism, scaling = f"row-panel x{world}", "strong"

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    a = torch.rand((rows, n), device=dev, generator=g) * 2 - 1          # uniform [-1,1) like random_matrix
    b = torch.empty((n, n), device=dev)
    if rank == 0:
        gb = torch.Generator(device=dev).manual_seed(99)
        b.copy_(torch.rand((n, n), device=dev, generator=gb) * 2 - 1)
    c = torch.empty((rows, n), device=dev)
    torch.cuda.synchronize()

    # ---- the clock ramp, traced: the first RAMP launches of this process, one event pair each ----
    # (before the broadcast on purpose: B's contents do not matter for timing, and the chip is as
    # cold here as it will ever be)
    trace = []
    if rows and args.ramp > 0:
        trace = mm.trace_sgemm(rows, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, count=min(25, args.ramp),
                               stream=stream)
        more = max(0, min(args.ramp - 25, int(RAMP_SECONDS / (max(trace[-1], 1e-3) * 1e-3)) - 25))
        if more:
            trace += mm.trace_sgemm(rows, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, count=more,
                                    stream=stream)
    if args.ramp_csv and rank == 0 and trace:
        with open(args.ramp_csv, "w") as f:
            f.write("launch,ms,tflops\n")
            for i, ms in enumerate(trace):
                f.write(f"{i + 1},{ms:.5f},{2.0 * rows * n * n / (ms * 1e-3) / 1e12:.2f}\n")

    bcast_ms = 0.0
    if sharded:
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sh.broadcast_b(b, src=0, always=True)
        torch.cuda.synchronize()
        dist.barrier()
        bcast_ms = (time.perf_counter() - t0) * 1e3
        # second broadcast = steady-state cost without communicator warm-up
        t0 = time.perf_counter()
        sh.broadcast_b(b, src=0, always=True)
        torch.cuda.synchronize()
        dist.barrier()
        bcast_ms = min(bcast_ms, (time.perf_counter() - t0) * 1e3)

    overlap_ms = None
    if sharded:
        # broadcast hidden behind the GEMM: B in --b-chunks K-chunks, consumed with accumulate (same bits)
        def gemm_acc(x, y, out, accumulate):
            mm.sgemm(x.shape[0], n, x.shape[1], x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0),
                     out.data_ptr(), out.stride(0), accumulate, stream)
        c_stream = torch.empty_like(c)
        streamed_error = None
        try:
            for rep in range(2):
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sh.gemm_with_streamed_b(gemm_acc, a, b, c_stream, src=0, chunks=args.b_chunks, always=True)
                torch.cuda.synchronize()
                dist.barrier()
                dt = (time.perf_counter() - t0) * 1e3
                overlap_ms = dt if overlap_ms is None else min(overlap_ms, dt)
        except Exception as e:   # an optional extra must never cost the run its headline number
            streamed_error, overlap_ms = f"{type(e).__name__}: {e}"[:200], None
        if dist:                 # every rank takes the same branch below
            okf = torch.tensor([0 if overlap_ms is None else 1], device=dev)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if not int(okf.item()):
                overlap_ms = None

    def step():
        if rows:
            mm.sgemm(rows, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, False, stream)

    # N > 1 / --force-shard: the broadcast phases above leave the chip idle between their host synchronisations, and W
    # warm-ups of a sub-millisecond launch do not bring the clock back: untimed launches for ~SETTLE_MS first -- the SAME
    # lead-in the single-GPU reference below gets, so that `scaling_efficiency` compares like with like (one rank at
    # N = 4096 read 0.919 without it: 139 TF after the broadcast phases against 152 TF right behind the timed region)
    def settle(fn):
        count, t_s = 0, time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < SETTLE_MS:
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
            count += 4
        return count
    settle_launches = settle(step) if (sharded and rows) else 0
    for _ in range(args.warmup):
        step()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    gflops = 2.0 * m * n * n * 1e-9 / (ms_per_step * 1e-3)

    # dominant-kernel duration: hipEvents on the launch stream around K back-to-back launches
    # (issued right behind the timed region so the clock state is the same)
    kern_ms = mm.time_sgemm(rows, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n,
                            warmup=1, reps=args.steps, stream=stream) if rows else 0.0
    launched = H.last_launch()

    # N > 1: every rank's own kernel time, and -- in the SAME process group, right behind the timed region -- the whole
    # problem on rank 0's GPU alone: the denominator of scaling_efficiency = value(N) / (N * value(1)).
    per_rank_ms, single_ms = None, None
    if sharded:
        t = torch.tensor([kern_ms], device=dev, dtype=torch.float64)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        per_rank_ms = [round(float(x.item()), 4) for x in parts]
        if not args.no_single_gpu_reference:
            if rank == 0:
                try:
                    a1 = torch.rand((m, n), device=dev) * 2 - 1
                    c1 = torch.empty((m, n), device=dev)
                    # the ranks' own protocol (warmup = 1, reps <= 5 read the one-rank line's
                    # efficiency as 1.0487): W warm-up steps, then K timed steps between two device syncs on the host's clock
                    def whole():
                        mm.sgemm(m, n, n, a1.data_ptr(), n, b.data_ptr(), n, c1.data_ptr(), n, False, stream)
                    settle(whole)
                    for _ in range(args.warmup):
                        whole()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        whole()
                    torch.cuda.synchronize()
                    single_ms = (time.perf_counter() - t1) * 1e3 / args.steps
                    del a1, c1
                except Exception:      # (out of memory beside the panel buffers: report none rather than fail the run)
                    single_ms = None
            dist.barrier()

    # --sweep: the reference's square sweep under this run's sharding (row panels over the ranks, B replicated
    # before the timed launches -- data placement, as above).  Every rank times its own panel with a hipEvent pair
    # around K launches after a barrier; the slowest rank sets the rate.
    sweep_sharded = None
    if args.sweep:
        sweep_sharded = {}
        from how_to_optimize_gemm_amd.shard import RowPanelShard
        sizes = [p for p in list(range(1024, 4097, 128)) + [8192, 16384] if p <= n]
        for p in sizes:
            shp = RowPanelShard(p, p, p, rank, world)
            pa = a[:max(shp.rows, 1), :p].contiguous()
            pb = b[:p, :p].contiguous()
            pc = torch.empty((max(shp.rows, 1), p), device=dev)
            if dist:
                dist.barrier()
            ms = 0.0
            if shp.rows:
                ms = mm.time_sgemm(shp.rows, p, p, pa.data_ptr(), p, pb.data_ptr(), p, pc.data_ptr(), p, warmup=args.warmup,
                                   reps=args.steps, stream=stream)
            if dist:
                t = torch.tensor([ms], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms = float(t.item())
            sweep_sharded[str(p)] = round(2.0 * p ** 3 * 1e-9 / (ms * 1e-3), 1) if ms else None
            del pa, pb, pc
    # Correctness of what was just timed, outside the timed region:
    #  (1) the FULL C of the timed kernel is bit-equal to the plain one-workgroup-per-128x128-tile
    #      launch (mfma_tiles) -- the configuration the parity tests pin to the oracle element by
    #      element at this size (tests/test_gpu_parity.py::test_headline_size_4096);
    #  (2) sampled rows against an fp64 contraction.
    bit_equal = None
    if rows:
        mm.set_kernel("mfma_tiles")
        c_ref = torch.empty_like(c)
        mm.sgemm(rows, n, n, a.data_ptr(), n, b.data_ptr(), n, c_ref.data_ptr(), n, False, stream)
        mm.set_kernel(args.kernel)
        bit_equal = bool(torch.equal(c, c_ref))
        assert bit_equal, f"rank {rank}: the timed kernel's C is not bit-equal to the 128x128-tile launch"
        del c_ref
        idx = torch.tensor([0, rows // 2, rows - 1], device=dev)
        want = a[idx].double() @ b.double()
        err = float((c[idx].double() - want).abs().max())
        assert err < 2e-7 * n + 1e-6, f"rank {rank}: sampled-row check failed ({err})"

    streamed_equal = True
    if sharded and rows and overlap_ms is not None:
        streamed_equal = bool(torch.equal(c_stream, c))
        if dist:
            flag = torch.tensor([1 if streamed_equal else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            streamed_equal = bool(flag.item())
    launch_flops = 2.0 * rows * n * n
    achieved = launch_flops / (kern_ms * 1e-3) / 1e12 if kern_ms else 0.0

    out = None
    if rank == 0:
        def tf(ms):
            return round(2.0 * rows * n * n / (ms * 1e-3) / 1e12, 2) if ms else None
        ref20 = sum(trace[:20]) / 20 if len(trace) >= 21 else None            # the reference's 20 launches, cold
        cold20 = sum(trace[1:21]) / 20 if len(trace) >= 21 else None          # ... without launch #1's one-offs
        tail = sorted(trace[-50:])[len(trace[-50:]) // 2] if len(trace) >= 100 else None
        settled = next((i + 1 for i, ms in enumerate(trace) if tail and ms <= 1.01 * tail), None)
        out = {
            "metric": METRIC, "value": round(gflops, 1), "unit": "GFLOPS", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic uniform [-1,1) fp32, seeded on device",
            "config": {"workload": workload, "m": m, "n": n, "k": n, "kernel": H.kernel_name(mm.get_kernel()),
                       "parallelism": parallelism, "rows_per_rank": rows},
            "ramp_launches": len(trace), "untimed_launches": len(trace) + settle_launches + args.warmup,
            "cold": {
                "what": f"per-launch hipEvent trace of this process's first {len(trace)} launches (rank 0's panel)",
                "launch_1_ms": round(trace[0], 4) if trace else None,
                "reference_convention_20_launches_no_warmup_tflops": tf(ref20),
                "launches_2_to_21_tflops": tf(cold20),
                "launches_2_to_21_pct_of_peak": round(100.0 * tf(cold20) / PEAK_FP32_MFMA_TFLOPS, 2) if cold20 else None,
                "sustained_median_last_50_tflops": tf(tail),
                "launches_until_within_1pct_of_sustained": settled,
            },
            "value_cold": round(2.0 * m * n * n * 1e-9 / (cold20 * 1e-3), 1) if cold20 else None,   # launches 2..21
            "pct_of_fp32_mfma_peak": round(100.0 * gflops / (world * PEAK_FP32_MFMA_TFLOPS * 1e3), 2),
            "bit_equal_to_128x128_tile_launch": bit_equal,
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         "traffic": pmc_traffic(n) if not sharded else None,
                         "traffic_source": "profiles/pmc_traffic.json (committed rocprofv3 --pmc pass of this command; "
                                           "not re-measured in-run)" if not sharded else None,
                         "kernel": launched,
                         "kernel_ms": round(kern_ms, 4),
                         "algorithmic_flops_per_launch": launch_flops,
                         "algorithmic_bytes_per_launch": 4.0 * (rows * n + n * n + rows * n)},
        }
        if sharded:
            out["settle_launches"] = settle_launches          # untimed, in front of the W warm-ups (and of the single-GPU reference's)
            out["rccl_ranks"] = dist.get_world_size()
            out["backend"] = str(dist.get_backend())
            out["bcast_ms"] = round(bcast_ms, 3)
            out["bcast_gbps"] = round(4.0 * n * n / (bcast_ms * 1e-3) / 1e9, 1) if bcast_ms else None
            out["gemm_ms_per_rank"] = per_rank_ms               # each rank's own hipEvent time per launch of its panel
            out["gemm_ms"] = max(per_rank_ms) if per_rank_ms else None
            if single_ms:
                v1 = 2.0 * m * n * n * 1e-9 / (single_ms * 1e-3)
                out["single_gpu_value"] = round(v1, 1)          # the whole problem on rank 0's GPU, same process group, same W / K / clock as `value`
                out["single_gpu_ms"] = round(single_ms, 4)
                out["scaling_efficiency"] = round(gflops / (world * v1), 4)
            # what the committed one-GPU dry run (profiles/r04_shard_dryrun.md) predicts for this line: every rank's
            # panel at the single-GPU rate, the broadcast at one xGMI link (153 GB/s) flat or over all links
            # (scatter + all-gather), the streamed form hiding all but one chunk of it
            link = 153e9
            flat_ms = 4.0 * n * n / link * 1e3
            out["model"] = {"what": "predictions beside the measurements: value = N x the one-GPU panel rate; broadcast of B over one "
                                    "153 GB/s xGMI link flat, or scattered over the N-1 links and all-gathered; streamed = GEMM + one chunk",
                            "bcast_flat_ms": round(flat_ms, 3) if world > 1 else 0.0,
                            "bcast_scatter_allgather_ms": round(2.0 * flat_ms / max(world - 1, 1) * (world - 1) / world, 3) if world > 1 else 0.0,
                            "b_chunks": args.b_chunks,
                            "streamed_ms": round(ms_per_step + (flat_ms / max(args.b_chunks, 1) if world > 1 else 0.0), 3),
                            "value_at_linear_scaling": round(world * out["single_gpu_value"], 1) if single_ms else None}
            out["value_incl_bcast"] = round(2.0 * m * n * n * 1e-9 / ((ms_per_step + bcast_ms) * 1e-3), 1)
            if overlap_ms is not None:
                out["bcast_overlapped_ms"] = round(overlap_ms, 3)
                out["value_incl_bcast_overlapped"] = round(2.0 * m * n * n * 1e-9 / (overlap_ms * 1e-3), 1)
            elif streamed_error:
                out["streamed_b_error"] = streamed_error
            out["streamed_equals_plain"] = bool(streamed_equal)
        if sweep_sharded is not None:
            out["sweep_gflops_sharded"] = {"what": f"square sweep, C row panels over {world} rank(s), kernel-only "
                                                   f"(B replicated beforehand), max over ranks of the mean of {args.steps} launches",
                                           "pct_of_peak_at_4096": (round(100.0 * sweep_sharded["4096"] / (world * PEAK_FP32_MFMA_TFLOPS * 1e3), 2)
                                                                   if sweep_sharded.get("4096") else None),
                                           "gflops": sweep_sharded}
        if not sharded and not args.no_extras:
            extras = {}
            try:
                # configs[1]/[2]: the square sweep at the BASELINE sizes, LDS-tiled VALU kernel
                # (K1), the MFMA kernel as shipped (AUTO: tile choice + stream-K), AUTO with the
                # opt-in split-K, rocBLAS and hipBLASLt
                sweep = {}
                SWEEP = list(SWEEP_SIZES)                       # cuda/parameters.h:5-7: the metric's 25 sizes
                FEW = (1024, 1536, 2048, 3072, 4096)
                # valu = configs[1] (LDS-tiled, MFMA-free); mfma_tiles / mfma_128x128_dma5 = configs[2]'s literal tile (one
                # workgroup per 128x128 tile: register-staged, and by loader waves' LDS-DMA) at its own size
                # (rocblas / hipblaslt here = the copies inside torch's wheel, a handful of sizes under their own keys; the
                # sweep's vendor rows come from the C++ harness below -- see vendor_sweep_via_harness)
                for kern in ("auto", "rocblas", "hipblaslt", "valu", "auto_splitk", "mfma_tiles", "mfma_128x128_dma5"):
                    if kern not in ("rocblas", "hipblaslt"):
                        mm.set_kernel("auto" if kern == "auto_splitk" else kern)
                        mm.set_splitk(1 if kern == "auto_splitk" else 0)
                    sizes = SWEEP if kern in ("auto", "valu") else (4096,) if kern.startswith("mfma_") else FEW
                    for p in sizes:
                        if p > n or (kern == "auto_splitk" and p >= 2048):
                            continue
                        pa, pb = (a, b) if p == n else (a[:p, :p].contiguous(), b[:p, :p].contiguous())
                        pc = torch.empty((p, p), device=dev)
                        if kern in ("rocblas", "hipblaslt"):   # the vendor comparators, behind the same C ABI
                            try:                                # (cuda/MMult_cuBLAS_1.cpp, cuda/MMult_cuBLAS_2.cpp)
                                ms = mm.time_comparator(kern, p, p, p, pa.data_ptr(), p, pb.data_ptr(), p, pc.data_ptr(), p,
                                                        warmup=3, reps=10, stream=stream)
                                # (50 ms of untimed calls, the harness's WARMUP_MS: hipBLASLt's stream-K launches read 15-20 % low
                                # behind 10-15 ms -- 2560: 120.5 against 148.4 TFLOP/s in the harness sweep; the comparators get
                                # the longer lead-in, not the kernel under test)
                                warm = max(3, int(50.0 / max(ms, 1e-3)))
                                ms = min(mm.time_comparator(kern, p, p, p, pa.data_ptr(), p, pb.data_ptr(), p, pc.data_ptr(), p,
                                                            warmup=warm, reps=20, stream=stream) for _ in range(3))
                            except H.MMultError:
                                continue
                        else:
                            # sustained, like `value`: a first burst sizes ~15 ms of untimed launches in front of the timed
                            # ones (three warm-ups of a 30 us kernel leave the chip at its idle clock: round 4's extras read
                            # the VALU rung 15 % and `auto` 5 % under their sustained rates at N = 1024)
                            ms = mm.time_sgemm(p, p, p, pa.data_ptr(), p, pb.data_ptr(), p, pc.data_ptr(), p,
                                               warmup=3, reps=10, stream=stream)
                            warm = max(3, int(15.0 / max(ms, 1e-3)))
                            ms = min(mm.time_sgemm(p, p, p, pa.data_ptr(), p, pb.data_ptr(), p, pc.data_ptr(), p,
                                                   warmup=warm, reps=20, stream=stream) for _ in range(3))   # best of three bursts
                        key = f"{kern}_in_process_torch_bundled_{p}" if kern in ("rocblas", "hipblaslt") else f"{kern}_{p}"
                        sweep[key] = round(2.0 * p ** 3 * 1e-9 / (ms * 1e-3), 1)
                        del pc
                mm.set_splitk(0)
                mm.set_kernel(args.kernel)
                extras["sweep_gflops"] = sweep
                extras["sweep_summary"] = sweep_summary(sweep)     # (again below, once the vendor rows are in)
                extras["probe_mfma_f32_tflops"] = round(mm.probe_mfma_f32(), 1)
                # the vector ALU's own roof (a bare v_pk_fma_f32 loop, 2 / 3 / 4 waves per SIMD): the denominator of the `valu_*` rows
                extras["probe_valu_pk_fma_f32_tflops"] = {f"{w}_waves_per_simd": round(mm.probe_valu_f32(True, w), 1) for w in (2, 3, 4)}
                extras["probe_hbm_copy_gbps"] = round(mm.probe_hbm_copy(1 << 30), 1)
                extras["probe_hbm_read_gbps"] = round(mm.probe_hbm_read(1 << 30), 1)
                extras["probe_lds_read_gbps"] = {w: round(mm.probe_lds_read(v), 1) for w, v in
                                                 (("b128", 16), ("b64", 8), ("b32", 4), ("b64_tr_b8", -8))}
                # configs[4]: int8 x int8 -> int32 at N = 4096 and 8192 (end to end), with its own roofline object
                try:
                    r8 = int8_roofline(mm, torch, dev, H)
                    extras["int8_roofline"] = r8
                    extras["int8_4096_tops"], extras["int8_8192_tops"] = r8["achieved"], r8["achieved_8192"]
                except H.MMultError:
                    pass
            except Exception as e:   # extras are optional: never let them cost the run its JSON line
                extras["error"] = f"{type(e).__name__}: {e}"[:300]
            out["extras"] = extras
        # The counter passes (three child processes under rocprofv3, GPU) run WHILE the host times REF_MMult (one core):
        # neither needs what the other uses, and the run stays inside its budget.
        pmc_thread, pmc_box = None, {}
        if not sharded and not args.no_live_traffic and not args.no_extras and "ROCPROFILER_" not in " ".join(os.environ):
            # (not under a profiler already: gpu_profile.sh runs this script under rocprofv3)
            import threading
            torch.cuda.synchronize()
            def background():
                pmc_box.update(zip(("live", "how"), live_counters(n, args.kernel)))
                for lib in ("rocblas", "hipblaslt"):        # (GPU work again: behind the counter passes, not beside them)
                    pmc_box[lib] = vendor_sweep_via_harness(lib, [p for p in SWEEP_SIZES if p <= n])
            pmc_thread = threading.Thread(target=background)
            pmc_thread.start()
        if not sharded and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n)
            except Exception as e:   # reported, never fatal (the contract wants the object; say why it is missing)
                out["cpu_baseline"] = {"value": None, "unit": "GFLOPS", "cores": 0, "kind": "reference",
                                       "sample": f"failed: {type(e).__name__}: {e}"[:300]}
        if pmc_thread is not None:
            pmc_thread.join()
            live, how = pmc_box.get("live") or {}, pmc_box.get("how", "")
            rl = out["roofline"]
            if live.get("traffic") is not None:
                rl["traffic_committed_pass"] = rl["traffic"]
                rl["traffic"], rl["traffic_source"] = live["traffic"], how
            else:
                rl["traffic_live_failed"] = how
            sw = out.get("extras", {}).get("sweep_gflops")
            if sw is not None:
                src = {}
                for lib in ("rocblas", "hipblaslt"):
                    rows = pmc_box.get(lib) or {}
                    for p_, v_ in rows.items():
                        sw[f"{lib}_{p_}"] = v_
                    src[lib] = ("harness/test_MMult.x KERNEL=%s REF=skip WARMUP_MS=50 TRIALS=3 (a C++ process: the image's ROCm "
                                "library, not the copy inside torch's wheel)" % lib) if rows else "unavailable (harness missing or failed)"
                    if not rows:        # fall back to the in-process rows where there are any, and say so
                        for key in [k_ for k_ in sw if k_.startswith(f"{lib}_in_process_torch_bundled_")]:
                            sw[f"{lib}_{key.rsplit('_', 1)[1]}"] = sw[key]
                out["extras"]["vendor_rows_source"] = src
                out["extras"]["sweep_summary"] = sweep_summary(sw)
            rl["mfma_busy_frac"] = live.get("mfma_busy_frac")
            rl["effective_clock_ghz_under_counters"] = live.get("effective_clock_ghz")
            rl["kernel_us_under_counters"] = live.get("kernel_us_under_counters")
        if not sharded:
            rl = out["roofline"]
            rl.setdefault("mfma_busy_frac", None)
            # HBM-side GB/s of the timed kernel: the counters' bytes per launch over the launch's duration
            rl["hbm_gbps"] = round(rl["traffic"] / (kern_ms * 1e-3) / 1e9, 1) if (rl.get("traffic") and kern_ms) else None
            rl["hbm_frac_of_8tbps"] = round(rl["hbm_gbps"] / 8000.0, 4) if rl["hbm_gbps"] else None
    mm.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    # fd 1 stays parked on stderr until the process exits (RCCL prints its version banner as late as
    # library teardown): the ONE line goes straight to the saved descriptor
    sys.stdout.flush()
    if rank == 0:
        os.write(saved_stdout, (json.dumps(out) + "\n").encode())
    os.close(saved_stdout)


if __name__ == "__main__":
    main()
