#!/usr/bin/env python3
"""shard_dryrun.py -- what every rank of the 8-GPU row-panel shard (BASELINE.json configs[3]) will run, timed on ONE GPU.

For G = 1, 2, 4, 8 the rank-0 panel of the N = 16384 problem (mmh_shard_rows: 16384 / G rows x 16384 x 16384, every
rank's panel has the same shape) goes through the same mmh_sgemm / MMH_KERNEL_AUTO call bench.py --gpus G times, and
the table lists the predicted whole-job rate G x (panel rate) beside the broadcast of B (1 GiB) under the xGMI model of
SURVEY.md section 8(e): a flat root -> 7 peers broadcast is bound by one link (~153 GB/s, all links in parallel);
a scatter + all-gather form moves 1/7 of B per link and phase.  No multi-GPU hardware is involved: this is the
single-GPU half of the scaling curve, the other half (RCCL over xGMI) is the model.  Writes markdown to stdout."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
LINK_GBPS = 153.0
mm = H.MMult(0, "auto")
stream = torch.cuda.current_stream().cuda_stream
b = torch.rand((N, N), device="cuda") * 2 - 1
a = torch.rand((N, N), device="cuda") * 2 - 1
c = torch.empty((N, N), device="cuda")
bytes_b = 4.0 * N * N
print(f"| G | rows per rank | panel ms | panel TFLOP/s | % of fp32 MFMA peak | predicted job GFLOPS (kernel only) | "
      f"bcast of B: flat {LINK_GBPS:.0f} GB/s per link | scatter + all-gather | predicted incl. flat bcast | launched |")
print("|---|---|---|---|---|---|---|---|---|---|")
rows_out = []
for G in (1, 2, 4, 8):
    row0, rows = H.shard_rows(N, G, 0)
    ms = min(mm.time_sgemm(rows, N, N, a.data_ptr(), N, b.data_ptr(), N, c.data_ptr(), N, warmup=2, reps=5, stream=stream)
             for _ in range(2))
    tf = 2.0 * rows * N * N / (ms * 1e-3) / 1e12
    job = G * tf * 1e3
    flat = 0.0 if G == 1 else bytes_b / (LINK_GBPS * 1e9) * 1e3
    sag = 0.0 if G == 1 else 2.0 * (bytes_b / max(G - 1, 1)) / (LINK_GBPS * 1e9) * 1e3
    incl = 2.0 * N ** 3 / ((ms + flat) * 1e-3) / 1e9
    rows_out.append({"G": G, "rows": rows, "panel_ms": round(ms, 3), "panel_tflops": round(tf, 1), "job_gflops": round(job, 0),
                     "bcast_flat_ms": round(flat, 2), "bcast_sag_ms": round(sag, 2), "launched": H.last_launch()})
    print(f"| {G} | {rows} | {ms:.2f} | {tf:.1f} | {100 * tf / 157.3:.1f} | {job:,.0f} | {flat:.1f} ms | {sag:.1f} ms | {incl:,.0f} | "
          f"{H.last_launch()[:60]} |")
print()
print("```json")
print(json.dumps(rows_out))
print("```")
