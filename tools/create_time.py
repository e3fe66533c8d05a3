#!/usr/bin/env python3
"""create_time.py -- wall time of mmh_create in a cold process (the role of cublasCreate in the reference's driver,
cuda/test_MMult.cpp:43-44): the HIP runtime is initialised first (a device allocation through torch), then mmh_create
is timed -- warmed (default) and with MMH_LAZY=1 -- and the first mmh_sgemm after it.  Each figure from a fresh
process, --runs times.  Needs a GPU."""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
import torch
x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
import how_to_optimize_gemm_amd as H
H.lib()
t0 = time.perf_counter(); mm = H.MMult(0, "auto"); torch.cuda.synchronize(); t1 = time.perf_counter()
n = 4096
a = torch.rand((n, n), device="cuda"); b = torch.rand((n, n), device="cuda"); c = torch.empty((n, n), device="cuda")
torch.cuda.synchronize()
t2 = time.perf_counter(); mm.matmul(a, b, out=c); torch.cuda.synchronize(); t3 = time.perf_counter()
mm.matmul(a, b, out=c); torch.cuda.synchronize(); t4 = time.perf_counter()
print(json.dumps({"create_ms": round((t1 - t0) * 1e3, 1), "first_sgemm_ms": round((t3 - t2) * 1e3, 2), "second_sgemm_ms": round((t4 - t3) * 1e3, 2)}))
mm.close()
''' % REPO


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=3)
    args = ap.parse_args()
    out = {}
    for label, env in (("warmed", {}), ("lazy", {"MMH_LAZY": "1"})):
        rows = []
        for _ in range(args.runs):
            e = dict(os.environ, **env)
            r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            rows.append(json.loads(line[-1]) if line else {"error": r.stderr[-300:]})
        out[label] = rows
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
