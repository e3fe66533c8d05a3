"""Row-panel shard of one large SGEMM across the GPUs of a node: one process
per GPU, torch.distributed over RCCL/xGMI (backend "nccl"), or gloo on CPU
for the host-logic tests.

BASELINE.json config 4 -- there is no reference analogue (the reference pins
device 0, cuda/test_MMult.cpp:24-25).  C[i,:] = A[i,:] * B: rank r owns the
rows mmh_shard_rows(m, world, r) of A and C (128-row aligned, so every panel
is whole block tiles), B is replicated by ONE broadcast from the root; the
C panels are disjoint, so there is no reduction and no further exchange.
Because each C element is the same k-ordered fmaf chain wherever it is
computed, the sharded product is bit-identical to the single-GPU product.
"""
from __future__ import annotations

from typing import Callable, Optional

from . import api


class RowPanelShard:
    def __init__(self, m: int, n: int, k: int, rank: int, world: int):
        self.m, self.n, self.k = m, n, k
        self.rank, self.world = rank, world
        self.row0, self.rows = api.shard_rows(m, world, rank)

    # -- the single exchange step ---------------------------------------------
    def broadcast_b(self, b, src: int = 0, chunks: int = 1, always: bool = False):
        """Replicate B (k x n) from `src` to every rank, in place.  `b` must be an
        allocated (k, n) tensor on every rank (contents only matter on src).
        chunks > 1 splits the broadcast along k so that a caller can overlap the
        first GEMM K-slices with the tail of the transfer; the default is the
        single collective the design calls for."""
        import torch.distributed as dist
        if self.world == 1 and not always:     # `always`: exercise the collective on one rank (tests)
            return b
        if chunks <= 1:
            dist.broadcast(b, src=src)
        else:
            step = (self.k + chunks - 1) // chunks
            for k0 in range(0, self.k, step):
                dist.broadcast(b[k0:k0 + step], src=src)
        return b

    # -- the independent unit of work ------------------------------------------
    def local_gemm(self, gemm: Callable, a_panel, b, c_panel):
        """c_panel (rows x n) = a_panel (rows x k) @ b.  `gemm(a, b, out)` is the
        product kernel (MMult.matmul on GPUs)."""
        if self.rows == 0:
            return c_panel
        return gemm(a_panel, b, c_panel)

    # -- exchange overlapped with compute ---------------------------------------------
    def gemm_with_streamed_b(self, gemm: Callable, a_panel, b, c_panel, src: int = 0, chunks: int = 8,
                             always: bool = False):
        """The same product with B's broadcast hidden behind the GEMM: B travels in
        `chunks` K-chunks (asynchronous broadcasts, RCCL's own stream) while the
        chunks that have already landed are consumed,
            C  = A[:, k0:k1] @ B[k0:k1]        (first chunk, overwrite)
            C += A[:, k1:k2] @ B[k1:k2] ...    (accumulate: C's value continues each chain)
        Every C element is still one chain over ascending k, so the result is
        bit-identical to broadcast-then-GEMM.  `gemm(a, b, out, accumulate)`.
        Chunks are whole 32-deep K-slices, so `chunks` larger than k/32 is reduced to that; k == 0
        is the empty contraction -- one overwrite call, which zeroes C exactly as mmh_sgemm does."""
        import torch.distributed as dist
        step = -(-self.k // max(chunks, 1))
        step = max(32, (step + 31) // 32 * 32)             # whole K-slices per chunk
        bounds = [(k0, min(k0 + step, self.k)) for k0 in range(0, self.k, step)]
        if not bounds:                                     # k == 0: nothing to broadcast, C = 0
            if self.rows:
                gemm(a_panel, b, c_panel, False)
            return c_panel
        works = []
        if self.world > 1 or always:
            works = [dist.broadcast(b[k0:k1], src=src, async_op=True) for (k0, k1) in bounds]
        for i, (k0, k1) in enumerate(bounds):
            if works:
                works[i].wait()                            # orders the current stream after chunk i
            if self.rows:
                gemm(a_panel[:, k0:k1], b[k0:k1], c_panel, i > 0)
        return c_panel

    # -- verification helper (tests / smoke only; not on the hot path) ----------
    def gather_c(self, c_panel, like):
        """Assemble the full C on every rank (all_gather of the padded panels)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return c_panel
        max_rows = max(api.shard_rows(self.m, self.world, r)[1] for r in range(self.world))
        pad = torch.zeros((max_rows, self.n), dtype=c_panel.dtype, device=c_panel.device)
        pad[:self.rows] = c_panel
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(parts, pad)
        full = torch.empty((self.m, self.n), dtype=like.dtype, device=c_panel.device)
        for r in range(self.world):
            r0, rows = api.shard_rows(self.m, self.world, r)
            full[r0:r0 + rows] = parts[r][:rows]
        return full
