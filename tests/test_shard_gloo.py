"""world_size-2 (and 3) gloo tests of the multi-GPU host logic on CPU: the
row-panel plan, the single broadcast of B, and the reassembly.  The local
GEMM is stood in for by the CPU oracle (tests may use it); on GPUs the same
RowPanelShard drives MMult.matmul (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, m, n, k, chunks, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from how_to_optimize_gemm_amd.shard import RowPanelShard
        from oracle import oracle as O
        a, b_full = O.harness_inputs(m, n, k, seed=4321)        # same on every rank (seeded)
        sh = RowPanelShard(m, n, k, rank, world)
        a_panel = torch.from_numpy(a[sh.row0:sh.row0 + sh.rows].copy())
        b = torch.from_numpy(b_full.copy()) if rank == 0 else torch.full((k, n), float("nan"))
        sh.broadcast_b(b, src=0, chunks=chunks)
        assert torch.equal(b, torch.from_numpy(b_full)), "B did not arrive intact"

        def gemm(x, y, out):
            out.copy_(torch.from_numpy(O.ref_mmult(x.numpy(), y.numpy(), fma=True)))
            return out

        c_panel = torch.empty((sh.rows, n))
        sh.local_gemm(gemm, a_panel, b, c_panel)
        full = sh.gather_c(c_panel, like=b)
        want = O.ref_mmult(a, b_full, fma=True)
        ok = bool(np.array_equal(full.numpy(), want))

        # the streamed form: B arrives in K-chunks, consumed with accumulate
        def gemm_acc(x, y, out, accumulate):
            c0 = out.numpy().copy() if accumulate else None
            out.copy_(torch.from_numpy(O.ref_mmult(np.ascontiguousarray(x.numpy()), y.numpy(), c0, fma=True)))
            return out

        b2 = torch.from_numpy(b_full.copy()) if rank == 0 else torch.full((k, n), float("nan"))
        c2 = torch.full((sh.rows, n), float("nan"))
        sh.gemm_with_streamed_b(gemm_acc, a_panel, b2, c2, src=0, chunks=3)
        ok = ok and torch.equal(b2, torch.from_numpy(b_full)) and torch.equal(c2, c_panel)
        q.put((rank, ok, sh.row0, sh.rows))
    finally:
        dist.destroy_process_group()


def test_streamed_form_with_an_empty_contraction_zeroes_c():
    """k == 0: nothing to broadcast; the streamed form makes the one overwrite call that mmh_sgemm
    answers with C = 0 (the same contract as local_gemm / mmh_sgemm), instead of leaving C as it was."""
    sys.path.insert(0, REPO)
    from how_to_optimize_gemm_amd.shard import RowPanelShard
    sh = RowPanelShard(256, 8, 0, 0, 1)
    calls = []

    def gemm(x, y, out, accumulate):
        calls.append((tuple(x.shape), tuple(y.shape), accumulate))
        if not accumulate:
            out.zero_()
        return out

    c = torch.full((256, 8), float("nan"))
    sh.gemm_with_streamed_b(gemm, torch.empty((256, 0)), torch.empty((0, 8)), c)
    assert calls == [((256, 0), (0, 8), False)] and torch.equal(c, torch.zeros_like(c))


@pytest.mark.parametrize("world,m,n,k,chunks", [(2, 256, 96, 64, 1), (2, 300, 72, 40, 1),
                                               (3, 520, 64, 48, 2), (2, 100, 33, 17, 1)])
def test_row_panel_shard_over_gloo(world, m, n, k, chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, n, k, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in results)
    assert sum(rows for _, _, _, rows in results) == m


def _gpu_worker(rank, world, port, m, n, k, q):
    """The same plan with the REAL local GEMM: every rank drives the HIP kernel on cuda:0 (fewer
    devices than ranks), the exchange runs over gloo on host copies."""
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import how_to_optimize_gemm_amd as H
        from how_to_optimize_gemm_amd.shard import RowPanelShard
        from oracle import oracle as O
        a, b_full = O.harness_inputs(m, n, k, seed=99)
        sh = RowPanelShard(m, n, k, rank, world)
        b = torch.from_numpy(b_full.copy()) if rank == 0 else torch.full((k, n), float("nan"))
        sh.broadcast_b(b, src=0)
        mm = H.MMult(0)
        da = torch.from_numpy(a[sh.row0:sh.row0 + sh.rows].copy()).cuda()
        db = b.cuda()

        def gemm(x, y, out):
            return mm.matmul(x, y, out=out)

        c_panel = torch.empty((sh.rows, n), device="cuda")
        sh.local_gemm(gemm, da, db, c_panel)
        full = sh.gather_c(c_panel.cpu(), like=b)
        ok = bool(np.array_equal(full.numpy(), O.ref_mmult(a, b_full, fma=True)))
        # and the single-device product of the same inputs has the same bits
        if rank == 0:
            whole = mm.matmul(torch.from_numpy(a).cuda(), db).cpu()
            ok = ok and bool(torch.equal(whole, full))
        q.put((rank, ok, sh.row0, sh.rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,m,n,k", [(2, 512, 384, 256), (3, 1000, 200, 136)])
def test_row_panel_shard_degrades_to_ranks_sharing_one_gpu(world, m, n, k):
    """the shard plan with more ranks than devices -- every rank runs the HIP GEMM on
    cuda:0, B travels over gloo -- reassembles the bits of the single-device product."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, m, n, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in results)
    assert sum(rows for _, _, _, rows in results) == m
