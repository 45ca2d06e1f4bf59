"""world_size-2 CPU (gloo) tests of the multi-GPU host logic: global-batch loss gradients equal the
single-process (nn.DataParallel-equivalent) gradients, and sharded-gallery ranking equals the unsharded one."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cvig_fov_oracle as O
from witw_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_encoders(we):
    torch.manual_seed(5)
    su_enc = torch.nn.Linear(24, 16 * 4 * we)
    ov_enc = torch.nn.Linear(24, 16 * 4 * 64)
    return su_enc, ov_enc


def _full_reference(xs, xo, we):
    su_enc, ov_enc = _toy_encoders(we)
    su = su_enc(xs).view(-1, 16, 4, we)
    ov = ov_enc(xo).view(-1, 16, 4, 64)
    _, d = O.match(ov, su)
    loss = O.triplet_loss(d)
    loss.backward()
    return loss.item(), [p.grad.clone() for p in list(su_enc.parameters()) + list(ov_enc.parameters())]


def _worker(rank, world, port, we, out_q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from witw_amd import cvig_fov, parallel
        torch.set_num_threads(2)
        B = 6
        xs = torch.from_numpy(synth.embeddings(3, 1, (B, 24)))
        xo = torch.from_numpy(synth.embeddings(3, 2, (B, 24)))
        b0, b1 = parallel.shard_range(B)
        su_enc, ov_enc = _toy_encoders(we)
        su = su_enc(xs[b0:b1]).view(-1, 16, 4, we)
        ov = ov_enc(xo[b0:b1]).view(-1, 16, 4, 64)
        su_all, ov_all = parallel.all_gather_embeddings(su, ov)
        assert su_all.shape[0] == B and ov_all.shape[0] == B
        _, d = O.match(ov_all, su_all)
        loss = O.triplet_loss(d)              # normaliser 2B(B-1) with the GLOBAL B
        loss.backward()
        params = list(su_enc.parameters()) + list(ov_enc.parameters())
        n = parallel.all_reduce_grads(params)
        assert n == sum(p.numel() for p in params)
        # sharded-gallery ranking, with the CPU oracle standing in for the HIP kernels
        N = 10
        gal = torch.from_numpy(synth.embeddings(4, 1, (N, 16, 4, 64)))
        qry = torch.stack([torch.roll(gal[i], -3 * i, dims=2)[:, :, :we] for i in range(N)]) \
            + 6.0 * torch.from_numpy(synth.embeddings(4, 2, (N, 16, 4, we)))
        g0, g1 = parallel.shard_range(N)
        r = cvig_fov.sharded_ranks(gal[g0:g1], qry, g0, query_chunk=4, _match=O.match,
                                   _count=lambda dd, thr: (dd <= thr[None, :]).sum(0).to(torch.int32))
        out_q.put((rank, loss.item(), [p.grad.numpy().copy() for p in params], np.asarray(r)))   # numpy: no shared-memory handles that die with the sender
    finally:
        dist.destroy_process_group()


def test_global_batch_gradients_and_sharded_ranks_world2():
    we = 12
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, we, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    B = 6
    xs = torch.from_numpy(synth.embeddings(3, 1, (B, 24)))
    xo = torch.from_numpy(synth.embeddings(3, 2, (B, 24)))
    loss_ref, grads_ref = _full_reference(xs, xo, we)
    for (_rank, loss, grads, _r) in res:
        assert abs(loss - loss_ref) < 1e-6
        for g, gr in zip(grads, grads_ref):
            np.testing.assert_allclose(np.asarray(g), gr.numpy(), rtol=1e-4, atol=1e-7)
    N = 10
    gal = torch.from_numpy(synth.embeddings(4, 1, (N, 16, 4, 64)))
    qry = torch.stack([torch.roll(gal[i], -3 * i, dims=2)[:, :, :we] for i in range(N)]) \
        + 6.0 * torch.from_numpy(synth.embeddings(4, 2, (N, 16, 4, we)))
    ranks_ref = O.ranks(gal, qry)
    assert ranks_ref.max() > 1
    for (_rank, _l, _g, r) in res:
        np.testing.assert_array_equal(r, ranks_ref)


class _CpuKernels(object):
    """The op set of cvig_fov.sharded_match_loss restated in CPU torch ops (oracle match; the slab formulas of
    witw_amd/csrc/loss.hip), so that the collective algebra runs under gloo without a GPU."""

    @staticmethod
    def match_fwd(ov, su, want_score=False, want_workspace=False):
        ori, dist = O.match(ov.detach(), su.detach())
        return ori, dist, torch.zeros_like(dist), torch.zeros(1)

    @staticmethod
    def match_bwd(ov, su, ori, score, ws, g_dist, need_ov=True, need_su=True):
        with torch.enable_grad():               # autograd.Function.backward runs with grad mode off
            ov = ov.detach().requires_grad_(True)
            su = su.detach().requires_grad_(True)
            _, d = O.match(ov, su)
            d.backward(g_dist)
        return ov.grad, su.grad

    @staticmethod
    def triplet_loss_slab_fwd(dist, diag, col0, alpha):
        b = dist.shape[1]
        t = torch.log(1 + torch.exp(alpha * (diag[col0:col0 + b][None, :] - dist))) + \
            torch.log(1 + torch.exp(alpha * (diag[:, None] - dist)))
        return t.sum().reshape(1)

    @staticmethod
    def triplet_loss_slab_sig(dist, diag, col0, alpha):
        b = dist.shape[1]
        rowsig = torch.sigmoid(alpha * (diag[:, None] - dist)).sum(1)
        colsig = torch.sigmoid(alpha * (diag[col0:col0 + b][None, :] - dist)).sum(0)
        return rowsig, colsig

    @staticmethod
    def triplet_loss_slab_bwd(dist, diag, rowsig, colsig, g_loss, col0, alpha):
        B, b = dist.shape
        g = -torch.sigmoid(alpha * (diag[col0:col0 + b][None, :] - dist)) - torch.sigmoid(alpha * (diag[:, None] - dist))
        idx = torch.arange(b)
        g[col0 + idx, idx] += colsig + rowsig[col0:col0 + b]
        return g * (g_loss * alpha / (2. * B * (B - 1)))


def _worker_sharded(rank, world, port, we, out_q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from witw_amd import cvig_fov, parallel
        torch.set_num_threads(2)
        B = 6
        xs = torch.from_numpy(synth.embeddings(3, 1, (B, 24)))
        xo = torch.from_numpy(synth.embeddings(3, 2, (B, 24)))
        b0, b1 = parallel.shard_range(B)
        su_enc, ov_enc = _toy_encoders(we)
        reducer = parallel.OverlappedGradReducer([su_enc, ov_enc])     # per-encoder buckets launched from grad hooks
        su = su_enc(xs[b0:b1]).view(-1, 16, 4, we)
        ov = ov_enc(xo[b0:b1]).view(-1, 16, 4, 64)
        loss, ori, d = cvig_fov.sharded_match_loss(ov, su, _kernels=_CpuKernels)
        assert tuple(d.shape) == (B, b1 - b0) and tuple(ori.shape) == (B, b1 - b0)
        loss.backward()
        params = list(su_enc.parameters()) + list(ov_enc.parameters())
        assert len(reducer.inflight) == 2                               # both all-reduces were started inside backward()
        assert reducer.wait() == sum(p.numel() for p in params)
        assert reducer.wait() == 0                                      # nothing left in flight
        out_q.put((rank, loss.item(), [p.grad.numpy().copy() for p in params]))
    finally:
        dist.destroy_process_group()


def test_sharded_match_loss_world2_equals_full_batch():
    """The column-sharded global-batch loss (one [B,b] slab per rank, reduce-scatter of the overhead-embedding gradients)
    gives the loss and the weight gradients of the single-process full-batch computation."""
    we = 12
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, we, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    B = 6
    xs = torch.from_numpy(synth.embeddings(3, 1, (B, 24)))
    xo = torch.from_numpy(synth.embeddings(3, 2, (B, 24)))
    loss_ref, grads_ref = _full_reference(xs, xo, we)
    for (_rank, loss, grads) in res:
        assert abs(loss - loss_ref) < 1e-6
        for g, gr in zip(grads, grads_ref):
            np.testing.assert_allclose(np.asarray(g), gr.numpy(), rtol=1e-4, atol=1e-7)


def test_shard_range_partitions():
    from witw_amd import parallel
    for n in (1, 7, 8, 1000003):
        for w in (1, 2, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
