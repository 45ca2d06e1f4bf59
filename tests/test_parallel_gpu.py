"""Two processes on ONE MI355X (gloo rendezvous, both ranks on cuda:0): the column-sharded global-batch loss on the HIP
kernels equals the full-batch match + triplet_loss — value and gradients w.r.t. each rank's own embeddings."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from witw_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, we, out_q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from witw_amd import cvig_fov
        torch.cuda.set_device(0)
        B, b = 48, 24
        ov = torch.from_numpy(synth.embeddings(95, 1, (B, 16, 4, 64))).cuda()
        su = torch.stack([torch.roll(ov[i], -int(5 * i % 64), dims=2)[:, :, :we] for i in range(B)]).contiguous() \
            + 1.5 * torch.from_numpy(synth.embeddings(95, 2, (B, 16, 4, we))).cuda()
        # full batch on this process
        ov_f, su_f = ov.clone().requires_grad_(True), su.clone().requires_grad_(True)
        ori_f, d_f = cvig_fov.match(ov_f, su_f)
        loss_f = cvig_fov.triplet_loss(d_f)
        loss_f.backward()
        # sharded: this rank owns pairs [rank*b, rank*b+b)
        sl = slice(rank * b, rank * b + b)
        ov_l, su_l = ov[sl].clone().requires_grad_(True), su[sl].clone().requires_grad_(True)
        loss, ori, d = cvig_fov.sharded_match_loss(ov_l, su_l)
        loss.backward()
        out_q.put((rank, loss.item(), loss_f.item(),
                   float((ov_l.grad - ov_f.grad[sl]).abs().max()), float(ov_f.grad[sl].abs().max()),
                   float((su_l.grad - su_f.grad[sl]).abs().max()), float(su_f.grad[sl].abs().max()),
                   bool(torch.equal(ori, ori_f[:, sl])), float((d - d_f.detach()[:, sl]).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('we', [64, 12])
def test_sharded_match_loss_two_ranks_one_gpu(we):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, we, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for (rank, loss, loss_full, dgo, go, dgs, gs, ori_equal, dd) in res:
        assert abs(loss - loss_full) <= 2e-6 * max(1.0, abs(loss_full)), (rank, loss, loss_full)
        assert ori_equal and dd == 0.0                       # same kernel on a column subset: bit-identical slab
        assert dgo <= 1e-5 * go + 1e-9 and dgs <= 1e-5 * gs + 1e-9, (rank, dgo, go, dgs, gs)
