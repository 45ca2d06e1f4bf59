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


def _train_worker(rank, world, port, root, out_q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['LOCAL_RANK'] = '0'                      # both ranks share the one GPU of the test box
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from witw_amd import cvig_fov
        os.chdir(root)
        best = cvig_fov.train(dataset='cvusa', fov=70, val_quantity=4, batch_size=2, num_workers=0, num_epochs=2,
                              csv_path=os.path.join(root, 'pairs.csv'), seed=3)
        cvig_fov.Globals.test_random_orientation = False
        table = cvig_fov.test(dataset='cvusa', fov=70, batch_size=4, num_workers=0, csv_path=os.path.join(root, 'pairs.csv'))
        out_q.put((rank, best, table))
    finally:
        dist.destroy_process_group()


def test_train_driver_two_ranks(tmp_path):
    """cvig_fov.train under torch.distributed (2 ranks): DistributedSampler shards, column-sharded global-batch loss,
    gradient all-reduce; both ranks report the same validation loss and only rank 0 writes the checkpoints."""
    from PIL import Image
    root = str(tmp_path)
    rows = []
    for i in range(12):
        Image.fromarray(synth.images_u8(96, i, (64, 64, 3)).astype(np.uint8)).save(os.path.join(root, 'ov_%d.png' % i))
        Image.fromarray(synth.images_u8(97, i, (48, 80, 3)).astype(np.uint8)).save(os.path.join(root, 'su_%d.png' % i))
        rows.append('ov_%d.png,su_%d.png' % (i, i))
    with open(os.path.join(root, 'pairs.csv'), 'w') as f:
        f.write('\n'.join(rows) + '\n')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, root, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] is not None and np.isfinite(res[0][1]) and res[0][1] == res[1][1]
    sd = torch.load(os.path.join(root, 'weights', 'fov_70_surface_best.pth'))
    assert 'model.features.27.weight' in sd
    # test() with the gallery sharded over the two ranks equals the single-process table on the same checkpoint
    assert res[0][2] == res[1][2]
    from witw_amd import cvig_fov
    cwd = os.getcwd()
    os.chdir(root)
    cvig_fov.Globals.test_random_orientation = False
    try:
        single = cvig_fov.test(dataset='cvusa', fov=70, batch_size=4, num_workers=0, csv_path=os.path.join(root, 'pairs.csv'))
    finally:
        os.chdir(cwd)
        del cvig_fov.Globals.test_random_orientation
    assert single == res[0][2]
