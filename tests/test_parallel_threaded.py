"""witw_amd.parallel / cvig_fov.sharded_match_loss / retrieve under a backend that is not gloo: 8 rank-threads on torch's
in-process 'threaded' group (tests/threaded_world.py), CPU kernels through the `_kernels=` hook. The branches taken are the ones
the RCCL job takes -- dist.all_gather_into_tensor and dist.reduce_scatter_tensor (witw_amd/parallel.py:39, :260) -- which the
gloo tests (test_parallel_gloo.py, test_parallel_world8_gloo.py) never reach. Reference semantics: the DataParallel global
batch of model/cvig_baseline.py:339-343 with the normaliser of model/cvig_fov.py:380."""
import numpy as np
import torch
import torch.distributed as dist

from oracle import cvig_fov_oracle as O
from witw_amd import synth

from .test_parallel_world8_gloo import CpuKernels, Fp64MatchKernels, fused_match
from .threaded_world import run_ranks

WORLD = 8


def test_collective_wrappers_take_the_non_gloo_branches():
    from witw_amd import parallel

    def fn(rank):
        assert dist.get_backend() == 'threaded' and parallel.world() == WORLD and parallel.rank() == rank
        t = torch.full((3, 2), float(rank))
        cat = parallel._all_gather_cat(t)
        rows = torch.arange(WORLD * 2 * 5, dtype=torch.float32).reshape(WORLD * 2, 5) * (rank + 1)
        mine = parallel.reduce_scatter_rows(rows.clone(), 2)
        rag = parallel.all_gather_ragged(torch.full((rank % 3, 2), float(rank)))
        c = torch.tensor([rank], dtype=torch.int32)
        parallel.all_reduce_sum_(c)
        return cat, mine, rag, int(c)
    res = run_ranks(WORLD, fn)
    scale = sum(range(1, WORLD + 1))
    base = torch.arange(WORLD * 2 * 5, dtype=torch.float32).reshape(WORLD * 2, 5)
    for rank, (cat, mine, rag, c) in enumerate(res):
        assert torch.equal(cat[:, 0], torch.arange(WORLD).repeat_interleave(3).float())
        assert torch.equal(mine, base[2 * rank:2 * rank + 2] * scale)
        assert torch.equal(rag[:, 0], torch.cat([torch.full((r % 3,), float(r)) for r in range(WORLD)]))
        assert c == sum(range(WORLD))


def test_sharded_match_loss_on_the_rccl_branches():
    """B = 256 over 8 ranks (b = 32), We = 24: loss, slabs and the gradients of every rank's own embeddings equal the
    single-process full-matrix result."""
    from witw_amd import cvig_fov
    B, b, we = 256, 32, 24
    ov = torch.from_numpy(synth.embeddings(51, 1, (B, 16, 4, 64)))
    su = (torch.roll(ov, -3, dims=3)[:, :, :, :we] + 2.0 * torch.from_numpy(synth.embeddings(51, 2, (B, 16, 4, we)))).contiguous()

    def fn(rank):
        sl = slice(rank * b, (rank + 1) * b)
        ov_l, su_l = ov[sl].clone().requires_grad_(True), su[sl].clone().requires_grad_(True)
        loss, ori, d = cvig_fov.sharded_match_loss(ov_l, su_l, _kernels=CpuKernels)
        loss.backward()
        return loss.item(), ov_l.grad, su_l.grad, ori, d
    res = run_ranks(WORLD, fn)
    ovf, suf = ov.clone().requires_grad_(True), su.clone().requires_grad_(True)
    ori_f, d_f = fused_match(ovf, suf)
    loss_f = O.triplet_loss(d_f)
    loss_f.backward()
    for rank, (loss, g_ov, g_su, ori, d) in enumerate(res):
        sl = slice(rank * b, (rank + 1) * b)
        assert abs(loss - loss_f.item()) <= 2e-6 * abs(loss_f.item())
        assert torch.equal(ori, ori_f[:, sl])
        np.testing.assert_allclose(d.numpy(), d_f.detach()[:, sl].numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(g_ov.numpy(), ovf.grad[sl].numpy(), rtol=0, atol=1e-5 * float(ovf.grad[sl].abs().max()))
        np.testing.assert_allclose(g_su.numpy(), suf.grad[sl].numpy(), rtol=0, atol=1e-5 * float(suf.grad[sl].abs().max()))


def test_sharded_retrieve_on_the_rccl_branches():
    """ragged gallery shards (one empty), k = 5: ranks and top-k equal the unsharded call bit for bit."""
    from witw_amd import cvig_fov
    split = [40, 0, 77, 13, 64, 1, 90, 18]
    G, Q, we = sum(split), 60, 12
    gal = torch.from_numpy(synth.embeddings(52, 1, (G, 16, 4, 64)))
    gal[250] = gal[2]
    qry = (torch.stack([torch.roll(gal[i], -5 * i, dims=2)[:, :, :we] for i in range(Q)])
           + 2.0 * torch.from_numpy(synth.embeddings(52, 2, (Q, 16, 4, we)))).contiguous()

    def fn(rank):
        g0 = sum(split[:rank])
        return cvig_fov.retrieve(gal[g0:g0 + split[rank]], qry, k=5, shard_begin=g0, query_chunk=32, _kernels=Fp64MatchKernels)
    res = run_ranks(WORLD, fn)
    r1, v1, i1 = cvig_fov.retrieve(gal, qry, k=5, query_chunk=32, _kernels=Fp64MatchKernels)
    np.testing.assert_array_equal(r1, O.ranks(gal, qry))
    for (r, v, i) in res:
        np.testing.assert_array_equal(r, r1)
        assert torch.equal(v, v1) and torch.equal(i, i1)
