"""The 8-rank partitionings of BASELINE configs[2] (cvig_fov bs = 1024 global batch over 8 ranks) and configs[4] (gallery
retrieval over 8 gallery shards) on the HIP kernels AT CONFIG SIZES: 8 rank-threads of one process sharing the one MI355X
(tests/threaded_world.py; the box allows at most 6 GPU processes, and the threaded group makes witw_amd.parallel take its RCCL
branches: all_gather_into_tensor / reduce_scatter_tensor).

* config 3: cvig_fov.sharded_match_loss at B = 1024 / b = 128 / We = 64 on witw_match_fwd / witw_triplet_loss_slab_* /
  witw_match_bwd -- against the single-process full-matrix HIP result and against the oracle's fused match + O.triplet_loss
  (model/cvig_fov.py:297-382 under the DataParallel global batch of model/cvig_baseline.py:339-343; normaliser :380).
* config 5: cvig_fov.retrieve direct and method='dft' over 8 ragged HIP shards (one empty) against the unsharded call."""
import numpy as np
import pytest
import torch

from oracle import cvig_fov_oracle as O
from witw_amd import synth

from .test_parallel_world8_gloo import C3_B, C3_WE, C3_b, _c3_embeddings, fused_match
from .threaded_world import run_ranks

pytestmark = pytest.mark.gpu
WORLD = 8


def test_config3_sharded_match_loss_1024_over_8_rank_threads_on_hip():
    from witw_amd import cvig_fov
    dev = torch.device('cuda:0')
    ov, su = _c3_embeddings()
    ov_d, su_d = ov.to(dev), su.to(dev)

    def fn(rank):
        torch.cuda.set_device(dev)
        sl = slice(rank * C3_b, (rank + 1) * C3_b)
        ov_l, su_l = ov_d[sl].clone().requires_grad_(True), su_d[sl].clone().requires_grad_(True)
        loss, ori, d = cvig_fov.sharded_match_loss(ov_l, su_l)
        assert tuple(d.shape) == (C3_B, C3_b) and tuple(ori.shape) == (C3_B, C3_b)
        loss.backward()
        torch.cuda.synchronize()
        return loss.item(), ov_l.grad.cpu(), su_l.grad.cpu(), ori.cpu(), d.cpu()
    res = run_ranks(WORLD, fn)
    # single process, full [1024, 1024] matrix on the HIP kernels
    ov1, su1 = ov_d.clone().requires_grad_(True), su_d.clone().requires_grad_(True)
    ori1, d1 = cvig_fov.match(ov1, su1)
    loss1 = cvig_fov.triplet_loss(d1)
    loss1.backward()
    # the oracle: the reference's arithmetic on the CPU (fused form of correlation / crop_overhead / l2_distance, triplet_loss)
    # Orientation: the CPU's arg-max wherever its two best scores are further apart than fp32 summation rounding; on the handful
    # of pairs inside that band either shift is a correct arg-max, and the distance / gradient comparison below takes the shift
    # the HIP kernel took (one flipped pair moves 4096 gradient entries by its share of the loss, ~1e-3 of the largest entry).
    ovc, suc = ov.clone().requires_grad_(True), su.clone().requires_grad_(True)
    ori_c, gap = fused_match(ov, su, want_gap=True)
    clear_all = gap > 8e-6
    assert float(clear_all.float().mean()) > 0.99 and torch.equal(ori1.cpu()[clear_all], ori_c[clear_all])
    assert (ori_c.diagonal() == (5 * torch.arange(C3_B)) % 64).float().mean() > 0.9
    _, d_c = fused_match(ovc, suc, ori=ori1.cpu())
    loss_c = O.triplet_loss(d_c)
    loss_c.backward()
    assert abs(loss1.item() - loss_c.item()) <= 2e-6 * abs(loss_c.item())
    gov1, gsu1 = ov1.grad.cpu(), su1.grad.cpu()
    for rank, (loss, g_ov, g_su, ori, d) in enumerate(res):
        sl = slice(rank * C3_b, (rank + 1) * C3_b)
        assert abs(loss - loss1.item()) <= 2e-6 * abs(loss1.item()), (rank, loss, loss1.item())
        assert abs(loss - loss_c.item()) <= 2e-6 * abs(loss_c.item()), (rank, loss, loss_c.item())
        assert torch.equal(ori, ori1.cpu()[:, sl])                       # the slab kernel's choice = the full-matrix kernel's
        assert torch.equal(d, d1.detach().cpu()[:, sl])
        np.testing.assert_allclose(d.numpy(), d_c.detach()[:, sl].numpy(), rtol=0, atol=1e-5)
        for got, one, cpu in ((g_ov, gov1[sl], ovc.grad[sl]), (g_su, gsu1[sl], suc.grad[sl])):
            np.testing.assert_allclose(got.numpy(), one.numpy(), rtol=0, atol=1e-5 * float(one.abs().max()))
            np.testing.assert_allclose(got.numpy(), cpu.numpy(), rtol=0, atol=1e-5 * float(cpu.abs().max()))


C5_SPLIT = [3000, 1, 5001, 0, 2048, 4097, 777, 6079]


@pytest.mark.parametrize('we', [64, 24])
def test_config5_sharded_retrieve_over_8_rank_threads_on_hip(we):
    from witw_amd import cvig_fov
    from .test_match_dft_gpu import _planted
    G, Q, k = sum(C5_SPLIT), 600, 10
    gallery, queries = _planted(G, Q, we, 4.0, seed=500 + we)
    r_direct, v_direct, i_direct = cvig_fov.retrieve(gallery, queries, k=k, query_chunk=256)
    r_dft, v_dft, i_dft = cvig_fov.retrieve(gallery, queries, k=k, query_chunk=256, method='dft')
    np.testing.assert_array_equal(r_dft, r_direct)
    assert torch.equal(i_dft, i_direct)

    def fn(rank):
        torch.cuda.set_device(gallery.device)
        g0 = sum(C5_SPLIT[:rank])
        shard = gallery[g0:g0 + C5_SPLIT[rank]].contiguous()
        a = cvig_fov.retrieve(shard, queries, k=k, shard_begin=g0, query_chunk=256)
        b = cvig_fov.retrieve(shard, queries, k=k, shard_begin=g0, query_chunk=256, method='dft')
        st = cvig_fov.last_retrieve_stats()
        torch.cuda.synchronize()
        return a, b, st
    res = run_ranks(WORLD, fn)
    rescored = 0
    for (ra, va, ia), (rb, vb, ib), st in res:
        np.testing.assert_array_equal(ra, r_direct)
        assert torch.equal(ia, i_direct) and torch.equal(va, v_direct)          # direct: bit-identical to the unsharded pass
        np.testing.assert_array_equal(rb, r_direct)
        assert torch.equal(ib, i_direct)                                        # spectral: index-exact
        assert float((vb - v_direct).abs().max()) <= st['eps']
        rescored += st['rescored_topk'] + st['rescored_rank']
    assert rescored > 0          # the planted near-ties were decided by exact re-scoring on the shards that own them
