"""BASELINE configs[4] at its FULL per-GPU size (1M / 8 = 125,000 gallery rows on one device) against the oracle
(VERDICT r04 weak #1b: until round 5 config 5 met the oracle at <= ~26 k rows and ran at 125 k x 10 k only inside bench.py).

One pass of each matching path over 125,000 x 1,024 (gallery row, query) pairs -- the direct kernel (witw_match_fwd, 524,288 FLOP
per pair) and the spectral one with its index-exact re-scoring (retrieve(method='dft')) -- then, on a 256-row x 64-query sample
that contains every sampled query's true row and its listed nearest rows, the oracle's match_fused (reference
model/cvig_fov.py:297-363 without the crop tensor, fp64 accumulation): orientation bit-exact wherever the two best shifts are not
within rounding, distance 1e-5, and the ranks of test() (model/cvig_fov.py:543-552) through the band contract: a row's side of a
query's true-match distance is the oracle's unless it lies within DISTANCE_EPS of it."""
import numpy as np
import pytest
import torch

from oracle import cvig_fov_oracle as O

pytestmark = pytest.mark.gpu


def _planted(G, Q, dev, seed=4321):
    """bench.py's config-5 data: N(0,1) gallery rows; query q = row q rolled by a per-query shift + noise (sigma 10)"""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    gallery = torch.randn((G, 16, 4, 64), generator=gen, device=dev)
    shifts = torch.randint(0, 64, (Q,), generator=gen, device=dev)
    col = (torch.arange(64, device=dev)[None, :] + shifts[:, None]) % 64
    queries = torch.gather(gallery[:Q], 3, col[:, None, None, :].expand(-1, 16, 4, -1)) \
        + 10.0 * torch.randn((Q, 16, 4, 64), generator=gen, device=dev)
    return gallery, queries.contiguous(), shifts


def test_config5_full_per_gpu_gallery_direct_and_dft_vs_oracle_sample():
    from witw_amd import cvig_fov, ops
    dev = torch.device('cuda:0')
    G, Q, k = 125000, 1024, 10
    gallery, queries, shifts = _planted(G, Q, dev)
    ranks_d, v_d, i_d = cvig_fov.retrieve(gallery, queries, k=k, method='direct')
    ranks_s, v_s, i_s = cvig_fov.retrieve(gallery, queries, k=k, method='dft')
    st = cvig_fov.last_retrieve_stats()
    # the spectral pass's contract: ranks and top-k INDICES equal the direct pass's exactly, distances to DISTANCE_EPS
    np.testing.assert_array_equal(ranks_s, ranks_d)
    assert torch.equal(i_s, i_d)
    assert float((v_s - v_d).abs().max()) <= ops.DISTANCE_EPS
    assert st['pairs'] == float(G) * Q and st['fallback_queries'] <= 8
    # the planted matches are found (the gallery is pure noise otherwise): most queries rank their true row first
    assert float(np.mean(ranks_d <= 1)) > 0.7 and float((i_d[:, 0].cpu() == torch.arange(Q)).float().mean()) > 0.7

    # the whole [G, Q] matrix of the direct kernel: retrieve()'s ranks and lists are those of this matrix
    ori, dist = ops.match_fwd(gallery, queries)
    qi = torch.arange(Q, device=dev)
    d_true = dist[qi, qi]
    np.testing.assert_array_equal((dist <= d_true[None, :]).sum(0).cpu().numpy(), ranks_d)
    assert torch.equal(torch.gather(dist, 0, i_d.t()).t(), v_d)

    # ---- oracle on a sample: 64 queries, 256 rows holding each sampled query's true row and its three nearest listed rows
    qs = np.unique(np.linspace(0, Q - 1, 64).astype(np.int64))
    g = np.random.Generator(np.random.Philox(key=[5, 125000]))
    must = set(int(q) for q in qs) | set(int(r) for r in i_d[torch.from_numpy(qs).to(dev), :3].cpu().numpy().ravel())
    extra = [int(r) for r in g.permutation(G) if int(r) not in must][:256 - len(must)]
    rows = np.array(sorted(must) + sorted(extra), dtype=np.int64)
    assert len(rows) == 256 and set(int(q) for q in qs) <= set(int(r) for r in rows)
    rows_t, qs_t = torch.from_numpy(rows).to(dev), torch.from_numpy(qs).to(dev)
    ov_c, su_c = gallery[rows_t].cpu(), queries[qs_t].cpu()
    o_ref, d_ref = O.match_fused(ov_c, su_c)                                       # [256, 64]
    d_gpu = dist[rows_t][:, qs_t].cpu()
    o_gpu = ori[rows_t][:, qs_t].cpu()
    np.testing.assert_allclose(d_gpu.numpy(), d_ref.numpy(), rtol=0, atol=1e-5)
    sc = O.correlation_scores(ov_c.double(), su_c.double())                        # [256, 64, 64] fp64
    top2 = torch.topk(sc, 2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-4 * top2[..., 0].abs().clamp_min(1.0)
    assert float(clear.float().mean()) > 0.99
    assert torch.equal(o_gpu[clear], o_ref[clear])
    # the planted shift is the orientation of every true pair
    pos = {int(r): n for n, r in enumerate(rows)}
    for n, q in enumerate(qs):
        assert int(o_gpu[pos[int(q)], n]) == int(shifts[int(q)]) == int(o_ref[pos[int(q)], n])
    # ranks through the band contract, on the sampled rows: which side of the true-match distance each row falls
    t_ref = torch.stack([d_ref[pos[int(q)], n] for n, q in enumerate(qs)])
    t_gpu = torch.stack([d_gpu[pos[int(q)], n] for n, q in enumerate(qs)])
    np.testing.assert_allclose(t_gpu.numpy(), d_true[qs_t].cpu().numpy(), rtol=0, atol=0)
    outside = (d_ref - t_ref[None, :]).abs() > ops.DISTANCE_EPS
    assert torch.equal((d_gpu <= t_gpu[None, :])[outside], (d_ref <= t_ref[None, :])[outside])
    # the band holds each query's TRUE row (difference 0: 64 of the 256 x 64 sampled pairs) and at most a handful of near ties: the
    # comparison above covers > 99.5 % of the sample (round 5 asserted > 90 %)
    inside = int((~outside).sum())
    assert len(qs) <= inside <= len(qs) + 0.001 * outside.numel(), (inside, outside.numel())
    # ... and the listed nearest rows of the sampled queries are the oracle's nearest among the sampled rows (each list's rows are in
    # the sample): same order wherever neighbours are further apart than the band
    for n, q in enumerate(qs):
        listed = [int(r) for r in i_d[int(q), :3].cpu()]
        dl = torch.stack([d_ref[pos[r], n] for r in listed])
        order_ref = torch.argsort(d_ref[:, n], stable=True)[:3]
        if float((torch.sort(d_ref[:, n]).values[1:4] - torch.sort(d_ref[:, n]).values[:3]).min()) > 2 * ops.DISTANCE_EPS:
            assert [int(rows[j]) for j in order_ref] == listed, (q, listed)
        assert float((dl - v_d[int(q), :3].cpu()).abs().max()) <= 1e-5
