"""End-to-end drivers on a tiny synthetic dataset written to disk: ImagePairDataset -> GPU transforms ->
train() (1 epoch) -> checkpoint with the reference's names/keys -> test() recall table; heat-map scoring."""
import os

import numpy as np
import pytest
import torch

from oracle import cvig_fov_oracle as O
from witw_amd import synth

pytestmark = pytest.mark.gpu


def _write_dataset(root, n):
    from PIL import Image
    rows = []
    for i in range(n):
        ov = synth.images_u8(50, i, (64, 64, 3)).astype(np.uint8)
        su = synth.images_u8(51, i, (48, 80, 3)).astype(np.uint8)
        Image.fromarray(ov).save(os.path.join(root, 'ov_%d.png' % i))
        Image.fromarray(su).save(os.path.join(root, 'su_%d.png' % i))
        rows.append('ov_%d.png,su_%d.png' % (i, i))           # cvusa format: overhead,surface, no header
    with open(os.path.join(root, 'pairs.csv'), 'w') as f:
        f.write('\n'.join(rows) + '\n')
    return os.path.join(root, 'pairs.csv')


def test_dataset_and_gpu_preprocess_match_oracle(tmp_path):
    from witw_amd import cvig_fov
    csv = _write_dataset(str(tmp_path), 3)
    ds = cvig_fov.ImagePairDataset('cvusa', csv)
    assert len(ds) == 3
    s = ds[1]
    assert s['surface'].shape == (3, 48, 80) and s['overhead'].shape == (3, 64, 64) and s['surface'].dtype == torch.float32
    prep = cvig_fov.GpuPreprocess('cvusa', fov=70, random_orientation=False)
    out = prep(cvig_fov.collate_raw([ds[0], ds[1]]))
    assert out['surface'].shape == (2, 3, 128, 99) and out['polar'].shape == (2, 3, 128, 512)
    rs, ro = O.resize_pair(ds[1]['surface'], ds[1]['overhead'], fov=70, panorama=True, start=0)
    np.testing.assert_allclose(out['surface'][1].cpu().numpy(), O.image_normalization(rs).numpy(), atol=2e-5)
    np.testing.assert_allclose(out['polar'][1].cpu().numpy(), O.polar_transform(O.image_normalization(ro)).numpy(), atol=2e-5)


def test_train_then_test_drivers(tmp_path, monkeypatch, capsys):
    from witw_amd import cvig_fov
    csv = _write_dataset(str(tmp_path), 6)
    monkeypatch.chdir(tmp_path)
    best = cvig_fov.train(dataset='cvusa', fov=70, val_quantity=2, batch_size=2, num_workers=0, num_epochs=1, csv_path=csv)
    assert best is not None and np.isfinite(best)
    for side in ('surface', 'overhead'):
        sd = torch.load(os.path.join('weights', 'fov_70_%s_best.pth' % side))
        assert any(k.endswith('features.27.weight') or k.endswith('features.27.layer.weight') for k in sd)
    table = cvig_fov.test(dataset='cvusa', fov=70, batch_size=4, num_workers=0, csv_path=csv)
    out = capsys.readouterr().out
    assert 'Top  1:' in out and 'Locations: 6' in out and 'new best' in out
    assert 0 <= table['top_1'] <= 100 and 1 <= table['median'] <= 6


def test_train_then_test_drivers_bf16(tmp_path, monkeypatch, capsys):
    """The same drivers with Globals.precision = 'bf16' (CLI --precision bf16): mixed-precision training step,
    fp32 checkpoints under the reference's names, bf16 evaluation."""
    from witw_amd import cvig_fov
    csv = _write_dataset(str(tmp_path), 6)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(cvig_fov.Globals, 'precision', 'bf16')
    best = cvig_fov.train(dataset='cvusa', fov=70, val_quantity=2, batch_size=2, num_workers=0, num_epochs=1, csv_path=csv)
    assert best is not None and np.isfinite(best)
    sd = torch.load(os.path.join('weights', 'fov_70_surface_best.pth'))
    assert all(v.dtype == torch.float32 for v in sd.values())
    table = cvig_fov.test(dataset='cvusa', fov=70, batch_size=4, num_workers=0, csv_path=csv)
    assert 'Top  1:' in capsys.readouterr().out and 1 <= table['median'] <= 6
    # evaluation on the fp16x3 arithmetic reads the same fp32 checkpoints
    monkeypatch.setattr(cvig_fov.Globals, 'precision', 'fp16x3')
    t3 = cvig_fov.test(dataset='cvusa', fov=70, batch_size=4, num_workers=0, csv_path=csv)
    assert 1 <= t3['median'] <= 6


def test_test_driver_bf16_with_loader_workers_captures_and_replays(tmp_path, monkeypatch, capsys):
    """test() in bf16 with DataLoader WORKERS and four equal-shaped batches: PairEmbedder captures its hipGraph on the second batch
    while the loader's worker / pinning / ring-reaper threads are alive (thread-local capture mode), replays it on the
    others, and the table equals the one of the eager path (num_workers=0, graphs off)."""
    from witw_amd import cvig_fov
    csv = _write_dataset(str(tmp_path), 16)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(cvig_fov.Globals, 'precision', 'bf16')
    monkeypatch.setattr(cvig_fov.Globals, 'test_random_orientation', False, raising=False)
    cvig_fov.train(dataset='cvusa', fov=70, val_quantity=2, batch_size=2, num_workers=0, num_epochs=1, csv_path=csv)
    made = []
    real = cvig_fov.PairEmbedder

    class Spy(real):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            made.append(self)
    monkeypatch.setattr(cvig_fov, 'PairEmbedder', Spy)
    t_workers = cvig_fov.test(dataset='cvusa', fov=70, batch_size=4, num_workers=2, csv_path=csv)
    st = made[-1].stats
    assert st['captures'] == 1 and st['graph_replay'] >= 2 and st['capture_failures'] == 0, st
    # the eager path: no graph, no second stream
    monkeypatch.setattr(cvig_fov, 'PairEmbedder', lambda se, oe: Spy(se, oe, graph_max=0, dual_max=0))
    t_eager = cvig_fov.test(dataset='cvusa', fov=70, batch_size=4, num_workers=0, csv_path=csv)
    assert made[-1].stats['graph_replay'] == 0 and made[-1].stats['eager'] == 4
    assert t_workers == t_eager, (t_workers, t_eager)


def test_pair_embedder_falls_back_to_eager_when_capture_fails(monkeypatch):
    """A capture that raises costs the key its graph, not the evaluation: same bits from the eager path, counted in stats."""
    from witw_amd import cvig_fov, parallel
    dev = torch.device('cuda:0')
    wts = synth.fov_dsm_weights(5)
    se = cvig_fov.FOV_DSM(False, weights=wts).to(dev).eval()
    oe = cvig_fov.FOV_DSM(True, weights=wts).to(dev).eval()
    se.precision = oe.precision = 'bf16'
    s = torch.from_numpy(synth.normalized_images(5, 0, (4, 3, 128, 99))).to(dev)
    p_ = torch.from_numpy(synth.normalized_images(5, 1, (4, 3, 128, 512))).to(dev)

    def boom(*a, **k):
        raise RuntimeError('capture invalidated (simulated)')
    monkeypatch.setattr(parallel, 'CapturedStep', boom)
    pe = cvig_fov.PairEmbedder(se, oe)
    with torch.no_grad(), pytest.warns(UserWarning, match='capture failed'):
        outs = [pe(s, p_) for _ in range(3)]
        ref = (se.forward_bf16(s), oe.forward_bf16(p_))
    assert pe.stats['capture_failures'] == 1 and pe.stats['captures'] == 0 and pe.stats['graph_replay'] == 0
    for su, ov in outs:
        assert torch.equal(su, ref[0]) and torch.equal(ov, ref[1])


def test_train_driver_on_f16x3(tmp_path, monkeypatch):
    """train() with Globals.precision = 'fp16x3': forward, dgrad and wgrad on the split-fp16 arithmetic, fp32 checkpoints."""
    from witw_amd import cvig_fov
    csv = _write_dataset(str(tmp_path), 6)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(cvig_fov.Globals, 'precision', 'fp16x3')
    best = cvig_fov.train(dataset='cvusa', fov=70, val_quantity=2, batch_size=2, num_workers=0, num_epochs=1, csv_path=csv)
    assert best is not None and np.isfinite(best)
    sd = torch.load(os.path.join('weights', 'fov_70_overhead_best.pth'))
    assert all(v.dtype == torch.float32 for v in sd.values())


def test_projector_dump_inverse_normalize_and_bilinear_interpolate(tmp_path, monkeypatch):
    """The reference's TensorBoard embedding-projector dump (model/cvig_fov.py:474-479, :534-540) with a recording
    writer, inverse_normalize's verbatim semantics (:151-154) and bilinear_interpolate (:156-183) on free coordinates."""
    from witw_amd import cvig_fov
    calls, texts = [], []

    class Rec(object):
        def add_scalar(self, *a, **k):
            pass

        def add_text(self, tag, text, global_step=None):
            texts.append((tag, global_step))

        def add_embedding(self, mat, metadata=None, metadata_header=None, label_img=None, global_step=None, tag=None):
            calls.append((tuple(mat.shape), len(metadata), metadata_header, tuple(label_img.shape), global_step, tag))

    monkeypatch.setattr(cvig_fov, '_writer', lambda path: Rec())
    csv = _write_dataset(str(tmp_path), 6)
    monkeypatch.chdir(tmp_path)
    cvig_fov.train(dataset='cvusa', fov=70, val_quantity=4, batch_size=2, num_workers=0, num_epochs=1, csv_path=csv)
    # 'best_loss' is logged at the step of the last validation iteration, epoch * len(val loader) + batch = 0 * 2 + 1 (:487)
    assert texts[0] == ('best_loss', 1), texts
    cvig_fov.test(dataset='cvusa', fov=70, batch_size=4, num_workers=0, csv_path=csv)
    e = 16 * 4 * 12
    assert calls[0] == ((4, e), 4, ['idx', 'type'], (4, 3, 128, 512), 1, 'val_embedding')
    assert calls[1] == ((4, e), 4, ['idx', 'type'], (4, 3, 128, 512), 0, 'test_embedding')       # last test batch: 2 pairs
    # inverse_normalize walks dim 0 like the reference
    t = torch.ones(5, 3, 2, 2)
    out = cvig_fov.inverse_normalize(t, [1., 2., 3.], [10., 20., 30.])
    assert out is t and [float(t[i, 0, 0, 0]) for i in range(5)] == [11., 22., 33., 1., 1.]
    # bilinear_interpolate vs the oracle's restatement on random coordinates (clipping before the weights)
    g = np.random.Generator(np.random.Philox(key=[3, 3]))
    im = torch.from_numpy(g.standard_normal((3, 17, 23), dtype=np.float32))
    x = g.uniform(-2, 25, size=(7, 9))
    y = g.uniform(-2, 19, size=(7, 9))
    got = cvig_fov.bilinear_interpolate(im, x, y).cpu().numpy()
    x0 = np.floor(x).astype(int); x1 = x0 + 1; y0 = np.floor(y).astype(int); y1 = y0 + 1
    x0 = np.clip(x0, 0, 22); x1 = np.clip(x1, 0, 22); y0 = np.clip(y0, 0, 16); y1 = np.clip(y1, 0, 16)
    f = lambda a: torch.from_numpy(a.astype(np.float32)).reshape(1, 7, 9)
    ref = f((x1 - x) * (y1 - y)) * im[:, y0, x0] + f((x1 - x) * (y - y0)) * im[:, y1, x0] + \
        f((x - x0) * (y1 - y)) * im[:, y0, x1] + f((x - x0) * (y - y0)) * im[:, y1, x1]
    np.testing.assert_allclose(got, ref.numpy(), rtol=0, atol=1e-6)


def test_sweep_scores_matches_heatmap_formula():
    from witw_amd import cvig_fov
    ov = torch.from_numpy(synth.embeddings(60, 1, (9, 16, 4, 64)))
    su = torch.from_numpy(synth.embeddings(60, 2, (1, 16, 4, 12)))
    ori_r, d_r = O.match(ov, su)
    o, d, sc = cvig_fov.sweep_scores(ov.cuda(), su.cuda())
    np.testing.assert_array_equal(o.cpu().numpy(), (ori_r.squeeze() * 360 / 64 - 180).numpy())
    np.testing.assert_allclose(d.cpu().numpy(), d_r.squeeze().numpy(), atol=1e-5)
    np.testing.assert_allclose(sc.cpu().numpy(), torch.exp(10. * (1. - d_r.squeeze())).numpy(), rtol=2e-4)


def test_sharded_ranks_single_rank_equals_ranks():
    from witw_amd import cvig_fov
    n, we = 40, 12
    ov = torch.from_numpy(synth.embeddings(61, 1, (n, 16, 4, 64)))
    su = torch.stack([torch.roll(ov[i], -int(i % 64), dims=2)[:, :, :we] for i in range(n)]) \
        + 7.0 * torch.from_numpy(synth.embeddings(61, 2, (n, 16, 4, we)))
    a = cvig_fov.ranks(ov.cuda(), su.contiguous().cuda())
    b = cvig_fov.sharded_ranks(ov.cuda(), su.contiguous().cuda(), 0, query_chunk=16)
    np.testing.assert_array_equal(a, b)
    assert a.max() > 1
    # two half-gallery shards evaluated one after the other add up to the same counts
    from witw_amd import ops
    _, dist = cvig_fov.match(ov.cuda(), su.contiguous().cuda())
    thr = torch.diagonal(dist).contiguous()
    c = ops.rank_count_thresh(dist[:17].contiguous(), thr) + ops.rank_count_thresh(dist[17:].contiguous(), thr)
    np.testing.assert_array_equal(c.cpu().numpy().astype('int64'), a)


def test_batched_preprocess_equals_the_per_sample_transforms(tmp_path):
    """GpuPreprocess (3 launches per batch: batched resize + normalise over a descriptor table per side, one polar transform)
    against the per-sample Compose[Resize, ImageNormalization, PolarTransform] (model/cvig_fov.py:393-397): BIT-identical, for
    float CHW tensors on the host, tensors already on the GPU, and the decoder's uint8 bytes packed by collate_packed; images of
    different sizes in one batch; panorama crops at per-sample offsets (wrapping) and the plain resize of the witw format."""
    from witw_amd import cvig_fov
    sizes = [((48, 80), (64, 64)), ((224, 224), (512, 512)), ((100, 333), (300, 200)), ((37, 51), (90, 77))]
    raw = [(synth.images_u8(61, 2 * i, s + (3,)).astype(np.uint8), synth.images_u8(61, 2 * i + 1, o + (3,)).astype(np.uint8))
           for i, (s, o) in enumerate(sizes)]
    as_float = [{'idx': i, 'surface': torch.from_numpy(s.astype(np.float32).transpose(2, 0, 1).copy()),
                 'overhead': torch.from_numpy(o.astype(np.float32).transpose(2, 0, 1).copy())} for i, (s, o) in enumerate(raw)]
    as_bytes = [{'idx': i, 'surface': s, 'overhead': o} for i, (s, o) in enumerate(raw)]
    starts = [0, 500, 17, 300]
    for dataset, fov in (('cvusa', 70), ('cvusa', 360), ('witw', 70)):
        prep = cvig_fov.GpuPreprocess(dataset, fov=fov, random_orientation=False)
        rs, nm, pt = cvig_fov.Resize(dataset, fov, False), cvig_fov.ImageNormalization(), cvig_fov.PolarTransform()
        ref_s, ref_p = [], []
        for smp, st in zip(as_float, starts):
            d = pt(nm(rs({'surface': smp['surface'].clone(), 'overhead': smp['overhead'].clone()}, start=st)))
            ref_s.append(d['surface'])
            ref_p.append(d['polar'])
        ref_s, ref_p = torch.stack(ref_s), torch.stack(ref_p)
        batches = {'host float': cvig_fov.collate_raw(as_float),
                   'resident float': cvig_fov.collate_raw([{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in smp.items()} for smp in as_float]),
                   'packed bytes': cvig_fov.collate_packed(as_bytes), 'packed float': cvig_fov.collate_packed(as_float)}
        assert batches['packed bytes']['surface_kind'] == 1 and batches['packed float']['surface_kind'] == 0
        for name, batch in batches.items():
            out = prep(batch, starts=starts)
            assert out['idx'] == [0, 1, 2, 3]
            assert torch.equal(out['surface'], ref_s), (dataset, fov, name)
            assert torch.equal(out['polar'], ref_p), (dataset, fov, name)
            assert 'overhead' not in out              # the fused overhead launch never forms the 256 x 256 image
        prep.keep_overhead = True                     # the reference's dict, overhead side as two launches
        out = prep(batches['packed bytes'], starts=starts)
        assert torch.equal(out['polar'], ref_p) and tuple(out['overhead'].shape) == (4, 3, 256, 256)
    # one batch ahead on a copy stream
    ds_like = [cvig_fov.collate_packed(as_bytes[:2]), cvig_fov.collate_packed(as_bytes[2:])]
    prep = cvig_fov.GpuPreprocess('witw', fov=70)
    got = [prep(st)['surface'] for st in cvig_fov.DevicePrefetcher(ds_like, prep)]
    assert torch.equal(torch.cat(got), ref_s)
    # grouped: the loader hands out part batches, DevicePrefetcher(group=g) stages g of them as ONE batch (every part keeps its
    # own device block; the descriptor tables are concatenated) -- same result as the whole batch, sample order kept
    parts = [cvig_fov.collate_packed(as_bytes[:1]), cvig_fov.collate_packed(as_bytes[1:3]), cvig_fov.collate_packed(as_bytes[3:])]
    staged = list(cvig_fov.DevicePrefetcher(parts, prep, group=3))
    assert len(staged) == 1 and staged[0].n == 4 and staged[0].idx == [0, 1, 2, 3]
    out = prep(staged[0])
    assert torch.equal(out['surface'], ref_s) and torch.equal(out['polar'], ref_p)
    two = list(cvig_fov.DevicePrefetcher(parts, prep, group=2))        # a short last group
    assert [st.n for st in two] == [3, 1] and torch.equal(torch.cat([prep(st)['surface'] for st in two]), ref_s)
    assert cvig_fov.loader_split(128, 12) == 4 and cvig_fov.loader_split(128, 0) == 1 and cvig_fov.loader_split(30, 4) == 1
    # a grey-scale image cannot feed a 3-channel model
    from witw_amd import _lib
    with pytest.raises(_lib.WitwError):
        prep(cvig_fov.collate_packed([{'surface': raw[0][0][:, :, :1], 'overhead': raw[0][1]}]))


def test_raw_dataset_and_workers(tmp_path):
    """ImagePairDataset(raw=True) + collate_packed through real DataLoader workers with pinned memory: the batch arrives as two
    byte blocks, and the embeddings' inputs equal those of the reference-format (float CHW) dataset bit for bit."""
    from witw_amd import cvig_fov
    csv = _write_dataset(str(tmp_path), 5)
    ds_raw, ds_ref = cvig_fov.ImagePairDataset('cvusa', csv, raw=True), cvig_fov.ImagePairDataset('cvusa', csv)
    assert ds_raw[0]['surface'].dtype == np.uint8 and ds_raw[0]['surface'].shape == (48, 80, 3)
    loader = torch.utils.data.DataLoader(ds_raw, batch_size=3, num_workers=2, collate_fn=cvig_fov.collate_packed, pin_memory=True)
    prep = cvig_fov.GpuPreprocess('cvusa', fov=70, random_orientation=False)
    outs = [prep(st) for st in cvig_fov.DevicePrefetcher(loader, prep)]
    assert [o['surface'].shape[0] for o in outs] == [3, 2] and outs[0]['idx'] == [0, 1, 2]
    ref = prep(cvig_fov.collate_raw([ds_ref[i] for i in range(5)]))
    assert torch.equal(torch.cat([o['polar'] for o in outs]), ref['polar'])
    assert torch.equal(torch.cat([o['surface'] for o in outs]), ref['surface'])
