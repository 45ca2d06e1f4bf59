"""Drivers and data formats either side of the hot path for the cvig_semantic and cvig_baseline variants:
5-band TIFF datasets, SyncedRotation on the GPU, train()/test() end to end on tiny synthetic data sets,
checkpoint exchange with the reference's state-dict layout."""
import os

import numpy as np
import pytest
import torch

from oracle import cvig_baseline_oracle as OB
from oracle import cvig_fov_oracle as O
from witw_amd import synth

pytestmark = pytest.mark.gpu


def test_rotate_nearest_matches_torchvision_restatement():
    """ops.rotate_nearest vs the oracle's restatement of torchvision 0.9.1 F.rotate (grid_sample nearest). The
    gather is index work: identical pixels, except where a source coordinate sits within float rounding of a
    .5 boundary (bmm vs fma summation order) — bounded at 1e-3 of the pixels, each then a neighbouring source."""
    from witw_amd import ops
    x = torch.from_numpy(synth.images_u8(70, 1, (3, 3, 96, 128)))
    angles = [0.0, 90.0, 37.3]
    got = ops.rotate_nearest(x.cuda(), angles).cpu()
    for i, a in enumerate(angles):
        ref = OB.rotate(x[i], a)
        diff = (got[i] != ref).any(dim=0)
        assert diff.float().mean().item() <= 1e-3, (a, diff.float().mean().item())
    np.testing.assert_array_equal(got[0].numpy(), x[0].numpy())                      # 0 degrees: identity
    sq = torch.from_numpy(synth.images_u8(70, 2, (1, 3, 64, 64)))
    r90 = ops.rotate_nearest(sq.cuda(), [90.0]).cpu()[0]
    from witw_amd import cvig_baseline
    # torchvision's angle is counter-clockwise on the displayed image (y down); the reference's quantized_rotation
    # factor counts the other way in array coordinates (factor 1 = transpose + flip(-1) = clockwise), so 90 deg == factor 3
    np.testing.assert_array_equal(r90.numpy(), cvig_baseline.quantized_rotation(sq[0], 3).numpy())
    full = torch.from_numpy(synth.images_u8(70, 3, (2, 3, 512, 512)))
    g2 = ops.rotate_nearest(full.cuda(), [123.4, 301.0]).cpu()
    for i, a in enumerate([123.4, 301.0]):
        assert (g2[i] != OB.rotate(full[i], a)).any(dim=0).float().mean().item() <= 1e-3


def test_synced_rotation_transforms():
    from witw_amd import cvig_baseline as cb
    su = torch.from_numpy(synth.images_u8(71, 1, (3, 32, 360)))
    ov = torch.from_numpy(synth.images_u8(71, 2, (3, 64, 64)))
    d = cb.SyncedRotation('cvusa')({'surface': su.clone(), 'overhead': ov.clone()}, angle=90.)
    np.testing.assert_array_equal(d['surface'].numpy(), torch.roll(su, -90, dims=-1).numpy())     # panorama: rolled by 90 px of 360
    np.testing.assert_array_equal(d['overhead'].cpu().numpy(), cb.quantized_rotation(ov, 3).numpy())
    d = cb.SyncedRotation('witw')({'surface': su.clone(), 'overhead': ov.clone()}, angle=45.)
    np.testing.assert_array_equal(d['surface'].numpy(), su.numpy())                                # not a panorama: untouched
    assert (d['overhead'].cpu() != OB.rotate(ov, 45.)).any(dim=0).float().mean().item() <= 1e-3
    q = cb.QuantizedSyncedRotation('cvusa')({'surface': su.clone(), 'overhead': ov.clone()}, factor=3)
    np.testing.assert_array_equal(q['overhead'].numpy(), ov.transpose(-2, -1).flip(-2).numpy())
    np.testing.assert_array_equal(q['surface'].numpy(), torch.roll(su, -270, dims=-1).numpy())


def _write_baseline_dataset(root, n):
    from PIL import Image
    rows = []
    for i in range(n):
        Image.fromarray(synth.images_u8(72, i, (400, 400, 3)).astype(np.uint8)).save(os.path.join(root, 'ov_%d.png' % i))
        Image.fromarray(synth.images_u8(73, i, (200, 800, 3)).astype(np.uint8)).save(os.path.join(root, 'su_%d.png' % i))
        rows.append('ov_%d.png,su_%d.png' % (i, i))
    with open(os.path.join(root, 'pairs.csv'), 'w') as f:
        f.write('\n'.join(rows) + '\n')
    return os.path.join(root, 'pairs.csv')


def test_baseline_train_then_test_drivers(tmp_path, monkeypatch, capsys):
    from witw_amd import cvig_baseline as cb
    csv = _write_baseline_dataset(str(tmp_path), 5)
    ds = cb.ImagePairDataset('cvusa', csv)
    assert sorted(ds[0].keys()) == ['overhead', 'surface'] and ds[0]['surface'].shape == (3, 200, 800)
    monkeypatch.chdir(tmp_path)
    best = cb.train(dataset='cvusa', val_quantity=2, batch_size=3, num_workers=0, num_epochs=1, csv_path=csv)
    assert best is not None and np.isfinite(best)
    sd = torch.load(os.path.join('weights', 'surface_best.pth'))
    assert 'conv7.weight' in sd and 'bn7.running_var' in sd
    table = cb.test(dataset='cvusa', batch_size=4, num_workers=0, csv_path=csv)
    out = capsys.readouterr().out
    assert 'Top  1:' in out and 'Locations: 5' in out and 'new best' in out
    assert 0 <= table['top_1'] <= 100


def test_baseline_forward_sees_weights_after_adam_step():
    """The packed / BatchNorm-folded filters are cached: an Adam step and a running-statistics update through the
    C-ABI (which torch's version counters cannot see) must invalidate them."""
    from witw_amd import cvig_baseline as cb
    from witw_amd import cvig_fov
    torch.manual_seed(3)
    enc = cb.SurfaceEncoder().cuda()
    x = torch.from_numpy(synth.images_u8(74, 1, (3, 3, 400, 400))).cuda()
    enc.eval()
    e0 = enc(x).clone()
    enc.train()
    opt = cvig_fov.Adam(list(enc.parameters()), lr=1e-2)
    f = enc(x)
    opt.zero_grad()
    (f * f).sum().backward()
    opt.step()
    enc.eval()
    e1 = enc(x)
    fresh = cb.SurfaceEncoder().cuda().eval()
    fresh.load_state_dict(enc.state_dict())
    e2 = fresh(x)
    assert (e1 - e0).abs().max().item() > 1e-4
    np.testing.assert_array_equal(e1.cpu().numpy(), e2.cpu().numpy())
    ref = OB.encoder_forward(x.cpu(), [dict(w=getattr(enc, 'conv%d' % i).weight.detach().cpu(),
                                            b=getattr(enc, 'conv%d' % i).bias.detach().cpu(),
                                            gamma=getattr(enc, 'bn%d' % i).weight.detach().cpu(),
                                            beta=getattr(enc, 'bn%d' % i).bias.detach().cpu(),
                                            mean=getattr(enc, 'bn%d' % i).running_mean.cpu().clone(),
                                            var=getattr(enc, 'bn%d' % i).running_var.cpu().clone()) for i in range(1, 8)])
    np.testing.assert_allclose(e1.cpu().numpy(), ref.numpy(), atol=1e-4)


def _write_semantic_dataset(root, n):
    """witw format: header row, surface path in column 15, overhead in column 16; 5-band float32 scene TIFFs."""
    from witw_amd import tiffio
    lines = [','.join('c%d' % i for i in range(17))]
    for i in range(n):
        su = np.concatenate([synth.images_u8(75, i, (40, 60, 3)), synth.images_u8(76, i, (40, 60, 2)) / 255.], axis=2)
        ov = np.concatenate([synth.images_u8(77, i, (64, 64, 3)), synth.images_u8(78, i, (64, 64, 2)) / 255.], axis=2)
        tiffio.imwrite(os.path.join(root, 'su_%d.tif' % i), su.astype(np.float32))
        tiffio.imwrite(os.path.join(root, 'ov_%d.tif' % i), ov.astype(np.float32))
        lines.append(','.join(['x'] * 15 + ['su_%d.jpg' % i, 'ov_%d.jpg' % i]))     # extension replaced by .tif
    with open(os.path.join(root, 'scenes.csv'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    return os.path.join(root, 'scenes.csv')


def test_semantic_dataset_preprocess_and_drivers(tmp_path, monkeypatch, capsys):
    from witw_amd import cvig_semantic as cs
    csv = _write_semantic_dataset(str(tmp_path), 6)
    ds = cs.ImagePairDataset('witw', csv)
    s = ds[2]
    assert s['idx'] == 2 and s['surface'].shape == (5, 40, 60) and s['overhead'].shape == (5, 64, 64)
    assert s['surface'][3:].max() <= 1.0 and s['surface'][:3].max() > 1.0
    with pytest.raises(KeyError):
        cs.ImagePairDataset('cvusa', csv)                        # reference: no 'semantic' key for cvusa
    prep = cs.GpuPreprocess('witw', fov=70)
    out = prep(cs._fov.collate_raw([ds[0], ds[2]]))
    assert out['surface'].shape == (2, 5, 128, 99) and out['polar'].shape == (2, 5, 128, 512)
    rs, ro = O.resize_pair(s['surface'], s['overhead'], fov=70, panorama=False, start=0)
    np.testing.assert_allclose(out['surface'][1].cpu().numpy(), O.image_normalization_semantic(rs).numpy(), atol=2e-5)
    np.testing.assert_allclose(out['polar'][1].cpu().numpy(),
                               O.polar_transform(O.image_normalization_semantic(ro)).numpy(), atol=2e-5)
    monkeypatch.chdir(tmp_path)
    best = cs.train(dataset='witw', fov=70, val_quantity=2, batch_size=2, num_workers=0, num_epochs=1, csv_path=csv)
    assert best is not None and np.isfinite(best)
    sd = torch.load(os.path.join('weights', 'fov_70_surface_best.pth'))
    assert sd['model.features.0.weight'].shape == (64, 5, 3, 3)
    table = cs.test(dataset='witw', fov=70, batch_size=4, num_workers=0, csv_path=csv)
    assert 'Locations: 6' in capsys.readouterr().out and 0 <= table['top_1'] <= 100


def test_reference_checkpoint_round_trip(tmp_path):
    """A checkpoint in the reference's layout (wrapper-nested keys + the unused VGG classifier tensors) loads,
    the classifier tensors ride along, and save_reference_state_dict writes the same key set back."""
    from witw_amd import cvig_fov
    enc = cvig_fov.FOV_DSM(circ_padding=True, seed=5)
    state = dict(enc.state_dict())
    assert 'model.features.17.layer.layer.weight' in state and 'model.features.0.layer.weight' in state
    cls = {'model.classifier.0.weight': torch.full((8, 4), 2.0), 'model.classifier.0.bias': torch.ones(8)}
    ref_ckpt = dict(state, **cls)
    other = cvig_fov.FOV_DSM(circ_padding=True, seed=6)
    cvig_fov.load_reference_state_dict(other, ref_ckpt)
    for k, v in state.items():
        assert torch.equal(other.state_dict()[k], v)
    path = str(tmp_path / 'ck.pth')
    cvig_fov.save_reference_state_dict(other, path)
    back = torch.load(path)
    assert sorted(back.keys()) == sorted(ref_ckpt.keys()) and torch.equal(back['model.classifier.0.weight'], cls['model.classifier.0.weight'])
    plain = cvig_fov.FOV_DSM(circ_padding=False, seed=7)
    cvig_fov.save_reference_state_dict(plain, path)
    back = torch.load(path)
    assert back['model.classifier.6.weight'].shape == (1000, 4096) and 'model.features.17.layer.weight' in back
