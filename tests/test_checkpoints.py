"""Checkpoint and weight exchange with the reference (CPU: no kernel runs here).

* torchvision-format VGG16 weights -> FOV_DSM, as the reference's __init__ does with torch.hub's model
  (model/cvig_fov.py:256-272; cvig_semantic.py:301-303 for the 5-channel first conv);
* checkpoints written for the reference's strict load_state_dict (model/cvig_fov.py:511-512): the key set is the one the
  reference's own modules have (tests/golden/encoder.npz: keys_surface / keys_overhead, recorded from the reference)."""
import os

import numpy as np
import torch

from witw_amd import cvig_fov, cvig_semantic, synth

VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


def vgg16_state_dict(seed, classifier=True):
    """A state_dict with torchvision vgg16's keys and feature shapes (cfg D); synthetic values; the classifier shrunk."""
    g = np.random.Generator(np.random.Philox(key=[seed, 3]))
    sd, i, cin = {}, 0, 3
    for v in VGG_CFG:
        if v == 'M':
            i += 1
            continue
        sd['features.%d.weight' % i] = torch.from_numpy((g.standard_normal((v, cin, 3, 3), dtype=np.float32) * 0.05).astype(np.float32))
        sd['features.%d.bias' % i] = torch.from_numpy((g.standard_normal((v,), dtype=np.float32) * 0.05).astype(np.float32))
        cin = v
        i += 2
    if classifier:
        for j, (o, n) in ((0, (6, 5)), (3, (6, 6)), (6, (4, 6))):
            sd['classifier.%d.weight' % j] = torch.full((o, n), float(j + 1))
            sd['classifier.%d.bias' % j] = torch.full((o,), float(j + 2))
    return sd


def test_vgg16_state_dict_ingestion(tmp_path):
    sd = vgg16_state_dict(1)
    assert sorted(int(k.split('.')[1]) for k in sd if k.endswith('weight') and k.startswith('features')) == \
        [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
    path = str(tmp_path / 'vgg16.pth')
    torch.save(sd, path)
    torch.manual_seed(5)
    for circ in (False, True):
        enc = cvig_fov.FOV_DSM.from_vgg16_state_dict(path, circ_padding=circ)
        for idx in cvig_fov.VGG16_CONVS:                       # features[:23] of the VGG
            conv = cvig_fov._conv_of(enc.model.features[idx])
            assert torch.equal(conv.weight, sd['features.%d.weight' % idx]) and torch.equal(conv.bias, sd['features.%d.bias' % idx])
            assert conv.weight.requires_grad == (idx >= 17)      # :275-278
        for idx, (co, ci) in ((23, (256, 512)), (25, (64, 256)), (27, (16, 64))):      # the extra convs: xavier_uniform, zero bias
            conv = cvig_fov._conv_of(enc.model.features[idx])
            bound = float(np.sqrt(6.0 / (ci * 9 + co * 9)))
            assert float(conv.weight.abs().max()) <= bound and float(conv.weight.abs().max()) > 0.9 * bound
            assert abs(float(conv.weight.std()) - bound / np.sqrt(3)) < 0.05 * bound
            assert float(conv.bias.abs().max()) == 0.0
        assert sorted(enc._reference_classifier) == sorted('model.classifier.%d.%s' % (j, k) for j in (0, 3, 6) for k in ('weight', 'bias'))
    # a cvig_fov checkpoint is not a VGG16 file
    import pytest
    from witw_amd import _lib
    with pytest.raises(_lib.WitwError):
        cvig_fov.load_vgg16_state_dict(cvig_fov.FOV_DSM(), dict(cvig_fov.FOV_DSM().state_dict()))


def test_vgg16_ingestion_semantic_first_conv():
    sd = vgg16_state_dict(2)
    torch.manual_seed(6)
    enc = cvig_semantic.FOV_DSM.from_vgg16_state_dict(sd, circ_padding=True)
    conv0 = cvig_fov._conv_of(enc.model.features[0])
    assert tuple(conv0.weight.shape) == (64, 5, 3, 3) and conv0.weight.requires_grad
    assert torch.equal(conv0.weight[:, :3], sd['features.0.weight'])              # model/cvig_semantic.py:303
    bound = 1.0 / np.sqrt(5 * 9)                                                     # a fresh Conv2d(5, 64, 3): kaiming_uniform(a=sqrt 5)
    extra = conv0.weight[:, 3:]
    assert 0.8 * bound < float(extra.abs().max()) <= bound and not torch.equal(conv0.bias, sd['features.0.bias'])
    assert torch.equal(cvig_fov._conv_of(enc.model.features[2]).weight, sd['features.2.weight'])


def test_reference_format_checkpoints(tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, 'encoder.npz'))
    for circ, keys in ((False, g['keys_surface']), (True, g['keys_overhead'])):
        enc = cvig_fov.FOV_DSM(circ_padding=circ, seed=3)
        path = str(tmp_path / ('ck%d.pth' % circ))
        cvig_fov.save_reference_state_dict(enc, path)
        assert os.path.getsize(path) < 40e6                       # 7.6 M feature weights; the classifier zeros cost nothing
        back = torch.load(path)
        assert sorted(back.keys()) == sorted(str(k) for k in keys)            # the reference's own key set, classifier included
        for j, shape in cvig_fov.VGG16_CLASSIFIER_SHAPES.items():
            assert tuple(back['model.classifier.%d.weight' % j].shape) == shape and tuple(back['model.classifier.%d.bias' % j].shape) == shape[:1]
        # a strict load into a module with that key set and those shapes (what the reference's load_state_dict does, :511-512)
        mirror = cvig_fov.FOV_DSM(circ_padding=circ, seed=4)
        mirror.model.classifier = torch.nn.Sequential(torch.nn.Linear(25088, 4096), torch.nn.ReLU(True), torch.nn.Dropout(),
                                                      torch.nn.Linear(4096, 4096), torch.nn.ReLU(True), torch.nn.Dropout(),
                                                      torch.nn.Linear(4096, 1000))
        missing = mirror.load_state_dict(back, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        assert torch.equal(mirror.state_dict()['model.features.27%s.weight' % ('.layer' if circ else '')],
                           enc.state_dict()['model.features.27%s.weight' % ('.layer' if circ else '')])
        assert float(mirror.model.classifier[0].weight.abs().max()) == 0.0
        # and back into an encoder of this package
        again = cvig_fov.FOV_DSM(circ_padding=circ, seed=5)
        cvig_fov.load_reference_state_dict(again, back)
        for k, v in enc.state_dict().items():
            assert torch.equal(again.state_dict()[k], v)
    # classifier tensors that came with real weights travel through
    enc = cvig_fov.FOV_DSM.from_vgg16_state_dict(vgg16_state_dict(7))
    path = str(tmp_path / 'ck_vgg.pth')
    cvig_fov.save_reference_state_dict(enc, path)
    back = torch.load(path)
    assert float(back['model.classifier.3.weight'][0, 0]) == 4.0 and float(back['model.classifier.6.bias'][0]) == 8.0


def test_reference_strict_loads_our_checkpoints(tmp_path, golden_dir):
    """tests/golden/checkpoint_interop.npz (gen_golden.py --checkpoint-interop) records what the REFERENCE's FOV_DSM held after
    `load_state_dict` (strict, model/cvig_fov.py:511-512; model/cvig_semantic.py:542-543) of a file written by
    save_reference_state_dict, and the embeddings it then computed. Here the file is re-created from the same seeds: same key
    order, shapes and per-tensor digests as the reference's loaded state, and the oracle on those weights reproduces the
    reference's embeddings."""
    import hashlib
    from oracle import cvig_fov_oracle as O
    g = np.load(os.path.join(golden_dir, 'checkpoint_interop.npz'))
    seed = int(g['seed'])
    for tag, mod, circ, c in (('surface', cvig_fov, False, 3), ('overhead', cvig_fov, True, 3), ('semantic_overhead', cvig_semantic, True, 5)):
        w = synth.fov_dsm_weights(seed, in_channels=c)
        enc = mod.FOV_DSM(circ_padding=circ, weights=w)
        path = str(tmp_path / (tag + '.pth'))
        cvig_fov.save_reference_state_dict(enc, path)
        assert abs(os.path.getsize(path) - int(g['file_bytes_' + tag])) < 1e5       # the archive's member names carry the file name
        back = torch.load(path, map_location='cpu')
        assert sorted(back.keys()) == sorted(str(k) for k in g['keys_' + tag])
        for k, shape, digest in zip(g['keys_' + tag], g['shapes_' + tag], g['digests_' + tag]):
            t = back[str(k)]
            assert ','.join(str(n) for n in t.shape) == str(shape), k
            got = hashlib.sha256(np.ascontiguousarray(t.contiguous().numpy()).tobytes()).hexdigest()[:16]
            assert got == str(digest), k                       # the reference held exactly these bytes after its strict load
        x = torch.from_numpy(synth.normalized_images(seed, int(g['stream_' + tag]), (2, c, 128, 512)))
        wt = {k: (torch.from_numpy(a), torch.from_numpy(b)) for k, (a, b) in w.items()}
        with torch.no_grad():
            e = O.fov_dsm_forward(x, wt, circ).numpy()
        np.testing.assert_allclose(e, g['embed_' + tag], rtol=0, atol=1e-6)
