"""JPEG back end on the GPU (csrc/jpeg.hip: dequantisation, integer inverse DCT, fancy chroma upsampling, YCbCr -> RGB behind
the host's entropy decoding) against Pillow, the decoder behind the reference's skimage.io.imread (model/cvig_fov.py:88-89):
BYTE-identical images, on the committed fixtures, on freshly written files of BASELINE's raw sizes, and through the data path
(ImagePairDataset(raw='jpeg') -> collate_packed -> GpuPreprocess) against the host-decoded path."""
import io
import os

import numpy as np
import pytest
import torch

from witw_amd import jpeg

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'jpeg')


def test_device_decode_equals_pillow_on_fixtures():
    from PIL import Image
    exp = np.load(os.path.join(HERE, 'expected.npz'))
    names = sorted(n[:-4] for n in os.listdir(HERE) if n.endswith('.jpg'))
    items = []
    for n in names:
        c = jpeg.read_coef(os.path.join(HERE, n + '.jpg'))
        items.append(c if c is not None else np.asarray(Image.open(os.path.join(HERE, n + '.jpg'))))      # progressive: Pillow's bytes ride along
    assert sum(isinstance(i, jpeg.JpegCoef) for i in items) == len(names) - 1
    out = jpeg.decode(items, torch.device('cuda:0'))              # ONE batch: mixed sizes, layouts, a raw image
    for n, o in zip(names, out):
        e = exp[n] if exp[n].ndim == 3 else exp[n][:, :, None]
        assert tuple(o.shape) == e.shape, n
        np.testing.assert_array_equal(o.cpu().numpy(), e, err_msg=n)


def test_device_decode_with_raw_images_of_any_size_in_the_same_batch():
    """files left to Pillow travel as raw images in the batch block; their sizes must not move the coefficient data of the JPEG
    entries behind them off the 128-byte block grid the device back end addresses (round 4: a 5 x 6 x 3 image in front of a JPEG
    entry made it decode from 96 bytes too early)"""
    exp = np.load(os.path.join(HERE, 'expected.npz'))
    names = ['s420_odd', 's444_q10', 's422_q50', 'gray_q85']
    g = np.random.Generator(np.random.Philox(key=[8, 4]))
    items, want = [], []
    for k, n in enumerate(names * 2):
        raw = g.integers(0, 256, size=(3 + 2 * k, 5 + k, 3), dtype=np.uint8)
        items += [raw, jpeg.read_coef(os.path.join(HERE, n + '.jpg'))]
        want += [raw, exp[n] if exp[n].ndim == 3 else exp[n][:, :, None]]
    out = jpeg.decode(items, torch.device('cuda:0'))
    for k, (o, e) in enumerate(zip(out, want)):
        np.testing.assert_array_equal(o.cpu().numpy(), e, err_msg=str(k))


def test_device_decode_fresh_files():
    from PIL import Image
    g = np.random.Generator(np.random.Philox(key=[8, 3]))
    items, refs = [], []
    for (h, w) in ((512, 512), (224, 224), (750, 333), (8, 8), (100, 2)):
        for sub, q, kw in ((2, 90, {}), (1, 70, {'optimize': True}), (0, 95, {}), (2, 40, {'restart_marker_blocks': 5})):
            small = g.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
            a = np.asarray(Image.fromarray(small).resize((w + 16, h + 16), Image.BICUBIC))[8:8 + h, 8:8 + w]
            a = np.clip(a.astype(np.int16) + g.integers(-15, 16, size=(h, w, 3)), 0, 255).astype(np.uint8)
            bio = io.BytesIO()
            Image.fromarray(a).save(bio, 'JPEG', quality=q, subsampling=sub, **kw)
            items.append(jpeg.read_coef(bio.getvalue()))
            refs.append(np.asarray(Image.open(io.BytesIO(bio.getvalue()))))
    out = jpeg.decode(items, torch.device('cuda:0'))
    for k, (o, r) in enumerate(zip(out, refs)):
        np.testing.assert_array_equal(o.cpu().numpy(), r, err_msg=str(k))


def test_data_path_with_device_decode_equals_host_decode(tmp_path):
    """ImagePairDataset(raw='jpeg') hands out entropy-decoded files, collate_packed one block per side, GpuPreprocess decodes on
    the device and goes on as before: the same 'surface' / 'polar' tensors, bit for bit, as with Pillow's bytes (raw=True)."""
    from PIL import Image
    from witw_amd import cvig_fov
    g = np.random.Generator(np.random.Philox(key=[9, 9]))
    root = str(tmp_path)
    rows = []
    for i, ((hs, ws), (ho, wo)) in enumerate([((224, 224), (512, 512)), ((100, 333), (300, 200)), ((48, 80), (64, 64)), ((224, 224), (512, 512))]):
        for tag, (h, w) in (('su', (hs, ws)), ('ov', (ho, wo))):
            a = g.integers(0, 256, size=(h // 4 + 1, w // 4 + 1, 3), dtype=np.uint8).repeat(4, 0).repeat(4, 1)[:h, :w]
            Image.fromarray(a).save(os.path.join(root, '%s_%d.jpg' % (tag, i)), quality=88, progressive=(i == 2 and tag == 'ov'))
        rows.append('ov_%d.jpg,su_%d.jpg' % (i, i))
    csv = os.path.join(root, 'pairs.csv')
    open(csv, 'w').write('\n'.join(rows) + '\n')
    prep = cvig_fov.GpuPreprocess('cvusa', fov=360, random_orientation=False)
    outs = {}
    for mode in (True, 'jpeg'):
        ds = cvig_fov.ImagePairDataset('cvusa', csv, raw=mode)
        batch = cvig_fov.collate_packed([ds[i] for i in range(4)])
        outs[mode] = prep(batch)
        if mode == 'jpeg':
            assert batch['overhead_kind'] == jpeg.KIND_JPEG and int(batch['overhead_desc'][:, 24].sum()) == 1      # the progressive file
    assert torch.equal(outs[True]['surface'], outs['jpeg']['surface']) and torch.equal(outs[True]['polar'], outs['jpeg']['polar'])
    # through a DataLoader with workers and the prefetcher
    ds = cvig_fov.ImagePairDataset('cvusa', csv, raw='jpeg')
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=2, collate_fn=cvig_fov.collate_packed, pin_memory=True)
    got = [prep(st) for st in cvig_fov.DevicePrefetcher(loader, prep)]
    assert torch.equal(torch.cat([d['polar'] for d in got]), outs[True]['polar'])
    # the workers building their blocks in the parent's page-locked shared ring (no pickling copy, no pinning thread); both decoders;
    # 16 batches over 6 slots, so slots are recycled behind the copy events; a slot too small for a batch falls back to plain blocks
    import functools
    from witw_amd import ring as ring_mod
    for mode in ('jpeg', True):
        for slot_bytes in (4 << 20, 1 << 16):
            ring = ring_mod.PinnedRing(slots=6, slot_bytes=slot_bytes)      # >= workers x prefetch_factor + 2 (see PinnedRing)
            assert ring.registered and ring.mem.is_shared()
            ds = cvig_fov.ImagePairDataset('cvusa', csv, raw=mode)
            prep_r = cvig_fov.GpuPreprocess('cvusa', fov=360, random_orientation=False, ring=ring)
            loader = torch.utils.data.DataLoader(torch.utils.data.ConcatDataset([ds] * 8), batch_size=2, shuffle=False, num_workers=2,
                                                 collate_fn=functools.partial(cvig_fov.collate_packed, ring=ring))
            used_ring = 0
            polar = []
            for raw in loader:
                used_ring += 'ring' in raw
                polar.append(prep_r(raw)['polar'])
            assert used_ring == (16 if slot_bytes > (1 << 20) else 0)
            assert torch.equal(torch.cat(polar), torch.cat([outs[True]['polar']] * 8))
            ring.close()
            assert ring.free.qsize() == 6


def _fresh(g, h, w, sub, q, **kw):
    from PIL import Image
    small = g.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    a = np.asarray(Image.fromarray(small).resize((w + 16, h + 16), Image.BICUBIC))[8:8 + h, 8:8 + w]
    a = np.clip(a.astype(np.int16) + g.integers(-15, 16, size=(h, w, 3)), 0, 255).astype(np.uint8)
    bio = io.BytesIO()
    Image.fromarray(a).save(bio, 'JPEG', quality=q, subsampling=sub, **kw)
    return bio.getvalue()


def test_device_entropy_decode_equals_the_host_coefficients():
    """round 6: files with restart markers are Huffman-decoded ON THE DEVICE (csrc/jpeg.hip jpeg_huffman_kernel, one thread per restart
    interval, the host only scans for the markers): the coefficient blocks are those of witw_jpeg_decode_coef bit for bit -- fixtures,
    fresh files of BASELINE's raw sizes, every sampling, optimised (per-file) Huffman tables, one / several / partial MCU rows per
    interval, more intervals than the 64 lanes of a wave."""
    from witw_amd import _lib, ops
    dev = torch.device('cuda:0')
    g = np.random.Generator(np.random.Philox(key=[13, 1]))
    files = [open(os.path.join(HERE, n), 'rb').read() for n in ('s420_rst.jpg', 's422_rst.jpg')]
    for (h, w, sub, q, kw) in ((512, 512, 2, 90, {'restart_marker_rows': 1}), (224, 224, 2, 90, {'restart_marker_rows': 1}),
                               (750, 333, 1, 70, {'restart_marker_blocks': 7, 'optimize': True}), (100, 2, 0, 95, {'restart_marker_blocks': 1}),
                               (8, 8, 2, 40, {'restart_marker_rows': 1}), (640, 640, 0, 99, {'restart_marker_blocks': 16}),
                               (300, 300, 2, 5, {'restart_marker_rows': 2})):
        files.append(_fresh(g, h, w, sub, q, **kw))
    items = [jpeg.open_file(f) for f in files]
    plans = [it.entropy_plan() for it in items]
    assert all(p is not None for p in plans)
    blocks = np.array([int(it.info[5]) for it in items], dtype=np.int64)
    first = np.cumsum(blocks) - blocks
    coef = torch.zeros((int(blocks.sum()), 64), dtype=torch.int16, device=dev)
    keep, rows = [], []
    for it, (plan, _qt), f0 in zip(items, plans, first):
        raw = np.zeros((it.data.size + 24 + 7) // 8 * 8, dtype=np.uint8)
        raw[:it.data.size] = it.data
        rb, pb = torch.from_numpy(raw).to(dev), torch.from_numpy(np.concatenate([plan, np.zeros(16, np.uint8)])).to(dev)
        keep += [rb, pb]
        assert rb.data_ptr() % 8 == 0 and pb.data_ptr() % 4 == 0
        rows.append((rb.data_ptr(), pb.data_ptr(), coef.data_ptr() + int(f0) * 128, it.data.size))
    files_t = torch.tensor(rows, dtype=torch.int64, device=dev)
    errors = torch.zeros((len(items),), dtype=torch.int32, device=dev)
    n_max = max(int(np.frombuffer(p[0][4:8].tobytes(), dtype=np.int32)[0]) for p in plans)
    assert n_max > 128                                                      # some file takes more than two waves
    _lib.check(_lib.load().witw_jpeg_huffman(files_t.data_ptr(), len(items), n_max, errors.data_ptr(), ops._stream()), 'witw_jpeg_huffman')
    torch.cuda.synchronize()
    assert int(errors.abs().sum()) == 0
    got = coef.cpu().numpy()
    for k, (raw, f0, nb) in enumerate(zip(files, first, blocks)):
        np.testing.assert_array_equal(got[f0:f0 + nb], jpeg.read_coef(raw).coef, err_msg='file %d' % k)
    # a damaged interval is flagged, the others of the file still decode, nothing is written outside the file's area
    bad = bytearray(files[2])
    p = bad.find(b'\xff\xd3')
    bad[p + 8:p + 40] = b'\xff' * 32
    it = jpeg.open_file(bytes(bad))
    plan = it.entropy_plan()
    if plan is not None:
        raw = np.zeros((it.data.size + 31) // 8 * 8, dtype=np.uint8)
        raw[:it.data.size] = it.data
        rb, pb = torch.from_numpy(raw).to(dev), torch.from_numpy(plan[0]).to(dev)
        c2 = torch.zeros((int(it.info[5]) + 8, 64), dtype=torch.int16, device=dev)
        e2 = torch.zeros((1,), dtype=torch.int32, device=dev)
        row = torch.tensor([[rb.data_ptr(), pb.data_ptr(), c2.data_ptr(), it.data.size]], dtype=torch.int64, device=dev)
        _lib.check(_lib.load().witw_jpeg_huffman(row.data_ptr(), 1, 4096, e2.data_ptr(), ops._stream()), 'witw_jpeg_huffman')
        torch.cuda.synchronize()
        assert int(e2.item()) == 1 and int(c2[int(it.info[5]):].abs().sum()) == 0


def test_device_entropy_path_gives_pillows_bytes_in_mixed_batches(monkeypatch):
    """pack() ships the FILE BYTES (descriptor column 26) and decode_packed entropy-decodes them on the device in front of the usual
    back end -- restart-marker files on the interval kernel, marker-less ones on the self-synchronising kernel (the small one on a
    single thread of the interval kernel); a progressive file (Pillow's bytes) and a raw array ride in the same batch: every image
    equals Pillow's decode byte for byte, and nothing was flagged."""
    from PIL import Image
    monkeypatch.setattr(jpeg, 'DEVICE_ENTROPY', 'all')
    del jpeg._ERRORS[:]
    g = np.random.Generator(np.random.Philox(key=[13, 2]))
    raws = [_fresh(g, 512, 512, 2, 90, restart_marker_rows=1), _fresh(g, 224, 224, 2, 90), _fresh(g, 224, 224, 2, 90, restart_marker_rows=1),
            open(os.path.join(HERE, 'prog_q85.jpg'), 'rb').read(), _fresh(g, 97, 131, 1, 60, restart_marker_blocks=3, optimize=True),
            open(os.path.join(HERE, 's422_rst.jpg'), 'rb').read(), _fresh(g, 64, 64, 0, 80)]
    items, refs = [], []
    for r in raws:
        it = jpeg.open_file(r)
        ref = np.asarray(Image.open(io.BytesIO(r)))
        items.append(it if it is not None else ref)
        refs.append(ref)
    items.insert(2, g.integers(0, 256, size=(7, 9, 3), dtype=np.uint8))
    refs.insert(2, items[2])
    buf, desc, _k = jpeg.pack(items)
    assert list(desc[:, 26].numpy()) == [1, 1, 0, 1, 0, 1, 1, 1] and int(desc[4, 24]) == 1
    assert list(desc[:, 29].numpy())[1] == 1 and list(desc[:, 29].numpy())[0] > 1      # no markers: one interval (self-synchronising kernel)
    out = jpeg.decode(items, torch.device('cuda:0'))
    for k, (o, r) in enumerate(zip(out, refs)):
        np.testing.assert_array_equal(o.cpu().numpy(), r if r.ndim == 3 else r[:, :, None], err_msg=str(k))
    assert jpeg.entropy_errors() == 0


def test_data_path_with_restart_marker_files_equals_host_decode(tmp_path):
    """The drivers' data path on a data set written with restart markers: DataLoader workers only scan for the markers, the GPU does
    Huffman decoding, IDCT, upsampling, colour conversion and the transforms -- the same 'surface' / 'polar' bits as Pillow's decode."""
    from PIL import Image
    from witw_amd import cvig_fov
    g = np.random.Generator(np.random.Philox(key=[13, 3]))
    root = str(tmp_path)
    rows = []
    for i in range(6):
        for tag, (h, w) in (('su', (224, 224)), ('ov', (512, 512))):
            open(os.path.join(root, '%s_%d.jpg' % (tag, i)), 'wb').write(_fresh(g, h, w, 2, 90, restart_marker_rows=1))
        rows.append('ov_%d.jpg,su_%d.jpg' % (i, i))
    csv = os.path.join(root, 'pairs.csv')
    open(csv, 'w').write('\n'.join(rows) + '\n')
    del jpeg._ERRORS[:]
    prep = cvig_fov.GpuPreprocess('cvusa', fov=360, random_orientation=False)
    ref = prep(cvig_fov.collate_packed([cvig_fov.ImagePairDataset('cvusa', csv, raw=True)[i] for i in range(6)]))
    ds = cvig_fov.ImagePairDataset('cvusa', csv, raw='jpeg')
    batch = cvig_fov.collate_packed([ds[i] for i in range(6)])
    assert int(batch['overhead_desc'][:, 26].sum()) == 6 and int(batch['surface_desc'][:, 26].sum()) == 6
    # the block that crosses PCIe holds file bytes, not coefficient blocks: an order of magnitude smaller
    assert batch['overhead_bytes'].numel() < 6 * 512 * 512 * 3 // 4
    got = prep(batch)
    assert torch.equal(got['surface'], ref['surface']) and torch.equal(got['polar'], ref['polar'])
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, num_workers=2, collate_fn=cvig_fov.collate_packed, pin_memory=True)
    outs = [prep(st) for st in cvig_fov.DevicePrefetcher(loader, prep)]
    assert torch.equal(torch.cat([d['polar'] for d in outs]), ref['polar']) and jpeg.entropy_errors() == 0


@pytest.mark.parametrize('threads', [512, 256, 1024])
def test_selfsync_decode_equals_the_host_coefficients(threads):
    """round 6, step two: files WITHOUT restart markers on jpeg_selfsync_kernel (one workgroup of `threads` threads per file: unstuff,
    synchronise as many subsequences round by round, number the blocks, write, DC prefix sums): coefficient blocks bit for bit those of
    witw_jpeg_decode_coef -- every fixture this decoder takes, fresh files of BASELINE's raw sizes, all samplings, optimised tables,
    very low and very high quality (long runs of EOB-only blocks / long codes), grey, sizes that are not multiples of the MCU."""
    from witw_amd import _lib, ops
    dev = torch.device('cuda:0')
    g = np.random.Generator(np.random.Philox(key=[14, 1]))
    files = [open(os.path.join(HERE, n), 'rb').read() for n in sorted(os.listdir(HERE)) if n.endswith('.jpg') and 'rst' not in n and 'prog' not in n]
    for (h, w, sub, q, kw) in ((512, 512, 2, 90, {}), (224, 224, 2, 90, {}), (750, 333, 1, 70, {'optimize': True}), (640, 640, 0, 99, {}),
                               (300, 300, 2, 3, {}), (1024, 768, 2, 85, {}), (97, 131, 1, 60, {'optimize': True}), (33, 999, 0, 50, {})):
        files.append(_fresh(g, h, w, sub, q, **kw))
    from PIL import Image
    bio = io.BytesIO()
    Image.fromarray(g.integers(0, 256, size=(200, 120), dtype=np.uint8)).save(bio, 'JPEG', quality=80)      # grey: one component
    files.append(bio.getvalue())
    items = [jpeg.open_file(f) for f in files]
    plans = [it.entropy_plan() for it in items]
    assert all(p is not None and int(np.frombuffer(p[0][4:8].tobytes(), dtype=np.int32)[0]) == 1 for p in plans)
    blocks = np.array([int(it.info[5]) for it in items], dtype=np.int64)
    first = np.cumsum(blocks) - blocks
    coef = torch.zeros((int(blocks.sum()), 64), dtype=torch.int16, device=dev)
    keep, rows = [], []
    for it, (plan, _qt), f0 in zip(items, plans, first):
        raw = np.zeros((it.data.size + 24 + 7) // 8 * 8, dtype=np.uint8)
        raw[:it.data.size] = it.data
        rb, pb = torch.from_numpy(raw).to(dev), torch.from_numpy(np.concatenate([plan, np.zeros(16, np.uint8)])).to(dev)
        sc = torch.empty(((it.data.size + 32 + 7) // 8 * 8,), dtype=torch.uint8, device=dev)
        keep += [rb, pb, sc]
        rows.append((rb.data_ptr(), pb.data_ptr(), coef.data_ptr() + int(f0) * 128, it.data.size, sc.data_ptr(), 0))
    files_t = torch.tensor(rows, dtype=torch.int64, device=dev)
    errors = torch.zeros((len(items),), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().witw_jpeg_huffman_selfsync_threads(files_t.data_ptr(), len(items), threads, errors.data_ptr(), ops._stream()),
               'witw_jpeg_huffman_selfsync_threads')
    torch.cuda.synchronize()
    assert int(errors.abs().sum()) == 0, errors
    got = coef.cpu().numpy()
    for k, (raw, f0, nb) in enumerate(zip(files, first, blocks)):
        ref = jpeg.read_coef(raw).coef
        bad = np.nonzero((got[f0:f0 + nb] != ref).any(axis=1))[0]
        assert bad.size == 0, 'file %d (%d bytes): %d of %d blocks differ, first %s' % (k, len(raw), bad.size, nb, bad[:5])


def test_device_flags_damaged_files_and_the_data_path_hands_them_to_pillow(tmp_path, monkeypatch):
    """a truncated file: the device decoder flags it (its block count does not come out), the staging step looks at the flags and lets
    Pillow decode the flagged file from the bytes still in the host block -- the image the host path (and the reference's imread with
    libjpeg's tolerance) produces; the other files of the batch are untouched"""
    from PIL import Image, ImageFile
    from witw_amd import cvig_fov
    monkeypatch.setattr(jpeg, 'DEVICE_ENTROPY', 'all')
    g = np.random.Generator(np.random.Philox(key=[14, 2]))
    good = [_fresh(g, 224, 224, 2, 90), _fresh(g, 224, 224, 2, 90, restart_marker_rows=1), _fresh(g, 224, 224, 2, 90)]
    cut = good[2][:len(good[2]) * 2 // 3]
    old = ImageFile.LOAD_TRUNCATED_IMAGES
    ImageFile.LOAD_TRUNCATED_IMAGES = True
    try:
        refs = [np.asarray(Image.open(io.BytesIO(b))) for b in (good[0], good[1], cut)]
        items = [jpeg.open_file(good[0]), jpeg.open_file(good[1]), jpeg.open_file(cut)]
        buf, desc, _k = jpeg.pack(items)
        assert list(desc[:, 26].numpy()) == [1, 1, 1]
        before = jpeg.REPAIRED[0]
        dbuf = buf.to('cuda:0')
        keep, table = jpeg.decode_packed(dbuf, desc, host_buf=buf)
        torch.cuda.synchronize()
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = old
    assert jpeg.REPAIRED[0] == before + 1
    for i, r in enumerate(refs):
        H, W, C = int(table[i, 1]), int(table[i, 2]), int(table[i, 4])
        src = next(t for t in keep if t.dtype == torch.uint8 and t.data_ptr() <= int(table[i, 0]) < t.data_ptr() + max(1, t.numel()))
        o = int(table[i, 0]) - src.data_ptr()
        np.testing.assert_array_equal(src.reshape(-1)[o:o + H * W * C].reshape(H, W, C).cpu().numpy(), r, err_msg=str(i))


def test_data_path_with_selfsync_mode_equals_host_decode(tmp_path, monkeypatch):
    """jpeg.DEVICE_ENTROPY = 'all': ordinary files (no restart markers) are Huffman-decoded by the self-synchronising kernel inside the
    drivers' data path -- loader workers included -- with the same 'surface' / 'polar' bits as Pillow's decode."""
    from witw_amd import cvig_fov
    monkeypatch.setattr(jpeg, 'DEVICE_ENTROPY', 'all')
    del jpeg._ERRORS[:]                                    # (flags of earlier tests' batches, e.g. the deliberately damaged file)
    g = np.random.Generator(np.random.Philox(key=[15, 1]))
    root = str(tmp_path)
    rows = []
    for i in range(4):
        for tag, (h, w) in (('su', (224, 224)), ('ov', (512, 512))):
            open(os.path.join(root, '%s_%d.jpg' % (tag, i)), 'wb').write(_fresh(g, h, w, 2, 90))
        rows.append('ov_%d.jpg,su_%d.jpg' % (i, i))
    csv = os.path.join(root, 'pairs.csv')
    open(csv, 'w').write('\n'.join(rows) + '\n')
    prep = cvig_fov.GpuPreprocess('cvusa', fov=360, random_orientation=False)
    ref = prep(cvig_fov.collate_packed([cvig_fov.ImagePairDataset('cvusa', csv, raw=True)[i] for i in range(4)]))
    ds = cvig_fov.ImagePairDataset('cvusa', csv, raw='jpeg')
    batch = cvig_fov.collate_packed([ds[i] for i in range(4)])
    assert int(batch['overhead_desc'][:, 26].sum()) == 4 and list(batch['overhead_desc'][:, 29].numpy()) == [1, 1, 1, 1]
    got = prep(batch)
    assert torch.equal(got['surface'], ref['surface']) and torch.equal(got['polar'], ref['polar'])
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=2, collate_fn=cvig_fov.collate_packed, pin_memory=True)
    outs = [prep(st) for st in cvig_fov.DevicePrefetcher(loader, prep)]
    assert torch.equal(torch.cat([d['polar'] for d in outs]), ref['polar']) and jpeg.entropy_errors() == 0


def test_one_set_of_launches_over_the_parts_of_a_batch_equals_part_by_part_decoding(monkeypatch):
    """decode_packed_multi: the blocks of a batch that arrived in pieces (separate allocations) are decoded by launches that cover all
    of them -- same images as decoding every block by itself, rows in part order; a damaged file in the SECOND part is re-decoded by
    Pillow from that part's host block and only its row is patched (deferred: nothing is read back before finish())."""
    from PIL import Image, ImageFile
    monkeypatch.setattr(jpeg, 'DEVICE_ENTROPY', 'all')
    g = np.random.Generator(np.random.Philox(key=[16, 3]))
    a = [_fresh(g, 224, 224, 2, 90, restart_marker_blocks=2), _fresh(g, 97, 131, 1, 60), g.integers(0, 256, size=(9, 7, 3), dtype=np.uint8)]
    whole = _fresh(g, 224, 224, 2, 85)
    b = [open(os.path.join(HERE, 'prog_q85.jpg'), 'rb').read(), whole[:len(whole) * 3 // 5], _fresh(g, 512, 512, 2, 90, restart_marker_blocks=1),
         _fresh(g, 64, 64, 0, 80)]
    old = ImageFile.LOAD_TRUNCATED_IMAGES
    ImageFile.LOAD_TRUNCATED_IMAGES = True
    try:
        def item(r):
            if isinstance(r, np.ndarray):
                return r, r
            ref = np.asarray(Image.open(io.BytesIO(r)))
            it = jpeg.open_file(r)
            return (it if it is not None else ref), (ref if ref.ndim == 3 else ref[:, :, None])
        parts, refs = [], []
        for raws in (a, b):
            items = [item(r) for r in raws]
            refs += [r for _i, r in items]
            buf, desc, _k = jpeg.pack([i for i, _r in items])
            parts.append((buf.to('cuda:0'), desc, buf))
        before = jpeg.REPAIRED[0]
        keep, table, finish = jpeg.decode_packed_multi(parts, defer=True)
        assert finish is not None and jpeg.REPAIRED[0] == before and table.shape[0] == len(refs)
        finish()
        torch.cuda.synchronize()
        assert jpeg.REPAIRED[0] == before + 1
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = old
    pool = [p[0] for p in parts] + keep
    for i, r in enumerate(refs):
        H, W, C = int(table[i, 1]), int(table[i, 2]), int(table[i, 4])
        assert (H, W, C) == r.shape, i
        src = next(t for t in pool if t.dtype == torch.uint8 and t.data_ptr() <= int(table[i, 0]) and int(table[i, 0]) + H * W * C <= t.data_ptr() + t.numel())
        o = int(table[i, 0]) - src.data_ptr()
        np.testing.assert_array_equal(src.reshape(-1)[o:o + H * W * C].reshape(H, W, C).cpu().numpy(), r, err_msg=str(i))


def test_grouped_prefetcher_decodes_both_sides_of_all_parts_together(tmp_path, monkeypatch):
    """DevicePrefetcher(group=2) over loader batches of 2: GpuPreprocess.stage queues the four blocks and decodes ground and overhead files
    of both parts with one set of launches -- the 'surface' / 'polar' bits of Pillow's decode, in loader order."""
    from witw_amd import cvig_fov
    monkeypatch.setattr(jpeg, 'DEVICE_ENTROPY', 'all')
    del jpeg._ERRORS[:]
    g = np.random.Generator(np.random.Philox(key=[17, 1]))
    root = str(tmp_path)
    rows = []
    for i in range(4):
        for tag, (h, w) in (('su', (224, 224)), ('ov', (512, 512))):
            kw = {'restart_marker_blocks': 2} if i % 2 else {}
            open(os.path.join(root, '%s_%d.jpg' % (tag, i)), 'wb').write(_fresh(g, h, w, 2, 90, **kw))
        rows.append('ov_%d.jpg,su_%d.jpg' % (i, i))
    csv = os.path.join(root, 'pairs.csv')
    open(csv, 'w').write('\n'.join(rows) + '\n')
    prep = cvig_fov.GpuPreprocess('cvusa', fov=360, random_orientation=False)
    ref = prep(cvig_fov.collate_packed([cvig_fov.ImagePairDataset('cvusa', csv, raw=True)[i] for i in range(4)]))
    ds = cvig_fov.ImagePairDataset('cvusa', csv, raw='jpeg')
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0, collate_fn=cvig_fov.collate_packed)
    n_launch = len(jpeg._ERRORS)
    outs = [prep(st) for st in cvig_fov.DevicePrefetcher(loader, prep, group=2)]
    assert len(outs) == 1 and outs[0]['surface'].shape[0] == 4
    assert len(jpeg._ERRORS) == n_launch + 1                  # ONE device-entropy decode for the 8 files
    assert torch.equal(outs[0]['surface'], ref['surface']) and torch.equal(outs[0]['polar'], ref['polar']) and jpeg.entropy_errors() == 0


@pytest.mark.parametrize('seed', [5, 6, 7])
def test_device_decode_of_randomly_parametrised_files_equals_pillow(seed):
    """the default data path (every baseline Huffman file entropy-decoded on the device) over 96 files whose size (8..700, mostly not a
    multiple of the MCU), sampling (4:4:4 / 4:2:2 / 4:2:0 / grey), quality (1..100), Huffman tables (standard / optimised) and restart
    interval (none, 1..5 MCU rows, 1..40 MCUs) are drawn at random: every image byte for byte Pillow's, nothing flagged. One batch, one set
    of launches -- interval kernel and self-synchronising kernel side by side."""
    from PIL import Image
    assert jpeg.DEVICE_ENTROPY == 'all'
    del jpeg._ERRORS[:]
    g = np.random.Generator(np.random.Philox(key=[18, seed]))
    raws = []
    for i in range(96):
        h, w = int(g.integers(8, 701)), int(g.integers(8, 701))
        if i % 7 == 0:
            h, w = int(g.integers(8, 40)), int(g.integers(8, 40))      # a few tiny ones (a single MCU or two)
        sub = int(g.integers(0, 4))
        kw = {'optimize': True} if g.integers(0, 2) else {}
        r = int(g.integers(0, 3))
        if r == 1:
            kw['restart_marker_rows'] = int(g.integers(1, 6))
        elif r == 2:
            kw['restart_marker_blocks'] = int(g.integers(1, 41))
        q = int(g.integers(1, 101))
        if sub == 3:      # grey
            small = g.integers(0, 256, size=(h // 8 + 2, w // 8 + 2), dtype=np.uint8)
            a = np.asarray(Image.fromarray(small).resize((w, h), Image.BICUBIC))
            a = np.clip(a.astype(np.int16) + g.integers(-15, 16, size=(h, w)), 0, 255).astype(np.uint8)
            bio = io.BytesIO()
            Image.fromarray(a).save(bio, 'JPEG', quality=q, **kw)
            raws.append(bio.getvalue())
        else:
            raws.append(_fresh(g, h, w, sub, q, **kw))
    items = [jpeg.open_file(r) for r in raws]
    assert all(it is not None for it in items)
    buf, desc, _k = jpeg.pack(items)
    d = desc.numpy()
    assert int(d[:, 26].sum()) == len(raws)                     # all of them travel as file bytes
    assert (d[:, 29] > 1).any() and (d[:, 29] == 1).any()       # both kernels have work
    out = jpeg.decode(items, torch.device('cuda:0'))
    for k, (o, r) in enumerate(zip(out, raws)):
        ref = np.asarray(Image.open(io.BytesIO(r)))
        np.testing.assert_array_equal(o.cpu().numpy(), ref if ref.ndim == 3 else ref[:, :, None], err_msg='file %d' % k)
    assert jpeg.entropy_errors() == 0


def test_device_decoders_survive_corrupted_entropy_data():
    """64 files with a few bytes of their entropy-coded data overwritten (never with FF: no new markers): both device decoders run to the
    end on whatever the bits now say (every store stays inside the file's coefficient area by construction: block addresses come from
    loop counters, zig-zag positions are checked, readers stop at the end of their data); a file they flag is re-decoded by Pillow and
    equals Pillow's reading, the others keep their shape. Pillow itself may refuse a file: that error surfaces from the staging step
    as it does from the reference's imread."""
    from PIL import Image, ImageFile
    assert jpeg.DEVICE_ENTROPY == 'all'
    g = np.random.Generator(np.random.Philox(key=[19, 2]))
    raws = []
    for i in range(64):
        kw = {} if i % 2 else {'restart_marker_blocks': int(g.integers(1, 9))}
        raw = bytearray(_fresh(g, int(g.integers(40, 400)), int(g.integers(40, 400)), int(g.integers(0, 3)), int(g.integers(30, 96)), **kw))
        sos = raw.rfind(bytes([0xff, 0xda]))
        first = sos + 2 + ((raw[sos + 2] << 8) | raw[sos + 3])
        for _ in range(int(g.integers(1, 4))):
            pos = int(g.integers(first, len(raw) - 2))
            if raw[pos] != 0xff and raw[pos - 1] != 0xff:      # leave markers and stuffed bytes alone
                raw[pos] = int(g.integers(0, 255))
        raws.append(bytes(raw))
    old = ImageFile.LOAD_TRUNCATED_IMAGES
    ImageFile.LOAD_TRUNCATED_IMAGES = True
    try:
        items = [jpeg.open_file(r) for r in raws]
        keep_idx = [i for i, it in enumerate(items) if isinstance(it, jpeg.JpegFile)]
        assert len(keep_idx) >= 48                               # (a corrupted restart marker sequence sends a file to the host path: not this test's)
        buf, desc, _k = jpeg.pack([items[i] for i in keep_idx])
        assert int(desc[:, 26].sum()) == len(keep_idx)
        before = jpeg.REPAIRED[0]
        dbuf = buf.to('cuda:0')
        try:
            keep, table = jpeg.decode_packed(dbuf, desc, host_buf=buf)
        except OSError:
            torch.cuda.synchronize()
            return                                               # Pillow refused a flagged file: the kernels had finished by then
        torch.cuda.synchronize()
        flagged = jpeg.REPAIRED[0] - before
        assert flagged >= 1                                      # some of the damage is always noticed
        for j, i in enumerate(keep_idx):
            ref = np.asarray(Image.open(io.BytesIO(raws[i])))
            H, W, C = int(table[j, 1]), int(table[j, 2]), int(table[j, 4])
            assert (H, W) == ref.shape[:2] and C == (ref.shape[2] if ref.ndim == 3 else 1)
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = old
