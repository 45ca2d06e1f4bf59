"""JPEG decode, CPU side: the host entropy decoder (witw_amd/csrc_host/jpeg_coef.cpp -> libwitw_jpeg.so) followed by the oracle's
restatement of libjpeg's back end (oracle/jpeg_oracle.py: dequantisation, 'islow' inverse DCT, fancy upsampling, YCbCr -> RGB)
against what Pillow -- the decoder behind the reference's skimage.io.imread, model/cvig_fov.py:88-89 -- makes of the same files:
byte for byte, on the committed fixtures (tests/golden/jpeg/, expected.npz written by gen_jpeg_fixtures.py) and on files written
here. This pins the oracle; tests/test_jpeg_gpu.py holds the HIP kernels to the same bytes."""
import io
import os

import numpy as np
import pytest

from oracle import jpeg_oracle as J
from witw_amd import jpeg

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'jpeg')


def fixture_names():
    return sorted(n[:-4] for n in os.listdir(HERE) if n.endswith('.jpg'))


@pytest.mark.parametrize('name', fixture_names())
def test_entropy_decoder_and_oracle_match_pillow_on_fixtures(name):
    exp = np.load(os.path.join(HERE, 'expected.npz'))[name]
    c = jpeg.read_coef(os.path.join(HERE, name + '.jpg'))
    if name.startswith('prog'):
        assert c is None                    # progressive: left to the host decoder
        return
    assert c is not None and c.shape[:2] == exp.shape[:2]
    out = J.decode(c.info, c.coef, c.qt)
    assert out.dtype == np.uint8 and out.shape == exp.shape
    np.testing.assert_array_equal(out, exp)


def test_fixtures_are_what_this_pillow_decodes():
    """the expected arrays came from the build container's Pillow; a box whose Pillow decodes differently would show here"""
    from PIL import Image
    exp = np.load(os.path.join(HERE, 'expected.npz'))
    for name in fixture_names():
        np.testing.assert_array_equal(np.asarray(Image.open(os.path.join(HERE, name + '.jpg'))), exp[name], err_msg=name)


def test_fresh_files_all_layouts():
    from PIL import Image
    g = np.random.Generator(np.random.Philox(key=[5, 77]))
    n = 0
    for (h, w) in ((224, 224), (130, 259), (9, 17), (16, 16)):
        for sub in (0, 1, 2):
            for q, kw in ((92, {}), (35, {'optimize': True}), (60, {'restart_marker_blocks': 2})):
                a = g.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
                a[h // 3:, w // 4:] //= 3                     # some structure, some noise
                bio = io.BytesIO()
                Image.fromarray(a).save(bio, 'JPEG', quality=q, subsampling=sub, **kw)
                ref = np.asarray(Image.open(io.BytesIO(bio.getvalue())))
                c = jpeg.read_coef(bio.getvalue())
                assert c is not None
                np.testing.assert_array_equal(J.decode(c.info, c.coef, c.qt), ref, err_msg=str((h, w, sub, q, kw)))
                n += 1
    assert n == 36


def test_garbage_and_truncated_streams_are_refused_not_crashed():
    assert jpeg.read_coef(b'') is None and jpeg.read_coef(b'\xff\xd8\xff') is None and jpeg.read_coef(b'not a jpeg at all') is None
    data = open(os.path.join(HERE, 's420_q90.jpg'), 'rb').read()
    assert jpeg.read_coef(data[:200]) is None                      # headers cut
    assert jpeg.read_coef(data[:len(data) // 2] + b'\xff\xd9') is None or True      # entropy data cut: refused or zero-filled, never a crash
    g = np.random.Generator(np.random.Philox(key=[6, 1]))
    for _ in range(200):                                           # random corruption of the entropy-coded part
        b = bytearray(data)
        for _k in range(5):
            b[int(g.integers(300, len(b)))] = int(g.integers(0, 256))
        jpeg.read_coef(bytes(b))


def test_pack_keeps_coefficient_entries_on_the_128_byte_block_grid():
    """the device back end addresses coefficients in 128-byte blocks from the start of the packed buffer: a raw image (a file left
    to Pillow) of any size in front of a JPEG entry must not move it off that grid"""
    c = jpeg.read_coef(os.path.join(HERE, 's420_odd.jpg'))
    for shape in ((5, 6, 3), (7, 9, 3), (1, 1, 3), (33, 17, 1)):
        raw = np.arange(int(np.prod(shape)), dtype=np.uint32).astype(np.uint8).reshape(shape)
        buf, desc, _k = jpeg.pack([c, raw, c, raw, c])
        d = desc.numpy()
        assert not (d[:, 0] % 128).any(), d[:, 0]
        planes, images, blk, pbytes, obytes, qt_base = jpeg.decode_tables(d[d[:, 24] == 0])
        b = buf.numpy()
        nb = [int(c.info[8 + 4 * k] * c.info[9 + 4 * k]) for k in range(int(c.info[2]))]      # blocks wide x blocks high per component
        starts = [sum(nb[:k]) for k in range(len(nb))]
        assert planes.shape[0] == 3 * len(nb)
        for k, row in enumerate(planes):      # image-major, component-minor: the block the device reads first is the block the host wrote
            got = b[int(row[0]) * 128:int(row[0]) * 128 + 128].view(np.int16)
            np.testing.assert_array_equal(got, c.coef[starts[k % len(nb)]])
    bad = d.copy()
    bad[2, 0] += 16
    with pytest.raises(ValueError):
        jpeg.decode_tables(bad[bad[:, 24] == 0])


def test_pack_layout():
    c = jpeg.read_coef(os.path.join(HERE, 's420_odd.jpg'))
    raw = np.arange(5 * 6 * 3, dtype=np.uint8).reshape(5, 6, 3)
    buf, desc, kind = jpeg.pack([c, raw, c])
    assert kind == jpeg.KIND_JPEG and tuple(desc.shape) == (3, jpeg.DESC_COLS)
    b = buf.numpy()
    for i in (0, 2):
        o, q = int(desc[i, 0]), int(desc[i, 1])
        assert o % 16 == 0 and q % 16 == 0 and desc[i, 24] == 0
        np.testing.assert_array_equal(b[o:o + c.coef.size * 2].view(np.int16).reshape(-1, 64), c.coef)
        np.testing.assert_array_equal(b[q:q + c.qt.size * 2].view(np.uint16).reshape(-1, 64), c.qt)
        assert list(desc[i, 2:5]) == [37, 51, 3]
    o = int(desc[1, 0])
    assert desc[1, 24] == 1 and desc[1, 25] == 3 and list(desc[1, 2:4]) == [5, 6]
    np.testing.assert_array_equal(b[o:o + raw.size].reshape(5, 6, 3), raw)


# ---- round 4: files the device path must leave to the host decoder, and hostile tables ------------------------------------------

def _dht_segment(counts, nsym=None):
    nsym = sum(counts) if nsym is None else nsym
    body = bytes([0x00]) + bytes(counts) + bytes((i & 255) for i in range(nsym))
    return b'\xff\xc4' + (len(body) + 2).to_bytes(2, 'big') + body


@pytest.mark.parametrize('pos', range(16))
def test_oversubscribed_huffman_table_is_refused_without_writing_past_the_tables(pos):
    """a DHT whose code lengths over-subscribe the code space (255 codes of ONE length: the symbol count passes the <= 256 check)
    used to be written into the look-up tables before it was rejected -- ~130 KB past them for length 1"""
    counts = [0] * 16
    counts[pos] = 255
    for tail in (b'', b'\xff\xd9'):
        assert jpeg.read_coef(b'\xff\xd8' + _dht_segment(counts) + tail) is None
    counts = [2] + [0] * 15                              # exactly full at length 1, then one more code at length 2
    counts[1] = 1
    assert jpeg.read_coef(b'\xff\xd8' + _dht_segment(counts) + b'\xff\xd9') is None


def _patch_sof(data, comp, hv):
    """the sampling byte of component `comp` in the SOF0 segment"""
    b = bytearray(data)
    i = b.index(b'\xff\xc0')
    b[i + 4 + 6 + 3 * comp + 1] = hv
    return bytes(b)


def test_440_sampling_is_left_to_the_host_decoder():
    """h1v2 (4:4:0, e.g. a losslessly rotated 4:2:2 file): the device back end has no such upsampler"""
    data = open(os.path.join(HERE, 's422_q50.jpg'), 'rb').read()
    assert jpeg.open_file(data) is not None
    assert jpeg.open_file(_patch_sof(data, 0, 0x12)) is None and jpeg.read_coef(_patch_sof(data, 0, 0x12)) is None
    from PIL import Image
    a = (np.arange(48 * 40 * 3, dtype=np.uint32).reshape(48, 40, 3) * 7 % 256).astype(np.uint8)
    try:
        bio = io.BytesIO()
        Image.fromarray(a).save(bio, 'JPEG', quality=90, subsampling='4:4:0')
    except (TypeError, ValueError, KeyError):
        return                                           # this Pillow cannot write 4:4:0
    assert jpeg.open_file(bio.getvalue()) is None


def test_rgb_colourspace_files_are_left_to_the_host_decoder():
    """libjpeg takes the colour space from the JFIF / Adobe markers and the component ids (jdapimin.c default_decompress_parms);
    the device back end always converts YCbCr -> RGB, so files stored as RGB must not reach it"""
    from PIL import Image
    a = np.zeros((32, 40, 3), dtype=np.uint8)
    a[..., 0], a[..., 1], a[..., 2] = 200, 40, 90
    bio = io.BytesIO()
    Image.fromarray(a).save(bio, 'JPEG', quality=95, subsampling=0, keep_rgb=True)
    data = bio.getvalue()
    ref = np.asarray(Image.open(io.BytesIO(data)))
    assert abs(int(ref[5, 5, 0]) - 200) < 6 and abs(int(ref[5, 5, 1]) - 40) < 6      # Pillow reads it as RGB
    assert jpeg.open_file(data) is None and jpeg.read_coef(data) is None
    # an ordinary file: JFIF => YCbCr => ours
    ycc = open(os.path.join(HERE, 's444_q100.jpg'), 'rb').read()
    assert jpeg.open_file(ycc) is not None
    # the same file behind an Adobe APP14 marker instead of JFIF: transform 1 = YCbCr (ours), 0 = RGB and 2 = YCCK (host decoder)
    i = ycc.index(b'\xff\xe0')
    ln = int.from_bytes(ycc[i + 2:i + 4], 'big')
    for transform, ours in ((1, True), (0, False), (2, False)):
        app14 = b'\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00' + bytes([transform])
        f = ycc[:i] + app14 + ycc[i + 2 + ln:]
        assert (jpeg.open_file(f) is not None) == ours, transform
        if ours:
            c = jpeg.read_coef(f)
            np.testing.assert_array_equal(J.decode(c.info, c.coef, c.qt), np.asarray(Image.open(io.BytesIO(f))))
    # no marker at all: the component ids decide
    bare = ycc[:i] + ycc[i + 2 + ln:]
    assert jpeg.open_file(bare) is not None
    b = bytearray(bare)
    j = b.index(b'\xff\xc0')
    k = b.index(b'\xff\xda')
    for c, ch in enumerate(b'RGB'):
        b[j + 4 + 6 + 3 * c] = ch
        b[k + 5 + 2 * c] = ch
    assert jpeg.open_file(bytes(b)) is None


def test_pack_hands_damaged_entropy_data_to_pillow_as_the_reference_would_read_it(monkeypatch):
    """open_file() sees only the header; a file whose entropy data turns out bad inside pack() is decoded by Pillow (libjpeg
    reads truncated files with a warning -- so does the reference's imread) and travels as a raw image of the same batch
    (the host-entropy path; with the device decoder the flags come back from the GPU: tests/test_jpeg_gpu.py)"""
    from PIL import Image, ImageFile
    monkeypatch.setattr(jpeg, 'DEVICE_ENTROPY', False)
    good = open(os.path.join(HERE, 's420_q90.jpg'), 'rb').read()
    cut = good[:len(good) * 2 // 3]
    f_good, f_cut = jpeg.open_file(good), jpeg.open_file(cut)
    assert f_good is not None and f_cut is not None
    old = ImageFile.LOAD_TRUNCATED_IMAGES
    ImageFile.LOAD_TRUNCATED_IMAGES = True
    try:
        ref = np.asarray(Image.open(io.BytesIO(cut)))
        buf, desc, kind = jpeg.pack([f_good, f_cut, f_good])
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = old
    d = desc.numpy()
    assert kind == jpeg.KIND_JPEG and list(d[:, 24]) == [0, 1, 0] and int(d[1, 25]) == 3
    assert (int(d[1, 2]), int(d[1, 3])) == ref.shape[:2]
    o = int(d[1, 0])
    np.testing.assert_array_equal(buf.numpy()[o:o + ref.size].reshape(ref.shape), ref)
    # the neighbours are untouched coefficient entries
    c = jpeg.read_coef(good)
    for i in (0, 2):
        o = int(d[i, 0])
        np.testing.assert_array_equal(buf.numpy()[o:o + c.coef.size * 2].view(np.int16).reshape(-1, 64), c.coef)
    # without Pillow's tolerance the batch fails as the reference's loader would: loudly, not silently
    if not old:
        with pytest.raises((OSError, ValueError)):
            jpeg.pack([f_cut])


def test_ring_slot_is_returned_when_a_batch_fails(monkeypatch):
    from witw_amd import cvig_fov

    class Ring(object):
        def __init__(self):
            self.out = 0

        def acquire(self):
            self.out += 1
            return 0

        def release(self, slot):
            self.out -= 1

        def allocator(self, slot):
            return lambda n: None

    r = Ring()
    monkeypatch.setattr(cvig_fov, '_pack_side', lambda images, alloc=None: (_ for _ in ()).throw(ValueError('unreadable')))
    with pytest.raises(ValueError):
        cvig_fov.collate_packed([{'surface': 0, 'overhead': 0}], ring=r)
    assert r.out == 0


def test_decode_tables_vectorised_form_equals_the_per_image_loop():
    """jpeg.decode_tables (whole-array numpy: the per-image Python loop cost as much host time per batch as the bf16 encoders
    take on the GPU) against the loop it replaced, on every fixture layout mixed in one batch, in several orders."""
    files = [jpeg.open_file(os.path.join(HERE, n + '.jpg')) for n in fixture_names() if not n.startswith('prog')]
    g = np.random.default_rng(4)
    for trial in range(6):
        pick = [files[i] for i in g.permutation(len(files))[:g.integers(1, len(files) + 1)]]
        _buf, desc, _k = jpeg.pack(pick)
        d = desc.numpy()
        planes, images, blk, pbytes, obytes, qt_base = jpeg.decode_tables(d)
        # the loop form
        rp, ri = [], np.zeros((len(pick), 12), dtype=np.int64)
        b, pb, ob = 0, 0, 0
        qb = int(d[:, 1].min())
        for n in range(len(pick)):
            H, W, ncomp, hmax, vmax = (int(v) for v in d[n, 2:7])
            cb = int(d[n, 0]) // 128
            offs = []
            for c in range(ncomp):
                bw, bh = int(d[n, 10 + 4 * c]), int(d[n, 11 + 4 * c])
                rp.append((cb, (int(d[n, 1]) - qb) // 128 + c, pb, bw, bh, b))
                offs.append((pb, bw * 8))
                cb += bw * bh
                b += bw * bh
                pb += bw * bh * 64
            mode = {(1, 1): 0, (2, 1): 1, (2, 2): 2}[(hmax, vmax)]
            if mode and -(-W // hmax) <= 2:
                mode += 2
            ri[n, :4] = (H, W, ncomp, mode)
            ri[n, 4], ri[n, 5] = offs[0]
            if ncomp == 3:
                ri[n, 6], ri[n, 7], ri[n, 8] = offs[1][0], offs[2][0], offs[1][1]
                ri[n, 9], ri[n, 10] = -(-H // vmax), -(-W // hmax)
            ri[n, 11] = ob
            ob += (H * W * ncomp + 15) // 16 * 16
        np.testing.assert_array_equal(planes, np.asarray(rp, dtype=np.int64))
        np.testing.assert_array_equal(images, ri)
        assert (blk, pbytes, obytes, qt_base) == (b, pb, ob, qb)
    bad = d.copy()
    bad[0, 5], bad[0, 6] = 1, 2
    with pytest.raises(ValueError):
        jpeg.decode_tables(bad)


# ---- entropy decoding on the device (round 6): what the HOST contributes is the plan; its layout is checked here with a plain-Python
# rendering of the kernel's algorithm (csrc/jpeg.hip jpeg_huffman_kernel): tables from the DHT counts, a bit reader that unstuffs
# FF 00 and stops at a marker, one independent decode per restart interval
def _decode_from_plan(data, plan, n_blocks):
    ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
          57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    h = np.frombuffer(plan[:128].tobytes(), dtype=np.int32)
    assert h[0] == 0x3157504A
    n_int, restart, mcux, mcuy, ncomp = (int(v) for v in h[1:6])
    tabs = []
    for t in range(4):
        d = plan[128 + 32 * t:128 + 32 * t + 32] if t < 2 else plan[192 + 272 * (t - 2):192 + 272 * (t - 2) + 272]
        codes, code, k = {}, 0, 0
        for ln in range(1, 17):
            for _ in range(int(d[ln - 1])):
                codes[(ln, code)] = int(d[16 + k])
                code += 1
                k += 1
            code <<= 1
        tabs.append(codes)
    ioff = np.frombuffer(plan[736:736 + 4 * n_int].tobytes(), dtype=np.uint32)
    coef = np.zeros((n_blocks, 64), dtype=np.int16)
    for iv in range(n_int):
        pos, end = int(ioff[iv]), (int(ioff[iv + 1]) if iv + 1 < n_int else int(h[27]))
        bits = []
        while pos < end:
            b = int(data[pos])
            if b == 0xFF:
                if pos + 1 < end and data[pos + 1] == 0:
                    pos += 2
                else:
                    break
            else:
                pos += 1
            bits.extend((b >> (7 - j)) & 1 for j in range(8))
        bits.extend([0] * 64)
        bp = 0

        def sym(tab):
            nonlocal bp
            code = 0
            for ln in range(1, 17):
                code = (code << 1) | bits[bp + ln - 1]
                if (ln, code) in tab:
                    bp += ln
                    return tab[(ln, code)]
            raise AssertionError('invalid code')

        def recv(s):
            nonlocal bp
            v = 0
            for _ in range(s):
                v = (v << 1) | bits[bp]
                bp += 1
            return v if v >= (1 << (s - 1)) else v - (1 << s) + 1
        pred = [0, 0, 0]
        for m in range(iv * restart, min((iv + 1) * restart, mcux * mcuy)):
            my, mx = divmod(m, mcux)
            for k in range(ncomp):
                c = int(h[28 + k])
                ch, cv, cbw, _cbh, coff, dcs, acs = (int(v) for v in h[6 + 7 * c:13 + 7 * c])
                for v in range(cv):
                    for hh in range(ch):
                        blk = coef[coff + (my * cv + v) * cbw + mx * ch + hh]
                        s = sym(tabs[dcs])
                        if s:
                            pred[c] += recv(s)
                        blk[0] = pred[c]
                        kk = 1
                        while kk < 64:
                            rs = sym(tabs[2 + acs])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r != 15:
                                    break
                                kk += 16
                                continue
                            kk += r
                            blk[ZZ[kk]] = recv(s)
                            kk += 1
    return coef


def test_entropy_plan_describes_the_restart_intervals():
    import io
    from PIL import Image
    g = np.random.Generator(np.random.Philox(key=[12, 1]))
    files = [open(os.path.join(HERE, n), 'rb').read() for n in ('s420_rst.jpg', 's422_rst.jpg')]
    for (hh, ww, sub, kw) in ((40, 56, 2, {'restart_marker_rows': 1}), (33, 17, 0, {'restart_marker_blocks': 2}),
                              (64, 48, 1, {'restart_marker_blocks': 3, 'optimize': True}), (24, 24, 2, {'restart_marker_blocks': 100})):
        a = g.integers(0, 256, size=(hh, ww, 3), dtype=np.uint8)
        bio = io.BytesIO()
        Image.fromarray(a).save(bio, 'JPEG', quality=85, subsampling=sub, **kw)
        files.append(bio.getvalue())
    for raw in files:
        f = jpeg.open_file(raw)
        pl = f.entropy_plan()
        assert pl is not None
        plan, qt = pl
        ref = jpeg.read_coef(raw)
        np.testing.assert_array_equal(qt, ref.qt)
        hdr = np.frombuffer(plan[:128].tobytes(), dtype=np.int32)
        n_int = int(hdr[1])
        assert n_int == -(-int(hdr[3]) * int(hdr[4]) // int(hdr[2])) and plan.size == 736 + 4 * n_int
        off = np.frombuffer(plan[736:].tobytes(), dtype=np.uint32)
        data = np.frombuffer(raw, dtype=np.uint8)
        for i in range(1, n_int):        # every interval but the first starts right behind its RSTn marker, numbered in sequence
            assert data[off[i] - 2] == 0xFF and data[off[i] - 1] == 0xD0 + ((i - 1) & 7)
        np.testing.assert_array_equal(_decode_from_plan(data, plan, int(f.info[5])), ref.coef)
    # no restart markers -> a plan of ONE interval (the self-synchronising kernel's input); markers out of sequence -> no plan
    raw = open(os.path.join(HERE, 's420_q90.jpg'), 'rb').read()
    f = jpeg.open_file(raw)
    plan, _qt = f.entropy_plan()
    hdr = np.frombuffer(plan[:128].tobytes(), dtype=np.int32)
    assert hdr[1] == 1 and hdr[2] == hdr[3] * hdr[4] and plan.size == 740
    np.testing.assert_array_equal(_decode_from_plan(np.frombuffer(raw, dtype=np.uint8), plan, int(f.info[5])), jpeg.read_coef(raw).coef)
    bad = bytearray(files[0])
    p = bad.find(b'\xff\xd1')
    bad[p + 1] = 0xD5
    assert jpeg.open_file(bytes(bad)).entropy_plan() is None


def test_pack_ships_file_bytes_for_device_entropy_decoding(monkeypatch):
    raw = open(os.path.join(HERE, 's420_rst.jpg'), 'rb').read()
    plain = open(os.path.join(HERE, 's420_q90.jpg'), 'rb').read()
    items = [jpeg.open_file(raw), jpeg.open_file(plain), jpeg.open_file(raw)]
    assert jpeg.DEVICE_ENTROPY == 'all'                         # the default: every baseline Huffman file goes to the device decoder
    monkeypatch.setattr(jpeg, 'DEVICE_ENTROPY', 'restart')      # only files with restart markers
    _b, desc_r, _k = jpeg.pack(items)
    assert list(desc_r[:, 26].numpy()) == [1, 0, 1]
    monkeypatch.setattr(jpeg, 'DEVICE_ENTROPY', 'all')
    buf, desc, kind = jpeg.pack(items)
    d = desc.numpy()
    assert kind == jpeg.KIND_JPEG and list(d[:, 26]) == [1, 1, 1] and list(d[:, 28]) == [len(raw), len(plain), len(raw)]
    assert list(d[:, 29]) == [int(np.frombuffer(buf.numpy()[d[0, 27] + 4:d[0, 27] + 8].tobytes(), dtype=np.int32)[0]), 1, d[0, 29]] and d[0, 29] > 1
    b = buf.numpy()
    for i in (0, 2):
        assert d[i, 0] % 128 == 0 and d[i, 27] % 16 == 0 and bytes(b[d[i, 0]:d[i, 0] + len(raw)]) == raw
        assert b[d[i, 27]:d[i, 27] + 4].view(np.int32)[0] == 0x3157504A
    assert bytes(b[d[1, 0]:d[1, 0] + len(plain)]) == plain
    # with the device decoder off every file is Huffman-decoded here, laid out as before
    monkeypatch.setattr(jpeg, 'DEVICE_ENTROPY', False)
    b2, desc2, _k = jpeg.pack([jpeg.open_file(raw), jpeg.open_file(plain)])
    d2 = desc2.numpy()
    assert list(d2[:, 26]) == [0, 0]
    np.testing.assert_array_equal(b2.numpy()[d2[1, 0]:d2[1, 0] + int(d2[1, 7]) * 128].view(np.int16).reshape(-1, 64), jpeg.read_coef(plain).coef)
