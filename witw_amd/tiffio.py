"""Minimal baseline-TIFF reader/writer for the multi-band scene files of cvig_semantic (the reference reads them
with tifffile.imread, model/cvig_semantic.py:93-94,108; tifffile is not in this image). Uses tifffile when it is
importable; otherwise handles what those files are: one image, uncompressed, strips, 8/16/32-bit unsigned or
float32 samples, any number of bands, chunky or planar. Returns HWC (H,W for one band) like tifffile.
Anything else (LZW/deflate, tiles, BigTIFF) raises — never a silent wrong read."""
import struct

import numpy as np

_TYPES = {1: ('B', 1), 2: ('c', 1), 3: ('H', 2), 4: ('I', 4), 5: ('II', 8), 16: ('Q', 8)}


def _entries(buf, bo):
    (magic,) = struct.unpack(bo + 'H', buf[2:4])
    if magic != 42:
        raise ValueError('not a classic TIFF (magic %d; BigTIFF is not supported)' % magic)
    (off,) = struct.unpack(bo + 'I', buf[4:8])
    (n,) = struct.unpack(bo + 'H', buf[off:off + 2])
    tags = {}
    for i in range(n):
        e = buf[off + 2 + 12 * i: off + 14 + 12 * i]
        tag, typ, cnt = struct.unpack(bo + 'HHI', e[:8])
        if typ not in _TYPES or typ == 5:
            continue
        fmt, size = _TYPES[typ]
        raw = e[8:12] if cnt * size <= 4 else buf[struct.unpack(bo + 'I', e[8:12])[0]:][:cnt * size]
        tags[tag] = list(struct.unpack(bo + fmt * cnt, raw[:cnt * size])) if typ != 2 else raw
    return tags


def imread(path):
    try:
        import tifffile
        return tifffile.imread(path)
    except ImportError:
        pass
    with open(path, 'rb') as f:
        buf = f.read()
    bo = {b'II': '<', b'MM': '>'}.get(buf[:2])
    if bo is None:
        raise ValueError('%s: not a TIFF file' % path)
    t = _entries(buf, bo)
    if t.get(259, [1])[0] != 1:
        raise ValueError('%s: compressed TIFF (compression %d) needs tifffile' % (path, t[259][0]))
    if 322 in t or 324 in t:
        raise ValueError('%s: tiled TIFF needs tifffile' % path)
    w, h = t[256][0], t[257][0]
    spp = t.get(277, [1])[0]
    bits = t.get(258, [1])
    fmt = t.get(339, [1])[0]
    if len(set(bits)) != 1:
        raise ValueError('%s: mixed sample widths %s' % (path, bits))
    kind = {(8, 1): 'u1', (16, 1): 'u2', (32, 1): 'u4', (32, 3): 'f4', (8, 2): 'i1', (16, 2): 'i2', (32, 2): 'i4',
            (64, 3): 'f8'}.get((bits[0], fmt))
    if kind is None:
        raise ValueError('%s: unsupported sample type (%d bits, format %d)' % (path, bits[0], fmt))
    dt = np.dtype(bo + kind)
    counts = t.get(279)
    if counts is None:
        raise ValueError('%s: no StripByteCounts' % path)
    data = b''.join(buf[o:o + c] for o, c in zip(t[273], counts))
    a = np.frombuffer(data, dtype=dt, count=w * h * spp)
    if spp == 1:
        return a.reshape(h, w).astype(dt.newbyteorder('='))
    if t.get(284, [1])[0] == 2:     # planar: [band][h][w]
        return a.reshape(spp, h, w).transpose(1, 2, 0).astype(dt.newbyteorder('='))
    return a.reshape(h, w, spp).astype(dt.newbyteorder('='))


def imwrite(path, arr):
    """One uncompressed little-endian strip, chunky; arr HW or HWC of uint8/uint16/uint32/float32."""
    a = np.ascontiguousarray(arr)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, spp = a.shape
    fmt = {'u': 1, 'i': 2, 'f': 3}[a.dtype.kind]
    bits = a.dtype.itemsize * 8
    a = a.astype(a.dtype.newbyteorder('<'))
    data = a.tobytes()
    entries = []          # (tag, type, values)
    extra = b''
    n_tags = 11
    ifd_off = 8
    extra_off = ifd_off + 2 + 12 * n_tags + 4

    def put(tag, typ, vals):
        nonlocal extra
        f, size = _TYPES[typ]
        raw = struct.pack('<' + f * len(vals), *vals)
        if len(raw) <= 4:
            entries.append(struct.pack('<HHI', tag, typ, len(vals)) + raw.ljust(4, b'\0'))
        else:
            entries.append(struct.pack('<HHII', tag, typ, len(vals), extra_off + len(extra)))
            extra += raw + (b'\0' if len(raw) % 2 else b'')

    bits_l, fmt_l = [bits] * spp, [fmt] * spp
    # sizes of out-of-line values are known before the strip offset is: BitsPerSample / SampleFormat only
    pre = sum((2 * spp + (2 * spp) % 2) for _ in range(2) if 2 * spp > 4)
    strip_off = extra_off + pre
    put(256, 4, [w])
    put(257, 4, [h])
    put(258, 3, bits_l)
    put(259, 3, [1])
    put(262, 3, [1 if spp < 3 else 2])
    put(273, 4, [strip_off])
    put(277, 3, [spp])
    put(278, 4, [h])
    put(279, 4, [len(data)])
    put(284, 3, [1])
    put(339, 3, fmt_l)
    assert len(entries) == n_tags and len(extra) == pre
    with open(path, 'wb') as f:
        f.write(b'II' + struct.pack('<HI', 42, ifd_off))
        f.write(struct.pack('<H', n_tags) + b''.join(entries) + struct.pack('<I', 0))
        f.write(extra)
        f.write(data)
