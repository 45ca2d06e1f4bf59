"""CPU restatement (test infrastructure: only tests/, bench.py's cpu legs and __graft_entry__.smoke() may import this) of the
BACK END of the JPEG decoder the reference reads its images with -- skimage.io.imread -> PIL -> libjpeg-turbo
(model/cvig_fov.py:88-89; requirements pin scikit_image 0.18.1, which defers to PIL) -- i.e. of libjpeg's published integer
algorithms (third-party, not under /root/reference; the build container has libjpeg-turbo API 6.2 inside Pillow 12.2):
  * dequantisation + the 'islow' inverse DCT (jidctint.c: 13-bit constants, two passes, range limit),
  * 'fancy' chroma upsampling h2v1 / h2v2 (jdsample.c: triangle filter 3/4 + 1/4 with alternating rounding, edges replicated),
  * YCbCr -> RGB (jdcolor.c: 16-bit fixed point tables).
Parity pinned: tests/test_jpeg.py holds this file to Pillow's decode of the committed fixtures (tests/golden/jpeg/) byte for
byte, and to Pillow itself on freshly written files. Input: the quantised coefficient blocks of witw_amd/csrc_host/jpeg_coef.cpp."""
import numpy as np

C = dict(F0_298=2446, F0_390=3196, F0_541=4433, F0_765=6270, F0_899=7373, F1_175=9633, F1_501=12299, F1_847=15137, F1_961=16069,
         F2_053=16819, F2_562=20995, F3_072=25172)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _pass(v, shift):
    """one 8-point pass of jpeg_idct_islow on the LAST axis of v (int64 [..., 8]) -> [..., 8]"""
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * C['F0_541']
    tmp2 = z1 + z3 * (-C['F1_847'])
    tmp3 = z1 + z2 * C['F0_765']
    z2, z3 = v[..., 0], v[..., 4]
    tmp0 = (z2 + z3) << 13
    tmp1 = (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * C['F1_175']
    tmp0, tmp1, tmp2, tmp3 = tmp0 * C['F0_298'], tmp1 * C['F2_053'], tmp2 * C['F3_072'], tmp3 * C['F1_501']
    z1, z2, z3, z4 = z1 * -C['F0_899'], z2 * -C['F2_562'], z3 * -C['F1_961'] + z5, z4 * -C['F0_390'] + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    out = [tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3]
    return np.stack([_descale(o, shift) for o in out], axis=-1)


def range_limit(x):
    """libjpeg's sample_range_limit table behind the IDCT, index masked to 10 bits (jdmaster.c prepare_range_limit_table)"""
    i = (x & 1023).astype(np.int64)
    return np.where(i < 128, i + 128, np.where(i < 512, 255, np.where(i < 896, 0, i - 896))).astype(np.uint8)


def idct_islow(coef, qt):
    """coef int16 [nblk, 64] natural order, qt uint16 [64] -> uint8 [nblk, 8, 8]"""
    d = coef.astype(np.int64).reshape(-1, 8, 8) * qt.astype(np.int64).reshape(1, 8, 8)
    ws = _pass(d.transpose(0, 2, 1), 13 - 2).transpose(0, 2, 1)         # pass 1: columns
    return range_limit(_pass(ws, 13 + 2 + 3))                           # pass 2: rows


def plane(coef, qt, bh, bw):
    return idct_islow(coef, qt).reshape(bh, bw, 8, 8).transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def upsample_h2v1(p):
    """p uint8 [h, w] (real samples only) -> [h, 2w] (jdsample.c h2v1_fancy_upsample)"""
    a = p.astype(np.int64)
    h, w = a.shape
    out = np.empty((h, 2 * w), dtype=np.int64)
    if w == 1:
        out[:, 0] = out[:, 1] = a[:, 0]
        return out.astype(np.uint8)
    left = np.concatenate((a[:, :1], a[:, :-1]), 1)
    right = np.concatenate((a[:, 1:], a[:, -1:]), 1)
    out[:, 0::2] = (3 * a + left + 1) >> 2
    out[:, 1::2] = (3 * a + right + 2) >> 2
    out[:, 0] = a[:, 0]
    out[:, -1] = a[:, -1]
    return out.astype(np.uint8)


def upsample_h2v2(p):
    """p uint8 [h, w] (real samples only) -> [2h, 2w] (jdsample.c h2v2_fancy_upsample; rows above the first / below the last
    real row are that row again: jdmainct.c context rows)"""
    a = p.astype(np.int64)
    h, w = a.shape
    up = np.concatenate((a[:1], a[:-1]), 0)
    dn = np.concatenate((a[1:], a[-1:]), 0)
    out = np.empty((2 * h, 2 * w), dtype=np.int64)
    for half, other in ((0, up), (1, dn)):
        cs = 3 * a + other                                   # column sums of the output row pair member
        last = np.concatenate((cs[:, :1], cs[:, :-1]), 1)
        nxt = np.concatenate((cs[:, 1:], cs[:, -1:]), 1)
        ev = (3 * cs + last + 8) >> 4
        od = (3 * cs + nxt + 7) >> 4
        ev[:, 0] = (4 * cs[:, 0] + 8) >> 4
        od[:, -1] = (4 * cs[:, -1] + 7) >> 4
        out[half::2, 0::2] = ev
        out[half::2, 1::2] = od
    return out.astype(np.uint8)


def ycc_to_rgb(y, cb, cr):
    y, cb, cr = y.astype(np.int64), cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack((r, g, b), -1), 0, 255).astype(np.uint8)


def decode(info, coef, qt):
    """info: the 22 ints of witw_jpeg_info; coef int16 [total blocks, 64]; qt uint16 [ncomp, 64] -> uint8 [H, W, 3] or [H, W]"""
    H, W, ncomp, hmax, vmax = (int(v) for v in info[:5])
    planes, off = [], 0
    for c in range(ncomp):
        h, v, bw, bh = (int(x) for x in info[6 + 4 * c:10 + 4 * c])
        planes.append((plane(coef[off:off + bw * bh], qt[c], bh, bw), h, v))
        off += bw * bh
    if ncomp == 1:
        return planes[0][0][:H, :W]
    out = [planes[0][0][:H, :W]]
    for p, h, v in planes[1:]:
        ch, cw = -(-H * v // vmax), -(-W * h // hmax)        # downsampled_height / width (ceil)
        p = p[:ch, :cw]
        if hmax == 2 and cw <= 2:          # jdsample.c jinit_upsampler: the fancy filters only for downsampled_width > 2,
            p = p.repeat(vmax, 0).repeat(2, 1)      # narrower planes are replicated (h2v1_upsample / h2v2_upsample)
        elif hmax == 2 and vmax == 2:
            p = upsample_h2v2(p)
        elif hmax == 2 and vmax == 1:
            p = upsample_h2v1(p)
        elif hmax != 1 or vmax != 1:
            raise ValueError('sampling %dx%d is left to the host decoder' % (hmax, vmax))
        out.append(p[:H, :W])
    return ycc_to_rgb(*out)
