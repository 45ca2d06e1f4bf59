#!/usr/bin/env python
"""Randomised parity sweep of the HIP kernels against the CPU oracle (test infrastructure, like tests/): random layer
shapes through the C ABI, compared with torch CPU ops in the reference's op order.

    python tools/fuzz_parity.py [seconds] [seed]

Covers: fp32 conv forward (stride, circular/zero padding, ReLU, fused pool, GEO / NW variants by shape), its dgrad
form, fp32 wgrad (+ bias), the 4-tap forms, bf16 conv forward / wgrad, the fused match (orientation exact, distance
1e-5) with ragged batch sizes and widths (direct and spectral forms), the fp16x3 conv forward / dgrad form (gate, Dropout2d scale, zero-interleaved
rows) / wgrad against fp64, the 4-tap conv's space-to-depth epilogue (bitwise) and its split-K mosaic form, the bf16 16x16x32 kernel against the 32x32x16 kernel. FUZZ_KINDS=10,11 restricts the sweep to the listed kinds. Prints one line per failure and a summary; exit code 1 on any failure.
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cvig_fov_oracle as O  # noqa: E402
from witw_amd import ops  # noqa: E402


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


N_KINDS = 20


def run(budget=120.0, seed=0, rounds=None, kinds=None):
    """Draw cases until `budget` seconds have passed (kinds at random), or — rounds given — exactly `rounds` cases of every
    kind in `kinds` (default all) in a fixed order: the deterministic form tests/test_fuzz_gpu.py runs. Returns
    (cases run per kind, list of failures)."""
    rng = np.random.default_rng(seed)
    if kinds is None and os.environ.get('FUZZ_KINDS'):
        kinds = [int(k) for k in os.environ['FUZZ_KINDS'].split(',')]
    kinds = list(range(N_KINDS)) if kinds is None else list(kinds)
    ran = {k: 0 for k in kinds}
    dev = torch.device('cuda:0')
    torch.manual_seed(seed)
    t0 = time.time()
    n, fails = 0, []

    def check(name, cfg, got, ref, tol):
        err = float((got - ref).abs().max())
        scale = max(1.0, float(ref.abs().max()))
        if not np.isfinite(err) or err > tol * scale:
            fails.append((name, cfg, err, scale))
            print('FAIL %s %s: max err %.3e (scale %.3e)' % (name, cfg, err, scale), flush=True)

    def draw_kinds():
        if rounds is not None:
            for _ in range(rounds):
                for k in kinds:
                    yield k
        else:
            while time.time() - t0 < budget:
                yield int(rng.choice(kinds))

    for kind in draw_kinds():
        n += 1
        ran[kind] += 1
        B = int(rng.integers(1, 5))
        H = int(rng.integers(1, 40))
        W = int(rng.integers(1, 140))
        cin = int(rng.choice([8, 16, 24, 64, 72, 128]))
        cout = int(rng.choice([8, 16, 64, 72, 128, 136, 256]))
        sh = int(rng.choice([1, 2]))
        circ = bool(rng.integers(0, 2))
        relu = bool(rng.integers(0, 2))
        x = torch.randn(B, cin, H, W)
        w = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn(cout) * 0.1
        cfg = (B, H, W, cin, cout, sh, circ, relu)
        try:
            if kind == 0:       # fp32 forward (+ pool)
                pool = sh == 1 and H >= 2 and W >= 2 and bool(rng.integers(0, 2))
                ref = O.conv3x3(x, w, b, sh, circ)
                if relu:
                    ref = torch.relu(ref)
                if pool:
                    ref = torch.nn.functional.max_pool2d(ref, 2, 2)
                y = ops.conv3x3_fwd(nhwc(x).to(dev), ops.PackedConv(w.to(dev), b.to(dev)), stride_h=sh, circular=circ, relu=relu,
                                    pool=pool)
                check('conv_fwd' + ('_pool' if pool else ''), cfg, y.cpu().permute(0, 3, 1, 2), ref, 3e-5)
            elif kind == 1:     # fp32 wgrad + dgrad
                xr = x.clone().requires_grad_(True)
                wr = w.clone().requires_grad_(True)
                br = b.clone().requires_grad_(True)
                yref = O.conv3x3(xr, wr, br, sh, circ)
                gy = torch.randn_like(yref)
                yref.backward(gy)
                dw, db = ops.conv3x3_wgrad(nhwc(x).to(dev), nhwc(gy).to(dev), cin, stride_h=sh, circular=circ)
                check('wgrad', cfg, dw.cpu(), wr.grad, 3e-5)
                check('bgrad', cfg, db.cpu(), br.grad, 3e-5)
                if cout % 8 == 0:
                    dx = ops.conv3x3_fwd(nhwc(gy).to(dev), ops.PackedConv(w.to(dev), None, transpose_flip=True), relu=False,
                                         circular=circ, dilate_h=(sh == 2), out_h=H if sh == 2 else None)
                    check('dgrad', cfg, dx.cpu().permute(0, 3, 1, 2), xr.grad, 3e-5)
            elif kind == 2:     # 4-tap forms vs the zero-filled 3x3 form
                w4 = w.clone()
                w4[:, :, 0, :] = 0
                w4[:, :, :, 0] = 0
                xd = nhwc(x).to(dev)
                y9 = ops.conv3x3_fwd(xd, ops.PackedConv(w4.to(dev), b.to(dev)), relu=relu)
                y4 = ops.conv3x3_fwd(xd, ops.PackedConv(w4.to(dev), b.to(dev), taps4=True), relu=relu)
                if not torch.equal(y9, y4):
                    check('taps4_fwd', cfg, y4.cpu(), y9.cpu(), 0.0)
                gy = torch.randn(B, H, W, cout).to(dev)
                a9, _ = ops.conv3x3_wgrad(xd, gy, cin)
                a4, _ = ops.conv3x3_wgrad(xd, gy, cin, taps4=True)
                if not torch.equal(a9[:, :, 1:, 1:], a4[:, :, 1:, 1:]):
                    check('taps4_wgrad', cfg, a4[:, :, 1:, 1:].cpu(), a9[:, :, 1:, 1:].cpu(), 0.0)
            elif kind == 14 and H >= 2 and W >= 2 and cout % 4 == 0:    # 4-tap conv with the space-to-depth epilogue vs conv + separate pass
                w4 = w.clone()
                w4[:, :, 0, :] = 0
                w4[:, :, :, 0] = 0
                xd = nhwc(x).to(dev)
                pk = ops.PackedConv(w4.to(dev), b.to(dev), taps4=True)
                sc, shf = (1 + 0.1 * torch.randn(cout)).to(dev), (0.1 * torch.randn(cout)).to(dev)
                vh = int(rng.integers(1, H + 1))
                vw = int(rng.integers(1, W + 1))
                vh -= (vh == H and H % 2 == 1)      # an odd valid size needs one more computed row / column
                vw -= (vw == W and W % 2 == 1)
                if vh >= 1 and vw >= 1:
                    y = ops.conv3x3_fwd(xd, pk, relu=False, lrelu_slope=0.2, post_scale=sc, post_shift=shf)
                    want = ops.space_to_depth2(y, valid_hw=(vh, vw), cpad=4 * cout)
                    got = ops.conv_taps4_s2d(xd, pk, (vh, vw), lrelu_slope=0.2, post_scale=sc, post_shift=shf)
                    if not torch.equal(got, want):
                        check('taps4_s2d', cfg + (vh, vw), got.cpu(), want.cpu(), 0.0)
            elif kind == 15 and cout % 4 == 0:    # split-K over a g x g mosaic vs the plain 4-tap conv per image
                g = int(rng.choice([1, 2, 4]))
                h = int(rng.choice([2, 3, 4, 8, 16])) if g > 1 else int(rng.integers(2, 20))
                cin2 = int(rng.choice([64, 128, 256, 512]))
                n_img = int(rng.integers(1, 12))
                x2 = torch.randn(n_img, h, h, cin2)
                w4 = torch.randn(cout, cin2, 3, 3) * (2.0 / (4 * cin2)) ** 0.5
                w4[:, :, 0, :] = 0
                w4[:, :, :, 0] = 0
                pk = ops.PackedConv(w4.to(dev), b.to(dev), taps4=True)
                y = ops.conv3x3_fwd(x2.to(dev), pk, relu=False, lrelu_slope=0.2)[:, :h - 1, :h - 1].contiguous()
                bm = (n_img + g * g - 1) // (g * g)
                xm = torch.zeros(bm, g * h, g * h, cin2)
                for i in range(n_img):
                    m_, cell = divmod(i, g * g)
                    cy, cx = divmod(cell, g)
                    xm[m_, cy * h:(cy + 1) * h, cx * h:(cx + 1) * h] = x2[i]
                nkc = cin2 // 8
                ks = int(rng.choice([0, 2, 3, 5, 8]))          # 0: the library's own choice
                if ks == 0 or (ks - 1) * ((nkc + ks - 1) // ks) >= nkc:
                    ks = None
                got = ops.conv_taps4_splitk(xm.to(dev), pk, n_img, g, (h - 1, h - 1), lrelu_slope=0.2, ksplit=ks)
                check('taps4_splitk', (n_img, g, h, cin2, cout, ks), got.cpu(), y.cpu(), 2e-5)
            elif kind == 16:    # bf16 forward: the 16x16x32 kernel against the 32x32x16 kernel on layers that take the 8-wave tile
                cin2 = int(rng.choice([32, 64, 96, 128, 160, 256]))
                cout2 = int(rng.choice([128, 144, 256, 384]))
                H2 = 8 * int(rng.integers(1, 5))
                W2 = int(rng.integers(1, 200))
                nt = (cout2 + 127) // 128
                B2 = (512 + nt * ((W2 + 63) // 64) * (H2 // 8) - 1) // (nt * ((W2 + 63) // 64) * (H2 // 8)) + int(rng.integers(0, 3))
                if B2 * H2 * W2 * max(cin2, cout2) > 6e8:
                    continue
                pool = bool(rng.integers(0, 2)) and H2 >= 2 and W2 >= 2
                x2 = torch.randn(B2, H2, W2, cin2, device=dev).bfloat16()
                pk = ops.PackedConvBf16((torch.randn(cout2, cin2, 3, 3) * (2.0 / (9 * cin2)) ** 0.5).to(dev), (torch.randn(cout2) * 0.1).to(dev))
                prev = ops.bf16_mfma16(False)
                try:
                    y32 = ops.conv3x3_bf16_fwd(x2, pk, circular=circ, relu=relu, pool=pool).float()
                    ops.bf16_mfma16(True)
                    y16 = ops.conv3x3_bf16_fwd(x2, pk, circular=circ, relu=relu, pool=pool).float()
                finally:
                    ops.bf16_mfma16(prev)
                bad = (y32 - y16).abs() > 2.0 ** -7 * torch.maximum(y32.abs(), y16.abs()) + 1e-5 * float(y32.abs().max())
                if bool(bad.any()):
                    check('bf16_mfma16', (B2, H2, W2, cin2, cout2, circ, relu, pool), y16.cpu(), y32.cpu(), 0.0)
            elif kind == 3:     # bf16 forward vs emulation
                cin16 = (cin + 15) // 16 * 16
                x = torch.randn(B, cin16, H, W).bfloat16().float()
                w = (torch.randn(cout, cin16, 3, 3) * (2.0 / (9 * cin16)) ** 0.5)
                ref = O.conv3x3(x, w.bfloat16().float(), b, sh, circ)
                if relu:
                    ref = torch.relu(ref)
                last = cout % 16 != 0
                y = ops.conv3x3_bf16_fwd(nhwc(x).to(dev).bfloat16(), ops.PackedConvBf16(w.to(dev), b.to(dev)), stride_h=sh,
                                         circular=circ, relu=relu, out_nchw_f32=last)
                if last:
                    check('bf16_fwd_f32out', cfg, y.cpu(), ref, 3e-5)
                else:
                    got = y.float().cpu().permute(0, 3, 1, 2)
                    r16 = ref.bfloat16().float()
                    bad = (got - r16).abs() > (2 ** -7) * r16.abs() + 1e-6
                    if bool(bad.any()):
                        check('bf16_fwd', cfg, got, r16, 2 ** -7)
            elif kind == 4:     # bf16 wgrad on bf16-exact operands
                cin8, cout8 = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
                x = torch.randn(B, cin8, H, W).bfloat16().float()
                wr = torch.zeros(cout8, cin8, 3, 3, requires_grad=True)
                br = torch.zeros(cout8, requires_grad=True)
                yref = O.conv3x3(x, wr, br, sh, circ)
                gy = torch.randn_like(yref).bfloat16().float()
                yref.backward(gy)
                dw, db = ops.conv3x3_wgrad_bf16(nhwc(x).to(dev).bfloat16(), nhwc(gy).to(dev).bfloat16(), cin8, stride_h=sh,
                                                circular=circ)
                check('wgrad_bf16', (B, H, W, cin8, cout8, sh, circ), dw.cpu(), wr.grad, 3e-5)
                check('bgrad_bf16', (B, H, W, cin8, cout8, sh, circ), db.cpu(), br.grad, 3e-5)
            elif kind == 6:     # first layer (C <= 4 -> 64), fp32 and bf16 forms
                c = int(rng.integers(1, 5))
                x = torch.randn(B, c, H, W)
                w = torch.randn(64, c, 3, 3) * (2.0 / (9 * c)) ** 0.5
                b = torch.randn(64) * 0.1
                ref = torch.relu(O.conv3x3(x, w, b, 1, circ))
                y = ops.conv3x3_first_fwd(x.to(dev), ops.PackedFirstConv(w.to(dev), b.to(dev)), circular=circ, relu=True)
                check('first_fwd', (B, c, H, W, circ), y.cpu().permute(0, 3, 1, 2), ref, 3e-5)
                ref16 = torch.relu(O.conv3x3(x.bfloat16().float(), w.bfloat16().float(), b, 1, circ)).bfloat16().float()
                y16 = ops.conv3x3_first_fwd(x.to(dev), ops.PackedFirstConv(w.to(dev), b.to(dev), bf16=True), circular=circ, relu=True)
                got = y16.float().cpu().permute(0, 3, 1, 2)
                if bool(((got - ref16).abs() > (2 ** -7) * ref16.abs() + 1e-6).any()):
                    check('first_fwd_bf16', (B, c, H, W, circ), got, ref16, 2 ** -7)
            elif kind == 7:     # resize + normalise, polar transform
                c = int(rng.integers(1, 6))
                hi, wi = int(rng.integers(2, 90)), int(rng.integers(2, 90))
                ho, wo = int(rng.integers(1, 70)), int(rng.integers(1, 70))
                img = torch.rand(B, c, hi, wi) * 255
                ref = torch.stack([O.resize_bilinear(img[i], (ho, wo)) for i in range(B)])
                y = ops.resize_bilinear(img.to(dev), (ho, wo))
                check('resize', (B, c, hi, wi, ho, wo), y.cpu(), ref, 2e-5)
                sq = torch.rand(B, c, 256, 256)
                refp = torch.stack([O.polar_transform(sq[i]) for i in range(min(B, 2))])
                yp = ops.polar_transform(sq[:min(B, 2)].to(dev))
                if not torch.equal(yp.cpu(), refp):
                    check('polar', (B, c), yp.cpu(), refp, 0.0)
            elif kind == 8:     # loss and rank counts on a random distance matrix
                nb = int(rng.integers(2, 200))
                d = torch.rand(nb, nb) * 4
                ref = O.triplet_loss(d)
                got = ops.triplet_loss_fwd(d.to(dev))
                got = got[0] if isinstance(got, (tuple, list)) else got
                check('triplet_loss', (nb,), got.cpu().reshape(()), ref.reshape(()), 2e-6)
                rk = ops.rank_count(d.to(dev))
                ref_rk = (d <= torch.diagonal(d)[None, :]).sum(0).to(torch.int32)
                if not torch.equal(rk.cpu(), ref_rk):
                    fails.append(('rank_count', (nb,), 0, 0))
                    print('FAIL rank_count', nb, flush=True)
            elif kind == 9:     # fp16x3 forward vs fp64
                cin8 = (cin + 7) // 8 * 8
                pool = sh == 1 and H >= 2 and W >= 2 and bool(rng.integers(0, 2))
                x = torch.randn(B, cin8, H, W)
                w = torch.randn(cout, cin8, 3, 3) * (2.0 / (9 * cin8)) ** 0.5
                ref = O.conv3x3(x.double(), w.double(), b.double(), sh, circ)
                if relu:
                    ref = torch.relu(ref)
                if pool:
                    ref = torch.nn.functional.max_pool2d(ref, 2, 2)
                last = cout % 8 != 0
                y = ops.conv3x3_f16x3_fwd(ops.nchw_to_split_f16(x.to(dev), cin8), ops.PackedConvF16x3(w.to(dev), b.to(dev)),
                                          stride_h=sh, circular=circ, relu=relu, pool=pool and not last, out_nchw_f32=last)
                if last and pool:
                    ref = O.conv3x3(x.double(), w.double(), b.double(), sh, circ)
                    ref = torch.relu(ref) if relu else ref
                got = y.cpu() if last else ops.split_f16_to_f32(y).cpu().permute(0, 3, 1, 2)
                check('f16x3_fwd', (B, H, W, cin8, cout, sh, circ, relu, pool), got, ref.float(), 5e-6)
            elif kind == 10:    # fp16x3 weight gradient vs fp64 autograd
                cin8, cout8 = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
                if 256 % (cout8 // 8):
                    cout8 = 64
                x = torch.randn(B, cin8, H, W)
                wr = torch.zeros(cout8, cin8, 3, 3, dtype=torch.float64, requires_grad=True)
                br = torch.zeros(cout8, dtype=torch.float64, requires_grad=True)
                yref = O.conv3x3(x.double(), wr, br, sh, circ)
                gy = torch.randn(tuple(yref.shape))
                yref.backward(gy.double())
                dw, db = ops.conv3x3_wgrad_f16x3(ops.nchw_to_split_f16(x.to(dev), cin8), ops.nchw_to_split_f16(gy.to(dev), cout8), cin8,
                                                 stride_h=sh, circular=circ)
                check('wgrad_f16x3', (B, H, W, cin8, cout8, sh, circ), dw.cpu(), wr.grad.float(), 4e-6)
                check('bgrad_f16x3', (B, H, W, cin8, cout8, sh, circ), db.cpu(), br.grad.float(), 4e-6)
            elif kind == 11:    # fp16x3 dgrad form: transposed filter, gate, dropout scale, zero-interleaved rows
                cin8, cout8 = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
                xr = torch.randn(B, cin8, H, W, dtype=torch.float64, requires_grad=True)
                w = torch.randn(cout8, cin8, 3, 3) * 0.05
                yref = O.conv3x3(xr, w.double(), torch.zeros(cout8, dtype=torch.float64), sh, circ)
                gy = torch.randn(tuple(yref.shape))
                yref.backward(gy.double())
                gate = torch.randn(B, cin8, H, W)
                scale = torch.rand(B, cin8) + 0.5
                # the gate is a stored split-fp16 activation: open iff its fp16 hi part is > 0 (values under 2^-25 were stored as 0)
                ref = xr.grad * scale[:, :, None, None].double() * (gate.half() > 0).double()
                dx = ops.conv3x3_f16x3_fwd(ops.nchw_to_split_f16(gy.to(dev), cout8), ops.PackedConvF16x3(w.to(dev), None, transpose_flip=True),
                                           stride_h=1, circular=circ, relu=False, drop_scale=scale.to(dev),
                                           gate=ops.nchw_to_split_f16(gate.to(dev), cin8), dilate_h=(sh == 2), out_h=H if sh == 2 else None)
                got = ops.split_f16_to_f32(dx).cpu().permute(0, 3, 1, 2)
                check('dgrad_f16x3', (B, H, W, cin8, cout8, sh, circ), got, ref.float(), 5e-6)
                if os.environ.get('FUZZ_DEBUG'):
                    bad = ((got - ref.float()).abs() > 1e-3).nonzero()
                    if len(bad):
                        b0 = tuple(bad[0].tolist())
                        print('   bad', len(bad), bad[:5].tolist(), 'got', float(got[b0]), 'ref', float(ref[b0]), 'gate', float(gate[b0]),
                              'scale', float(scale[b0[0], b0[1]]), 'ungated', float(xr.grad[b0]), flush=True)
            elif kind == 17:    # cvig_baseline block 1 in one launch vs float64 conv2d (normalisation, LeakyReLU, affine, s2d layout)
                c = int(rng.choice([1, 3, 5]))
                hh, ww = int(rng.integers(4, 140)), int(rng.integers(4, 200))
                xr = torch.from_numpy(rng.integers(0, 256, (B, c, hh, ww)).astype(np.float32))
                wf = torch.randn(64, c, 4, 4) * 0.1
                bf = torch.randn(64) * 0.1
                sc, shf = 1 + 0.1 * torch.randn(64), 0.1 * torch.randn(64)
                yv = ops.conv4x4s2_first(xr.to(dev), wf.to(dev), bf.to(dev), sc.to(dev), shf.to(dev))
                vh, vw = (hh - 4) // 2 + 1, (ww - 4) // 2 + 1
                ref = F.leaky_relu(F.conv2d(-1.0 + 2.0 * (xr.double() / 255.0), wf.double(), bf.double(), stride=2), 0.2)
                ref = ref * sc.double()[None, :, None, None] + shf.double()[None, :, None, None]
                full = torch.zeros((B, 64, 2 * ((vh + 1) // 2), 2 * ((vw + 1) // 2)), dtype=torch.float64)
                full[:, :, :vh, :vw] = ref
                want = torch.cat([full[:, :, dy::2, dx::2] for dy in (0, 1) for dx in (0, 1)], dim=1).permute(0, 2, 3, 1)
                check('first4x4', (B, c, hh, ww), yv.cpu().double(), want, 3e-6)
            elif kind == 18:    # 2x2-tap weight gradient with few input channels (the packed (tap, channel) tile) vs float64 autograd
                c4 = int(rng.choice([4, 8, 12, 16]))
                co = int(rng.choice([8, 64, 72]))
                hh, ww = int(rng.integers(2, 30)), int(rng.integers(2, 100))
                xs = torch.randn(B, hh, ww, c4)
                dz = torch.randn(B, hh, ww, co)
                dw, db = ops.conv3x3_wgrad(xs.to(dev), dz.to(dev), c4, taps4=True)
                wr = torch.zeros(co, c4, 3, 3, dtype=torch.float64, requires_grad=True)
                yr = F.conv2d(xs.double().permute(0, 3, 1, 2), wr, None, padding=1)
                yr.backward(dz.double().permute(0, 3, 1, 2))
                check('wgrad_taps4_small_cin', (B, hh, ww, c4, co), dw.cpu().double()[:, :, 1:, 1:], wr.grad[:, :, 1:, 1:], 2e-5)
                check('wgrad_taps4_bias', (B, hh, ww, c4, co), db.cpu().double(), dz.double().sum((0, 1, 2)), 2e-5)
            elif kind == 19:    # GeM pooling (phases over pixels) and the channel-quad BatchNorm backward vs torch in float64
                cc = int(rng.choice([8, 64, 72, 512]))
                hh, ww = int(rng.integers(1, 20)), int(rng.integers(1, 20))
                a = torch.rand(B, hh + 1, ww + 2, cc) + 0.05
                fo = torch.zeros(B, cc + 8, device=dev)
                ops.gem_pool(a.to(dev), (hh, ww), fo, 4, 3.0)
                ref = a[:, :hh, :ww].double().clamp(min=0).pow(3.0).mean((1, 2)).pow(1.0 / 3.0)
                check('gem_pool', (B, hh, ww, cc), fo[:, 4:4 + cc].cpu().double(), ref, 1e-5)
                if B * hh * ww >= 8:       # a handful of values per channel: the normalisation amplifies fp32 rounding of mean / variance
                    z = torch.randn(B, hh + 1, ww + 2, cc)
                    gam, bet = 1 + 0.1 * torch.randn(cc), 0.1 * torch.randn(cc)
                    dy = torch.randn(B, hh + 1, ww + 2, cc)
                    dy[:, hh:] = 0
                    dy[:, :, ww:] = 0
                    zr = z[:, :hh, :ww].double().permute(0, 3, 1, 2).clone().requires_grad_(True)
                    yb = F.batch_norm(F.leaky_relu(zr, 0.2), None, None, gam.double(), bet.double(), True, 0.1, 1e-5)
                    yb.backward(dy[:, :hh, :ww].double().permute(0, 3, 1, 2))
                    act = F.leaky_relu(z, 0.2).to(dev)
                    mean, invstd, _s, _t = ops.bn_train_stats(act, (hh, ww), gam.to(dev), bet.to(dev))
                    dzv, _dg, _db = ops.bn_lrelu_bwd(act, dy.to(dev), (hh, ww), mean, invstd, gam.to(dev), 0.2)
                    var_ok = float(zr.detach().var(dim=(0, 2, 3), unbiased=False).min()) > 1e-2      # tiny variances amplify fp32 rounding
                    if var_ok:
                        check('bn_lrelu_bwd', (B, hh, ww, cc), dzv.cpu().double()[:, :hh, :ww], zr.grad.permute(0, 2, 3, 1), 5e-5)
            else:               # fused match (kinds 5, 12, 13)
                bo, bs, we = int(rng.integers(1, 40)), int(rng.integers(1, 150)), int(rng.integers(1, 65))
                ov = torch.randn(bo, 16, 4, 64)
                su = torch.randn(bs, 16, 4, we)
                ori_r, d_r = O.match(ov, su)
                sc = O.correlation_scores(ov, su)
                top2 = torch.topk(sc, min(2, sc.shape[-1]), dim=-1).values
                clear = (top2[..., 0] - top2[..., -1]) > 1e-3 if sc.shape[-1] > 1 else torch.ones_like(ori_r, dtype=torch.bool)
                spectral = bool(kind & 1)     # odd kinds: the spectral form (witw_match_fwd_dft), same checks
                ori, d = (ops.match_fwd_dft if spectral else ops.match_fwd)(ov.to(dev), su.to(dev))[:2]
                if not torch.equal(ori.cpu()[clear], ori_r[clear]):
                    fails.append(('match_ori', (bo, bs, we), 0, 0))
                    print('FAIL match orientation', (bo, bs, we), flush=True)
                check('match_dist', (bo, bs, we), d.cpu()[clear], d_r[clear], 2e-5)
        except Exception as e:      # an error return from the C ABI on a legal shape is a failure too
            fails.append(('exception', cfg, str(e)[:200], kind))
            print('EXC kind %d %s: %s' % (kind, cfg, str(e)[:300]), flush=True)
    print('fuzz_parity: %d cases in %.0f s, %d failures' % (n, time.time() - t0, len(fails)), flush=True)
    return ran, fails


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    _, fails = run(budget, seed)
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()
