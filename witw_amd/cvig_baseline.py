#!/usr/bin/env python
"""MI355X-native drop-in for the hot path of the reference's model/cvig_baseline.py (Liu & Li CVPR'19-style
baseline): 7 x [Conv2d(4,2,0) -> LeakyReLU(0.2) -> BatchNorm2d] encoders with GeM-like
multi-scale pooling, exhaustive minibatch triplet loss on squared Euclidean distances, Euclidean ranking.
Inference (BatchNorm running statistics folded into the conv epilogue) and training (batch statistics, full backward
through BatchNorm / LeakyReLU / the 4x4 convolutions / GeM pooling / the exhaustive loss) run on the HIP kernels.
"""
import math

import torch
import torch.nn as nn

from . import _lib, ops
from . import cvig_fov as _fov
from .cvig_fov import Adam, recall_table  # noqa: F401  (same table, model/cvig_baseline.py:461-466)

device = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')   # model/cvig_baseline.py:20
device_parallel = False
device_ids = None


class Globals:
    """model/cvig_baseline.py:24-48"""
    dataset_paths = {
        'cvusa': {'train': './data/train-19zl.csv', 'test': './data/val-19zl.csv'},
        'witw': {'train': './data2/train.csv', 'test': './data2/test.csv'},
    }
    path_formats = {
        'cvusa': {'path_columns': [0, 1], 'path_names': ['overhead', 'surface'], 'header': None, 'panorama': True},
        'witw': {'path_columns': [15, 16], 'path_names': ['surface', 'overhead'], 'header': 0, 'panorama': False},
    }


# how many units make one full turn of the panorama, per first letter of the unit name; None: the shift already is in pixels
_FULL_TURN = {'p': None, 'f': 1., 'd': 360., 'r': 2 * math.pi}
_UNIT_NAMES = {first + rest + plural for first, rest in (('p', 'ixel'), ('f', 'raction'), ('d', 'egree'), ('r', 'adian'))
               for plural in ('', 's')} | set(_FULL_TURN)


def horizontal_shift(img, shift, unit='pixels'):
    """Turn a 360-degree panorama by `shift` (the viewer turning clockwise = the columns moving left), behaviour of
    model/cvig_baseline.py:97-113: units pixel(s)/p, fraction(s)/f, degree(s)/d, radian(s)/r in any case, the column count
    rounded half-to-even, anything else raises. Pinned by tests/golden/augment.npz."""
    name = unit.lower()
    if name not in _UNIT_NAMES:
        raise Exception('! Invalid unit in horizontal_shift()')
    turn = _FULL_TURN[name[0]]
    columns = shift if turn is None else shift * img.size(-1) / turn
    return img.roll(-round(columns), -1)


def quantized_rotation(img, factor):
    """Quarter turns counter-clockwise (model/cvig_baseline.py:116-128); any integer factor, a view where torch allows."""
    return torch.rot90(img, factor % 4, (-1, -2))


class ImagePairDataset(_fov.ImagePairDataset):
    """model/cvig_baseline.py:51-94: {'surface','overhead'} CPU float32 CHW tensors (no 'idx')."""
    _with_idx = False

    @classmethod
    def _globals(cls):
        return Globals


def _on_device(t):
    if not t.is_cuda:
        if device.type != 'cuda':
            raise _lib.WitwError('no gfx950 device: the WITW transforms run on the GPU only')
        t = t.to(device)
    return t


class SyncedRotation(object):
    """model/cvig_baseline.py:130-144: rotate the overhead image by a random angle; a panoramic surface image is
    rolled by the same angle. `angle` can be injected (tests)."""

    def __init__(self, dataset):
        self.dataset = dataset

    def __call__(self, data, angle=None):
        if angle is None:
            angle = torch.rand(()).item() * 360.
        if Globals.path_formats[self.dataset]['panorama']:
            data['surface'] = horizontal_shift(data['surface'], angle, unit='degrees')
        o = _on_device(data['overhead'])
        squeeze = o.dim() == 3
        o = ops.rotate_nearest(o.unsqueeze(0) if squeeze else o, [angle] * (1 if squeeze else o.shape[0]))
        data['overhead'] = o.squeeze(0) if squeeze else o
        return data


class QuantizedSyncedRotation(object):
    """model/cvig_baseline.py:147-160: multiples of 90 degrees (exact transposes / flips)."""

    def __init__(self, dataset):
        self.dataset = dataset

    def __call__(self, data, factor=None):
        if factor is None:
            factor = torch.randint(4, ()).item()
        if Globals.path_formats[self.dataset]['panorama']:
            data['surface'] = horizontal_shift(data['surface'], factor * 90, unit='degrees')
        data['overhead'] = quantized_rotation(data['overhead'], factor)
        return data


class GpuPreprocess(object):
    """Compose[SyncedRotation, SurfaceResize] (model/cvig_baseline.py:324-328) over a batch of raw images, on the
    GPU: -> {'surface' [B,3,Hs,Ws], 'overhead' [B,3,Ho,Wo]} (values stay 0..255, the encoders rescale)."""

    def __init__(self, dataset):
        self.rotation = SyncedRotation(dataset)
        self.resize = SurfaceResize(dataset)

    def __call__(self, batch):
        s, o = [], []
        for su, ov in zip(batch['surface'], batch['overhead']):
            d = self.resize(self.rotation({'surface': _on_device(su[:3]), 'overhead': ov[:3]}))
            s.append(d['surface'])
            o.append(d['overhead'])
        return {'surface': torch.stack(s), 'overhead': torch.stack(o)}


class SurfaceResize(object):
    """model/cvig_baseline.py:208-225 on the GPU."""

    def __init__(self, dataset):
        self.dataset = dataset

    def __call__(self, data):
        s = data['surface']
        if self.dataset == 'cvusa':
            data['surface'] = torch.repeat_interleave(s, 2, dim=-2)
        elif self.dataset == 'witw':
            x = s.to(device) if not s.is_cuda else s
            squeeze = x.dim() == 3
            x = ops.resize_bilinear((x.unsqueeze(0) if squeeze else x).contiguous(), (500, 500))
            data['surface'] = x.squeeze(0) if squeeze else x
        else:
            raise Exception('! Invalid dataset type in ' + type(self).__name__ + '().')
        return data


class SurfaceEncoder(nn.Module):
    """model/cvig_baseline.py:228-279; parameters live in nn.Conv2d / nn.BatchNorm2d children named
    conv1..7 / bn1..7 (same state-dict keys), the forward runs on the HIP kernels."""

    def __init__(self, orientation=False, bands=3, p=3.):
        super().__init__()
        self.orientation, self.bands, self.p = orientation, bands, p
        self.inputs = self.bands + 2 * self.orientation
        widths = [self.inputs, 64, 128, 256, 512, 512, 512, 512]
        for i in range(1, 8):
            conv = nn.Conv2d(widths[i - 1], widths[i], kernel_size=4, stride=2, padding=0)
            bn = nn.BatchNorm2d(widths[i], momentum=0.1, affine=True, track_running_stats=True)
            torch.nn.init.normal_(conv.weight, mean=0.0, std=0.02)     # :255-262
            torch.nn.init.normal_(conv.bias, mean=0.0, std=0.02)
            torch.nn.init.normal_(bn.weight, mean=1.0, std=0.02)
            torch.nn.init.normal_(bn.bias, mean=0.0, std=0.02)
            setattr(self, 'conv%d' % i, conv)
            setattr(self, 'bn%d' % i, bn)
        self._packed = {}
        self.fused_first = True       # eval forward: block 1 by witw_conv4x4s2_first_fwd (False: space-to-depth pass + 2x2-tap conv)

    def _layer(self, i, fold=True):
        """-> (packed 2x2-tap filter, eval-mode BatchNorm scale, shift, padded input channels). fold=False (the training forward:
        batch statistics) skips the scale / shift of the running statistics."""
        conv, bn = getattr(self, 'conv%d' % i), getattr(self, 'bn%d' % i)
        # torch's version counters see in-place torch ops; the C-ABI Adam bumps _witw_version instead
        wkey = tuple(t._version for t in (conv.weight, conv.bias)) + tuple(getattr(t, '_witw_version', 0) for t in (conv.weight, conv.bias)) + \
            (conv.weight.data_ptr(),)
        hit = self._packed.get(i)
        if hit is None or hit[0] != wkey:
            with torch.no_grad():
                co, ci = conv.weight.shape[:2]
                cpad = (4 * ci + 7) // 8 * 8
                # 4x4/s2 filter -> 3x3 filter over space-to-depth channels (dy*2+dx)*ci+c; tap (kh,kw) in {1,2}^2 holds
                # W[:, :, 2(kh-1)+dy, 2(kw-1)+dx], row / column 0 of the taps stay zero (witw_conv4x4_to_k3: one launch)
                k3 = ops.conv4x4_to_k3(conv.weight, cpad)
                packed = ops.PackedConv(k3, conv.bias, taps4=True, reuse=hit[1] if hit else None)      # only the 2x2 live taps
            hit = [wkey, packed, cpad, k3, None]
            self._packed[i] = hit
        if not fold:
            return hit[1], None, None, hit[2]
        bkey = tuple(t._version for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)) + \
            tuple(getattr(t, '_witw_version', 0) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
        if hit[4] is None or hit[4][0] != bkey:
            with torch.no_grad():
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)               # eval-mode BatchNorm2d
                shift = bn.bias - bn.running_mean * scale
            hit[4] = (bkey, scale.contiguous(), shift.contiguous())
        return hit[1], hit[4][1], hit[4][2], hit[2]

    def _layer_k3(self, i):
        self._layer(i, fold=False)
        return self._packed[i][3]

    def _layer_t(self, i):
        """the data-gradient filter of block i (transposed, taps rotated), re-packed in place when the weights have changed"""
        self._layer(i, fold=False)
        wkey, k3 = self._packed[i][0], self._packed[i][3]
        hit = self._packed.get(('t', i))
        if hit is None or hit[0] != wkey:
            hit = (wkey, ops.PackedConv(k3, None, transpose_flip=True, taps4=True, reuse=hit[1] if hit else None))
            self._packed[('t', i)] = hit
        return hit[1]

    def train_params(self):
        out = []
        for i in range(1, 8):
            conv, bn = getattr(self, 'conv%d' % i), getattr(self, 'bn%d' % i)
            out += [conv.weight, conv.bias, bn.weight, bn.bias]
        return out

    def forward(self, x, lrelu_acts=None):
        """model/cvig_baseline.py:264-284. Parity hook (train mode): `lrelu_acts` {block i: NHWC tensor} replaces, in the
        BACKWARD's LeakyReLU gate only, the activation this forward recorded (its sign is the gate); the dict is read when
        the backward runs, so it may be filled after the forward (tests/test_baseline_gpu.py: a conv output within rounding of
        zero can fall on the other side of the kink than it did on the CPU reference)."""
        if not x.is_cuda:
            raise _lib.WitwError('SurfaceEncoder.forward needs a GPU tensor (no CPU fallback)')
        B, _c, H, W = x.shape
        if min(H, W) < 382:
            raise _lib.WitwError('cvig_baseline encoder needs sides >= 382 px, got %dx%d' % (H, W))
        if self.training:
            self._bwd_override = {} if lrelu_acts is None else lrelu_acts
            return _BaselineEncoderFn.apply(x, self, *self.train_params())
        with torch.no_grad():
            f = torch.empty((B, 1536), dtype=torch.float32, device=x.device)
            vh, vw = H, W
            g = 1                  # images per mosaic side of the current layer input h
            # block 1 straight from the raw image (normalisation :265-266, conv1, LeakyReLU, bn1 in one launch) when the second
            # block reads exactly its 4 x 64 space-to-depth channels; otherwise the re-layout pass + the generic 2x2-tap form
            fused_first = self.fused_first and self._layer(2)[3] == 256 and self.inputs <= 5
            h = None if fused_first else ops.space_to_depth2(x.contiguous(), in_nchw=True, normalize=True, cpad=self._layer(1)[3])
            for i in range(1, 8):
                packed, scale, shift, _cp = self._layer(i)
                vh, vw = (vh - 4) // 2 + 1, (vw - 4) // 2 + 1
                if i == 1 and fused_first:
                    h = ops.conv4x4s2_first(x.contiguous(), self.conv1.weight, self.conv1.bias, scale, shift, normalize=True, lrelu_slope=0.2)
                    continue
                if i < 5 and 4 * packed.cout == self._layer(i + 1)[3]:
                    # blocks 1-4: the epilogue writes the next block's space-to-depth input directly
                    h = ops.conv_taps4_s2d(h, packed, (vh, vw), lrelu_slope=0.2, post_scale=scale, post_shift=shift)
                    continue
                # blocks 5-7 (maps of 16x16 and below, K = 4 x 2048): split-K over a mosaic that fills the 16x16 tile
                y = ops.conv_taps4_splitk(h, packed, B, g, (vh, vw), lrelu_slope=0.2, post_scale=scale, post_shift=shift)
                if i >= 5:
                    ops.gem_pool(y, (vh, vw), f, 512 * (i - 5), self.p)                                      # :276-282
                if i < 7:
                    nh, nw = (vh + 1) // 2, (vw + 1) // 2
                    g = max(1, min(16 // nh, 16 // nw))
                    h = ops.space_to_depth2_mosaic(y, g)
            return ops.embed_normalize_(f)                                                                 # :283-284


class OverheadEncoder(SurfaceEncoder):
    pass


def _mosaic_g(x):
    """images per mosaic side for an NHWC batch of small maps: as many cells as fit a 16 x 16 tile"""
    return max(1, min(16 // x.shape[1], 16 // x.shape[2]))


def _to_mosaic(x, g):
    """[B,h,w,C] -> [ceil(B/g^2), g*h, g*w, C]: g x g images side by side (missing images zero); pure data movement on maps of at
    most 16 x 16 (the layout ops.space_to_depth2_mosaic writes directly on the eval path)"""
    if g == 1:
        return x
    B, h, w, C = x.shape
    Bm = (B + g * g - 1) // (g * g)
    if Bm * g * g != B:
        x = torch.cat((x, x.new_zeros((Bm * g * g - B, h, w, C))), 0)
    return x.reshape(Bm, g, g, h, w, C).permute(0, 1, 3, 2, 4, 5).reshape(Bm, g * h, g * w, C).contiguous()


def _pad_hw(y, H, W):
    """[B,vh,vw,C] -> [B,H,W,C], zeros outside (the layout conv3x3_fwd gives an activation: its input's spatial size)"""
    if y.shape[1] == H and y.shape[2] == W:
        return y
    return torch.nn.functional.pad(y, (0, 0, 0, W - y.shape[2], 0, H - y.shape[1]))


class _BaselineEncoderFn(torch.autograd.Function):
    """Train-mode forward (BatchNorm2d batch statistics, running-stat update) and backward of one encoder call."""

    @staticmethod
    def forward(ctx, x, enc, *params):
        B, _c, H, W = x.shape
        with torch.no_grad():
            h = ops.space_to_depth2(x.contiguous(), in_nchw=True, normalize=True, cpad=enc._layer(1, fold=False)[3])
            g = torch.empty((B, 1536), dtype=torch.float32, device=x.device)
            vh, vw = H, W
            saved = []
            for i in range(1, 8):
                packed, _es, _et, _cp = enc._layer(i, fold=False)
                bn = getattr(enc, 'bn%d' % i)
                vh, vw = (vh - 4) // 2 + 1, (vw - 4) // 2 + 1
                if i >= 5:      # maps of 16 x 16 and below: split-K over a mosaic of g x g images, as the eval path (15 of 16 lanes of
                    gm = _mosaic_g(h)                     # the narrow-geometry kernel idle otherwise); zero-padded to the input's size
                    a = _pad_hw(ops.conv_taps4_splitk(_to_mosaic(h, gm), packed, B, gm, (vh, vw), lrelu_slope=0.2), h.shape[1], h.shape[2])
                else:
                    a = ops.conv3x3_fwd(h, packed, relu=False, lrelu_slope=0.2)              # LeakyReLU(conv), :267-275
                mean, invstd, scale, shift = ops.bn_train_stats(a, (vh, vw), bn.weight, bn.bias, bn.running_mean,
                                                                bn.running_var, bn.eps, bn.momentum)
                bn.num_batches_tracked += 1
                for t in (bn.running_mean, bn.running_var):      # updated through the C-ABI: invalidate the eval fold
                    t._witw_version = getattr(t, '_witw_version', 0) + 1
                saved.append((h, a, (vh, vw), mean, invstd, scale, shift))
                if i >= 5:
                    ops.gem_pool(a, (vh, vw), g, 512 * (i - 5), enc.p, scale, shift)
                if i < 7:
                    h = ops.space_to_depth2(a, valid_hw=(vh, vw), cpad=enc._layer(i + 1, fold=False)[3], scale=scale, shift=shift)
            f = ops.embed_normalize_(g.clone())
        ctx.enc, ctx.saved, ctx.g = enc, saved, g
        ctx.override = getattr(enc, '_bwd_override', {})
        enc._bwd_override = {}
        enc._last_saved = saved if getattr(enc, 'keep_activations', False) else None     # diagnostics / parity tests
        return f

    @staticmethod
    def backward(ctx, df):
        enc, saved, g = ctx.enc, ctx.saved, ctx.g
        dg = ops.embed_normalize_bwd(g, df.contiguous())
        grads = [None] * 28
        dx_s2d = None
        for i in range(7, 0, -1):
            h, a, valid, mean, invstd, scale, shift = saved[i - 1]
            conv, bn = getattr(enc, 'conv%d' % i), getattr(enc, 'bn%d' % i)
            s2d_grad = False
            if i == 7:
                dy = ops.gem_pool_bwd(a, scale, shift, g, dg, valid, 1024, enc.p)
            elif i >= 5:
                dy = ops.depth_to_space2(dx_s2d, a, valid)
                ops.gem_pool_bwd(a, scale, shift, g, dg, valid, 512 * (i - 5), enc.p, out=dy)
            else:       # blocks 1-4 (the large maps): the BatchNorm backward reads the space-to-depth gradient in place
                dy, s2d_grad = dx_s2d, True
            dz, dgamma, dbeta = ops.bn_lrelu_bwd(ctx.override.get(i, a), dy, valid, mean, invstd, bn.weight, 0.2, dy_s2d=s2d_grad)
            k3 = enc._layer_k3(i)
            dk3, dbias = ops.conv3x3_wgrad(h, dz, k3.shape[1], stride_h=1, circular=False, taps4=True)
            # tap (ta+1, tb+1) of the 3x3 filter holds W[:, :, 2ta+dy, 2tb+dx] as channel (dy,dx,c): gathered back in one launch
            dw = ops.k3_to_conv4x4(dk3, conv.weight.shape[1])
            grads[4 * (i - 1):4 * i] = [dw, dbias, dgamma, dbeta]
            if i > 1:
                pk_t = enc._layer_t(i)
                if i >= 5:      # the same small maps in the data gradient: mosaic + split-K (dz is zero outside its valid region,
                    gm = _mosaic_g(dz)                    # so the last row / column of every cell is the zero border a window may touch)
                    dx_s2d = ops.conv_taps4_splitk(_to_mosaic(dz, gm), pk_t, dz.shape[0], gm, (dz.shape[1], dz.shape[2]))
                else:
                    dx_s2d = ops.conv3x3_fwd(dz, pk_t, relu=False)
        ctx.saved = ctx.g = None
        return (None, None) + tuple(grads)


class _ExhaustiveLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e1, e2, soft_margin, alpha, margin):
        e1, e2 = e1.contiguous(), e2.contiguous()
        D = ops.pairwise_sqdist(e1, e2)
        ctx.save_for_backward(e1, e2, D)
        ctx.cfg = (soft_margin, alpha, margin)
        return ops.exhaustive_triplet_loss(D, soft_margin, alpha, margin)

    @staticmethod
    def backward(ctx, gl):
        e1, e2, D = ctx.saved_tensors
        de1, de2 = ops.exhaustive_triplet_loss_bwd(e1, e2, D, gl.contiguous(), *ctx.cfg)
        return de1, de2, None, None, None


def exhaustive_minibatch_triplet_loss(embed1, embed2, soft_margin=False, alpha=10., margin=1.):
    """model/cvig_baseline.py:286-315 (all valid (a,p,n) combinations of the minibatch); differentiable."""
    if torch.is_grad_enabled() and (embed1.requires_grad or embed2.requires_grad):
        return _ExhaustiveLossFn.apply(embed1, embed2, bool(soft_margin), float(alpha), float(margin))
    D = ops.pairwise_sqdist(embed1.contiguous(), embed2.contiguous())
    return ops.exhaustive_triplet_loss(D, soft_margin, alpha, margin)


def ranks(overhead_embed, surface_embed):
    """model/cvig_baseline.py:454-460: Euclidean distances, rank = #{gallery : d <= d_true}."""
    D = ops.pairwise_sqdist(overhead_embed.contiguous(), surface_embed.contiguous(), take_sqrt=True)   # [gallery, query]
    return ops.rank_count(D, 0).cpu().numpy().astype('int64')


# ----------------------------------------------------------------------------- drivers
def train(dataset='cvusa', val_quantity=1000, batch_size=16, num_workers=4, num_epochs=999999, csv_path=None):
    """model/cvig_baseline.py:318-404 on the HIP kernels: same flow, prints and checkpoint names; Adam with torch's
    defaults (lr 1e-3) over every encoder parameter. Single GPU: the reference's optional nn.DataParallel (:338-342)
    would change the BatchNorm statistics per replica and is not reproduced."""
    import pathlib
    import time
    pathlib.Path('./weights').mkdir(parents=True, exist_ok=True)
    csv_path = csv_path or Globals.dataset_paths[dataset]['train']
    prep = GpuPreprocess(dataset)
    trainval_set = ImagePairDataset(dataset=dataset, csv_path=csv_path)
    train_set, val_set = torch.utils.data.random_split(trainval_set, [len(trainval_set) - val_quantity, val_quantity])
    train_loader = torch.utils.data.DataLoader(train_set, batch_size=batch_size, shuffle=True, drop_last=True,
                                               num_workers=num_workers, collate_fn=_fov.collate_raw)
    val_loader = torch.utils.data.DataLoader(val_set, batch_size=batch_size, shuffle=False, drop_last=False,
                                             num_workers=num_workers, collate_fn=_fov.collate_raw)
    surface_encoder = SurfaceEncoder().to(device)
    overhead_encoder = OverheadEncoder().to(device)
    loss_func = exhaustive_minibatch_triplet_loss
    optimizer = Adam(list(surface_encoder.parameters()) + list(overhead_encoder.parameters()))

    best_loss = None
    for epoch in range(num_epochs):
        print('Epoch %d, %s' % (epoch + 1, time.ctime(time.time())))
        for phase in ['train', 'val']:
            running_count = 0
            running_loss = 0.
            loader = train_loader if phase == 'train' else val_loader
            surface_encoder.train(phase == 'train')
            overhead_encoder.train(phase == 'train')
            for batch, raw in enumerate(loader):
                data = prep(raw)
                with torch.set_grad_enabled(phase == 'train'):
                    surface_embed = surface_encoder(data['surface'])
                    overhead_embed = overhead_encoder(data['overhead'])
                    loss = loss_func(surface_embed, overhead_embed)
                    if phase == 'train':
                        optimizer.zero_grad()
                        loss.backward()
                        optimizer.step()
                count = surface_embed.size(0)
                running_count += count
                running_loss += loss.item() * count
                print('epoch = {} {}, iter = {}, count = {}, loss = {:.4f}'.format(epoch + 1, phase, batch, running_count,
                                                                                 loss.item()))
            print('  %5s: avg loss = %f' % (phase, running_loss / max(1, running_count)))
        if running_count and (best_loss is None or running_loss / running_count < best_loss):
            print('-------> new best')
            best_loss = running_loss / running_count
            torch.save(surface_encoder.state_dict(), './weights/surface_best.pth')
            torch.save(overhead_encoder.state_dict(), './weights/overhead_best.pth')
    return best_loss


def test(dataset='cvusa', batch_size=16, num_workers=4, csv_path=None):
    """model/cvig_baseline.py:405-475: embed the test set (SyncedRotation stays on, as in the reference :410-414),
    rank every query against the whole gallery on the GPU, print the recall table."""
    csv_path = csv_path or Globals.dataset_paths[dataset]['test']
    prep = GpuPreprocess(dataset)
    test_set = ImagePairDataset(dataset=dataset, csv_path=csv_path)
    test_loader = torch.utils.data.DataLoader(test_set, batch_size=batch_size, shuffle=False, drop_last=False,
                                              num_workers=num_workers, collate_fn=_fov.collate_raw)
    surface_encoder = SurfaceEncoder().to(device)
    overhead_encoder = OverheadEncoder().to(device)
    surface_encoder.load_state_dict(torch.load('./weights/surface_best.pth', map_location='cpu'))
    overhead_encoder.load_state_dict(torch.load('./weights/overhead_best.pth', map_location='cpu'))
    surface_encoder.eval()
    overhead_encoder.eval()
    su_parts, ov_parts = [], []
    for raw in test_loader:
        data = prep(raw)
        with torch.no_grad():
            su_parts.append(surface_encoder(data['surface']))
            ov_parts.append(overhead_encoder(data['overhead']))
    rk = ranks(torch.cat(ov_parts, dim=0), torch.cat(su_parts, dim=0))
    t = recall_table(rk)
    print('Top  1: {:.2f}%'.format(t['top_1']))
    print('Top  5: {:.2f}%'.format(t['top_5']))
    print('Top 10: {:.2f}%'.format(t['top_10']))
    print('Top 1%: {:.2f}%'.format(t['top_1pct']))
    print('Avg. Rank: {:.2f}'.format(t['mean']))
    print('Med. Rank: {:.2f}'.format(t['median']))
    print('Locations: {}'.format(len(rk)))
    return t


def main(argv=None):
    """CLI of model/cvig_baseline.py:478-492."""
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument('--mode', default='train', choices=['train', 'test'], help='Run mode. [Default = train]')
    parser.add_argument('--dataset', default='cvusa', choices=['cvusa', 'witw'], help='Dataset to use. [Default = cvusa]')
    args = parser.parse_args(argv)
    if args.mode == 'train':
        train(dataset=args.dataset)
    elif args.mode == 'test':
        test(dataset=args.dataset)


if __name__ == '__main__':
    main()
