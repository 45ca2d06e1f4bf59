#!/usr/bin/env python
"""MI355X-native drop-in for the hot path of the reference's model/cvig_fov.py.

Same public names and argument meaning as the reference module (Globals, FOV_DSM,
correlation, crop_overhead, l2_distance, triplet_loss, device ...); the arithmetic runs in
hand-written HIP kernels (witw_amd/csrc) behind the C ABI of include/witw_hip.h. There is no
CPU path: tensors must live on the gfx950 device.
"""
import torch

from . import _lib, ops, synth


class Globals:
    """model/cvig_fov.py:19-51"""
    surface_height_max = 128
    surface_width_max = 512
    overhead_size = 256

    img_mean = [0.485, 0.456, 0.406]
    img_std = [0.229, 0.224, 0.225]

    dataset_paths = {
        'cvusa': {'train': './data/train-19zl.csv', 'test': './data/val-19zl.csv'},
        'witw': {'train': './data2/train.csv', 'test': './data2/test.csv'},
    }

    path_formats = {
        'cvusa': {'path_columns': [0, 1], 'path_names': ['overhead', 'surface'], 'header': None, 'panorama': True},
        'witw': {'path_columns': [15, 16], 'path_names': ['surface', 'overhead'], 'header': 0, 'panorama': False},
    }


device = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')  # model/cvig_fov.py:578


class HorizCircPadding(torch.nn.Module):
    """Parameter container with the reference's nesting (model/cvig_fov.py:212-231): the wrapped
    conv is `.layer`. Circular-W / zero-H padding itself is a halo-load policy of the HIP conv."""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer


class AddDropout(torch.nn.Module):
    """Parameter container mirroring model/cvig_fov.py:234-245 (`.layer`, p). Dropout2d is applied
    as a per-(sample,channel) scale in the conv epilogue, before the ReLU."""

    def __init__(self, layer, p=0.5):
        super().__init__()
        self.layer = layer
        self.p = p


class _VGGShell(torch.nn.Module):
    def __init__(self, features):
        super().__init__()
        self.features = features


def _conv_of(m):
    while not isinstance(m, torch.nn.Conv2d):
        m = m.layer
    return m


class FOV_DSM(torch.nn.Module):
    """VGG16 features[:23] + 3 extra convs (model/cvig_fov.py:248-294), forward on HIP kernels.

    State-dict keys follow the reference (`model.features.{i}[.layer[.layer]].weight`).
    Weights come from `weights` ({features idx: (w, b)} numpy) or a seeded synthetic set —
    torch.hub's pretrained VGG16 (reference :256) is unreachable without network; load real
    weights with load_state_dict. The unused VGG classifier of the reference is not kept.
    """
    in_channels = 3

    def __init__(self, circ_padding=False, weights=None, seed=0):
        super().__init__()
        if weights is None:
            weights = synth.fov_dsm_weights(seed, in_channels=self.in_channels)
        mods = []
        self.layer_specs = []
        for (idx, cin, cout, sh, relu, pool, drop) in synth.FOV_LAYERS:
            if idx == 0:
                cin = self.in_channels
            while len(mods) < idx:
                mods.append(torch.nn.Identity())
            conv = torch.nn.Conv2d(cin, cout, 3, (sh, 1), padding=1)
            w, b = weights[idx]
            with torch.no_grad():
                conv.weight.copy_(torch.as_tensor(w))
                conv.bias.copy_(torch.as_tensor(b))
            if idx < synth.TRAINABLE_FROM:   # model/cvig_fov.py:275-278
                conv.weight.requires_grad = False
                conv.bias.requires_grad = False
            m = conv
            if circ_padding:
                m = HorizCircPadding(m)
            if drop:
                m = AddDropout(m, 0.2)
            mods.append(m)
            self.layer_specs.append((idx, sh, relu, pool, drop))
        # indices of ReLU / MaxPool of the reference hold parameter-free placeholders
        self.model = _VGGShell(torch.nn.Sequential(*mods))
        self.circ_padding = circ_padding
        self._packed = {}

    def _pack(self, idx):
        conv = _conv_of(self.model.features[idx])
        key = (conv.weight.data_ptr(), conv.weight._version, conv.bias._version, getattr(conv.weight, '_witw_version', 0),
               getattr(conv.bias, '_witw_version', 0))
        hit = self._packed.get(idx)
        if hit is None or hit[0] != key:
            hit = (key, ops.PackedConv(conv.weight, conv.bias))
            self._packed[idx] = hit
        return hit[1]

    def _pack_t(self, idx):
        """dgrad filter (transpose + 180-degree tap rotation) of layer idx."""
        conv = _conv_of(self.model.features[idx])
        key = (conv.weight.data_ptr(), conv.weight._version, getattr(conv.weight, '_witw_version', 0))
        hit = self._packed.get(('t', idx))
        if hit is None or hit[0] != key:
            hit = (key, ops.PackedConv(conv.weight, None, transpose_flip=True))
            self._packed[('t', idx)] = hit
        return hit[1]

    def _draw_scales(self, x, dropout_scales):
        scales = {}
        for (idx, sh, relu, pool, drop) in self.layer_specs:
            if drop and self.training:
                if dropout_scales is not None:
                    scales[idx] = dropout_scales[idx].contiguous()
                else:   # Dropout2d(p=0.2): whole channels, scale 1/(1-p) (reference :241,288)
                    conv = _conv_of(self.model.features[idx])
                    keep = torch.rand((x.shape[0], conv.out_channels), device=x.device) >= 0.2
                    scales[idx] = keep.float() / 0.8
        return scales

    def _run(self, x, scales, keep_from=None):
        """Layer stack; returns (embedding NCHW, {idx: NHWC output} for idx >= keep_from, input of keep_from)."""
        h = ops.nchw_to_nhwc8(x.contiguous())
        last = self.layer_specs[-1][0]
        kept, first_in = {}, None
        for (idx, sh, relu, pool, drop) in self.layer_specs:
            if keep_from is not None and idx == keep_from:
                first_in = h
            h = ops.conv3x3_fwd(h, self._pack(idx), stride_h=sh, circular=self.circ_padding, relu=relu, pool=pool,
                                out_nchw=(idx == last), drop_scale=scales.get(idx))
            if keep_from is not None and idx >= keep_from:
                kept[idx] = h
        return h, kept, first_in

    def trainable_convs(self):
        return [(idx, _conv_of(self.model.features[idx])) for (idx, *_r) in self.layer_specs
                if _conv_of(self.model.features[idx]).weight.requires_grad]

    def forward(self, x, dropout_scales=None):
        """x [B,C,128,W] NCHW fp32 on the GPU -> [B,16,4,W/8] NCHW (reference :292-294).
        In train() mode Dropout2d scales are drawn per call unless `dropout_scales`
        ({17|19|21: [B,C]}) injects them. With grad enabled the call is recorded for autograd
        (weight / bias gradients of the trainable layers, computed by the HIP backward kernels)."""
        if not x.is_cuda:
            raise _lib.WitwError('FOV_DSM.forward needs a GPU tensor (no CPU fallback)')
        scales = self._draw_scales(x, dropout_scales)
        tr = self.trainable_convs()
        if torch.is_grad_enabled() and tr:
            if tr[0][0] != synth.TRAINABLE_FROM or self.in_channels != 3:
                raise _lib.WitwError('backward is implemented for the cvig_fov trainable set (layers >= 17) only')
            params = []
            for _i, c in tr:
                params += [c.weight, c.bias]
            return _EncoderFn.apply(x, self, scales, *params)
        return self._run(x, scales)[0]


class _EncoderFn(torch.autograd.Function):
    """autograd node of one FOV_DSM call: forward = the 13 fused conv launches; backward = per
    trainable layer one wgrad launch and one dgrad launch (the forward kernel on the transposed,
    tap-rotated filter with the ReLU / Dropout2d gate fused into its epilogue)."""

    @staticmethod
    def forward(ctx, x, enc, scales, *params):
        out, kept, first_in = enc._run(x, scales, keep_from=synth.TRAINABLE_FROM)
        ctx.enc, ctx.scales, ctx.kept, ctx.first_in = enc, scales, kept, first_in
        return out

    @staticmethod
    def backward(ctx, grad_out):
        enc, scales, kept = ctx.enc, ctx.scales, ctx.kept
        specs = [sp for sp in enc.layer_specs if sp[0] >= synth.TRAINABLE_FROM]
        circ = enc.circ_padding
        last = specs[-1][0]
        cout_last = _conv_of(enc.model.features[last]).out_channels
        dz = ops.nchw_to_nhwc(grad_out.contiguous(), (cout_last + 7) // 8 * 8)   # layer 27 has no ReLU
        grads = {}
        for n in range(len(specs) - 1, -1, -1):
            idx, sh, relu, pool, drop = specs[n]
            x_in = ctx.first_in if n == 0 else kept[specs[n - 1][0]]
            conv = _conv_of(enc.model.features[idx])
            grads[idx] = ops.conv3x3_wgrad(x_in, dz, conv.in_channels, stride_h=sh, circular=circ)
            if n > 0:   # gradient at the previous layer's conv output: dgrad, gated by its ReLU and dropout
                pidx = specs[n - 1][0]
                dz = ops.conv3x3_fwd(dz, enc._pack_t(idx), stride_h=1, circular=circ, relu=False, pool=False,
                                     drop_scale=scales.get(pidx), gate=kept[pidx], dilate_h=(sh == 2),
                                     out_h=x_in.shape[1] if sh == 2 else None)
        ctx.kept = ctx.first_in = None
        flat = []
        for (idx, *_r) in specs:
            flat += [grads[idx][0], grads[idx][1]]
        return (None, None, None) + tuple(flat)


# ----------------------------------------------------------------------------- transforms
def _batched(t):
    return (t.unsqueeze(0), True) if t.dim() == 3 else (t, False)


def _to_device(t):
    if not t.is_cuda:
        if device.type != 'cuda':
            raise _lib.WitwError('no gfx950 device: the WITW transforms run on the GPU only')
        t = t.to(device)
    return t.contiguous()


class Resize(object):
    """model/cvig_fov.py:100-134 on the GPU. Accepts per-sample CHW or batched NCHW tensors in
    the data dict. `start` fixes the random FoV crop offset (reference draws torch.randint, :121)."""

    def __init__(self, dataset, fov=360, random_orientation=True):
        self.fov = fov
        self.surface_width = int(self.fov / 360 * Globals.surface_width_max)
        self.panorama = Globals.path_formats[dataset]['panorama']
        self.random_orientation = random_orientation

    def __call__(self, data, start=None):
        s, squeeze = _batched(_to_device(data['surface']))
        if self.panorama:
            s = ops.resize_bilinear(s, (Globals.surface_height_max, Globals.surface_width_max))
            if start is None:
                start = int(torch.randint(0, Globals.surface_width_max, ())) if self.random_orientation else 0
            end = start + self.surface_width
            if end < Globals.surface_width_max:
                s = s[:, :, :, start:end]
            else:
                s = torch.cat((s[:, :, :, start:], s[:, :, :, :end - Globals.surface_width_max]), dim=3)
            s = s.contiguous()
        else:
            s = ops.resize_bilinear(s, (Globals.surface_height_max, self.surface_width))
        o, _ = _batched(_to_device(data['overhead']))
        o = ops.resize_bilinear(o, (Globals.overhead_size, Globals.overhead_size))
        data['surface'] = s.squeeze(0) if squeeze else s
        data['overhead'] = o.squeeze(0) if squeeze else o
        return data


class ImageNormalization(object):
    """model/cvig_fov.py:137-149."""

    def __init__(self, mean=None, std=None, n_div255=None):
        self.keys = ['surface', 'overhead']
        self.mean = Globals.img_mean if mean is None else mean
        self.std = Globals.img_std if std is None else std
        self.n_div255 = n_div255

    def __call__(self, data):
        for key in self.keys:
            t, squeeze = _batched(_to_device(data[key]))
            t = ops.normalize(t, self.mean, self.std, self.n_div255)
            data[key] = t.squeeze(0) if squeeze else t
        return data


class PolarTransform(object):
    """model/cvig_fov.py:186-209."""

    def __call__(self, data):
        t, squeeze = _batched(_to_device(data['overhead']))
        p = ops.polar_transform(t, Globals.surface_height_max, Globals.surface_width_max)
        data['polar'] = p.squeeze(0) if squeeze else p
        return data


# ----------------------------------------------------------------------------- matching + loss
def correlation(overhead_embed, surface_embed):
    """model/cvig_fov.py:297-315 -> int64 [Bo,Bs]."""
    return ops.match_fwd(overhead_embed.contiguous(), surface_embed.contiguous())[0]


def crop_overhead(overhead_embed, orientation, surface_width):
    """model/cvig_fov.py:318-343 (materialising; the drivers use match() instead)."""
    return ops.crop_overhead(overhead_embed.contiguous(), orientation.contiguous(), surface_width)


def l2_distance(overhead_cropped, surface_embed):
    """model/cvig_fov.py:346-363."""
    return ops.l2_distance(overhead_cropped.contiguous(), surface_embed.contiguous())


class _MatchFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, overhead_embed, surface_embed):
        ov, su = overhead_embed.contiguous(), surface_embed.contiguous()
        ori, dist, score, ws = ops.match_fwd(ov, su, want_score=True, want_workspace=True)
        ctx.save_for_backward(ov, su, ori, score, ws)
        ctx.mark_non_differentiable(ori)
        return ori, dist

    @staticmethod
    def backward(ctx, _g_ori, g_dist):
        ov, su, ori, score, ws = ctx.saved_tensors
        gov, gsu = ops.match_bwd(ov, su, ori, score, ws, g_dist.contiguous(), ctx.needs_input_grad[0],
                                 ctx.needs_input_grad[1])
        return gov, gsu


def match(overhead_embed, surface_embed):
    """correlation -> crop_overhead -> l2_distance fused (no crop tensor): (orientation, distance).
    Differentiable w.r.t. both embeddings (the arg-max orientation is a constant, as in the reference)."""
    if torch.is_grad_enabled() and (overhead_embed.requires_grad or surface_embed.requires_grad):
        return _MatchFn.apply(overhead_embed, surface_embed)
    return ops.match_fwd(overhead_embed.contiguous(), surface_embed.contiguous())


class Adam(object):
    """torch.optim.Adam(params, lr) as used at model/cvig_fov.py:416-418 (defaults betas=(0.9,0.999),
    eps=1e-8, no weight decay), one fused HIP launch per parameter. Parameters that never receive a
    gradient (the frozen layers) are skipped, as torch does."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.state = {}

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self):
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = {'step': 0, 'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p)}
            st['step'] += 1
            with torch.no_grad():
                ops.adam_step(p.data, p.grad.contiguous(), st['exp_avg'], st['exp_avg_sq'], st['step'], self.lr,
                              self.betas[0], self.betas[1], self.eps)
                p._witw_version = getattr(p, '_witw_version', 0) + 1   # packed-weight caches key on this


class _TripletLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, distances, alpha):
        d = distances.contiguous()
        loss, ws = ops.triplet_loss_fwd(d, alpha)
        ctx.save_for_backward(d, ws)
        ctx.alpha = alpha
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad):
        d, ws = ctx.saved_tensors
        return ops.triplet_loss_bwd(d, ws, grad.contiguous(), ctx.alpha), None


def triplet_loss(distances, alpha=10.):
    """model/cvig_fov.py:366-382."""
    return _TripletLoss.apply(distances, float(alpha))


def ranks(overhead_embed, surface_embed):
    """Ranking loop of test() (model/cvig_fov.py:543-552) for all queries at once: int64 [N] on
    the host, rank = #{gallery : d <= d_true} with gallery index == query index."""
    _, dist = match(overhead_embed, surface_embed)
    return ops.rank_count(dist, 0).cpu().numpy().astype('int64')


def sharded_ranks(overhead_shard, surface_all, shard_begin, query_chunk=4096, _match=None, _count=None):
    """Ranking with the GALLERY sharded by rows across ranks: this rank holds
    overhead_shard = gallery rows [shard_begin, shard_begin+n); queries (replicated) match gallery row
    == query index. The owner of each true row publishes its distance (all-reduce of a vector that is
    zero elsewhere), every rank counts d <= d_true over its shard, counts are summed. Returns int64 [N]
    on the host, identical on every rank and identical to ranks() on one GPU."""
    from . import parallel
    _match = _match or ops.match_fwd                 # injectable so the collective algebra is testable on CPU/gloo
    _count = _count or ops.rank_count_thresh
    n_q = surface_all.shape[0]
    n_g = overhead_shard.shape[0]
    out = torch.zeros((n_q,), dtype=torch.int32, device=surface_all.device)
    for q0 in range(0, n_q, query_chunk):
        q1 = min(n_q, q0 + query_chunk)
        _, dist = _match(overhead_shard.contiguous(), surface_all[q0:q1].contiguous())   # [n_g, q]
        qi = torch.arange(q0, q1, device=dist.device)
        own = (qi >= shard_begin) & (qi < shard_begin + n_g)
        row = (qi - shard_begin).clamp(0, n_g - 1)
        d_true = torch.where(own, dist[row, qi - q0], torch.zeros_like(dist[0]))
        parallel.all_reduce_sum_(d_true)
        out[q0:q1] = _count(dist, d_true.contiguous())
    parallel.all_reduce_sum_(out)
    return out.cpu().numpy().astype('int64')


def recall_table(ranks_arr):
    """model/cvig_fov.py:553-558."""
    import numpy as np
    count = len(ranks_arr)
    return {
        'top_1': np.sum(ranks_arr <= 1) / count * 100,
        'top_5': np.sum(ranks_arr <= 5) / count * 100,
        'top_10': np.sum(ranks_arr <= 10) / count * 100,
        'top_1pct': np.sum(ranks_arr * 100 <= count) / count * 100,
        'mean': float(np.mean(ranks_arr)),
        'median': float(np.median(ranks_arr)),
    }
