"""CPU oracle for the cvig_fov hot path — TEST INFRASTRUCTURE ONLY.

This is a restatement, in plain CPU torch/numpy ops, of the algorithm of the reference's
model/cvig_fov.py (and the 5-channel variant model/cvig_semantic.py). It exists to check the
HIP kernels; nothing under witw_amd/ may import it. Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg use it.

Parity pin: every function here is checked bit-for-bit / to 1e-6 against outputs of the
reference itself (imported in the build container with stub modules for the missing
torchvision / skimage / tensorboard) through the fixtures under tests/golden/ made by
tests/golden/gen_golden.py — see tests/test_oracle_golden.py.
Not pinned (third-party arithmetic absent from /root/reference and from this image):
torchvision==0.9.1 transforms.functional.resize — restated as bilinear, align_corners=False,
no antialias.

Each function cites the reference file:line it follows.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SURFACE_HEIGHT_MAX = 128   # model/cvig_fov.py:20
SURFACE_WIDTH_MAX = 512    # :21
OVERHEAD_SIZE = 256        # :22
IMG_MEAN = [0.485, 0.456, 0.406]   # :24
IMG_STD = [0.229, 0.224, 0.225]    # :25
SEM_MEAN = [0.485, 0.456, 0.406, 0.45, 0.45]   # model/cvig_semantic.py:25
SEM_STD = [0.229, 0.224, 0.225, 0.22, 0.22]    # model/cvig_semantic.py:26


# ----------------------------------------------------------------------------- transforms
def resize_bilinear(img, size):
    """torchvision 0.9.1 F.resize on a float CHW tensor (call sites model/cvig_fov.py:119,131,133):
    bilinear, align_corners=False, no antialias. UNPINNED (torchvision absent)."""
    return F.interpolate(img.unsqueeze(0), size=size, mode='bilinear', align_corners=False).squeeze(0)


def surface_width(fov):
    """model/cvig_fov.py:113"""
    return int(fov / 360 * SURFACE_WIDTH_MAX)


def resize_pair(surface, overhead, fov=360, panorama=False, start=0):
    """Resize.__call__, model/cvig_fov.py:117-134. `start` replaces the torch.randint draw (:121)."""
    ws = surface_width(fov)
    if panorama:
        s = resize_bilinear(surface, (SURFACE_HEIGHT_MAX, SURFACE_WIDTH_MAX))
        end = start + ws
        if end < SURFACE_WIDTH_MAX:
            s = s[:, :, start:end]
        else:
            s = torch.cat((s[:, :, start:], s[:, :, :end - SURFACE_WIDTH_MAX]), dim=2)
    else:
        s = resize_bilinear(surface, (SURFACE_HEIGHT_MAX, ws))
    o = resize_bilinear(overhead, (OVERHEAD_SIZE, OVERHEAD_SIZE))
    return s, o


def image_normalization(img, mean=IMG_MEAN, std=IMG_STD):
    """ImageNormalization.__call__, model/cvig_fov.py:146-149: Normalize(x / 255.)."""
    m = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return (img / 255. - m) / s


def image_normalization_semantic(img, mean=SEM_MEAN, std=SEM_STD):
    """model/cvig_semantic.py:172-176: only channels 0-2 are divided by 255."""
    x = img.clone()
    x[:3] = x[:3] / 255.
    m = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return (x - m) / s


def polar_grid(h_s=SURFACE_HEIGHT_MAX, w_s=SURFACE_WIDTH_MAX, s_o=OVERHEAD_SIZE):
    """Sampling coordinates of PolarTransform, model/cvig_fov.py:197-201 (fp64)."""
    xx, yy = np.meshgrid(range(w_s), range(h_s))
    yy_o = (s_o / 2) + (s_o / 2) * (h_s - 1 - yy) / h_s * np.cos(2 * math.pi * xx / w_s)
    xx_o = (s_o / 2) - (s_o / 2) * (h_s - 1 - yy) / h_s * np.sin(2 * math.pi * xx / w_s)
    return xx_o, yy_o


def polar_taps(size=OVERHEAD_SIZE, h_s=SURFACE_HEIGHT_MAX, w_s=SURFACE_WIDTH_MAX):
    """The 4 tap indices and fp32 weights of bilinear_interpolate, model/cvig_fov.py:163-181.
    Indices are clipped BEFORE the weights are formed, so taps on the last row/column get
    weight 0 (the reference's behaviour, kept)."""
    x, y = polar_grid(h_s, w_s, size)
    x0 = np.floor(x).astype(int)
    x1 = x0 + 1
    y0 = np.floor(y).astype(int)
    y1 = y0 + 1
    x0 = np.clip(x0, 0, size - 1)
    x1 = np.clip(x1, 0, size - 1)
    y0 = np.clip(y0, 0, size - 1)
    y1 = np.clip(y1, 0, size - 1)
    wa = ((x1 - x) * (y1 - y)).astype(np.float32)
    wb = ((x1 - x) * (y - y0)).astype(np.float32)
    wc = ((x - x0) * (y1 - y)).astype(np.float32)
    wd = ((x - x0) * (y - y0)).astype(np.float32)
    return (x0, x1, y0, y1), (wa, wb, wc, wd)


def polar_transform(overhead):
    """PolarTransform.__call__ + bilinear_interpolate, model/cvig_fov.py:156-209.
    overhead: [C,256,256] fp32 -> [C,128,512] fp32, combine order wa*Ia+wb*Ib+wc*Ic+wd*Id (:183)."""
    size = overhead.shape[-1]
    (x0, x1, y0, y1), (wa, wb, wc, wd) = polar_taps(size)
    Ia = overhead[:, y0, x0]
    Ib = overhead[:, y1, x0]
    Ic = overhead[:, y0, x1]
    Id = overhead[:, y1, x1]
    wa, wb, wc, wd = (torch.from_numpy(w).unsqueeze(0) for w in (wa, wb, wc, wd))
    return wa * Ia + wb * Ib + wc * Ic + wd * Id


# ----------------------------------------------------------------------------- encoder
FOV_LAYERS = [  # (features index, stride_h, relu, pool_after, dropout); model/cvig_fov.py:256-288
    (0, 1, True, False, False), (2, 1, True, True, False),
    (5, 1, True, False, False), (7, 1, True, True, False),
    (10, 1, True, False, False), (12, 1, True, False, False), (14, 1, True, True, False),
    (17, 1, True, False, True), (19, 1, True, False, True), (21, 1, True, False, True),
    (23, 2, True, False, False), (25, 2, True, False, False), (27, 1, False, False, False),
]


def conv3x3(x, w, b, stride_h, circ):
    """Conv2d(3x3, padding=1, stride=(stride_h,1)); with circ: HorizCircPadding
    (model/cvig_fov.py:212-231) = zero pad rows, wrap columns."""
    if circ:
        x = F.pad(x, (0, 0, 1, 1))                    # prelayer ConstantPad2d (:222-223)
        x = F.pad(x, (1, 1, 0, 0), mode='circular')   # padding_mode='circular', padding=(0,1) (:225-227)
        return F.conv2d(x, w, b, stride=(stride_h, 1), padding=0)
    return F.conv2d(x, w, b, stride=(stride_h, 1), padding=1)


def fov_dsm_forward(x, weights, circ_padding, dropout_scales=None, return_activations=False):
    """FOV_DSM.forward, model/cvig_fov.py:292-294 over the layer stack built at :256-288.
    weights: {features index: (w, b)} torch tensors. dropout_scales: None (eval) or
    {17|19|21: [B,C] scale} for train mode with injected Dropout2d masks (applied to the conv
    output BEFORE the ReLU, :287-288, :243-244)."""
    acts = {}
    for (idx, sh, relu, pool, drop) in FOV_LAYERS:
        w, b = weights[idx]
        x = conv3x3(x, w, b, sh, circ_padding)
        if drop and dropout_scales is not None:
            x = x * dropout_scales[idx][:, :, None, None]
        if relu:
            x = F.relu(x)
        if pool:
            x = F.max_pool2d(x, 2, 2)
        if return_activations:
            acts[idx] = x
    return (x, acts) if return_activations else x


# ----------------------------------------------------------------------------- matching
def correlation_scores(overhead_embed, surface_embed):
    """The conv2d of correlation(), model/cvig_fov.py:302-312 -> [Bo,Bs,64] scores."""
    n = surface_embed.shape[3] - 1
    x = torch.cat((overhead_embed, overhead_embed[:, :, :, :n]), axis=3)
    out = F.conv2d(x, surface_embed, stride=1)
    return torch.squeeze(out, -2)


def spectral_scores(overhead_embed, surface_embed):
    """The algebra of witw_match_fwd_dft (witw_amd/csrc/match_dft.hip) restated in numpy fp64 -- test infrastructure for that
    kernel, not a reference function: the scores of correlation_scores() through the 64-point DFT of every (channel,row) line.
    Slots t = 0..32 hold [P | Q] = [Re X_t | Im X_t] per line (Q = 0 for t = 0, 32), X_t = sum_k x[k] e^{-2 pi i t k / 64};
    per slot Re C = P_s.P_o + Q_s.Q_o and Im C = -Q_s.P_o + P_s.Q_o (the kernel's two K halves, the sign applied once);
    even slots accumulate E[shift < 32], odd slots O[shift < 32] with the coefficient table of match_dft_table_kernel;
    score[shift] = E + O, score[shift + 32] = E - O. -> float64 [Bo,Bs,64]."""
    import numpy as np
    ov = overhead_embed.numpy().astype(np.float64).reshape(overhead_embed.shape[0], 64, 64)
    we = surface_embed.shape[3]
    su = np.zeros((surface_embed.shape[0], 64, 64))
    su[:, :, :we] = surface_embed.numpy().astype(np.float64).reshape(surface_embed.shape[0], 64, we)
    k = np.arange(64)
    t = np.arange(33)
    ang = 2.0 * np.pi * ((t[:, None] * k[None, :]) % 64) / 64.0
    cs, sn = np.cos(ang), np.sin(ang)                                  # [33,64]

    def spectrum(x):                                                   # [B,64 lines,64] -> P, Q [B,33,64 lines]
        p = np.einsum('blk,tk->btl', x, cs)
        q = -np.einsum('blk,tk->btl', x, sn)
        q[:, 0] = 0.0
        q[:, 32] = 0.0
        return p.astype(np.float32).astype(np.float64), q.astype(np.float32).astype(np.float64)      # rounded once to fp32

    po, qo = spectrum(ov)
    ps, qs = spectrum(su)
    re = np.einsum('stl,otl->ost', ps, po) + np.einsum('stl,otl->ost', qs, qo)       # [Bo,Bs,33]
    im = -np.einsum('stl,otl->ost', qs, po) + np.einsum('stl,otl->ost', ps, qo)
    shift = np.arange(32)
    a2 = 2.0 * np.pi * ((t[:, None] * shift[None, :]) % 64) / 64.0
    d_re = np.cos(a2) / 32.0
    d_im = -np.sin(a2) / 32.0
    d_re[0], d_im[0] = 1.0 / 64.0, 0.0
    d_re[32], d_im[32] = np.where(shift % 2 == 0, 1.0, -1.0) / 64.0, 0.0
    contrib = re[..., None] * d_re[None, None] + im[..., None] * d_im[None, None]   # [Bo,Bs,33,32]
    e = contrib[:, :, 0::2].sum(axis=2)
    o = contrib[:, :, 1::2].sum(axis=2)
    return np.concatenate([e + o, e - o], axis=2)


def correlation(overhead_embed, surface_embed):
    """model/cvig_fov.py:297-315 -> int64 [Bo,Bs] (argmax, first index on ties)."""
    return torch.argmax(correlation_scores(overhead_embed, surface_embed), -1)


def crop_overhead(overhead_embed, orientation, surface_width):
    """model/cvig_fov.py:318-343: cropped[o,s,c,h,k] = ov[o,c,h,(k+orientation[o,s]) % w], k < surface_width."""
    w = overhead_embed.shape[3]
    k = torch.arange(surface_width)
    idx = torch.fmod(k[None, None, :] + orientation[:, :, None], w)          # [Bo,Bs,k]
    ov = overhead_embed[:, None].expand(-1, orientation.shape[1], -1, -1, -1)  # [Bo,Bs,c,h,w]
    idx = idx[:, :, None, None, :].expand(-1, -1, ov.shape[2], ov.shape[3], -1)
    return torch.gather(ov, 4, idx)


def l2_distance(overhead_cropped, surface_embed):
    """model/cvig_fov.py:346-363 (no epsilon in the norms)."""
    bo, bs = overhead_cropped.shape[:2]
    on = overhead_cropped.reshape(bo, bs, -1)
    on = torch.div(on, torch.linalg.norm(on, ord=2, dim=-1).unsqueeze(-1))
    sn = surface_embed.reshape(bs, -1)
    sn = torch.div(sn, torch.linalg.norm(sn, ord=2, dim=-1).unsqueeze(-1))
    return 2 * (1 - torch.sum(on * sn.unsqueeze(0), -1))


def match(overhead_embed, surface_embed):
    """correlation -> crop_overhead -> l2_distance as called at model/cvig_fov.py:450-453."""
    ori = correlation(overhead_embed, surface_embed)
    crop = crop_overhead(overhead_embed, ori, surface_embed.shape[3])
    return ori, l2_distance(crop, surface_embed)


def match_fused(overhead_embed, surface_embed):
    """Same result without the [Bo,Bs,C,H,We] crop tensor (the identity of SURVEY.md §4):
    distance = 2*(1 - max_corr / (|window| * |surface|)). float64 accumulation; used for
    large shapes where the materialised crop does not fit."""
    ov = overhead_embed.double()
    su = surface_embed.double()
    we = su.shape[3]
    w = ov.shape[3]
    sc = correlation_scores(ov, su)                                  # [Bo,Bs,64]
    ori = torch.argmax(correlation_scores(overhead_embed, surface_embed), -1)
    col = (ov * ov).sum(dim=(1, 2))                                  # [Bo,64]
    col2 = torch.cat((col, col[:, :we - 1]), dim=1) if we > 1 else col
    win = col2.unfold(1, we, 1)[:, :w].sum(-1)                       # [Bo,64] window energy per shift
    best = torch.gather(sc, 2, ori[:, :, None]).squeeze(-1)
    wn = torch.gather(win, 1, ori).sqrt()
    sn = su.reshape(su.shape[0], -1).norm(dim=1)
    return ori, (2 * (1 - best / (wn * sn[None, :]))).float()


def triplet_loss(distances, alpha=10.):
    """model/cvig_fov.py:366-382 (diagonal terms included, naive log(1+exp))."""
    bsz = distances.shape[0]
    m = torch.diagonal(distances)
    s2o = m - distances
    o2s = m.unsqueeze(1) - distances
    l1 = torch.sum(torch.log(1. + torch.exp(alpha * s2o)))
    l2 = torch.sum(torch.log(1. + torch.exp(alpha * o2s)))
    return (l1 + l2) / (2. * bsz * (bsz - 1))


# ----------------------------------------------------------------------------- ranking
def ranks(overhead_embed, surface_embed):
    """The loop body of test(), model/cvig_fov.py:545-552: one query at a time,
    rank = #{gallery : d <= d_true} (ties counted pessimistically)."""
    count = surface_embed.shape[0]
    out = np.zeros([count], dtype=np.int64)
    for idx in range(count):
        q = surface_embed[idx:idx + 1]
        _, d = match(overhead_embed, q)
        d = torch.squeeze(d, 1)
        out[idx] = torch.sum(torch.le(d, d[idx])).item()
    return out


def recall_table(ranks_arr):
    """model/cvig_fov.py:553-558."""
    count = len(ranks_arr)
    return {
        'top_1': np.sum(ranks_arr <= 1) / count * 100,
        'top_5': np.sum(ranks_arr <= 5) / count * 100,
        'top_10': np.sum(ranks_arr <= 10) / count * 100,
        'top_1pct': np.sum(ranks_arr * 100 <= count) / count * 100,
        'mean': float(np.mean(ranks_arr)),
        'median': float(np.median(ranks_arr)),
    }


# ----------------------------------------------------------------------------- training step
TRAINABLE = (17, 19, 21, 23, 25, 27)   # model/cvig_fov.py:275-278


def train_step(surface, overhead, w_surface, w_overhead, drop_surface=None, drop_overhead=None, lr=1.E-5,
               adam_state=None, trainable=None):
    """One iteration of the loop body model/cvig_fov.py:444-461 with torch autograd on the CPU:
    forward (train mode, injected Dropout2d scales) -> correlation/crop/l2_distance -> triplet_loss ->
    backward -> Adam(lr) step over the trainable layers. w_*: {idx: (w, b)} torch tensors (updated in place).
    Returns (loss, orientation, distance, grads {('s'|'o', idx): (dw, db)})."""
    trainable = TRAINABLE if trainable is None else trainable      # cvig_semantic: (0,) + TRAINABLE (model/cvig_semantic.py:308)
    leaves = {}
    for tag, w in (('s', w_surface), ('o', w_overhead)):
        for idx in trainable:
            for t in w[idx]:
                t.requires_grad_(True)
                t.grad = None
            leaves[(tag, idx)] = w[idx]
    s_emb = fov_dsm_forward(surface, w_surface, False, dropout_scales=drop_surface)
    o_emb = fov_dsm_forward(overhead, w_overhead, True, dropout_scales=drop_overhead)
    ori, dist = match(o_emb, s_emb)
    loss = triplet_loss(dist)
    loss.backward()
    grads = {k: (v[0].grad.clone(), v[1].grad.clone()) for k, v in leaves.items()}
    params = [t for v in leaves.values() for t in v]
    opt = torch.optim.Adam(params, lr=lr)
    if adam_state is not None:
        opt.load_state_dict(adam_state)
    opt.step()
    for t in params:
        t.requires_grad_(False)
    return loss.detach(), ori, dist.detach(), grads


def fov_dsm_forward_bf16_emulated(x, weights, circ_padding):
    """What the bf16 MFMA path computes, emulated on the CPU: inputs, filters and every stored activation are
    rounded to bfloat16 (round-to-nearest-even), products/accumulation/bias/ReLU/max-pool in fp32, the last
    layer's output stays fp32. (No reference counterpart: the reference has no reduced-precision path; this
    pins the GPU kernel to 'the fp32 algorithm of model/cvig_fov.py:292-294 with bf16 storage'.)"""
    def r(t):
        return t.bfloat16().float()
    x = r(x)
    last = FOV_LAYERS[-1][0]
    for (idx, sh, relu, pool, drop) in FOV_LAYERS:
        w, b = weights[idx]
        x = conv3x3(x, r(w), b, sh, circ_padding)
        if relu:
            x = F.relu(x)
        if pool:
            x = F.max_pool2d(x, 2, 2)
        if idx != last:
            x = r(x)
    return x


def fov_dsm_forward_f16x3_emulated(x, weights, circ_padding):
    """What the fp16x3 path computes, emulated on the CPU: every layer input and filter is split into hi = fp16(v) and
    lo = fp16(v - hi) (fp16 subnormals kept, as the MI355X matrix core does) and the convolution is the sum of the
    hi*hi, lo*hi and hi*lo convolutions in fp32; bias / ReLU / max-pool in fp32. (No reference counterpart; the path is
    ALSO held directly to the reference goldens at the fp32 tolerance.)"""
    def split(t):
        hi = t.half().float()
        return hi, (t - hi).half().float()
    for (idx, sh, relu, pool, drop) in FOV_LAYERS:
        w, b = weights[idx]
        xh, xl = split(x)
        wh, wl = split(w)
        z = torch.zeros_like(b)
        x = conv3x3(xh, wh, b, sh, circ_padding) + conv3x3(xl, wh, z, sh, circ_padding) + conv3x3(xh, wl, z, sh, circ_padding)
        if relu:
            x = F.relu(x)
        if pool:
            x = F.max_pool2d(x, 2, 2)
    return x
