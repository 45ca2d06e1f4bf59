#!/usr/bin/env python
"""MI355X-native drop-in for the hot path of the reference's model/cvig_fov.py.

Same public names and argument meaning as the reference module (Globals, FOV_DSM,
correlation, crop_overhead, l2_distance, triplet_loss, device ...); the arithmetic runs in
hand-written HIP kernels (witw_amd/csrc) behind the C ABI of include/witw_hip.h. There is no
CPU path: tensors must live on the gfx950 device.
"""
import os

import torch

from . import _lib, build as _build, ops, synth


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

    # not in the reference (which downloads them, :256): torchvision VGG16 weights for train() -- a path to a state_dict file
    # with the keys `features.{i}.weight|bias` / `classifier.*` (torchvision.models.vgg16().state_dict()), or None = seeded
    # synthetic weights (benchmarks / tests: there is no network here). CLI: --vgg16 PATH.
    vgg16_weights = None
    # train() writes checkpoints the reference's strict load_state_dict accepts (the unused VGG classifier keys included)
    reference_checkpoints = True

    # not in the reference: arithmetic of the encoders built by train() / test() / the CLI (`--precision`).
    # 'fp32' = the reference's arithmetic (parity path); 'bf16' = bf16 MFMA operands, fp32 accumulate / weights / Adam.
    precision = 'fp32'
    # not in the reference: the data path of train() / test(). device_jpeg: DataLoader workers only entropy-decode JPEG files and
    # the GPU finishes them (byte-identical to Pillow, witw_amd/jpeg.py); pinned_ring: workers build their batches in page-locked
    # shared memory (witw_amd/ring.py) instead of torch's pickling + pinning thread.
    device_jpeg = True
    pinned_ring = True
    # not in the reference: how test() ranks the queries against the gallery. 'direct' = the reference's sum on every pair,
    # 'dft' = the spectral pass with exact re-scoring (same ranks), 'auto' = 'dft' from cvig_fov.SPECTRAL_FROM pairs on.
    match_method = 'auto'


def _default_device():
    # model/cvig_fov.py:578 uses cuda:0; under torch.distributed.run every process takes the GPU of its LOCAL_RANK
    import os
    if not torch.cuda.is_available():
        return torch.device('cpu')
    n = torch.cuda.device_count()
    return torch.device('cuda:%d' % (int(os.environ.get('LOCAL_RANK', '0')) % max(1, n)))


device = _default_device()


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
    # bf16 inference: layers 0 and 2 as one launch (False: the two separate kernels, same bits). None = decided at the first bf16
    # forward, i.e. AFTER the library has been built and loaded: off when build.py found the fused kernel compiled with a register
    # allocation the parity tests have not seen (hand-counted LDS waits), unless WITW_F2 is set (_lib.guards()['first2'])
    fuse_first2 = None
    dropout_seed = None       # None: torch.initial_seed()
    # 'fp32' = the reference's arithmetic on the fp32 MFMA kernels (parity path). 'bf16' = mixed precision on the bf16
    # MFMA kernels: bf16 activations / filters / activation gradients, fp32 accumulate, fp32 weight gradients, master
    # weights and Adam (BASELINE config "bf16 MFMA"; not bit-comparable, see tests/test_bf16_train_gpu.py). 'fp16x3' =
    # evaluation with fp32-grade products from fp16 hi/lo pairs on the fp16 MFMA (forward_f16x3; same goldens, same 1e-4).
    precision = 'fp32'

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
        # Dropout2d mask stream of this encoder: a STABLE id per side (0 surface, 1 overhead), not the count of encoders the
        # process has built (an encoder made earlier by test() / a bench block must not change a seeded run's masks); two
        # encoders of one side in a process that need distinct masks set `dropout_stream` themselves. The step counter advances
        # per training call; train() sets it from the global step so that a re-started run continues the mask sequence.
        self.dropout_stream = 1 if circ_padding else 0
        self._drop_step = 0

    def _pack(self, idx):
        conv = _conv_of(self.model.features[idx])
        key = (conv.weight.data_ptr(), conv.weight._version, conv.bias._version, getattr(conv.weight, '_witw_version', 0),
               getattr(conv.bias, '_witw_version', 0))
        hit = self._packed.get(idx)
        if hit is None or hit[0] != key:
            hit = (key, ops.PackedConv(conv.weight, conv.bias, reuse=hit[1] if hit else None))
            self._packed[idx] = hit
        return hit[1]

    def _pack_t(self, idx):
        """dgrad filter (transpose + 180-degree tap rotation) of layer idx."""
        conv = _conv_of(self.model.features[idx])
        key = (conv.weight.data_ptr(), conv.weight._version, getattr(conv.weight, '_witw_version', 0))
        hit = self._packed.get(('t', idx))
        if hit is None or hit[0] != key:
            hit = (key, ops.PackedConv(conv.weight, None, transpose_flip=True, reuse=hit[1] if hit else None))
            self._packed[('t', idx)] = hit
        return hit[1]

    def _draw_scales(self, x, dropout_scales):
        """Dropout2d(p=0.2) scales of this call (whole channels, 1/(1-p) on the kept ones; reference :241,288): injected, or
        drawn by the counter-based generator of csrc/loss.hip from (seed, encoder, step, rank, layer, sample, channel) in one
        launch. seed = self.dropout_seed, else torch.initial_seed() (so torch.manual_seed still names the run); the step
        counter advances per training call."""
        layers = [idx for (idx, sh, relu, pool, drop) in self.layer_specs if drop]
        if not self.training or not layers:
            return {}
        if dropout_scales is not None:
            return {idx: dropout_scales[idx].contiguous() for idx in layers}
        from . import parallel
        seed = self.dropout_seed if self.dropout_seed is not None else torch.initial_seed()
        ch = _conv_of(self.model.features[layers[0]]).out_channels
        p = float(self.model.features[layers[0]].p)
        for idx in layers[1:]:      # one launch draws all layers: they have to agree on what is drawn
            if _conv_of(self.model.features[idx]).out_channels != ch or float(self.model.features[idx].p) != p:
                raise _lib.WitwError('Dropout2d layers of one encoder must share channel count and p (layer %d differs)' % idx)
        sc = ops.dropout2d_scales(seed, self.dropout_stream & 0xFFFF, self._drop_step, parallel.rank(), layers, x.shape[0], ch, p,
                                  x.device)
        self._drop_step += 1
        return {idx: sc[i] for i, idx in enumerate(layers)}

    def _run(self, x, scales, keep_from=None):
        """Layer stack. Returns (embedding NCHW, kept) where kept[idx] = (layer input NHWC, layer output NHWC,
        max-pool arg-max codes or None) for every layer idx >= keep_from (what the backward needs)."""
        fast0 = self.in_channels <= 4 and (keep_from is None or keep_from > 0)
        h = x.contiguous() if fast0 else ops.nchw_to_nhwc8(x.contiguous())
        last = self.layer_specs[-1][0]
        kept = {}
        for (idx, sh, relu, pool, drop) in self.layer_specs:
            if idx == 0 and fast0:     # C<=4 -> 64 straight from NCHW (layer 0 is frozen: nothing to keep)
                h = ops.conv3x3_first_fwd(h, self._pack_first(False), circular=self.circ_padding, relu=relu)
                continue
            keep = keep_from is not None and idx >= keep_from
            out = ops.conv3x3_fwd(h, self._pack(idx), stride_h=sh, circular=self.circ_padding, relu=relu, pool=pool,
                                  out_nchw=(idx == last), drop_scale=scales.get(idx), want_pool_code=(keep and pool))
            y, code = out if (keep and pool) else (out, None)
            if keep:
                kept[idx] = (h, y, code)
            h = y
        return h, kept

    def _pack_first(self, bf16):
        conv = _conv_of(self.model.features[0])
        key = (conv.weight.data_ptr(), conv.weight._version, conv.bias._version, getattr(conv.weight, '_witw_version', 0),
               getattr(conv.bias, '_witw_version', 0))
        hit = self._packed.get(('first', bf16))
        if hit is None or hit[0] != key:
            hit = (key, ops.PackedFirstConv(conv.weight, conv.bias, bf16=bf16))
            self._packed[('first', bf16)] = hit
        return hit[1]

    def _pack_bf16(self, idx):
        conv = _conv_of(self.model.features[idx])
        key = (conv.weight.data_ptr(), conv.weight._version, conv.bias._version, getattr(conv.weight, '_witw_version', 0),
               getattr(conv.bias, '_witw_version', 0))
        hit = self._packed.get(('bf16', idx))
        if hit is None or hit[0] != key:
            hit = (key, ops.PackedConvBf16(conv.weight, conv.bias, reuse=hit[1] if hit else None))
            self._packed[('bf16', idx)] = hit
        return hit[1]

    def _pack_t_bf16(self, idx):
        """bf16 dgrad filter (transpose + 180-degree tap rotation) of layer idx."""
        conv = _conv_of(self.model.features[idx])
        key = (conv.weight.data_ptr(), conv.weight._version, getattr(conv.weight, '_witw_version', 0))
        hit = self._packed.get(('t_bf16', idx))
        if hit is None or hit[0] != key:
            hit = (key, ops.PackedConvBf16(conv.weight, None, transpose_flip=True, reuse=hit[1] if hit else None))
            self._packed[('t_bf16', idx)] = hit
        return hit[1]

    def _refresh_packed_bf16(self, first):
        """Every STALE bf16 filter image the step from layer `first` on will use -- forward images of all layers, dgrad images
        (transposed, tap-rotated) of the layers behind `first` -- re-packed by one launch (ops.PackedConvBf16.batch) instead of one
        launch + one bias copy per image as _pack_bf16 / _pack_t_bf16 would do them on first use: after an Adam update that is
        11 images per encoder (model/cvig_fov.py:275-278: the trainable layers). The cache keys are those of the lazy methods."""
        todo = []
        for (idx, sh, relu, pool, drop) in self.layer_specs:
            conv = _conv_of(self.model.features[idx])
            fkey = (conv.weight.data_ptr(), conv.weight._version, conv.bias._version, getattr(conv.weight, '_witw_version', 0),
                    getattr(conv.bias, '_witw_version', 0))
            hit = self._packed.get(('bf16', idx))
            if (hit is None or hit[0] != fkey) and not (idx == 0 and self.in_channels <= 8):     # layer 0 runs on the first-layer kernels: _pack_first
                todo.append((('bf16', idx), fkey, (conv.weight, conv.bias, False, hit[1] if hit else None)))
            if idx > first:
                tkey = (conv.weight.data_ptr(), conv.weight._version, getattr(conv.weight, '_witw_version', 0))
                hit = self._packed.get(('t_bf16', idx))
                if hit is None or hit[0] != tkey:
                    todo.append((('t_bf16', idx), tkey, (conv.weight, None, True, hit[1] if hit else None)))
        if len(todo) > 1:
            for (slot, key, _item), pk in zip(todo, ops.PackedConvBf16.batch([t[2] for t in todo])):
                self._packed[slot] = (key, pk)

    def _run_bf16(self, x, scales, keep_from=None):
        """The layer stack on the bf16 MFMA kernels (bf16 NHWC activations, fp32 accumulate, fp32 NCHW embedding).
        Returns (embedding, kept) with kept[idx] = (layer input, layer output, max-pool arg-max codes or None) bf16 NHWC
        for idx >= keep_from."""
        fast0 = self.in_channels <= 8 and (keep_from is None or keep_from > 0)      # the bf16 first-layer kernel takes up to 8 channels
        x_nchw = x.contiguous()
        h = x_nchw if fast0 else ops.nchw_to_nhwc_bf16(x_nchw, 16)
        sp0 = self.layer_specs[0]
        first_direct = (not fast0 and self.in_channels <= 8 and sp0[0] == 0 and sp0[1] == 1 and not sp0[3] and 0 not in scales
                        and keep_from is not None and keep_from <= 0)
        last = self.layer_specs[-1][0]
        kept = {}
        # layers 0 and 2 in one kernel, as forward_bf16 runs them, when neither is kept for a backward (cvig_fov: the frozen trunk,
        # model/cvig_fov.py:275-278 -- the backward starts at layer 17) nor carries a Dropout2d scale: the same bits as the two launches
        fuse = _lib.guards()['first2']['hand_scheduled_kernel'] if self.fuse_first2 is None else self.fuse_first2
        fused = (fast0 and fuse and (keep_from is None or keep_from > 2) and 0 not in scales and 2 not in scales and
                 self.layer_specs[0][:4] == (0, 1, True, False) and self.layer_specs[1][:4] == (2, 1, True, True))
        # ... and when the backward DOES cross both layers (layer 0 trains: cvig_semantic, model/cvig_semantic.py:301-309) the training
        # form of the same kernel: neither 64-channel activation is written -- the backward gets the max-pool's arg-max codes, layer
        # 2's gate is its pooled output, layer 0's gate one bit per output (67 MB instead of the 1.07 GB map at 128 images). Needs
        # the data gradient of layer 2 on the weight-resident kernel (the one that reads bit gates); smaller batches keep the two launches.
        B_, _C, H_, W_ = x_nchw.shape
        fused_train = (first_direct and fuse and keep_from == 0 and 2 not in scales and H_ % 2 == 0 and W_ % 2 == 0 and
                       self.layer_specs[0][:4] == (0, 1, True, False) and self.layer_specs[1][:4] == (2, 1, True, True) and
                       not _conv_of(self.model.features[2]).weight.requires_grad and ops.gatebits_dgrad_ok(B_, H_, W_, 64, 64))
        for (idx, sh, relu, pool, drop) in self.layer_specs:
            if idx == 0 and fused:
                h = ops.conv_first2_bf16(h, self._pack_first(True), self._pack_bf16(2), circular=self.circ_padding)
                continue
            if idx == 2 and (fused or fused_train):
                continue
            if idx == 0 and fused_train:
                y2, code2, bits0 = ops.conv_first2_bf16_train(x_nchw, self._pack_first(True), self._pack_bf16(2), circular=self.circ_padding)
                kept[0] = (h, _GateBits(bits0), None)                      # h: the NHWC bf16 image, what layer 0's weight gradient reads
                kept[2] = (_ShapeOnly((B_, H_, W_, 64)), y2, code2)         # layer 2 is frozen: nobody reads its input
                h = y2
                continue
            if idx == 0 and fast0:
                h = ops.conv3x3_first_fwd(h, self._pack_first(True), circular=self.circ_padding, relu=relu)
                continue
            keep = keep_from is not None and idx >= keep_from
            if idx == 0 and first_direct:
                # a TRAINABLE layer 0 (cvig_semantic, model/cvig_semantic.py:301-309): its forward still runs on the first-layer kernel
                # straight from the NCHW image (the generic kernel on the 16-channel NHWC copy is bound by that copy's pixel stride:
                # 628 against ~300 us at 128 images); the NHWC bf16 copy h is only what the weight gradient reads
                y = ops.conv3x3_first_fwd(x_nchw, self._pack_first(True), circular=self.circ_padding, relu=relu)
                kept[idx] = (h, y, None)
                h = y
                continue
            out = ops.conv3x3_bf16_fwd(h, self._pack_bf16(idx), stride_h=sh, circular=self.circ_padding, relu=relu, pool=pool,
                                       out_nchw_f32=(idx == last), drop_scale=scales.get(idx), want_pool_code=(keep and pool))
            y, code = out if (keep and pool) else (out, None)
            if keep:
                kept[idx] = (h, y, code)
            h = y
        return h, kept

    def forward_bf16(self, x):
        """Inference on the bf16 MFMA kernels (bf16 activations and filters, fp32 accumulate, fp32 embedding
        out): the 'bf16 MFMA' configuration of BASELINE.json. Not bit-comparable with the fp32 path — see
        tests/test_bf16_gpu.py for the stated tolerance. Eval only."""
        if not x.is_cuda:
            raise _lib.WitwError('FOV_DSM.forward_bf16 needs a GPU tensor (no CPU fallback)')
        if self.training:
            raise _lib.WitwError('forward_bf16 is an inference path; call .eval()')
        with torch.no_grad():
            fast0 = self.in_channels <= 8
            h = x.contiguous() if fast0 else ops.nchw_to_nhwc_bf16(x.contiguous(), 16)
            last = self.layer_specs[-1][0]
            # layers 0 and 2 in one kernel (the 64-channel map between them stays on the chip): csrc/conv_first2_bf16.hip
            fuse = _lib.guards()['first2']['hand_scheduled_kernel'] if self.fuse_first2 is None else self.fuse_first2
            fused = fast0 and fuse and self.layer_specs[0][:4] == (0, 1, True, False) and self.layer_specs[1][:4] == (2, 1, True, True)
            for (idx, sh, relu, pool, drop) in self.layer_specs:
                if idx == 0 and fused:
                    h = ops.conv_first2_bf16(h, self._pack_first(True), self._pack_bf16(2), circular=self.circ_padding)
                    continue
                if idx == 2 and fused:
                    continue
                if idx == 0 and fast0:
                    h = ops.conv3x3_first_fwd(h, self._pack_first(True), circular=self.circ_padding, relu=relu)
                    continue
                h = ops.conv3x3_bf16_fwd(h, self._pack_bf16(idx), stride_h=sh, circular=self.circ_padding, relu=relu,
                                         pool=pool, out_nchw_f32=(idx == last))
        return h

    def _pack_f16x3(self, idx, transpose_flip=False):
        conv = _conv_of(self.model.features[idx])
        key = (conv.weight.data_ptr(), conv.weight._version, conv.bias._version, getattr(conv.weight, '_witw_version', 0),
               getattr(conv.bias, '_witw_version', 0))
        slot = ('f16x3_t' if transpose_flip else 'f16x3', idx)
        hit = self._packed.get(slot)
        if hit is None or hit[0] != key:
            hit = (key, ops.PackedConvF16x3(conv.weight, None if transpose_flip else conv.bias, transpose_flip=transpose_flip,
                                            reuse=hit[1] if hit else None))
            self._packed[slot] = hit
        return hit[1]

    def _run_f16x3(self, x, scales, keep_from=None):
        """The layer stack on the fp16x3 kernels (split-fp16 activations, fp32-grade products, fp32 NCHW embedding).
        Returns (embedding, kept) with kept[idx] = (layer input, layer output, max-pool arg-max codes or None) in the split
        layout for idx >= keep_from."""
        fast0 = self.in_channels <= 4 and (keep_from is None or keep_from > 0)
        h = x.contiguous() if fast0 else ops.nchw_to_split_f16(x.contiguous(), 8)
        last = self.layer_specs[-1][0]
        kept = {}
        for (idx, sh, relu, pool, drop) in self.layer_specs:
            if idx == 0 and fast0:      # C<=4 -> 64 in exact fp32 straight from NCHW, output already split
                h = ops.conv3x3_first_fwd(h, self._pack_first(False), circular=self.circ_padding, relu=relu, split_f16=True)
                continue
            keep = keep_from is not None and idx >= keep_from
            out = ops.conv3x3_f16x3_fwd(h, self._pack_f16x3(idx), stride_h=sh, circular=self.circ_padding, relu=relu, pool=pool,
                                        out_nchw_f32=(idx == last), drop_scale=scales.get(idx), want_pool_code=(keep and pool))
            y, code = out if (keep and pool) else (out, None)
            if keep:
                kept[idx] = (h, y, code)
            h = y
        return h, kept

    def forward_f16x3(self, x):
        """Inference with fp32-grade accuracy on the fp16 MFMA (csrc/conv3x3_f16x3.hip): every activation and filter
        value is carried as fp16 hi + fp16 lo, products are hi*hi + lo*hi + hi*lo with fp32 accumulation. Held to the SAME
        reference goldens and tolerance (1e-4) as the fp32 path (tests/test_f16x3_gpu.py). Eval only."""
        if not x.is_cuda:
            raise _lib.WitwError('FOV_DSM.forward_f16x3 needs a GPU tensor (no CPU fallback)')
        if self.training:
            raise _lib.WitwError('forward_f16x3 is an inference path; call .eval()')
        with torch.no_grad():
            return self._run_f16x3(x, {})[0]

    def trainable_convs(self):
        return [(idx, _conv_of(self.model.features[idx])) for (idx, *_r) in self.layer_specs
                if _conv_of(self.model.features[idx]).weight.requires_grad]

    @classmethod
    def from_vgg16_state_dict(cls, state, circ_padding=False):
        """The encoder the reference builds from the pretrained VGG16 (model/cvig_fov.py:256-290): `state` is a
        torchvision-format VGG16 state_dict or a path to one (load_vgg16_state_dict)."""
        return load_vgg16_state_dict(cls(circ_padding=circ_padding), state)

    def forward(self, x, dropout_scales=None, relu_gates=None, pool_codes=None):
        """x [B,C,128,W] NCHW fp32 on the GPU -> [B,16,4,W/8] NCHW (reference :292-294).
        In train() mode Dropout2d scales are drawn per call unless `dropout_scales`
        ({17|19|21: [B,C]}) injects them. With grad enabled the call is recorded for autograd
        (weight / bias gradients of the trainable layers, computed by the HIP backward kernels).
        Parity hooks, like dropout_scales: `relu_gates` {layer idx: NHWC float tensor, > 0 where the ReLU behind that
        layer passes} and `pool_codes` {layer idx: uint8 arg-max codes of the fused max-pool} replace, in the BACKWARD only,
        the gates / routes this forward would record itself -- a pre-activation within rounding of zero (or a tie inside a
        pooling window) may fall the other way on the CPU reference, and the gradient is only piecewise continuous there."""
        if not x.is_cuda:
            raise _lib.WitwError('FOV_DSM.forward needs a GPU tensor (no CPU fallback)')
        scales = self._draw_scales(x, dropout_scales)
        tr = self.trainable_convs()
        if torch.is_grad_enabled() and tr:
            params = []
            for _i, c in tr:
                params += [c.weight, c.bias]
            fn = _EncoderFnBf16 if self.precision == 'bf16' else _EncoderFnF16x3 if self.precision == 'fp16x3' else _EncoderFn
            # the dicts are read when the backward runs, so a caller may fill them between forward and backward
            self._bwd_override = ({} if relu_gates is None else relu_gates, {} if pool_codes is None else pool_codes)
            bucket = getattr(self, '_grad_bucket', None)
            if bucket is not None:      # parallel.GradBucket: how many backward nodes of this encoder the step will run
                bucket.nodes += 1
            return fn.apply(x, self, scales, *params)
        if self.precision == 'fp16x3' and not self.training:
            return self.forward_f16x3(x)
        if self.precision == 'bf16':
            return self._run_bf16(x, scales)[0]
        return self._run(x, scales)[0]


class _EncoderFn(torch.autograd.Function):
    """autograd node of one FOV_DSM call: forward = the 13 fused conv launches; backward walks the layers from
    27 down to the first trainable one: per trainable layer one wgrad launch, per layer one dgrad launch (the
    forward kernel on the transposed, tap-rotated filter with the previous layer's ReLU / Dropout2d gate fused
    into its epilogue, zero-interleaved rows for the stride-(2,1) layers) and, behind a fused max-pool, the
    arg-max scatter. cvig_fov stops at layer 17; cvig_semantic (layer 0 trainable) goes all the way down."""

    @staticmethod
    def forward(ctx, x, enc, scales, *params):
        first = min(i for i, _c in enc.trainable_convs())
        out, kept = enc._run(x, scales, keep_from=first)
        ctx.enc, ctx.scales, ctx.kept, ctx.first = enc, scales, kept, first
        ctx.override = getattr(enc, '_bwd_override', ({}, {}))
        enc._bwd_override = ({}, {})
        enc._last_kept = kept if getattr(enc, 'keep_activations', False) else None      # diagnostics: gates of the last call
        return out

    @staticmethod
    def backward(ctx, grad_out):
        enc, scales, kept = ctx.enc, ctx.scales, ctx.kept
        specs = [sp for sp in enc.layer_specs if sp[0] >= ctx.first]
        circ = enc.circ_padding
        last = specs[-1][0]
        cout_last = _conv_of(enc.model.features[last]).out_channels
        dz = ops.nchw_to_nhwc(grad_out.contiguous(), (cout_last + 7) // 8 * 8)   # layer 27 has no ReLU
        grads = {}
        bucket = getattr(enc, '_grad_bucket', None)      # parallel.GradBucket: the wgrad kernels write into its views
        direct = bucket is not None and bucket.direct()
        for n in range(len(specs) - 1, -1, -1):
            idx, sh, relu, pool, drop = specs[n]
            x_in = kept[idx][0]
            conv = _conv_of(enc.model.features[idx])
            if conv.weight.requires_grad:
                out = (conv.weight._witw_grad_view, conv.bias._witw_grad_view) if direct else None
                grads[idx] = ops.conv3x3_wgrad(x_in, dz, conv.in_channels, stride_h=sh, circular=circ, out=out)
            if n > 0:   # gradient at the previous layer's conv output
                pidx, _psh, _prelu, ppool, _pdrop = specs[n - 1]
                p_in, p_out, p_code = kept[pidx]
                p_gate = ctx.override[0].get(pidx, p_out)
                p_code = ctx.override[1].get(pidx, p_code)
                dy = ops.conv3x3_fwd(dz, enc._pack_t(idx), stride_h=1, circular=circ, relu=False, pool=False,
                                     drop_scale=scales.get(pidx), gate=p_gate, dilate_h=(sh == 2),
                                     out_h=x_in.shape[1] if sh == 2 else None)
                dz = ops.maxpool2x2_bwd(dy, p_code, (p_in.shape[1], p_in.shape[2])) if ppool else dy
        ctx.kept = None
        if direct:      # the gradients already sit in the parameters' .grad views: nothing for autograd to accumulate
            bucket.notify()
            return (None, None, None) + (None,) * (2 * len(enc.trainable_convs()))
        flat = []
        for (idx, _c) in enc.trainable_convs():
            flat += [grads[idx][0], grads[idx][1]]
        return (None, None, None) + tuple(flat)


class _EncoderFnF16x3(torch.autograd.Function):
    """_EncoderFn with fp32-grade products on the fp16 MFMA (FOV_DSM.precision = 'fp16x3'): the forward (frozen trunk and
    trainable layers, Dropout2d scale in the epilogue) and every dgrad launch run on the fp16x3 kernels with split-fp16
    activations and activation gradients; the weight gradients come from witw_conv3x3_wgrad_f16x3 (the same split
    products over the batch-octet layout), in fp32. Parity: tests/test_f16x3_gpu.py against the fp32 path."""

    @staticmethod
    def forward(ctx, x, enc, scales, *params):
        first = min(i for i, _c in enc.trainable_convs())
        out, kept = enc._run_f16x3(x, scales, keep_from=first)
        ctx.enc, ctx.scales, ctx.kept, ctx.first = enc, scales, kept, first
        return out

    @staticmethod
    def backward(ctx, grad_out):
        enc, scales, kept = ctx.enc, ctx.scales, ctx.kept
        specs = [sp for sp in enc.layer_specs if sp[0] >= ctx.first]
        circ = enc.circ_padding
        last = specs[-1][0]
        cout_last = _conv_of(enc.model.features[last]).out_channels
        dz = ops.nchw_to_split_f16(grad_out.contiguous(), (cout_last + 7) // 8 * 8)      # layer 27 has no ReLU
        grads = {}
        for n in range(len(specs) - 1, -1, -1):
            idx, sh, relu, pool, drop = specs[n]
            x_in = kept[idx][0]
            conv = _conv_of(enc.model.features[idx])
            if conv.weight.requires_grad:
                dw, db = ops.conv3x3_wgrad_f16x3(x_in, dz, conv.in_channels, stride_h=sh, circular=circ)
                grads[idx] = (dw[:conv.out_channels].contiguous(), db[:conv.out_channels].contiguous())
            if n > 0:   # gradient at the previous layer's conv output
                pidx, _psh, _prelu, ppool, _pdrop = specs[n - 1]
                p_in, p_out, p_code = kept[pidx]
                dy = ops.conv3x3_f16x3_fwd(dz, enc._pack_f16x3(idx, transpose_flip=True), stride_h=1, circular=circ, relu=False,
                                           pool=False, drop_scale=scales.get(pidx), gate=p_out, dilate_h=(sh == 2),
                                           out_h=x_in.shape[1] if sh == 2 else None)
                dz = ops.maxpool2x2_bwd_split(dy, p_code, (p_in.shape[1], p_in.shape[2])) if ppool else dy
        ctx.kept = None
        flat = []
        for (idx, _c) in enc.trainable_convs():
            flat += [grads[idx][0], grads[idx][1]]
        return (None, None, None) + tuple(flat)


class _GateBits(object):
    """A ReLU gate kept as one bit per output (ops.conv_first2_bf16_train) in the place of the activation tensor"""
    __slots__ = ('bits',)

    def __init__(self, bits):
        self.bits = bits


class _ShapeOnly(object):
    """Stands for a kept activation of which the backward only asks the shape"""
    __slots__ = ('shape',)

    def __init__(self, shape):
        self.shape = tuple(shape)


class _EncoderFnBf16(torch.autograd.Function):
    """_EncoderFn on the bf16 MFMA kernels (FOV_DSM.precision = 'bf16'): mixed-precision training step with bf16
    activations / filters / activation gradients, fp32 accumulation, fp32 weight gradients (the fp32 master weights and
    Adam are untouched). Per trainable layer one witw_conv3x3_wgrad_bf16 (operands re-laid out to the batch-octet
    layout), per layer one dgrad launch = the bf16 forward kernel on the transposed, tap-rotated filter with the
    ReLU / Dropout2d gate in its epilogue. No reference counterpart (the reference trains in fp32 only): parity is
    stated against the fp32 HIP path in tests/test_bf16_train_gpu.py."""

    @staticmethod
    def forward(ctx, x, enc, scales, *params):
        first = min(i for i, _c in enc.trainable_convs())
        enc._refresh_packed_bf16(first)
        out, kept = enc._run_bf16(x, scales, keep_from=first)
        ctx.enc, ctx.scales, ctx.kept, ctx.first = enc, scales, kept, first
        return out

    @staticmethod
    def backward(ctx, grad_out):
        enc, scales, kept = ctx.enc, ctx.scales, ctx.kept
        specs = [sp for sp in enc.layer_specs if sp[0] >= ctx.first]
        circ = enc.circ_padding
        last = specs[-1][0]
        cout_last = _conv_of(enc.model.features[last]).out_channels
        dz = ops.nchw_to_nhwc_bf16(grad_out.contiguous(), (cout_last + 15) // 16 * 16)   # layer 27 has no ReLU
        grads = {}
        bucket = getattr(enc, '_grad_bucket', None)      # parallel.GradBucket: the wgrad kernels write into its views (as _EncoderFn)
        # ... when no layer's gradient tensor carries padded output channels (bf16 activations are stored in multiples of 16)
        direct = bucket is not None and bucket.direct() and all(c.out_channels % 16 == 0 for _i, c in enc.trainable_convs())
        for n in range(len(specs) - 1, -1, -1):
            idx, sh, relu, pool, drop = specs[n]
            x_in = kept[idx][0]
            conv = _conv_of(enc.model.features[idx])
            if conv.weight.requires_grad:
                if direct:
                    grads[idx] = ops.conv3x3_wgrad_bf16(x_in, dz, conv.in_channels, stride_h=sh, circular=circ,
                                                        out=(conv.weight._witw_grad_view, conv.bias._witw_grad_view))
                else:
                    dw, db = ops.conv3x3_wgrad_bf16(x_in, dz, conv.in_channels, stride_h=sh, circular=circ)
                    grads[idx] = (dw[:conv.out_channels].contiguous(), db[:conv.out_channels].contiguous())
            if n > 0:   # gradient at the previous layer's conv output
                pidx, _psh, _prelu, ppool, _pdrop = specs[n - 1]
                p_in, p_out, p_code = kept[pidx]
                if isinstance(p_out, _GateBits):      # the fused first-two-layers forward kept this gate as bits
                    dy = ops.conv3x3_bf16_dgrad_gatebits(dz, enc._pack_t_bf16(idx), p_out.bits, circular=circ)
                else:
                    dy = ops.conv3x3_bf16_fwd(dz, enc._pack_t_bf16(idx), stride_h=1, circular=circ, relu=False, pool=False,
                                              drop_scale=scales.get(pidx), gate=p_out, dilate_h=(sh == 2),
                                              out_h=x_in.shape[1] if sh == 2 else None)
                dz = ops.maxpool2x2_bwd_bf16(dy, p_code, (p_in.shape[1], p_in.shape[2])) if ppool else dy
        ctx.kept = None
        if direct:      # the gradients already sit in the parameters' .grad views: nothing for autograd to accumulate
            bucket.notify()
            return (None, None, None) + (None,) * (2 * len(enc.trainable_convs()))
        flat = []
        for (idx, _c) in enc.trainable_convs():
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


def inverse_normalize(tensor, mean, std):
    """model/cvig_fov.py:151-154, verbatim semantics: zip() walks the FIRST dimension of `tensor`, so on the [N,C,H,W] batch
    the reference passes (:477, :536) slices 0..len(mean)-1 along N are rescaled in place, each as a whole — kept as is,
    since the result only labels TensorBoard's embedding projector."""
    for t, m, s in zip(tensor, mean, std):
        t.mul_(s).add_(m)
    return tensor


def bilinear_interpolate(im, x, y):
    """model/cvig_fov.py:156-183 on the GPU (ops.bilinear_interpolate): im [C,H,W], x / y coordinate arrays."""
    return ops.bilinear_interpolate(_to_device(im), x, y)


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
    eps=1e-8, no weight decay), one HIP launch per 48 parameter tensors (witw_adam_step_multi). Parameters that never
    receive a gradient (the frozen layers) are skipped, as torch does."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.state = {}

    def zero_grad(self):
        seen = set()
        for p in self.params:
            b = getattr(p, '_witw_bucket', None)      # parallel.GradBucket: .grad stays a view into the flat buffer
            if b is None:
                p.grad = None
            elif id(b) not in seen:
                seen.add(id(b))
                b.zero()

    def step(self):
        live, grads = [], []
        for p in self.params:
            if p.grad is None:
                continue
            b = getattr(p, '_witw_bucket', None)
            if b is not None and not b.received(p):      # its .grad view was only cleared: no gradient this step
                continue
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = {'step': 0, 'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p)}
            st['step'] += 1
            live.append(p)
            grads.append(p.grad.contiguous())
        if not live:
            return
        with torch.no_grad():
            states = [self.state[p] for p in live]
            ops.adam_step_multi([p.data for p in live], grads, [s['exp_avg'] for s in states], [s['exp_avg_sq'] for s in states],
                                [s['step'] for s in states], self.lr, self.betas[0], self.betas[1], self.eps)
            for p in live:
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


SPECTRAL_FROM = 8192      # evaluation sets from this many pairs on rank through the spectral pass under match_method 'auto'


def evaluation_ranks(overhead_embed, surface_embed, shard_begin=0, world=1, method='auto'):
    """The ranks test() tabulates (model/cvig_fov.py:543-552), int64 [N] on the host, identical on every rank. `overhead_embed` /
    `surface_embed` are THIS rank's rows (world > 1: queries are replicated, gallery rows stay sharded). method:
    'direct' = the fused correlation kernel on every (gallery, query) pair (2*64*E FLOP each); 'dft' = the spectral pass
    (21 k FLOP per pair) with the exact re-scoring of retrieve(): the SAME ranks (index-exact, tests/test_match_dft_gpu.py) about
    17x faster at retrieval sizes; 'auto' = 'dft' from SPECTRAL_FROM pairs on. 8,884 CVUSA test pairs: 0.55 s direct, 10^5 pairs:
    36 s direct / 2 s spectral."""
    from . import parallel
    if method not in ('auto', 'direct', 'dft'):
        raise _lib.WitwError("match_method must be 'auto', 'direct' or 'dft', got %r" % (method,))
    surface_all = parallel.all_gather_ragged(surface_embed) if world > 1 else surface_embed
    if method == 'auto':
        method = 'dft' if surface_all.shape[0] >= SPECTRAL_FROM else 'direct'
    if method == 'dft':
        return retrieve(overhead_embed, surface_all, k=1, shard_begin=shard_begin, method='dft')[0]
    if world > 1:
        return sharded_ranks(overhead_embed, surface_all, shard_begin)
    return ranks(overhead_embed, surface_embed)


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
        if n_g == 0:         # a rank without gallery rows still takes part in the exchanges
            parallel.all_reduce_sum_(torch.zeros((q1 - q0,), dtype=torch.float32, device=surface_all.device))
            continue
        _, dist = _match(overhead_shard.contiguous(), surface_all[q0:q1].contiguous())   # [n_g, q]
        qi = torch.arange(q0, q1, device=dist.device)
        own = (qi >= shard_begin) & (qi < shard_begin + n_g)
        row = (qi - shard_begin).clamp(0, n_g - 1)
        d_true = torch.where(own, dist[row, qi - q0], torch.zeros_like(dist[0]))
        parallel.all_reduce_sum_(d_true)
        out[q0:q1] = _count(dist, d_true.contiguous())
    parallel.all_reduce_sum_(out)
    return out.cpu().numpy().astype('int64')


def retrieve_topk(overhead_shard, surface_all, k=10, shard_begin=0, query_chunk=4096, method='direct', _kernels=None):
    """Top-k retrieval (BASELINE config C5): for every query the k nearest gallery rows by the fused
    orientation-search chord distance, ordered by (distance, gallery index). With the gallery sharded over
    ranks each rank ranks its shard, the [N,k] candidate lists are all-gathered and merged by the same kernel.
    -> (distances f32 [N,k], gallery indices int64 [N,k]) on the device, identical on every rank."""
    return retrieve(overhead_shard, surface_all, k, shard_begin, query_chunk, method, _kernels, _want_ranks=False)[1:]


def _merge_topk(v, i, k, _kernels=None):
    """All-gather per-shard [N,k] candidate lists and merge them on (distance, global gallery index)."""
    from . import parallel
    kn = _kernels or ops
    w = parallel.world()
    n_q = v.shape[0]
    v_all = parallel._all_gather_cat(v.unsqueeze(0))            # [w, N, k]
    i_all = parallel._all_gather_cat(i.unsqueeze(0))
    cand_v = v_all.permute(0, 2, 1).reshape(w * k, n_q).contiguous()   # candidates as a [w*k, N] "distance matrix"
    cand_i = i_all.permute(0, 2, 1).reshape(w * k, n_q)
    cand_v = torch.where(cand_i < 0, torch.full_like(cand_v, float('inf')), cand_v)
    # order candidate rows by global index first so that the kernel's row-number tie-break equals the
    # gallery-index tie-break
    order = torch.argsort(torch.where(cand_i < 0, torch.full_like(cand_i, 2 ** 62), cand_i), dim=0, stable=True)
    cand_v = torch.gather(cand_v, 0, order).contiguous()
    cand_i = torch.gather(cand_i, 0, order)
    v, pos = kn.topk_smallest(cand_v, k)
    return v, torch.gather(cand_i.t(), 1, pos.clamp(min=0))


def retrieve(overhead_shard, surface_all, k=10, shard_begin=0, query_chunk=4096, method='direct', _kernels=None,
             _want_ranks=True):
    """sharded_ranks + retrieve_topk from ONE matching pass per query chunk (config C5: the pass is
    2*64*E FLOP per (gallery row, query) and dominates). -> (ranks int64 [N] on the host, distances f32 [N,k],
    gallery indices int64 [N,k] on the device), identical on every rank. The gallery may be sharded raggedly (any
    number of rows per rank, including none); `_kernels` swaps the op set (CPU tests of the collective algebra).
    method='dft': the pass runs through the row spectra (21k instead of 524k FLOP per pair) and every decision that the
    spectral distances leave within fp32 rounding -- a row within DISTANCE_EPS of a query's true-match distance, neighbours
    in a top-k list closer than 2 DISTANCE_EPS -- is re-made on distances from ops.match_pairs, which are bit-identical to the
    direct kernel's: ranks and top-k INDICES equal method='direct' exactly (the listed distances agree to DISTANCE_EPS)."""
    from . import parallel
    kn = _kernels or ops
    if method == 'dft' and k + DFT_MARGIN > 32:      # no room for the candidate margin in a 32-wide list: the top-k comes from the
        ranks_dft = None                              # direct pass, the rank counts (no list involved) still from the spectral one
        if _want_ranks:
            ranks_dft = _retrieve_dft(overhead_shard, surface_all, 1, shard_begin, query_chunk, kn, True)[0]
        _r, v, i = retrieve(overhead_shard, surface_all, k, shard_begin, query_chunk, 'direct', _kernels, _want_ranks=False)
        return ranks_dft, v, i
    if method == 'dft':
        return _retrieve_dft(overhead_shard, surface_all, k, shard_begin, query_chunk, kn, _want_ranks)
    if method != 'direct':
        raise ValueError("retrieve: method must be 'direct' or 'dft'")
    n_q, n_g = surface_all.shape[0], overhead_shard.shape[0]
    counts = torch.zeros((n_q,), dtype=torch.int32, device=surface_all.device)
    vals, idxs = [], []
    gallery = overhead_shard.contiguous()
    for q0 in range(0, n_q, query_chunk):
        q1 = min(n_q, q0 + query_chunk)
        nq = q1 - q0
        if n_g == 0:        # a rank without gallery rows (more ranks than rows): nothing to match, empty candidate lists
            dist = torch.empty((0, nq), dtype=torch.float32, device=surface_all.device)
        else:
            _, dist = kn.match_fwd(gallery, surface_all[q0:q1].contiguous())            # [n_g, q]
        if _want_ranks:
            qi = torch.arange(q0, q1, device=dist.device)
            own = (qi >= shard_begin) & (qi < shard_begin + n_g)
            if n_g:
                row = (qi - shard_begin).clamp(0, n_g - 1)
                d_true = torch.where(own, dist[row, qi - q0], torch.zeros_like(dist[0]))
            else:
                d_true = torch.zeros((nq,), dtype=torch.float32, device=dist.device)
            parallel.all_reduce_sum_(d_true)
            if n_g:
                counts[q0:q1] = kn.rank_count_thresh(dist, d_true.contiguous())
        if n_g:
            v, i = kn.topk_smallest(dist, k, shard_begin)
        else:
            v = torch.full((nq, k), float('inf'), dtype=torch.float32, device=dist.device)
            i = torch.full((nq, k), -1, dtype=torch.int64, device=dist.device)
        vals.append(v)
        idxs.append(i)
    if _want_ranks:
        parallel.all_reduce_sum_(counts)
    v, i = torch.cat(vals), torch.cat(idxs)
    if parallel.world() > 1:
        v, i = _merge_topk(v, i, k, _kernels)
    return (counts.cpu().numpy().astype('int64') if _want_ranks else None), v, i


def _sort_by_value_then_index(v, i):
    """rows of (v, i) ordered by (value, index) ascending; missing candidates (index < 0) last."""
    big = torch.where(i < 0, torch.full_like(i, 2 ** 62), i)
    o1 = torch.argsort(big, dim=1, stable=True)
    v, i = torch.gather(v, 1, o1), torch.gather(i, 1, o1)
    o2 = torch.argsort(v, dim=1, stable=True)
    return torch.gather(v, 1, o2), torch.gather(i, 1, o2)


_RETRIEVE_TLS = __import__('threading').local()


def last_retrieve_stats():
    """Re-scoring statistics of the calling thread's last retrieve(method='dft') (retrieve.last_stats is process-wide)."""
    return dict(getattr(_RETRIEVE_TLS, 'stats', None) or {})


DFT_MARGIN = 6      # candidates kept beyond place k by the spectral top-k (place k+1 must exist to decide place k)


def _retrieve_dft(overhead_shard, surface_all, k, shard_begin, query_chunk, kn, want_ranks):
    """retrieve() on the spectral pass, index-exact (see retrieve). eps = ops.DISTANCE_EPS bounds |d_dft - d_direct| at full
    width; narrower surfaces (We < 64: the window norm, hence the distance, depends on the chosen shift) first have every pair
    whose two best spectral scores are within rounding re-scored, so that the same bound holds for what is left."""
    from . import parallel
    eps = float(getattr(kn, 'DISTANCE_EPS', ops.DISTANCE_EPS))
    dev = surface_all.device
    n_q, n_g, we = surface_all.shape[0], overhead_shard.shape[0], surface_all.shape[3]
    kc = k + DFT_MARGIN                                  # local candidates per query, by spectral distance (retrieve: <= 32)
    assert kc <= 32
    gallery = overhead_shard.contiguous()
    spec_g = kn.match_spectrum(gallery, overhead=True) if n_g else None
    counts = torch.zeros((n_q,), dtype=torch.int32, device=dev)
    vals, idxs, sns = [], [], []
    band_checks = []
    wn = None
    stats = {'method': 'dft', 'pairs': float(n_g) * n_q, 'rescored_rank': 0, 'rescored_topk': 0, 'rescored_true': 0,
             'rescored_orientation': 0, 'fallback_queries': 0}
    for q0 in range(0, n_q, query_chunk):
        q1 = min(n_q, q0 + query_chunk)
        nq = q1 - q0
        su = surface_all[q0:q1].contiguous()
        if n_g:
            if we < 64:
                dist, ws, n_fix = _dft_pass_narrow(kn, gallery, su, spec_g)
                stats['rescored_orientation'] += n_fix
            else:
                _, dist, ws = kn.match_fwd_dft(gallery, su, spec_ov=spec_g, want_orientation=False, want_workspace=True)
            wn, sn = ws[:n_g * 64], ws[n_g * 64:n_g * 64 + nq]
        else:
            dist = torch.empty((0, nq), dtype=torch.float32, device=dev)
            sn = torch.zeros((nq,), dtype=torch.float32, device=dev)
        sns.append(sn)
        if q0 == 0 and we < 64:
            # with the shift settled, d = 2 (1 - score / (|window| |su|)): a score known to 2 SCORE_ROUNDING |ov| |su| gives a
            # distance known to 4 SCORE_ROUNDING |ov| / |window| -- take the gallery's worst window (the same on every rank)
            ratio = torch.zeros((1,), dtype=torch.float32, device=dev)
            if n_g:
                wmin = wn.reshape(n_g, 64).min(dim=1).values
                ratio = (gallery.reshape(n_g, -1).norm(dim=1) / wmin).max().reshape(1)
            if parallel.world() > 1:
                import torch.distributed as dist_
                dist_.all_reduce(ratio, op=dist_.ReduceOp.MAX)
            rounding = float(getattr(kn, 'SCORE_ROUNDING', ops.SCORE_ROUNDING))
            eps = max(eps, 1.25 * 4 * rounding * float(ratio.item()))
        if want_ranks:
            # the queries of this chunk whose true row lives in this shard: a contiguous range, known on the host (no mask, no
            # device round trip)
            lo, hi = max(q0, shard_begin), min(q1, shard_begin + n_g)
            d_true = torch.zeros((nq,), dtype=torch.float32, device=dev)
            if n_g and hi > lo:                          # the owner's EXACT distance of every true pair
                po = torch.arange(lo - shard_begin, hi - shard_begin, dtype=torch.int32, device=dev)
                ps = torch.arange(lo - q0, hi - q0, dtype=torch.int32, device=dev)
                d_true[lo - q0:hi - q0] = kn.match_pairs(gallery, su, wn, sn, po, ps, want_orientation=False)[1]
                stats['rescored_true'] += hi - lo
            parallel.all_reduce_sum_(d_true)
            if n_g:
                d_true = d_true.contiguous()
                if hasattr(kn, 'rank_count_resolved'):
                    # band list, exact re-scoring and the count update in one stream sequence; the list's length stays on the device
                    # and is looked at once, behind the last chunk (band_checks)
                    c, n_band, cap = kn.rank_count_resolved(dist, d_true, eps, gallery, su, wn, sn)
                    band_checks.append((n_band, cap, q0, q1, dist if len(range(0, n_q, query_chunk)) == 1 else None, su, d_true, sn))
                else:
                    c, po, ps = kn.rank_count_band(dist, d_true, eps)
                    if po.numel():
                        d_x = kn.match_pairs(gallery, su, wn, sn, po, ps, want_orientation=False)[1]
                        c.index_add_(0, ps.long(), (d_x <= d_true[ps.long()]).to(torch.int32))
                        stats['rescored_rank'] += int(po.numel())
                counts[q0:q1] = c
        if n_g:
            v, i = kn.topk_smallest(dist, kc, shard_begin)
        else:
            v = torch.full((nq, kc), float('inf'), dtype=torch.float32, device=dev)
            i = torch.full((nq, kc), -1, dtype=torch.int64, device=dev)
        vals.append(v)
        idxs.append(i)
    if band_checks:
        # one look at the band lists' lengths for the whole pass; a list that overflowed (more pairs within eps of a threshold than
        # 1/4096 of the chunk: not seen on real or synthetic data) sends its chunk through the two-step form again
        n_host = torch.cat([b[0] for b in band_checks]).tolist()
        for got, (_n, cap, q0, q1, dist_kept, su, d_true, sn) in zip(n_host, band_checks):
            stats['rescored_rank'] += min(int(got), cap)
            if got > cap:
                dist_c = dist_kept
                if dist_c is None:
                    dist_c = (_dft_pass_narrow(kn, gallery, su, spec_g)[0] if we < 64 else
                              kn.match_fwd_dft(gallery, su, spec_ov=spec_g, want_orientation=False)[1])
                c, po, ps = kn.rank_count_band(dist_c, d_true, eps)
                d_x = kn.match_pairs(gallery, su, wn, sn, po, ps, want_orientation=False)[1]
                c.index_add_(0, ps.long(), (d_x <= d_true[ps.long()]).to(torch.int32))
                counts[q0:q1] = c
        band_checks = []
    if want_ranks:
        parallel.all_reduce_sum_(counts)
    v, i, sn_all = torch.cat(vals), torch.cat(idxs), torch.cat(sns)
    # ---- candidates of all shards, ordered by (spectral distance, index)
    w = parallel.world()
    outsider = v[:, kc - 1].clone()                      # no row outside a shard's list has a smaller spectral distance
    if w > 1:
        v = parallel._all_gather_cat(v.unsqueeze(0)).permute(1, 0, 2).reshape(n_q, w * kc)
        i = parallel._all_gather_cat(i.unsqueeze(0)).permute(1, 0, 2).reshape(n_q, w * kc)
        outsider = parallel._all_gather_cat(outsider.unsqueeze(0)).min(dim=0).values
    v = torch.where(i < 0, torch.full_like(v, float('inf')), v)
    v, i = _sort_by_value_then_index(v.contiguous(), i.contiguous())
    ncand = v.shape[1]
    m = min(ncand, k + DFT_MARGIN)                       # candidates re-scored for an undecided query
    inf_col = torch.full((n_q, 1), float('inf'), dtype=torch.float32, device=dev)
    vp = torch.cat((v, inf_col), dim=1)
    gaps = vp[:, 1:k + 1] - vp[:, :k]                    # between places 1..k+1
    undecided = ((gaps <= 2 * eps) & torch.isfinite(vp[:, :k])).any(dim=1)
    # a row outside every shard's list (spectral distance >= outsider) must not come within rounding of place k either
    undecided |= torch.isfinite(vp[:, k - 1]) & (vp[:, k - 1] + 2 * eps >= outsider) & torch.isfinite(outsider)
    rows = torch.nonzero(undecided).squeeze(1)           # identical on every rank: computed from gathered data
    fallback = torch.zeros((0,), dtype=torch.int64, device=dev)
    if rows.numel():
        ci = i[rows, :m]                                                     # [r, m] global gallery rows
        mine = (ci >= shard_begin) & (ci < shard_begin + n_g)
        exact = torch.zeros(ci.shape, dtype=torch.float32, device=dev)
        if n_g and bool(mine.any()):
            po = (ci - shard_begin)[mine].to(torch.int32).contiguous()
            ps = rows[:, None].expand(-1, m)[mine].to(torch.int32).contiguous()
            exact[mine] = kn.match_pairs(gallery, surface_all.contiguous(), wn, sn_all, po, ps, want_orientation=False)[1]
            stats['rescored_topk'] += int(po.numel())
        parallel.all_reduce_sum_(exact)
        exact = torch.where(ci < 0, torch.full_like(exact, float('inf')), exact)
        ev, ei = _sort_by_value_then_index(exact, ci)
        # nothing outside the re-scored set can belong to the first k: its direct distance is >= its spectral one - eps
        beyond = torch.minimum(outsider[rows], vp[rows, m]) - eps
        kth = ev[:, min(k, m) - 1]
        safe = (kth < beyond) | ~torch.isfinite(beyond)
        v[rows, :m], i[rows, :m] = ev, ei
        fallback = rows[~safe]
    if fallback.numel():     # more near-ties than candidates kept (not seen on real data): those queries take the direct pass
        stats['fallback_queries'] = int(fallback.numel())
        _r, fv, fi = retrieve(overhead_shard, surface_all[fallback].contiguous(), k, shard_begin, query_chunk, 'direct',
                              None if kn is ops else kn, _want_ranks=False)
        v[fallback, :k], i[fallback, :k] = fv, fi
    stats['eps'] = eps
    retrieve.last_stats = stats
    _RETRIEVE_TLS.stats = stats
    return (counts.cpu().numpy().astype('int64') if want_ranks else None), v[:, :k].contiguous(), i[:, :k].contiguous()


def _dft_pass_narrow(kn, gallery, su, spec_g):
    """The spectral pass for surfaces narrower than the overhead embedding (We < 64): the window norm depends on the shift, so
    a pair whose two best scores are within rounding could take the other shift in the direct kernel and land on a different
    distance. Those pairs (top-2 score gap <= 4 SCORE_ROUNDING |ov| |su|) are re-scored exactly and patched into the matrix.
    -> (distance [n_g, nq], workspace, pairs re-scored)."""
    n_g, nq = gallery.shape[0], su.shape[0]
    _, dist, gap, ws = kn.match_fwd_dft(gallery, su, spec_ov=spec_g, want_orientation=False, want_workspace=True, want_gap=True)
    wn, sn = ws[:n_g * 64], ws[n_g * 64:n_g * 64 + nq]
    ov_norm = gallery.reshape(n_g, -1).norm(dim=1)
    rounding = float(getattr(kn, 'SCORE_ROUNDING', ops.SCORE_ROUNDING))
    close = gap <= (4 * rounding) * ov_norm[:, None] * sn[None, :]
    pairs = torch.nonzero(close)
    if pairs.numel():
        po, ps = pairs[:, 0].to(torch.int32).contiguous(), pairs[:, 1].to(torch.int32).contiguous()
        dist[pairs[:, 0], pairs[:, 1]] = kn.match_pairs(gallery, su, wn, sn, po, ps, want_orientation=False)[1]
    return dist, ws, int(pairs.shape[0])


class _ShardedMatchLossFn(torch.autograd.Function):
    """Global-batch match + soft-margin triplet loss with the distance matrix sharded by COLUMNS over the ranks:
    rank r evaluates all B overheads against its own b surfaces ([B,b] slab; the full [B,B] matrix is never built on
    one GPU). Exchanges: forward = all-gather of the overhead embeddings and of the diagonal (B floats), all-reduce
    of the loss partial; backward = all-reduce of the row sigmoid sums (B floats) and a reduce-scatter of the
    overhead-embedding gradients (every rank holds the part that flows through ITS surfaces). The surface gradients
    are complete locally. Same value and gradients as match + triplet_loss on the gathered batch."""

    @staticmethod
    def forward(ctx, overhead_local, surface_local, alpha, k):
        from . import parallel
        b = surface_local.shape[0]
        col0 = parallel.rank() * b
        with parallel.phase('overhead_all_gather'):
            ov_all = parallel._all_gather_cat(overhead_local.contiguous())
        su = surface_local.contiguous()
        with parallel.phase('slab_match'):
            ori, dist, score, ws = k.match_fwd(ov_all, su, want_score=True, want_workspace=True)
        B = ov_all.shape[0]
        with parallel.phase('diagonal_all_gather'):
            diag = parallel._all_gather_cat(dist[col0:col0 + b].diagonal().contiguous())
        with parallel.phase('loss_partial_all_reduce'):
            part = k.triplet_loss_slab_fwd(dist, diag, col0, alpha)
            parallel.all_reduce_sum_(part)
        ctx.save_for_backward(ov_all, su, ori, score, ws, dist, diag)
        ctx.cfg = (col0, b, float(alpha), k)
        ctx.mark_non_differentiable(ori, dist)
        return (part / (2. * B * (B - 1))).reshape(()), ori, dist

    @staticmethod
    def backward(ctx, g_loss, _g_ori, _g_dist):
        from . import parallel
        ov_all, su, ori, score, ws, dist, diag = ctx.saved_tensors
        col0, b, alpha, k = ctx.cfg
        with parallel.phase('row_sigmoid_all_reduce'):
            rowsig, colsig = k.triplet_loss_slab_sig(dist, diag, col0, alpha)
            parallel.all_reduce_sum_(rowsig)
        with parallel.phase('slab_match_backward'):
            g_dist = k.triplet_loss_slab_bwd(dist, diag, rowsig, colsig, g_loss.contiguous(), col0, alpha)
            gov_all, gsu = k.match_bwd(ov_all, su, ori, score, ws, g_dist, True, True)
        with parallel.phase('overhead_grad_reduce_scatter'):
            gov = parallel.reduce_scatter_rows(gov_all, b)
        return gov, gsu, None, None


def sharded_match_loss(overhead_local, surface_local, alpha=10., _kernels=None):
    """(loss, orientation [B,b], distance [B,b]) of the GLOBAL batch from this rank's b pairs; every rank must call it
    with the same b. On one rank it is match + triplet_loss. `_kernels` swaps the op set (CPU tests of the
    collective algebra)."""
    from . import parallel
    if parallel.world() == 1 and _kernels is None:
        ori, dist = match(overhead_local, surface_local)
        return triplet_loss(dist, alpha), ori, dist.detach()
    return _ShardedMatchLossFn.apply(overhead_local, surface_local, float(alpha), _kernels or ops)


class PairEmbedder(object):
    """The two encoders' EVAL forwards of one batch (model/cvig_fov.py:524-527 in test(), :447-449 in the validation phase), the
    way that serves the reference's own default batch sizes (64 in cvig_fov :385,490; 32 in cvig_semantic :416) well:

      * two streams: the surface and the overhead encoder share nothing, so up to `dual_max` pairs their launches are issued on
        two HIP streams forked from the current one and joined behind it -- a layer whose grid leaves CUs idle at that batch (the
        stride-(2,1) tail: 64 workgroups at 16 pairs; the first layers of a small batch) shares the chip with its twin;
      * one hipGraph: up to `graph_max` pairs the bf16 step is bound by its ~35 launches, not by their kernels; once a (shape,
        precision, weights) key has been seen twice its two forwards are captured (parallel.CapturedStep, both streams inside)
        and later batches of that key replay the graph. A weight update (validation after a training epoch) changes the key:
        the stale graph is dropped and a fresh one captured on the second batch.
    Thresholds as measured (whole eval step, same box; docs/experiments.md): two streams help up to 32 pairs with the bf16 encoders
    and up to 64 with the fp32 ones (+1.2 % at 64; at 64 bf16 pairs they cost 2 % inside a graph); the graph helps bf16 up to 64 pairs
    (+2.4 % at 64, without the second stream there), fp32 never (its kernels are long).

    Either way the kernels, their order within an encoder and therefore the BITS are those of calling the encoders one after
    the other (tests/test_fullsize_properties_gpu.py). Gradient-recording calls and training-mode encoders take the plain path."""

    def __init__(self, surface_encoder, overhead_encoder, graph_max=64, dual_max=None):
        """dual_max None: 32 pairs with bf16 encoders, 64 with fp32 ones (decided per call from the encoders' precision)"""
        self.se, self.oe = surface_encoder, overhead_encoder
        self.graph_max, self._dual_max = graph_max, dual_max
        self._streams = None
        self._seen = {}
        self._graphs = {}
        self._uncapturable = set()      # keys whose capture failed once: they stay on the eager path (same bits)
        self.stats = {'eager': 0, 'dual_stream': 0, 'graph_replay': 0, 'captures': 0, 'capture_failures': 0}

    def _key(self, surface, polar):
        sig = tuple((q.data_ptr(), q._version, getattr(q, '_witw_version', 0)) for enc in (self.se, self.oe) for q in enc.parameters())
        return (tuple(surface.shape), tuple(polar.shape), surface.dtype, self.se.precision, self.oe.precision, hash(sig))

    def _plain(self, surface, polar):
        return self.se(surface), self.oe(polar)

    def _dual(self, surface, polar):
        cur = torch.cuda.current_stream()
        if self._streams is None:
            self._streams = (torch.cuda.Stream(), torch.cuda.Stream())
        s1, s2 = self._streams
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            su = self.se(surface)
        with torch.cuda.stream(s2):
            ov = self.oe(polar)
        cur.wait_stream(s1)
        cur.wait_stream(s2)
        if not torch.cuda.is_current_stream_capturing():      # (a capture's pool is private to the graph)
            for t in (su, ov):      # allocated on a side stream, consumed on the current one: keep the allocator from re-using them early
                t.record_stream(cur)
            surface.record_stream(s1)
            polar.record_stream(s2)
        return su, ov

    def dual_for(self, B):
        """two streams at this batch? (the measured thresholds, see the class comment)"""
        bf16 = self.se.precision == 'bf16' and self.oe.precision == 'bf16'
        return B <= (self._dual_max if self._dual_max is not None else (32 if bf16 else 64))

    def __call__(self, surface, polar):
        from . import parallel
        B = surface.shape[0]
        plain = (torch.is_grad_enabled() or self.se.training or self.oe.training or not surface.is_cuda or B != polar.shape[0])
        bf16 = self.se.precision == 'bf16' and self.oe.precision == 'bf16'
        graph_max = self.graph_max if bf16 else 0
        dual = self.dual_for(B)
        if plain or (B > graph_max and not dual):
            self.stats['eager'] += 1
            return self._plain(surface, polar)
        body = self._dual if dual else self._plain
        if B <= graph_max:
            key = self._key(surface, polar)
            g = self._graphs.get(key)
            if g is None and self._seen.get(key, 0) >= 1 and key not in self._uncapturable:
                self._graphs = {k: v for k, v in self._graphs.items() if k[:5] != key[:5]}      # graphs of older weights of this shape
                # Other threads of a driver process make HIP calls meanwhile (ring.PinnedRing's reaper synchronises events, a
                # DataLoader pin_memory thread allocates): the capture is thread-local so that they cannot invalidate it, and a
                # capture that fails all the same costs this key its graph, not the evaluation -- the eager path gives the same bits
                try:
                    g = self._graphs[key] = parallel.CapturedStep(lambda a, b: body(a, b), [surface, polar], warmup=1,
                                                                  capture_error_mode='thread_local')
                    self.stats['captures'] += 1
                except Exception as e:      # noqa: BLE001 -- whatever the runtime raises for a broken capture
                    import warnings
                    warnings.warn('PairEmbedder: hipGraph capture failed (%s: %s); this shape stays on the eager path'
                                  % (type(e).__name__, str(e)[:200]))
                    self._graphs.pop(key, None)
                    self._uncapturable.add(key)
                    self.stats['capture_failures'] += 1
                    torch.cuda.synchronize()
                    g = None
            if g is not None:
                su, ov = g(surface, polar)
                self.stats['graph_replay'] += 1
                return su.clone(), ov.clone()       # the graph's static outputs are overwritten by the next replay
            self._seen = {key: self._seen.get(key, 0) + 1}
        self.stats['dual_stream' if dual else 'eager'] += 1
        return body(surface, polar)


def evaluate_global_batch(overhead_all, surface_local, col0, alpha=10.):
    """Inference-time similarity for a minibatch sharded over ranks (no gradients): this rank matches ALL
    overhead embeddings of the global batch against its OWN surfaces (column slab [B, b]), which is all that
    the rank counts of its queries and its share of the global-batch loss need; the only exchanges are the
    diagonal (B floats) and the loss partial. -> (loss scalar tensor, ranks int32 [b], orientation [B,b], distance [B,b])."""
    from . import parallel
    with parallel.phase('slab_match'):
        ori, dist = ops.match_fwd(overhead_all.contiguous(), surface_local.contiguous())
    b = surface_local.shape[0]
    B = overhead_all.shape[0]
    with parallel.phase('diagonal_all_gather'):
        diag_local = dist[col0:col0 + b].diagonal().contiguous()
        diag = parallel._all_gather_cat(diag_local) if parallel.world() > 1 else diag_local
    with parallel.phase('loss_partial_all_reduce'):
        part = ops.triplet_loss_slab_fwd(dist, diag, col0, alpha)
        parallel.all_reduce_sum_(part)
    loss = part / (2. * B * (B - 1))
    return loss.reshape(()), ops.rank_count(dist, col0), ori, dist


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


# ----------------------------------------------------------------------------- dataset + drivers
class ImagePairDataset(torch.utils.data.Dataset):
    """model/cvig_fov.py:54-97: pairs of images (one surface, one overhead) from a CSV. Returns CPU
    float32 CHW tensors {'idx','surface','overhead'}; `transform` (if any) is applied per sample — the
    drivers pass none here and run Resize/ImageNormalization/PolarTransform batched on the GPU instead
    (DataLoader worker processes must not touch the device)."""

    _with_idx = True

    @classmethod
    def _globals(cls):
        return Globals

    def __init__(self, dataset, csv_path, base_path=None, transform=None, raw=False):
        import os
        import pandas as pd
        self.raw = raw      # not in the reference: True = hand over the decoder's uint8 HWC arrays (collate_packed / GpuPreprocess
        #                     convert them on the GPU, exactly); 'jpeg' = hand over entropy-decoded JPEG files, the GPU does the
        #                     rest of the decode too (same bytes); False = the reference's float32 CHW tensors
        self.csv_path = csv_path
        self.base_path = base_path if base_path is not None else os.path.dirname(csv_path)
        self.transform = transform
        path_format = self._globals().path_formats[dataset]
        file_paths = pd.read_csv(self.csv_path, header=path_format['header'], names=path_format['path_names'],
                                 usecols=path_format['path_columns'])
        self.file_paths = file_paths.map(
            lambda x: os.path.join(self.base_path, x) if isinstance(x, str) and len(x) > 0 and x[0] != '/' else x)

    def __len__(self):
        return len(self.file_paths)

    @staticmethod
    def _read(path):
        import numpy as np
        from PIL import Image
        a = np.asarray(Image.open(path))
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(a.astype(np.float32).transpose((2, 0, 1)).copy())

    @staticmethod
    def _read_raw(path):
        import numpy as np
        from PIL import Image
        a = np.asarray(Image.open(path))
        if a.ndim == 2:
            a = a[:, :, None]
        return a if a.dtype == np.uint8 else ImagePairDataset._read(path)

    @staticmethod
    def _read_jpeg(path):
        """raw='jpeg': a JPEG file is only ENTROPY-decoded here (witw_amd/jpeg.py: quantised DCT coefficient blocks); the GPU
        finishes it into the bytes Pillow would have produced. Other formats, and JPEG flavours the device path leaves alone,
        are decoded by Pillow as with raw=True."""
        from . import jpeg
        if str(path).lower().endswith(('.jpg', '.jpeg')):
            c = jpeg.open_file(path)      # header only; collate_packed entropy-decodes straight into the batch block
            if c is not None:
                return c
        return ImagePairDataset._read_raw(path)

    def __getitem__(self, idx):
        read = self._read_jpeg if self.raw == 'jpeg' else self._read_raw if self.raw else self._read
        data = {'surface': read(self.file_paths.iloc[idx]['surface']),
                'overhead': read(self.file_paths.iloc[idx]['overhead'])}
        if self._with_idx:
            data = dict(idx=idx, **data)
        if self.transform is not None:
            data = self.transform(data)
        return data


def collate_raw(samples):
    """Raw images differ in size: keep them as lists; GpuPreprocess batches them on the device."""
    out = {'surface': [s['surface'] for s in samples], 'overhead': [s['overhead'] for s in samples]}
    if samples and 'idx' in samples[0]:
        out['idx'] = [s['idx'] for s in samples]
    return out


def _pack_side(images, alloc=None):
    """A batch of differently sized images -> ONE contiguous byte buffer + [B,4] int64 {byte offset, H, W, channels stored per
    pixel}. All uint8 HWC (decoder output) -> kind 1, bytes as they are; anything else -> float32 planar CHW, kind 0.
    alloc: build the block in caller-provided memory (ring.PinnedRing.allocator); the first result is then (offset, nbytes),
    or the whole result None when the memory is too small."""
    import numpy as np
    from . import jpeg
    if any(isinstance(a, jpeg.JpegCoef) for a in images):      # entropy-decoded JPEG files (+ uint8 images of files left to Pillow)
        return jpeg.pack(images, shared=torch.utils.data.get_worker_info() is not None, alloc=alloc)
    raw = all(isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.ndim == 3 for a in images)
    parts, desc, off = [], [], 0
    for a in images:
        if raw:
            h, w, cs = a.shape
            flat = np.ascontiguousarray(a).reshape(-1)
        else:
            t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a, dtype=np.float32).transpose((2, 0, 1)))
            t = t.to(torch.float32).contiguous()
            cs, h, w = t.shape
            flat = t.reshape(-1).numpy().view(np.uint8)
        desc.append((off, h, w, cs))
        parts.append(flat)
        off += (flat.size + 15) // 16 * 16                 # images start 16-byte aligned
    # in a DataLoader worker the block is built in shared memory, where the parent reads it (no second copy when it is pickled);
    # not zero-filled: the few alignment bytes between images are never read
    if alloc is not None:
        got = alloc(off)
        if got is None:
            return None
        t, buf = (got[0], off), got[1]
    else:
        t = jpeg._shared_bytes(off) if torch.utils.data.get_worker_info() is not None else torch.empty((off,), dtype=torch.uint8)
        buf = t.numpy()
    for (o, _h, _w, _c), f in zip(desc, parts):
        buf[o:o + f.size] = f
    return t, torch.tensor(desc, dtype=torch.int64), 1 if raw else 0


def collate_packed(samples, ring=None):
    """collate_fn of the fast path (runs in the DataLoader worker): each side of the batch becomes one byte buffer + a
    descriptor table, so that the batch crosses the process boundary, the pinning thread and PCIe as two large blocks
    instead of 2 x B tensors. ring (ring.PinnedRing, bound with functools.partial): the two blocks are built in a slot of the
    parent's page-locked shared memory and only {slot, offsets} travel back -- no copy into shared memory, no pinning thread."""
    if ring is not None:
        slot = ring.acquire()
        try:
            alloc = ring.allocator(slot)
            s_side = _pack_side([s['surface'] for s in samples], alloc)
            o_side = _pack_side([s['overhead'] for s in samples], alloc) if s_side is not None else None
        except BaseException:
            ring.release(slot)      # an unreadable file must not cost the ring a slot for the rest of the epoch
            raise
        if o_side is not None:
            (s_off, s_len), sd, sk = s_side
            (o_off, o_len), od, ok = o_side
            out = {'ring': (slot, s_off, s_len, o_off, o_len), 'surface_desc': sd, 'surface_kind': sk, 'overhead_desc': od,
                   'overhead_kind': ok, 'packed': True}
            if samples and 'idx' in samples[0]:
                out['idx'] = [s['idx'] for s in samples]
            return out
        ring.release(slot)          # the batch does not fit a slot: the ordinary blocks
    sb, sd, sk = _pack_side([s['surface'] for s in samples])
    ob, od, ok = _pack_side([s['overhead'] for s in samples])
    out = {'surface_bytes': sb, 'surface_desc': sd, 'surface_kind': sk, 'overhead_bytes': ob, 'overhead_desc': od,
           'overhead_kind': ok, 'packed': True}
    if samples and 'idx' in samples[0]:
        out['idx'] = [s['idx'] for s in samples]
    return out


class StagedBatch(object):
    """A batch whose images sit in device memory with their descriptor tables built (GpuPreprocess.stage)."""

    def __init__(self):
        self.keep = []          # tensors the descriptors point into
        self.event = None       # recorded on the copy stream once everything has been queued there
        self.n = 0
        self.idx = None

    def wait(self):
        """Make the current stream wait for the staging copies and keep the allocator from recycling them early."""
        if self.event is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self.event)
            for t in self.keep:
                if t.is_cuda:
                    t.record_stream(cur)
        return self


class GpuPreprocess(object):
    """Compose[Resize, ImageNormalization, PolarTransform] (model/cvig_fov.py:393-397) over a batch of raw images, on the
    GPU: -> {'surface' [B,3,128,Ws], 'polar' [B,3,128,512]} in TWO launches whatever the batch size: one batched resize +
    normalise of the ground side and one fused resize + normalise + polar transform of the overhead side, each over a descriptor
    table of individually sized images (keep_overhead = True adds the reference's 'overhead' [B,3,256,256] entry and runs the
    overhead side as two launches; the drivers never read it).
    stage() moves a batch to the device (one copy per side from pinned memory when the batch is packed) and may run on a
    side stream ahead of time (DevicePrefetcher); __call__ accepts a raw batch or a staged one."""

    channels = 3
    normalization = None      # class used for the normalisation step (cvig_semantic overrides both)
    fused = True              # overhead side through witw_polar_from_raw (False: resize + polar transform as two launches)
    keep_overhead = False     # True: also return the reference's 'overhead' entry (the resized + normalised image; two launches)

    def __init__(self, dataset, fov=360, random_orientation=True, device=None, ring=None):
        self.resize = Resize(dataset, fov, random_orientation)
        self.norm = (self.normalization or ImageNormalization)()
        self.polar = PolarTransform()
        self.device = device      # where host images go (train() / test() pass their module's `device`); None = cvig_fov.device
        self.ring = ring          # ring.PinnedRing the loader's collate_packed builds its batch blocks in (None: ordinary blocks)

    def _dev(self):
        dev = self.device if self.device is not None else device
        if dev.type != 'cuda':
            raise _lib.WitwError('no gfx950 device: the WITW transforms run on the GPU only')
        return dev

    def _copy_side(self, st, dev, packed, pinned=False):
        """one packed block: pinned (unless it sits in the ring already) and queued for the device -> (dbuf, desc, kind, host block)"""
        buf, desc, kind = packed
        if not pinned and not buf.is_pinned():
            buf = buf.pin_memory()
        dbuf = buf.to(dev, non_blocking=True)
        st.keep += [dbuf] if pinned else [dbuf, buf]
        return dbuf, desc, kind, buf

    def _table_side(self, st, copied):
        """copied: the _copy_side results of ONE side of a batch (one per part) -> (descriptor table over all parts, image kind, finish).
        JPEG parts (jpeg.KIND_JPEG: file bytes or coefficient blocks -> [Huffman decoding,] dequantise, inverse DCT, upsample,
        colour-convert on the device) are decoded by launches that cover every part; finish (None or a callable) must run before the
        table is used: it reads the device decoder's damage flags and patches the rows of re-decoded files."""
        kinds = set(c[2] for c in copied)
        if kinds == {2}:
            from . import jpeg
            keep, table, finish = jpeg.decode_packed_multi([(dbuf, desc, buf) for dbuf, desc, _k, buf in copied], defer=True)
            st.keep += keep
            if int(table[:, 4].min()) < self.channels:
                raise _lib.WitwError('an image of the batch has %d channels, the model takes %d' % (int(table[:, 4].min()), self.channels))
            return table, 1, finish      # from here on: uint8 HWC images on the device
        if len(kinds) > 1:
            raise _lib.WitwError('the parts of a grouped batch hold different image kinds (fp32 CHW / u8 HWC)')
        tables = []
        for dbuf, desc, kind, _buf in copied:
            if int(desc[:, 3].min()) < self.channels:
                raise _lib.WitwError('an image of the batch has %d channels, the model takes %d' % (int(desc[:, 3].min()), self.channels))
            table = torch.empty((desc.shape[0], 5), dtype=torch.int64)
            table[:, 0] = desc[:, 0] + dbuf.data_ptr()
            table[:, 1:3] = desc[:, 1:3]
            table[:, 3] = 0
            table[:, 4] = desc[:, 3]
            tables.append(table)
        return (tables[0] if len(tables) == 1 else torch.cat(tables)), copied[0][2], None

    def _stage_side(self, st, dev, images):
        """a list of images (resident tensors, or host tensors / JpegCoef / JpegFile objects to pack here) -> (table, kind, finish)"""
        if all(isinstance(t, torch.Tensor) and t.is_cuda for t in images):        # already resident: point at them
            rows = []
            for t in images:
                t = t.to(torch.float32).contiguous()
                st.keep.append(t)
                rows.append((t.data_ptr(), t.shape[1], t.shape[2], 0, t.shape[0]))
            return torch.tensor(rows, dtype=torch.int64).to(dev, non_blocking=True), 0, None
        packed = _pack_side([t.cpu() if isinstance(t, torch.Tensor) else t for t in images])
        return self._table_side(st, [self._copy_side(st, dev, packed)])

    def stage(self, batch, starts=None):
        """batch: one raw / packed batch, or a LIST of them that become ONE staged batch (DevicePrefetcher(group=g): the loader
        hands out batch_size/g samples at a time so that the first full batch is ready after 1/g of the decode time; every
        part keeps its own device block, the descriptor tables hold absolute pointers and are simply concatenated)."""
        dev = self._dev()
        st = StagedBatch()
        parts = batch if isinstance(batch, (list, tuple)) else [batch]
        s_tabs, o_tabs, idx = [], [], []
        s_copied, o_copied, ring_slots, finishers = [], [], [], []
        for part in parts:
            if part.get('packed'):
                ring_slot = None
                if 'ring' in part:        # the blocks sit in a slot of the page-locked shared ring: DMA straight from there
                    if self.ring is None:
                        raise _lib.WitwError('a batch built in a PinnedRing reached a GpuPreprocess without `ring`')
                    ring_slot, s_off, s_len, o_off, o_len = part['ring']
                    part = dict(part, surface_bytes=self.ring.view(ring_slot, s_off, s_len), overhead_bytes=self.ring.view(ring_slot, o_off, o_len))
                    ring_slots.append(ring_slot)
                # every part's blocks are queued for the device first; the decode launches below then cover all parts at once
                s_copied.append(self._copy_side(st, dev, (part['surface_bytes'], part['surface_desc'], part['surface_kind']), pinned=ring_slot is not None))
                o_copied.append(self._copy_side(st, dev, (part['overhead_bytes'], part['overhead_desc'], part['overhead_kind']), pinned=ring_slot is not None))
            else:
                if s_copied:
                    raise _lib.WitwError('the parts of a grouped batch are packed and unpacked')
                s_tab, s_kind, f1 = self._stage_side(st, dev, part['surface'])
                o_tab, o_kind, f2 = self._stage_side(st, dev, part['overhead'])
                finishers += [f for f in (f1, f2) if f is not None]
                if s_tabs and (s_kind != st.s_kind or o_kind != st.o_kind):
                    raise _lib.WitwError('the parts of a grouped batch hold different image kinds (fp32 CHW / u8 HWC)')
                st.s_kind, st.o_kind = s_kind, o_kind
                s_tabs.append(s_tab)
                o_tabs.append(o_tab)
            if part.get('idx') is not None:
                idx += list(part['idx'])
        if s_copied:
            if s_tabs:
                raise _lib.WitwError('the parts of a grouped batch are packed and unpacked')
            if all(c[2] == 2 for c in s_copied + o_copied):
                # JPEG on both sides: ONE set of decode launches for the ground and the overhead files of every part (the device
                # Huffman decoders are bound by a thread's chain of symbols, not by how many files a launch holds)
                both, _kind, f1 = self._table_side(st, s_copied + o_copied)
                n_s = sum(int(c[1].shape[0]) for c in s_copied)
                s_tab, o_tab, f2 = both[:n_s], both[n_s:], None      # (views: a re-decoded file's row is patched in place)
                st.s_kind = st.o_kind = 1
            else:
                s_tab, st.s_kind, f1 = self._table_side(st, s_copied)
                o_tab, st.o_kind, f2 = self._table_side(st, o_copied)
            finishers += [f for f in (f1, f2) if f is not None]
            s_tabs, o_tabs = [s_tab], [o_tab]
        for f in finishers:      # the device decoder's damage flags: ONE host wait per batch, behind every launch of the staging
            f()
        if ring_slots:           # (the host blocks were still needed for files Pillow re-decodes)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            for slot in ring_slots:
                self.ring.release_after(slot, ev)
        st.idx = idx if idx else None
        if len(s_tabs) == 1:
            s_tab, o_tab = s_tabs[0], o_tabs[0]
        else:
            s_tab, o_tab = torch.cat(s_tabs), torch.cat(o_tabs)
        st.n = s_tab.shape[0]
        if self.resize.panorama:      # random FoV crop offset per sample (reference: torch.randint per call, :121)
            if starts is None:
                starts = torch.randint(0, Globals.surface_width_max, (st.n,)) if self.resize.random_orientation else torch.zeros(st.n, dtype=torch.int64)
            s_tab = s_tab.cpu() if s_tab.is_cuda else s_tab
            s_tab[:, 3] = torch.as_tensor(starts, dtype=torch.int64)
        st.s_desc = s_tab if s_tab.is_cuda else s_tab.pin_memory().to(dev, non_blocking=True)
        st.o_desc = o_tab if o_tab.is_cuda else o_tab.pin_memory().to(dev, non_blocking=True)
        st.keep += [st.s_desc, st.o_desc]
        st.event = torch.cuda.Event()
        st.event.record(torch.cuda.current_stream())
        return st

    def __call__(self, batch, starts=None):
        st = batch if isinstance(batch, StagedBatch) else self.stage(batch, starts)
        st.wait()
        c, r = self.channels, self.resize
        hs, wmax, so = Globals.surface_height_max, Globals.surface_width_max, Globals.overhead_size
        nd = self.norm.n_div255
        surface = ops.resize_batched(st.s_desc, st.n, c, (hs, r.surface_width), wfull=wmax if r.panorama else None, kind=st.s_kind,
                                     mean=self.norm.mean, std=self.norm.std, n_div255=nd)
        if self.fused and not self.keep_overhead and ops.polar_tiles(st.o_desc.device, so, hs, wmax) is not None:
            # Resize -> ImageNormalization -> PolarTransform of the overhead side in one launch (the same bits; the 256 x 256
            # intermediate is never written)
            polar = ops.polar_from_raw(desc=st.o_desc, kind=st.o_kind, batch=st.n, channels=c, mean=self.norm.mean, std=self.norm.std,
                                       n_div255=nd, size=so, h_s=hs, w_s=wmax)
            return {'idx': st.idx, 'surface': surface, 'polar': polar}
        overhead = ops.resize_batched(st.o_desc, st.n, c, (so, so), kind=st.o_kind, mean=self.norm.mean, std=self.norm.std, n_div255=nd)
        return self.polar({'idx': st.idx, 'surface': surface, 'overhead': overhead})


class DevicePrefetcher(object):
    """Iterates a DataLoader one batch ahead: batch n+1 is staged (H2D copies from the loader's pinned buffers, descriptor
    tables) on a copy stream while the caller's kernels for batch n run on the compute stream. Yields StagedBatch."""

    def __init__(self, loader, prep, group=1):
        self.loader, self.prep, self.group = loader, prep, max(1, int(group))
        self.stream = torch.cuda.Stream()

    def __len__(self):
        return (len(self.loader) + self.group - 1) // self.group

    def _stage(self, raw):
        with torch.cuda.stream(self.stream):
            return self.prep.stage(raw)

    def _pull(self, it):
        """The next `group` loader batches (fewer at the end of the data) staged as one batch, or None."""
        parts = []
        for _ in range(self.group):
            try:
                parts.append(next(it))
            except StopIteration:
                break
        if not parts:
            return None
        return self._stage(parts if self.group > 1 else parts[0])

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._pull(it)
        while nxt is not None:
            cur = nxt
            nxt = self._pull(it)
            yield cur


def make_staging(dataset_obj, loader_batch, num_workers, n_loaders=1, prefetch_factor=2):
    """How a driver's DataLoaders hand their batches to the GPU. With worker processes and Globals.pinned_ring: a ring.PinnedRing
    sized from the first files of the data set (slots for every batch the workers can have in flight) -> (ring, collate_fn bound
    to it, pin_memory=False); else torch's own path -> (None, collate_packed, pin_memory=True). A batch larger than a slot (files
    much bigger than the sampled ones) falls back to ordinary blocks by itself."""
    import functools
    if not (num_workers > 0 and getattr(Globals, 'pinned_ring', True) and device.type == 'cuda' and len(dataset_obj) > 0):
        return None, collate_packed, True
    from . import ring as ring_mod
    n = min(len(dataset_obj), 8)
    probe = collate_packed([dataset_obj[i * (len(dataset_obj) // n)] for i in range(n)])
    per_pair = (probe['surface_bytes'].numel() + probe['overhead_bytes'].numel()) / n
    slot_bytes = int(1.5 * per_pair * loader_batch) + (1 << 16)
    slots = n_loaders * num_workers * prefetch_factor + 4
    if slots * slot_bytes > (8 << 30):          # do not lock more than 8 GB of host memory
        return None, collate_packed, True
    ring = ring_mod.PinnedRing(slots=slots, slot_bytes=slot_bytes)
    return ring, functools.partial(collate_packed, ring=ring), False


def loader_split(batch_size, num_workers):
    """Parts a batch is decoded in (DevicePrefetcher(group=...)): 4 for batches of 32 and more that divide evenly, when there
    are worker processes to decode them side by side; else 1."""
    return 4 if (num_workers > 0 and batch_size >= 32 and batch_size % 4 == 0) else 1


def projector_dump(writer, surface, overhead, surface_embed, overhead_embed, global_step, tag, img_mean, img_std):
    """The TensorBoard embedding-projector dump at the end of a validation epoch / of test() (model/cvig_fov.py:474-479,
    :534-540): surface embeddings and the diagonal of the orientation-aligned overhead crops, labelled by the images of
    the last batch. Nothing is computed for the null writer (tensorboard absent)."""
    if isinstance(writer, _NullWriter):
        return False
    n = surface_embed.shape[0]
    labels = [[i, 0] for i in range(n)] + [[i, 1] for i in range(overhead_embed.shape[0])]
    label_header = ['idx', 'type']
    original_images = inverse_normalize(
        torch.cat((torch.nn.functional.pad(surface, (0, overhead.shape[-1] - surface.shape[-1])), overhead), dim=0),
        mean=img_mean, std=img_std)
    with torch.no_grad():
        orientation = correlation(overhead_embed.detach(), surface_embed.detach())
        cropped = crop_overhead(overhead_embed.detach(), orientation, surface_embed.shape[3])
    cropped = cropped[range(n), range(n)]
    writer.add_embedding(torch.cat((surface_embed.detach().reshape(n, -1), cropped.reshape(n, -1)), dim=0).cpu(),
                         metadata=[[l, 0] for l in labels], metadata_header=label_header, label_img=original_images.cpu(),
                         global_step=global_step, tag=tag)
    return True


class _NullWriter(object):
    def add_scalar(self, *a, **k):
        pass

    add_text = add_embedding = add_scalar


def _writer(path):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(path)
    except Exception:
        return _NullWriter()


def load_reference_state_dict(encoder, state):
    """Load a checkpoint written by the reference (model/cvig_fov.py:485-486): same keys, plus the unused
    VGG classifier tensors (model.classifier.*), which are dropped."""
    encoder._reference_classifier = {k: v for k, v in state.items() if k.startswith('model.classifier')} or None
    state = {k: v for k, v in state.items() if not k.startswith('model.classifier')}
    return encoder.load_state_dict(state, strict=True)


VGG16_CLASSIFIER_SHAPES = {0: (4096, 25088), 3: (4096, 4096), 6: (1000, 4096)}      # torchvision vgg16().classifier Linear layers


def save_reference_state_dict(encoder, path):
    """Write a checkpoint the reference's `load_state_dict` accepts (model/cvig_fov.py:511-512, strict): this
    encoder's tensors plus the VGG classifier tensors the reference carries along (`self.model = model` keeps the whole
    VGG, :290) -- the ones a previous load_reference_state_dict / load_vgg16_state_dict saw, else zeros of the VGG16 shapes
    (they never reach the forward, :258). The zeros are stored as ONE element expanded to the shape (stride 0): the file
    stays a few MB instead of 530 MB, and load_state_dict copies from it like from any tensor of that shape."""
    state = {k: v.detach().cpu() for k, v in encoder.state_dict().items()}
    extra = getattr(encoder, '_reference_classifier', None)
    if extra is None:
        extra = {}
        for i, (o, n) in VGG16_CLASSIFIER_SHAPES.items():
            extra['model.classifier.%d.weight' % i] = torch.zeros((1, 1)).expand(o, n)
            extra['model.classifier.%d.bias' % i] = torch.zeros((1,)).expand(o)
    state.update(extra)
    torch.save(state, path)


VGG16_CONVS = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21)     # conv indices of torchvision vgg16().features[:23] (cfg D)


def load_vgg16_state_dict(encoder, state):
    """What the reference's FOV_DSM.__init__ does with torch.hub's pretrained VGG16 (model/cvig_fov.py:256-272), from a
    torchvision-format state_dict (or a path to one): `features.{i}.weight|bias` of the ten convs of features[:23] go to the
    same-numbered layers (through the HorizCircPadding / AddDropout wrappers), the three extra convs 23 / 25 / 27 get
    xavier_uniform weights and zero bias, `classifier.*` is remembered for save_reference_state_dict. cvig_semantic's
    5-channel first conv (model/cvig_semantic.py:301-303): a freshly initialised Conv2d(5,64) whose first three input
    channels take the VGG filter (its bias stays the fresh one, as in the reference)."""
    if isinstance(state, (str, bytes)) or hasattr(state, '__fspath__'):
        state = torch.load(state, map_location='cpu')
    if any(k.startswith('model.features') for k in state):
        raise _lib.WitwError('this is a cvig_fov checkpoint (model.features.*): use load_reference_state_dict')
    with torch.no_grad():
        for idx in VGG16_CONVS:
            conv = _conv_of(encoder.model.features[idx])
            w, b = state['features.%d.weight' % idx], state['features.%d.bias' % idx]
            if idx == 0 and conv.in_channels != w.shape[1]:
                fresh = torch.nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=3, stride=1, padding=1)
                conv.weight.copy_(fresh.weight)
                conv.bias.copy_(fresh.bias)
                conv.weight[:, :w.shape[1]] = w.to(conv.weight.device)
            else:
                if tuple(w.shape) != tuple(conv.weight.shape):
                    raise _lib.WitwError('features.%d.weight has shape %s, VGG16 layer %d is %s' % (idx, tuple(w.shape), idx,
                                                                                             tuple(conv.weight.shape)))
                conv.weight.copy_(w)
                conv.bias.copy_(b)
            for t in (conv.weight, conv.bias):
                t._witw_version = getattr(t, '_witw_version', 0) + 1
        for idx in (23, 25, 27):
            conv = _conv_of(encoder.model.features[idx])
            torch.nn.init.xavier_uniform_(conv.weight)
            torch.nn.init.zeros_(conv.bias)
    cls = {'model.' + k: v for k, v in state.items() if k.startswith('classifier.')}
    encoder._reference_classifier = cls or None
    return encoder


def train(dataset='cvusa', fov=360, val_quantity=1000, batch_size=64, num_workers=12, num_epochs=999999, csv_path=None,
          seed=0, _mod=None):
    """model/cvig_fov.py:385-487 on the HIP kernels (same flow, checkpoint names and prints). `_mod` is the
    module whose Globals / FOV_DSM / ImagePairDataset / GpuPreprocess are used (cvig_semantic passes itself)."""
    import pathlib
    import sys
    import time
    from datetime import datetime
    m = _mod or sys.modules[__name__]
    Globals, FOV_DSM, ImagePairDataset, GpuPreprocess, device = m.Globals, m.FOV_DSM, m.ImagePairDataset, m.GpuPreprocess, m.device
    from . import parallel
    world, rank = parallel.world(), parallel.rank()
    if device.type == 'cuda':
        torch.cuda.set_device(device)
    pathlib.Path('./weights').mkdir(parents=True, exist_ok=True)
    writer = _writer('runs/{}/train/{}/{}'.format(dataset, fov, datetime.now().strftime("%Y%m%d-%H%M%S"))) if rank == 0 \
        else _NullWriter()
    csv_path = csv_path or Globals.dataset_paths[dataset]['train']
    # raw='jpeg': a worker only entropy-decodes JPEG files, the GPU finishes them (witw_amd/jpeg.py; the same bytes); True:
    # Pillow's bytes; either way converted / resized / normalised on the GPU
    trainval_set = ImagePairDataset(dataset=dataset, csv_path=csv_path, raw='jpeg' if getattr(Globals, 'device_jpeg', True) else True)
    split_gen = torch.Generator().manual_seed(seed) if world > 1 else None      # every rank must draw the same split
    train_set, val_set = torch.utils.data.random_split(trainval_set, [len(trainval_set) - val_quantity, val_quantity],
                                                       generator=split_gen)
    # batch_size is per process; with N ranks (python -m torch.distributed.run ... ) the global batch is N * batch_size,
    # every rank reads its own shard of each epoch and the loss still couples the whole global batch
    train_sampler = torch.utils.data.distributed.DistributedSampler(train_set, shuffle=True, drop_last=True) if world > 1 else None
    val_sampler = torch.utils.data.distributed.DistributedSampler(val_set, shuffle=False) if world > 1 else None
    # workers decode and pack each batch into two byte blocks (collate_packed) built in place in page-locked shared memory
    # (make_staging / ring.PinnedRing; else the loader's pinning thread), DevicePrefetcher copies batch n+1 to the GPU on a side
    # stream while batch n computes
    ring, collate, pin = m.make_staging(trainval_set, batch_size, num_workers, n_loaders=2) if hasattr(m, 'make_staging') else (None, collate_packed, True)
    prep = GpuPreprocess(dataset, fov, device=device, ring=ring)
    train_loader = torch.utils.data.DataLoader(train_set, batch_size=batch_size, shuffle=(world == 1), drop_last=True,
                                               sampler=train_sampler, num_workers=num_workers, collate_fn=collate, pin_memory=pin)
    val_loader = torch.utils.data.DataLoader(val_set, batch_size=batch_size, shuffle=False, drop_last=False,
                                             sampler=val_sampler, num_workers=num_workers, collate_fn=collate, pin_memory=pin)
    surface_encoder = FOV_DSM(circ_padding=False, seed=seed)
    overhead_encoder = FOV_DSM(circ_padding=True, seed=seed)
    if getattr(Globals, 'vgg16_weights', None):      # the reference's starting point (:256-272); else seeded synthetic weights
        load_vgg16_state_dict(surface_encoder, Globals.vgg16_weights)
        load_vgg16_state_dict(overhead_encoder, Globals.vgg16_weights)
    surface_encoder, overhead_encoder = surface_encoder.to(device), overhead_encoder.to(device)
    surface_encoder.precision = overhead_encoder.precision = Globals.precision
    parallel.broadcast_parameters([surface_encoder, overhead_encoder])
    all_params = list(surface_encoder.parameters()) + list(overhead_encoder.parameters())
    optimizer = Adam(all_params, lr=1.E-5)
    reducer = parallel.OverlappedGradReducer([surface_encoder, overhead_encoder])
    embed = PairEmbedder(surface_encoder, overhead_encoder)

    def say(*a):
        if rank == 0:
            print(*a)

    best_loss = None
    for epoch in range(num_epochs):
        say('Epoch %d, %s' % (epoch + 1, time.ctime(time.time())))
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)
        for phase in ['train', 'val']:
            running_count = 0
            running_loss = 0.
            loader = train_loader if phase == 'train' else val_loader
            surface_encoder.train(phase == 'train')
            overhead_encoder.train(phase == 'train')
            for batch, raw in enumerate(DevicePrefetcher(loader, prep)):
                data = prep(raw)
                surface = data['surface']
                overhead = data['polar']
                if phase == 'train':      # Dropout2d masks keyed on the global step, not on how many calls this process made
                    surface_encoder._drop_step = overhead_encoder._drop_step = epoch * len(loader) + batch
                with torch.set_grad_enabled(phase == 'train'):
                    # validation: PairEmbedder (small batches on two streams, bf16 as one hipGraph); training: the plain calls
                    surface_embed, overhead_embed = embed(surface, overhead)
                    # correlation -> crop_overhead -> l2_distance -> triplet_loss (:450-454) over the GLOBAL batch
                    loss, orientation_estimate, distance = sharded_match_loss(overhead_embed, surface_embed)
                    if phase == 'train':
                        optimizer.zero_grad()
                        loss.backward()          # per-encoder gradient all-reduce overlapped with the other encoder's backward
                        reducer.wait()
                        optimizer.step()
                count = surface_embed.size(0) * world
                running_count += count
                running_loss += loss.item() * count
                say('epoch = {} {}, iter = {}, count = {}, loss = {:.4f}'.format(epoch + 1, phase, batch, running_count,
                                                                                loss.item()))
                writer.add_scalar('{} loss'.format(phase), running_loss / running_count, epoch * len(loader) + batch)
            say('  %5s: avg loss = %f' % (phase, running_loss / max(1, running_count)))
        if running_count and getattr(m, 'PROJECTOR_DUMP', True):      # last validation batch -> embedding projector (:474-479)
            projector_dump(writer, surface, overhead, surface_embed, overhead_embed, epoch + 1, 'val_embedding',
                           Globals.img_mean, Globals.img_std)
        if running_count and (best_loss is None or running_loss / running_count < best_loss):
            say('-------> new best')
            best_loss = running_loss / running_count
            if rank == 0:
                for enc, side in ((surface_encoder, 'surface'), (overhead_encoder, 'overhead')):
                    path = './weights/fov_{}_{}_best.pth'.format(int(fov), side)
                    if getattr(Globals, 'reference_checkpoints', True):      # the layout the reference's strict load expects (:511-512)
                        save_reference_state_dict(enc, path)
                    else:
                        torch.save(enc.state_dict(), path)
            writer.add_text('best_loss', 'new best loss: {}, epoch: {}'.format(best_loss, epoch + 1),
                            epoch * len(loader) + batch)        # the last validation iteration's step (:487)
    if ring is not None:
        del train_loader, val_loader
        ring.close()
    return best_loss


def test(dataset='cvusa', fov=360, batch_size=64, num_workers=8, csv_path=None, _mod=None):
    """model/cvig_fov.py:490-575: embed the test set, rank every query against the whole gallery (all
    queries at once on the GPU instead of the O(N) Python loop), print the recall table."""
    import sys
    from datetime import datetime
    m = _mod or sys.modules[__name__]
    Globals, FOV_DSM, ImagePairDataset, GpuPreprocess, device = m.Globals, m.FOV_DSM, m.ImagePairDataset, m.GpuPreprocess, m.device
    from . import parallel
    world, rank = parallel.world(), parallel.rank()
    if device.type == 'cuda':
        torch.cuda.set_device(device)
    writer = _writer('runs/{}/test/{}/{}'.format(dataset, fov, datetime.now().strftime("%Y%m%d-%H%M%S"))) if rank == 0 \
        else _NullWriter()
    csv_path = csv_path or Globals.dataset_paths[dataset]['test']
    # the reference crops test panoramas at a random orientation too (:495-499); Globals.test_random_orientation = False
    # makes the evaluation repeatable
    test_set = ImagePairDataset(dataset=dataset, csv_path=csv_path, raw='jpeg' if getattr(Globals, 'device_jpeg', True) else True)
    # under torch.distributed every rank embeds a contiguous shard of the test set and keeps its gallery rows
    shard_begin, shard_end = parallel.shard_range(len(test_set))
    shard = torch.utils.data.Subset(test_set, range(shard_begin, shard_end)) if world > 1 else test_set
    split = loader_split(batch_size, num_workers)      # workers decode quarter batches: the first batch arrives 4x sooner
    ring, collate, pin = m.make_staging(test_set, batch_size // split, num_workers) if hasattr(m, 'make_staging') else (None, collate_packed, True)
    prep = GpuPreprocess(dataset, fov, getattr(Globals, 'test_random_orientation', True), device=device, ring=ring)
    test_loader = torch.utils.data.DataLoader(shard, batch_size=batch_size // split, shuffle=False, drop_last=False,
                                              num_workers=num_workers, collate_fn=collate, pin_memory=pin)
    surface_encoder = FOV_DSM(circ_padding=False).to(device)
    overhead_encoder = FOV_DSM(circ_padding=True).to(device)
    surface_encoder.precision = overhead_encoder.precision = Globals.precision
    load_reference_state_dict(surface_encoder, torch.load('./weights/fov_{}_surface_best.pth'.format(int(fov))))
    load_reference_state_dict(overhead_encoder, torch.load('./weights/fov_{}_overhead_best.pth'.format(int(fov))))
    surface_encoder.eval()
    overhead_encoder.eval()
    su_parts, ov_parts = [], []
    data = None
    embed = PairEmbedder(surface_encoder, overhead_encoder)      # small batches: both encoders at once, bf16 as one hipGraph
    for raw in DevicePrefetcher(test_loader, prep, group=split):
        data = prep(raw)
        with torch.no_grad():
            su, ov = embed(data['surface'], data['polar'])
            su_parts.append(su)
            ov_parts.append(ov)
    if ring is not None:
        del test_loader
        ring.close()
    if data is not None and getattr(m, 'PROJECTOR_DUMP', True):        # last batch -> embedding projector (:534-540)
        projector_dump(writer, data['surface'], data['polar'], su_parts[-1], ov_parts[-1], 0, 'test_embedding',
                       Globals.img_mean, Globals.img_std)
    surface_embed = torch.cat(su_parts, dim=0)
    overhead_embed = torch.cat(ov_parts, dim=0)
    if Globals.precision == 'fp16x3' and ops.f16x3_overflowed(surface_embed.device):
        raise _lib.WitwError("an activation left the fp16 range (|v| > 65504) on the fp16x3 kernels: evaluate these weights with "
                             "precision 'fp32'")
    rk = evaluation_ranks(overhead_embed, surface_embed, shard_begin, world, getattr(Globals, 'match_method', 'auto'))
    t = recall_table(rk)
    count = len(rk)
    if rank != 0:
        return t
    lines = ['Top  1: {:.2f}%'.format(t['top_1']), 'Top  5: {:.2f}%'.format(t['top_5']), 'Top 10: {:.2f}%'.format(t['top_10']),
             'Top 1%: {:.2f}%'.format(t['top_1pct']), 'Avg. Rank: {:.2f}'.format(t['mean']),
             'Med. Rank: {:.2f}'.format(t['median']), 'Locations: {}'.format(count)]
    for tag, line in zip(['top_1', 'top_5', 'top_10', 'top_1%', 'avg_rank', 'med_rank', 'locations'], lines):
        print(line)
        writer.add_text(tag, line)
    return t


def sweep_scores(overhead_embed, surface_embed, output_width_max=64):
    """The scoring block of tools/heatmap/heatmap.py:172-178 (one photo against N satellite tiles):
    -> (orientation in degrees, dissimilarity, score = exp(10*(1-d))), each [N] (or [N,Bs])."""
    ori, dist = match(overhead_embed, surface_embed)
    orientations = torch.squeeze(ori) * 360 / output_width_max - 180
    distances = torch.squeeze(dist)
    return orientations, distances, torch.exp(10. * (1. - distances))


def main(argv=None):
    """CLI of model/cvig_fov.py:580-601."""
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument('--mode', default='train', choices=['train', 'test'], help='Run mode. [Default = train]')
    parser.add_argument('--dataset', default='cvusa', choices=['cvusa', 'witw'], help='Dataset to use. [Default = cvusa]')
    parser.add_argument('--fov', type=int, default=360, choices=range(6, 361), metavar='{6-360}',
                        help='The field of view for cropping street level images. [Default = 360]')
    parser.add_argument('--precision', default='fp32', choices=['fp32', 'bf16', 'fp16x3'],
                        help='Encoder arithmetic (not in the reference): fp32, bf16 MFMA mixed precision, or (test mode) fp16x3 = '
                             'fp32-grade products on the fp16 MFMA. [Default = fp32]')
    parser.add_argument('--vgg16', default=None, metavar='PATH',
                        help='train mode: torchvision VGG16 state_dict file to start from (the reference downloads it through '
                             'torch.hub). [Default = seeded synthetic weights]')
    args = parser.parse_args(argv)
    print(args)
    Globals.precision = args.precision
    Globals.vgg16_weights = args.vgg16
    init_distributed()
    if args.mode == 'train':
        train(dataset=args.dataset, fov=args.fov)
    elif args.mode == 'test':
        test(dataset=args.dataset, fov=args.fov)


def init_distributed(backend='nccl'):
    """Under `python -m torch.distributed.run --nproc-per-node N -m witw_amd.cvig_fov ...`: one process per GPU over RCCL
    (backend name 'nccl'); a plain `python` launch stays single-process like the reference."""
    import os
    import torch.distributed as dist
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if device.type == 'cuda':
            torch.cuda.set_device(device)
        dist.init_process_group(backend, device_id=device if backend == 'nccl' else None)


if __name__ == '__main__':
    main()
