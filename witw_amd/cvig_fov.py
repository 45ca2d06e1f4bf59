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
        key = (conv.weight.data_ptr(), conv.weight._version, conv.bias._version)
        hit = self._packed.get(idx)
        if hit is None or hit[0] != key:
            hit = (key, ops.PackedConv(conv.weight, conv.bias))
            self._packed[idx] = hit
        return hit[1]

    def forward(self, x, dropout_scales=None):
        """x [B,C,128,W] NCHW fp32 on the GPU -> [B,16,4,W/8] NCHW (reference :292-294).
        In train() mode Dropout2d scales are drawn per call unless `dropout_scales`
        ({17|19|21: [B,C]}) injects them."""
        if not x.is_cuda:
            raise _lib.WitwError('FOV_DSM.forward needs a GPU tensor (no CPU fallback)')
        h = ops.nchw_to_nhwc8(x.contiguous())
        last = self.layer_specs[-1][0]
        for (idx, sh, relu, pool, drop) in self.layer_specs:
            scale = None
            if drop and self.training:
                if dropout_scales is not None:
                    scale = dropout_scales[idx]
                else:
                    conv = _conv_of(self.model.features[idx])
                    keep = torch.rand((x.shape[0], conv.out_channels), device=x.device) >= 0.2
                    scale = keep.float() / 0.8
            h = ops.conv3x3_fwd(h, self._pack(idx), stride_h=sh, circular=self.circ_padding, relu=relu, pool=pool,
                                out_nchw=(idx == last), drop_scale=scale)
        return h
