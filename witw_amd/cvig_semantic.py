#!/usr/bin/env python
"""MI355X-native drop-in for the hot path of the reference's model/cvig_semantic.py: cvig_fov with 5-channel
inputs (RGB + 2 semantic channels). Differences from cvig_fov (reference diff):
  * Globals.img_mean/img_std have 5 entries (model/cvig_semantic.py:25-26);
  * ImageNormalization divides only channels 0-2 by 255 (:172-176);
  * FOV_DSM's first conv is Conv2d(5,64) and is trainable (:301-309).
Everything else (matching, loss, ranking, transforms) is shared with witw_amd.cvig_fov.
Forward and backward run on the HIP kernels; because layer 0 trains, the backward walks all 13 layers
(dgrad through the frozen VGG layers, arg-max scatter behind the three fused max-pools, layer-0 wgrad).
"""
import torch

from . import cvig_fov as _fov
from . import synth
from .cvig_fov import (Adam, PolarTransform, Resize, correlation, crop_overhead, l2_distance, match, ranks,  # noqa: F401
                       recall_table, sweep_scores, triplet_loss)


class Globals(_fov.Globals):
    img_mean = [0.485, 0.456, 0.406, 0.45, 0.45]
    img_std = [0.229, 0.224, 0.225, 0.22, 0.22]
    dataset_paths = {
        'cvusa': {'train': './data/train-19zl.csv', 'test': './data/val-19zl.csv'},
        'witw': {'train': './data4/train_scenes.csv', 'test': './data4/test_scenes.csv', 'semantic': True},
    }


device = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')  # reference uses cuda:1 (:609)


class ImageNormalization(_fov.ImageNormalization):
    """model/cvig_semantic.py:167-176."""

    def __init__(self, mean=None, std=None):
        super().__init__(Globals.img_mean if mean is None else mean, Globals.img_std if std is None else std, n_div255=3)


class FOV_DSM(_fov.FOV_DSM):
    """model/cvig_semantic.py:275-325."""
    in_channels = 5

    def __init__(self, circ_padding=False, weights=None, seed=0):
        super().__init__(circ_padding, weights, seed)
        conv0 = _fov._conv_of(self.model.features[0])
        conv0.weight.requires_grad = True     # `torch_layer_num != 0` exemption, :308
        conv0.bias.requires_grad = True
