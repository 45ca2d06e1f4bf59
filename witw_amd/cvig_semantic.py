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

from .cvig_fov import (AddDropout, Adam, HorizCircPadding, PolarTransform, Resize, bilinear_interpolate,  # noqa: F401
                       correlation, crop_overhead, load_reference_state_dict, load_vgg16_state_dict, save_reference_state_dict,
                       inverse_normalize, l2_distance, match, ranks, recall_table, sweep_scores, triplet_loss)

PROJECTOR_DUMP = False      # the reference has the embedding-projector dump commented out here (model/cvig_semantic.py:508-510, :567-571)


class Globals(_fov.Globals):
    img_mean = [0.485, 0.456, 0.406, 0.45, 0.45]
    img_std = [0.229, 0.224, 0.225, 0.22, 0.22]
    dataset_paths = {
        'cvusa': {'train': './data/train-19zl.csv', 'test': './data/val-19zl.csv'},
        'witw': {'train': './data4/train_scenes.csv', 'test': './data4/test_scenes.csv', 'semantic': True},
    }


device = _fov.device      # the reference pins cuda:1 (:609); here ONE device per process, cuda:LOCAL_RANK, shared with cvig_fov's transforms


class ImageNormalization(_fov.ImageNormalization):
    """model/cvig_semantic.py:167-176."""

    def __init__(self, mean=None, std=None):
        super().__init__(Globals.img_mean if mean is None else mean, Globals.img_std if std is None else std, n_div255=3)


class ImagePairDataset(_fov.ImagePairDataset):
    """model/cvig_semantic.py:56-123. With Globals.dataset_paths[dataset]['semantic'] the 5-band scene TIFFs next
    to the listed files are read (`<stem>.tif`, :91-94); otherwise the RGB files plus, when present, the road mask
    `cresi_uint8/<overhead stem>.tif` as data['cresi'] (bands 0,1,2,-1; :104-113). Like the reference, a dataset
    without the 'semantic' key (cvusa) raises KeyError."""

    @classmethod
    def _globals(cls):
        return Globals

    def __init__(self, dataset, csv_path, base_path=None, transform=None, raw=False):
        super().__init__(dataset, csv_path, base_path, transform, raw)
        self.semantic = Globals.dataset_paths[dataset]['semantic']

    @staticmethod
    def _read_tiff(path):
        import numpy as np
        from . import tiffio
        a = tiffio.imread(path)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32).transpose((2, 0, 1))))

    def __getitem__(self, idx):
        import os
        row = self.file_paths.iloc[idx]
        if self.semantic:
            data = {'idx': idx, 'surface': self._read_tiff(os.path.splitext(row['surface'])[0] + '.tif'),
                    'overhead': self._read_tiff(os.path.splitext(row['overhead'])[0] + '.tif')}
        else:
            read = self._read_raw if self.raw else self._read
            data = {'idx': idx, 'surface': read(row['surface']), 'overhead': read(row['overhead'])}
            cresi_path = os.path.join(self.base_path, 'cresi_uint8',
                                      os.path.splitext(os.path.basename(row['overhead']))[0] + '.tif')
            data['cresi'] = self._read_tiff(cresi_path)[[0, 1, 2, -1], :, :] if os.path.exists(cresi_path) else None
        if self.transform is not None:
            data = self.transform(data)
        return data


class GpuPreprocess(_fov.GpuPreprocess):
    """Compose[Resize, ImageNormalization, PolarTransform] (model/cvig_semantic.py:424-428) on 5-band batches."""
    channels = 5
    normalization = ImageNormalization


class FOV_DSM(_fov.FOV_DSM):
    """model/cvig_semantic.py:275-325."""
    in_channels = 5

    def __init__(self, circ_padding=False, weights=None, seed=0):
        super().__init__(circ_padding, weights, seed)
        conv0 = _fov._conv_of(self.model.features[0])
        conv0.weight.requires_grad = True     # `torch_layer_num != 0` exemption, :308
        conv0.bias.requires_grad = True


def train(dataset='cvusa', fov=360, val_quantity=1000, batch_size=32, num_workers=8, num_epochs=999999, csv_path=None, seed=0):
    """model/cvig_semantic.py:416-518 (defaults :416): the cvig_fov loop on the 5-band encoders and dataset."""
    import sys
    return _fov.train(dataset, fov, val_quantity, batch_size, num_workers, num_epochs, csv_path, seed, _mod=sys.modules[__name__])


def test(dataset='cvusa', fov=360, batch_size=64, num_workers=8, csv_path=None):
    """model/cvig_semantic.py:521-606."""
    import sys
    return _fov.test(dataset, fov, batch_size, num_workers, csv_path, _mod=sys.modules[__name__])


def main(argv=None):
    """CLI of model/cvig_semantic.py:611-634."""
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument('--mode', default='train', choices=['train', 'test'], help='Run mode. [Default = train]')
    parser.add_argument('--dataset', default='cvusa', choices=['cvusa', 'witw'], help='Dataset to use. [Default = cvusa]')
    parser.add_argument('--fov', type=int, default=360, choices=range(6, 361), metavar='{6-360}',
                        help='The field of view for cropping street level images. [Default = 360]')
    parser.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                        help='Encoder arithmetic (not in the reference): fp32, or bf16 MFMA mixed precision. [Default = fp32]')
    parser.add_argument('--vgg16', default=None, metavar='PATH',
                        help='train mode: torchvision VGG16 state_dict file to start from (the RGB slice of the 5-channel first '
                             'conv takes the VGG filter, model/cvig_semantic.py:301-303). [Default = seeded synthetic weights]')
    args = parser.parse_args(argv)
    print(args)
    Globals.precision = args.precision
    Globals.vgg16_weights = args.vgg16
    _fov.init_distributed()
    if args.mode == 'train':
        train(dataset=args.dataset, fov=args.fov)
    elif args.mode == 'test':
        test(dataset=args.dataset, fov=args.fov)


if __name__ == '__main__':
    main()
