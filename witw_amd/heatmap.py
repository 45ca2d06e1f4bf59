#!/usr/bin/env python
"""Heat-map sweep on the MI355X hot path: one surface photo against a grid of satellite tiles -> per tile the
estimated orientation, dissimilarity and score, written as the reference's CSV (tools/heatmap/heatmap.py:113-187,
SURVEY §8f-1). Same tiling, transforms and score formula; differences by design:
  * the satellite strip is uploaded once and every tile window is cropped, resized, normalised and
    polar-transformed on the GPU in batches (the reference makes one gdal.Translate call per tile on the CPU);
  * the encoders and the correlation / crop / distance chain are the HIP kernels (cvig_fov.sweep_scores);
  * the raster source is an interface (`read_strip`): GdalTileSource needs osgeo.gdal (absent from this image),
    ArrayTileSource takes any in-memory array with a north-up geotransform.
"""
import os

import numpy as np
import torch

from . import cvig_fov as cvig
from . import ops

Globals = cvig.Globals

names = ['01_rio', '02_vegas', '03_paris', '04_shanghai', '05_khartoum', '06_atlanta', '07_moscow', '08_mumbai', '09_san',
         '10_dar', '11_rotterdam']          # SpaceNet AOIs, tools/heatmap/heatmap.py:20-32


def tile_windows(bounds, edge, offset):
    """tools/heatmap/heatmap.py:115-125: centres and [ulx, uly, lrx, lry] windows of the tile grid, eastings outer,
    northings (descending) inner. bounds = (left, bottom, right, top)."""
    center_eastings, center_northings, windows = [], [], []
    e2 = edge / 2.
    for easting in np.arange(bounds[0] - e2, bounds[2] - e2, offset):
        for northing in np.arange(bounds[3] + e2, bounds[1] + e2, -offset):
            center_eastings.append(easting + e2)
            center_northings.append(northing - e2)
            windows.append([easting, northing, easting + edge, northing - edge])
    return center_eastings, center_northings, windows


class ArrayTileSource(object):
    """A north-up raster held in memory: array [C,H,W]; (origin_x, origin_y) = map coordinates of the upper-left
    corner of pixel (0,0); pixel_size in map units. Windows are cut on whole pixels the way gdal.Translate's
    projWin does for a north-up raster (offset = floor((ul - origin)/pixel + 0.001), size = int(extent/pixel + 0.5));
    parts of a window outside the raster read as zero."""

    def __init__(self, array, origin_x, origin_y, pixel_size):
        self.array = torch.as_tensor(np.asarray(array, dtype=np.float32)) if not torch.is_tensor(array) else array.float()
        self.origin_x, self.origin_y, self.pixel_size = float(origin_x), float(origin_y), float(pixel_size)

    def pixel_window(self, window):
        ulx, uly, lrx, lry = window
        x0 = int(np.floor((ulx - self.origin_x) / self.pixel_size + 0.001))
        y0 = int(np.floor((self.origin_y - uly) / self.pixel_size + 0.001))
        w = int((lrx - ulx) / self.pixel_size + 0.5)
        h = int((uly - lry) / self.pixel_size + 0.5)
        return x0, y0, w, h

    def read_strip(self, device):
        return self.array.to(device)


class GdalTileSource(ArrayTileSource):
    """A GeoTIFF strip read once through osgeo.gdal (tools/heatmap/heatmap.py:128-129)."""

    def __init__(self, path):
        from osgeo import gdal          # not in this image: raises ImportError, the caller passes another source
        ds = gdal.Open(path)
        gt = ds.GetGeoTransform()
        if gt[2] != 0 or gt[4] != 0 or abs(gt[1] + gt[5]) > 1e-9 * abs(gt[1]):
            raise ValueError('%s: only north-up rasters with square pixels are supported' % path)
        super().__init__(ds.ReadAsArray(), gt[0], gt[3], gt[1])


# ---- the reference tool's dataset / transform classes (tools/heatmap/heatmap.py:35-110), same names and dict -> dict
# protocol ({'image': CHW float tensor}); the transforms run on the GPU through the C-ABI ops. sweep() below does
# not need them (it batches the same steps), they are here for code written against the reference tool.
class ImageDataset(torch.utils.data.Dataset):
    """tools/heatmap/heatmap.py:35-47: photos by path -> {'image': CHW fp32}."""

    def __init__(self, paths, transform=None):
        self.paths = paths
        self.transform = transform

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, idx):
        data = {'image': cvig.ImagePairDataset._read(self.paths[idx])}
        if self.transform is not None:
            data = self.transform(data)
        return data


class TileDataset(torch.utils.data.Dataset):
    """tools/heatmap/heatmap.py:50-66 with a tile source (ArrayTileSource / GdalTileSource) in place of the
    per-tile gdal.Translate call: window idx -> {'image': [C,h,w]} cut from the strip."""

    def __init__(self, source, windows, transform=None):
        self.source = source
        self.windows = windows
        self.transform = transform
        self._strip = None

    def __len__(self):
        return len(self.windows)

    def __getitem__(self, idx):
        if self._strip is None:
            self._strip = self.source.read_strip(cvig.device)
        data = {'image': _cut_tiles(self._strip, self.source, [self.windows[idx]])[0]}
        if self.transform is not None:
            data = self.transform(data)
        return data


class ResizeSurface(object):
    """tools/heatmap/heatmap.py:69-78."""

    def __init__(self, fov=360):
        self.fov = fov
        self.surface_width = int(self.fov / 360 * Globals.surface_width_max)

    def __call__(self, data):
        x = data['image'].to(cvig.device).float().contiguous()
        data['image'] = ops.resize_bilinear(x.unsqueeze(0), (Globals.surface_height_max, self.surface_width))[0]
        return data


class ResizeOverhead(object):
    """tools/heatmap/heatmap.py:81-87."""

    def __call__(self, data):
        x = data['image'].to(cvig.device).float().contiguous()
        data['image'] = ops.resize_bilinear(x.unsqueeze(0), (Globals.overhead_size, Globals.overhead_size))[0]
        return data


class ImageNormalization(object):
    """tools/heatmap/heatmap.py:90-101 (the ImageNet statistics are hard-coded there too)."""

    def __call__(self, data):
        x = data['image'].to(cvig.device).float().contiguous()
        data['image'] = ops.normalize(x.unsqueeze(0), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])[0]
        return data


class PolarTransform(object):
    """tools/heatmap/heatmap.py:104-110: {'image'} -> cvig.PolarTransform's dict ({'overhead', 'polar'})."""

    def __init__(self):
        self.transform = cvig.PolarTransform()

    def __call__(self, data):
        return self.transform({'overhead': data['image']})


def _cut_tiles(strip, source, windows):
    """[N,C,h,w] tile stack cut from the device-resident strip (zero outside the raster)."""
    C, H, W = strip.shape
    x0, y0, w, h = source.pixel_window(windows[0])
    tiles = torch.zeros((len(windows), C, h, w), dtype=torch.float32, device=strip.device)
    for i, win in enumerate(windows):
        x0, y0, wi, hi = source.pixel_window(win)
        if (wi, hi) != (w, h):
            raise ValueError('tile windows differ in pixel size: %s vs %s' % ((wi, hi), (w, h)))
        xa, xb, ya, yb = max(x0, 0), min(x0 + w, W), max(y0, 0), min(y0 + h, H)
        if xa < xb and ya < yb:
            tiles[i, :, ya - y0:yb - y0, xa - x0:xb - x0] = strip[:, ya:yb, xa:xb]
    return tiles


def embed_photo(surface_encoder, photo, fov):
    """ResizeSurface + ImageNormalization (tools/heatmap/heatmap.py:68-102) + surface encoder. photo: [3,H,W] 0..255."""
    ws = int(fov / 360 * Globals.surface_width_max)
    x = ops.resize_bilinear(photo[:3].unsqueeze(0).float().to(cvig.device).contiguous(), (Globals.surface_height_max, ws),
                            Globals.img_mean, Globals.img_std)
    with torch.no_grad():
        return surface_encoder(x)


def embed_tiles(overhead_encoder, source, windows, batch_size=64):
    """ResizeOverhead + ImageNormalization + PolarTransform (:79-110) + overhead encoder over all windows."""
    strip = source.read_strip(cvig.device)[:3]
    parts = []
    with torch.no_grad():
        for i in range(0, len(windows), batch_size):
            tiles = _cut_tiles(strip, source, windows[i:i + batch_size])
            # resize + normalise + polar transform in one launch (witw_polar_from_raw: the same bits as the three transforms)
            polar = ops.polar_from_raw(tiles, mean=Globals.img_mean, std=Globals.img_std, size=Globals.overhead_size,
                                       h_s=Globals.surface_height_max, w_s=Globals.surface_width_max)
            parts.append(overhead_encoder(polar))
    return torch.cat(parts, dim=0)


def sweep(aoi, bounds, edge, offset, fov, sat_dir, photo_path, csv_path, tile_source=None, surface_encoder=None,
          overhead_encoder=None, weights_dir='../../model', photo=None, batch_size=64):
    """tools/heatmap/heatmap.py:113-187. Extra keyword arguments inject the raster source, already-loaded encoders
    or an in-memory photo (tests, services); by default everything is read from the reference's paths."""
    import pandas as pd
    center_eastings, center_northings, windows = tile_windows(bounds, edge, offset)
    if tile_source is None:
        tile_source = GdalTileSource(os.path.join(sat_dir, names[aoi - 1] + '.tif'))
    if surface_encoder is None:
        surface_encoder = cvig.FOV_DSM(circ_padding=False).to(cvig.device)
        cvig.load_reference_state_dict(surface_encoder, torch.load(
            os.path.join(weights_dir, 'fov_{}_surface_best.pth'.format(int(fov))), map_location='cpu'))
    if overhead_encoder is None:
        overhead_encoder = cvig.FOV_DSM(circ_padding=True).to(cvig.device)
        cvig.load_reference_state_dict(overhead_encoder, torch.load(
            os.path.join(weights_dir, 'fov_{}_overhead_best.pth'.format(int(fov))), map_location='cpu'))
    surface_encoder.eval()
    overhead_encoder.eval()
    if photo is None:
        photo = cvig.ImagePairDataset._read(photo_path)
    surface_embed = embed_photo(surface_encoder, photo, fov)
    overhead_embed = embed_tiles(overhead_encoder, tile_source, windows, batch_size)
    orientations, distances, scores = cvig.sweep_scores(overhead_embed, surface_embed)
    df = pd.DataFrame({'x': center_eastings, 'y': center_northings, 'orientation': orientations.cpu().numpy().reshape(-1),
                       'dissimilarity': distances.cpu().numpy().reshape(-1), 'score': scores.cpu().numpy().reshape(-1)})
    df.to_csv(csv_path, index=False)
    return df


def main(argv=None):
    """CLI of tools/heatmap/heatmap.py:197-246 (the -i/--image layer export is a pure GDAL call and stays there)."""
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument('-a', '--aoi', type=int, choices=range(1, 12), default=3, help='SpaceNet AOI of satellite image')
    parser.add_argument('-b', '--bounds', type=float, nargs=4, default=(447665.8, 5411329.8, 448184.8, 5411814.8),
                        metavar=('left', 'bottom', 'right', 'top'), help='Bounds as UTM coordinates')
    parser.add_argument('-e', '--edge', type=float, default=225, help='Edge length of satellite imagery tiles [m]')
    parser.add_argument('-o', '--offset', type=float, default=56.25, help='Offset between centers of adjacent tiles [m]')
    parser.add_argument('-f', '--fov', type=int, default=70, help='Field of view assumed for photo (deg, rounded)')
    parser.add_argument('-s', '--satdir', default='/local_data/geoloc/sat/utm', help='Folder containing satellite images')
    parser.add_argument('-p', '--photopath', default='img.jpg', help='Path to surface photo to analyze')
    parser.add_argument('-c', '--csvpath', default='./geomatch.csv', help='Path to output CSV file path')
    args = parser.parse_args(argv)
    sweep(args.aoi, args.bounds, args.edge, args.offset, args.fov, args.satdir, args.photopath, args.csvpath)


if __name__ == '__main__':
    main()
