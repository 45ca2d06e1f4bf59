"""Heat-map sweep (tools/heatmap/heatmap.py:113-187) on the HIP path vs the oracle run tile by tile."""
import numpy as np
import pytest
import torch

from oracle import cvig_fov_oracle as O
from witw_amd import synth


def test_tile_windows_grid():
    from witw_amd import heatmap
    ce, cn, win = heatmap.tile_windows((100., 200., 220., 300.), 40., 30.)
    # eastings 80,110,140,170 (outer) x northings 320,290,260,230 (inner, descending)
    assert len(win) == 16 and win[0] == [80., 320., 120., 280.] and win[1] == [80., 290., 120., 250.]
    assert ce[:2] == [100., 100.] and cn[:2] == [300., 270.] and ce[4] == 130.
    src = heatmap.ArrayTileSource(np.zeros((3, 50, 60), np.float32), origin_x=90., origin_y=330., pixel_size=2.)
    assert src.pixel_window(win[0]) == (-5, 5, 20, 20) and src.pixel_window(win[5]) == (10, 20, 20, 20)


@pytest.mark.gpu
def test_sweep_matches_oracle(tmp_path):
    import pandas as pd
    from witw_amd import cvig_fov, heatmap
    fov = 70
    wts = synth.fov_dsm_weights(91)
    se = cvig_fov.FOV_DSM(circ_padding=False, weights=wts).cuda().eval()
    oe = cvig_fov.FOV_DSM(circ_padding=True, weights=wts).cuda().eval()
    strip = synth.images_u8(92, 1, (3, 150, 170))
    photo = torch.from_numpy(synth.images_u8(92, 2, (3, 90, 120)))
    src = heatmap.ArrayTileSource(strip, origin_x=1000., origin_y=5000., pixel_size=1.5)
    bounds, edge, offset = (1030., 4840., 1130., 4960.), 90., 45.     # windows partly off the raster's left/top edge
    csv = str(tmp_path / 'geomatch.csv')
    df = heatmap.sweep(3, bounds, edge, offset, fov, None, None, csv, tile_source=src, surface_encoder=se,
                       overhead_encoder=oe, photo=photo, batch_size=4)
    back = pd.read_csv(csv)
    assert list(back.columns) == ['x', 'y', 'orientation', 'dissimilarity', 'score'] and len(back) == len(df) == 9
    ce, cn, windows = heatmap.tile_windows(bounds, edge, offset)
    np.testing.assert_allclose(back['x'], ce)
    np.testing.assert_allclose(back['y'], cn)
    # oracle, tile by tile, with its own window cut
    wt = {k: (torch.from_numpy(v[0]), torch.from_numpy(v[1])) for k, v in wts.items()}
    su = O.fov_dsm_forward(O.image_normalization(O.resize_bilinear(photo, (128, O.surface_width(fov)))).unsqueeze(0), wt, False)
    ovs = []
    st = torch.from_numpy(strip)
    for (ulx, uly, lrx, lry) in windows:
        x0, y0 = int(np.floor((ulx - 1000.) / 1.5 + 0.001)), int(np.floor((5000. - uly) / 1.5 + 0.001))
        tile = torch.zeros((3, 60, 60))
        for yy in range(60):
            for xx in range(60):
                if 0 <= y0 + yy < 150 and 0 <= x0 + xx < 170:
                    tile[:, yy, xx] = st[:, y0 + yy, x0 + xx]
        ovs.append(O.polar_transform(O.image_normalization(O.resize_bilinear(tile, (256, 256)))))
    ov = O.fov_dsm_forward(torch.stack(ovs), wt, True)
    ori, dist = O.match(ov, su)
    np.testing.assert_array_equal(back['orientation'].to_numpy(), (ori.squeeze() * 360 / 64 - 180).numpy())
    np.testing.assert_allclose(back['dissimilarity'].to_numpy(), dist.squeeze().numpy(), atol=1e-4)
    np.testing.assert_allclose(back['score'].to_numpy(), torch.exp(10. * (1. - dist.squeeze())).numpy(), rtol=2e-3)


@pytest.mark.gpu
def test_reference_tool_classes_reproduce_the_batched_path(tmp_path):
    """ImageDataset / TileDataset / ResizeSurface / ResizeOverhead / ImageNormalization / PolarTransform of the reference
    tool (tools/heatmap/heatmap.py:35-110), composed per sample, give the tensors the batched sweep feeds the encoders."""
    import torch
    from PIL import Image
    from witw_amd import cvig_fov, heatmap, ops
    g = np.random.Generator(np.random.Philox(key=[12, 1]))
    strip = g.integers(0, 256, size=(3, 90, 120)).astype(np.float32)
    src = heatmap.ArrayTileSource(strip, 1000.0, 2000.0, 0.5)
    _ce, _cn, windows = heatmap.tile_windows((1005.0, 1960.0, 1045.0, 1995.0), 20.0, 10.0)
    compose = [heatmap.ResizeOverhead(), heatmap.ImageNormalization(), heatmap.PolarTransform()]
    ds = heatmap.TileDataset(src, windows)
    assert len(ds) == len(windows)
    d = ds[3]
    for t in compose:
        d = t(d)
    tiles = heatmap._cut_tiles(src.read_strip(cvig_fov.device), src, [windows[3]])
    x = ops.resize_bilinear(tiles, (256, 256), cvig_fov.Globals.img_mean, cvig_fov.Globals.img_std)
    ref = ops.polar_transform(x, 128, 512)[0]
    np.testing.assert_allclose(d['polar'].cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-6)
    # photo side
    photo = g.integers(0, 256, size=(48, 80, 3)).astype(np.uint8)
    path = str(tmp_path / 'photo.png')
    Image.fromarray(photo).save(path)
    s = heatmap.ImageDataset([path], transform=None)[0]
    for t in (heatmap.ResizeSurface(70), heatmap.ImageNormalization()):
        s = t(s)
    ref_s = ops.resize_bilinear(torch.from_numpy(photo.astype(np.float32).transpose(2, 0, 1)).unsqueeze(0).to(cvig_fov.device).contiguous(),
                                (128, 99), cvig_fov.Globals.img_mean, cvig_fov.Globals.img_std)[0]
    assert s['image'].shape == (3, 128, 99)
    np.testing.assert_allclose(s['image'].cpu().numpy(), ref_s.cpu().numpy(), rtol=0, atol=2e-6)
