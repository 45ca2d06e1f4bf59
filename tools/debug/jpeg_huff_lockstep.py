"""How much of jpeg_huffman_kernel's time is divergence? Files whose MCUs are all identical (a 16 x 16 texture tiled over the image: every
restart interval holds the same bits, the 64 lanes of a wave run in lockstep) against ordinary files of about the same size."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from PIL import Image
from witw_amd import jpeg

g = np.random.Generator(np.random.Philox(key=[3, 4]))
dev = torch.device('cuda:0')
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def ordinary(h, w):
    small = g.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(small).resize((w + 16, h + 16), Image.BICUBIC))[8:8 + h, 8:8 + w]
    return np.clip(img.astype(np.int16) + g.integers(-12, 13, size=(h, w, 3)), 0, 255).astype(np.uint8)
def tiled(h, w):
    t = ordinary(16 * blocks, 16)      # one restart interval's worth of MCUs (blocks MCUs side by side would be 16*blocks wide; vertical is fine for the bits)
    t = ordinary(16, 16 * blocks)
    return np.tile(t, (h // 16, w // (16 * blocks) + 1, 1))[:, :w]
for name, make in (('ordinary', ordinary), ('identical intervals', tiled)):
    for (h, w) in ((512, 512),):
        files = []
        for i in range(16):
            b = io.BytesIO(); Image.fromarray(make(h, w)).save(b, 'JPEG', quality=90, restart_marker_blocks=blocks); files.append(b.getvalue())
        items = [jpeg.open_file(files[i % 16]) for i in range(128)]
        buf, desc, _k = jpeg.pack(items)
        dbuf = buf.pin_memory().to(dev)
        for _ in range(2): keep, table = jpeg.decode_packed(dbuf, desc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): keep, table = jpeg.decode_packed(dbuf, desc)
        e1.record(); torch.cuda.synchronize()
        print('%s, restart every %d MCUs, %dx%d: %.1f KB per file, decode_packed %.3f ms per 128 files, errors %d' % (name, blocks, h, w, len(files[0]) / 1e3, e0.elapsed_time(e1) / 5, jpeg.entropy_errors()))
