"""jpeg_huffman_kernel alone: 128 synthetic pairs with restart markers, packed as the data path packs them; HIP events around the
device entropy decode and around the whole decode_packed of each side."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from PIL import Image
from witw_amd import jpeg, _lib, ops

def picture(g, h, w):
    small = g.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(small).resize((w + 16, h + 16), Image.BICUBIC))[8:8 + h, 8:8 + w]
    fine = g.integers(-12, 13, size=(h, w, 3))
    return np.clip(img.astype(np.int16) + fine, 0, 255).astype(np.uint8)

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = np.random.Generator(np.random.Philox(key=[1, 2]))
dev = torch.device('cuda:0')
for (h, w) in ((512, 512), (224, 224)):
    files = []
    for i in range(16):
        b = io.BytesIO(); Image.fromarray(picture(g, h, w)).save(b, 'JPEG', quality=90, **({'restart_marker_blocks': blocks} if blocks else {'restart_marker_rows': rows})); files.append(b.getvalue())
    items = [jpeg.open_file(files[i % 16]) for i in range(128)]
    t0 = time.perf_counter(); buf, desc, _k = jpeg.pack(items); t_pack = time.perf_counter() - t0
    dbuf = buf.pin_memory().to(dev)
    for _ in range(2): keep, table = jpeg.decode_packed(dbuf, desc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): keep, table = jpeg.decode_packed(dbuf, desc)
    e1.record(); torch.cuda.synchronize()
    print('rows %d blocks %d ' % (rows, blocks) + '%dx%d: %d files, %.1f KB each, block %.1f MB, host scan+pack %.2f ms, device decode (huffman + idct + rgb) %.3f ms per batch, errors %d'
          % (h, w, len(items), len(files[0]) / 1e3, buf.numel() / 1e6, t_pack * 1e3, e0.elapsed_time(e1) / 5, jpeg.entropy_errors()))
