"""jpeg_selfsync_kernel alone: 128 marker-less synthetic pairs packed as the data path packs them (jpeg.DEVICE_ENTROPY = 'all')."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from PIL import Image
from witw_amd import jpeg
jpeg.DEVICE_ENTROPY = 'all'
g = np.random.Generator(np.random.Philox(key=[1, 2]))
def picture(h, w):
    small = g.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(small).resize((w + 16, h + 16), Image.BICUBIC))[8:8 + h, 8:8 + w]
    return np.clip(img.astype(np.int16) + g.integers(-12, 13, size=(h, w, 3)), 0, 255).astype(np.uint8)
dev = torch.device('cuda:0')
for (h, w) in ((512, 512), (224, 224)):
    files = []
    for i in range(16):
        b = io.BytesIO(); Image.fromarray(picture(h, w)).save(b, 'JPEG', quality=90); files.append(b.getvalue())
    items = [jpeg.open_file(files[i % 16]) for i in range(128)]
    t0 = time.perf_counter(); buf, desc, _k = jpeg.pack(items); t_pack = time.perf_counter() - t0
    dbuf = buf.pin_memory().to(dev)
    for _ in range(2): keep, table = jpeg.decode_packed(dbuf, desc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): keep, table = jpeg.decode_packed(dbuf, desc)
    e1.record(); torch.cuda.synchronize()
    print('threads/file %s  %dx%d: host parse+pack %.2f ms, device decode (self-sync huffman + idct + rgb) %.3f ms per batch of 128, errors %d'
          % (os.environ.get('WITW_SELFSYNC_THREADS', '512'), h, w, t_pack * 1e3, e0.elapsed_time(e1) / 5, jpeg.entropy_errors()))
