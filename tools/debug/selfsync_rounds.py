"""DIAGNOSTIC (library built with -DWITW_SS_DIAG=1, WITW_LIB=...): rounds and re-decoded subsequences of jpeg_selfsync_kernel per file."""
import io, os, sys
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
    items = [jpeg.open_file(f) for f in files]
    buf, desc, _k = jpeg.pack(items)
    keep, table = jpeg.decode_packed(buf.to(dev), desc)
    torch.cuda.synchronize()
    e = jpeg._ERRORS[-1].cpu().numpy()
    print('%dx%d threads %s: rounds %s' % (h, w, os.environ.get('WITW_SELFSYNC_THREADS', '512'), list(e >> 20)))
    print('   re-decoded subsequences in all %s' % list(e & 0xfffff))
