import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from witw_amd import ops, synth
dev = torch.device('cuda:0')
mean, std = [0.485, 0.456, 0.406, 0.45, 0.45], [0.229, 0.224, 0.225, 0.22, 0.22]
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, C, Hi, Wi) in ((128, 3, 512, 512), (128, 5, 512, 512), (3, 3, 500, 470), (2, 1, 300, 777)):
    x = torch.from_numpy(synth.images_u8(7, 1, (B, C, Hi, Wi))).to(dev)
    nd = 3 if C == 5 else None
    ref = ops.polar_transform(ops.resize_bilinear(x, (256, 256), mean[:C], std[:C], nd))
    got = ops.polar_from_raw(x, mean=mean[:C], std=std[:C], n_div255=nd)
    print((B, C, Hi, Wi), 'equal', torch.equal(ref, got), float((ref - got).abs().max()))
    t_sep = timed(lambda: ops.polar_transform(ops.resize_bilinear(x, (256, 256), mean[:C], std[:C], nd)))
    t_f = timed(lambda: ops.polar_from_raw(x, mean=mean[:C], std=std[:C], n_div255=nd))
    nbytes = x.numel() * 4 + ref.numel() * 4
    print('   separate %.1f us, fused %.1f us, fused %.2f TB/s (raw + out bytes %.0f MB)' % (t_sep, t_f, nbytes / t_f / 1e6, nbytes / 1e6))
# descriptor sources: u8 HWC of mixed sizes, fp32 CHW of mixed sizes
g = np.random.Generator(np.random.Philox(key=[5, 5]))
for kind in (1, 0):
    imgs, rows, keep = [], [], []
    for (h, w) in ((512, 512), (300, 411), (750, 750), (256, 256), (513, 200)) * 8:
        if kind == 1:
            t = torch.from_numpy(g.integers(0, 256, (h, w, 3), dtype=np.uint8)).to(dev)
            rows.append((t.data_ptr(), h, w, 0, 3))
        else:
            t = torch.from_numpy(g.integers(0, 256, (3, h, w)).astype(np.float32)).to(dev)
            rows.append((t.data_ptr(), h, w, 0, 3))
        keep.append(t)
    desc = torch.tensor(rows, dtype=torch.int64).to(dev)
    ref = ops.polar_transform(ops.resize_batched(desc, len(rows), 3, (256, 256), kind=kind, mean=mean[:3], std=std[:3]))
    got = ops.polar_from_raw(desc=desc, kind=kind, batch=len(rows), channels=3, mean=mean[:3], std=std[:3])
    print('desc kind', kind, 'equal', torch.equal(ref, got))
