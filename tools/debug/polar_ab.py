import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from witw_amd import ops, synth
dev = torch.device('cuda:0')
x = torch.from_numpy(synth.images_u8(7, 1, (128, 3, 512, 512))).to(dev)
mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
def timed(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(os.environ.get('WITW_LIB', 'head'), 'fused %.1f us' % timed(lambda: ops.polar_from_raw(x, mean=mean, std=std)),
      'resize %.1f us' % timed(lambda: ops.resize_bilinear(x, (256, 256), mean, std)))
