"""What does this box deliver for plain streaming writes / copies? (calibration for the HBM-bound kernels' fractions)"""
import torch
dev = torch.device('cuda:0')
def timed(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (512, 2048):
    a = torch.empty(mb * 1024 * 1024 // 4, device=dev); b = torch.empty_like(a)
    t = timed(lambda: a.fill_(1.0)); print('fill  %4d MB: %.2f TB/s written' % (mb, a.numel() * 4 / t / 1e12))
    t = timed(lambda: b.copy_(a)); print('copy  %4d MB: %.2f TB/s read + written (%.2f each way)' % (mb, 2 * a.numel() * 4 / t / 1e12, a.numel() * 4 / t / 1e12))
    t = timed(lambda: a.sum()); print('sum   %4d MB: %.2f TB/s read' % (mb, a.numel() * 4 / t / 1e12))
