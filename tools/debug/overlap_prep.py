"""cvig_semantic / cvig_fov bf16 inference step at B=128: the overhead side's preprocessing launch (polar_from_raw, HBM-bound) on a side
stream beside the ground encoder, against the plain order. Alternating, same process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from witw_amd import ops

dev = torch.device('cuda:0')
model = sys.argv[1] if len(sys.argv) > 1 else 'semantic'
sb = bench.StepBench(model, 'infer', 'bf16', 128, 360, 0, 1, dev)
side = torch.cuda.Stream()

def plain():
    with torch.no_grad():
        surface, polar = sb.preprocess(sb.ground_raw, sb.ov_raw)
        su, ov = sb.embed(surface, polar)
        return sb.cvig_fov.evaluate_global_batch(ov, su, 0)

def overlapped():
    with torch.no_grad():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            polar = ops.polar_from_raw(sb.ov_raw, mean=sb.mean, std=sb.std, n_div255=sb.ndiv)
        surface = ops.resize_bilinear(sb.ground_raw, (128, sb.ws), sb.mean, sb.std, sb.ndiv)
        su = sb.se.forward_bf16(surface)
        cur.wait_stream(side)
        polar.record_stream(cur)
        ov = sb.oe.forward_bf16(polar)
        return sb.cvig_fov.evaluate_global_batch(ov, su, 0)

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out

a = plain(); b = overlapped(); torch.cuda.synchronize()
print('same loss', float(a[0]) == float(b[0]), float(a[0]), float(b[0]))
for r in range(3):
    tp, _ = t(plain); to, _ = t(overlapped)
    print('%s bf16 B=128 round %d: plain %.3f ms, overhead preprocessing beside the ground encoder %.3f ms' % (model, r, tp, to))
