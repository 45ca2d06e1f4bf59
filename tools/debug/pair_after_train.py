"""Why is PairEmbedder's bf16 hipGraph at B = 64 slow after a training block ran in the same process? (round-6 batch_sweep anomaly)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from witw_amd import cvig_fov, ops, synth, parallel
dev = torch.device('cuda:0')

def timed(sb, n=10):
    for _ in range(3): sb.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): sb.step()
    torch.cuda.synchronize(); return sb.B * n / (time.perf_counter() - t0)

def point(tag):
    a = bench.StepBench('fov', 'infer', 'bf16', 64, 360, 0, 1, dev, pair=False)
    b = bench.StepBench('fov', 'infer', 'bf16', 64, 360, 0, 1, dev, pair=True)
    print(tag, 'plain eager %.0f  pair-embedder %.0f  stats %s' % (timed(a), timed(b), b.pair.stats), flush=True)
    del a, b; torch.cuda.empty_cache()

mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
point('before')
if mode == 'fp32train':
    t = bench.StepBench('fov', 'train', 'fp32', 128, 360, 0, 1, dev); timed(t, 3); del t
elif mode == 'bf16train':
    t = bench.StepBench('fov', 'train', 'bf16', 128, 360, 0, 1, dev); timed(t, 3); del t
elif mode == 'semtrain':
    t = bench.StepBench('semantic', 'train', 'bf16', 128, 360, 0, 1, dev); timed(t, 3); del t
elif mode == 'seminfer':
    t = bench.StepBench('semantic', 'infer', 'bf16', 128, 360, 0, 1, dev); timed(t, 3); del t
elif mode == 'baseline':
    bench.baseline_bench(type('A', (), {'no_cpu_baseline': True})(), dev, full=False)
torch.cuda.empty_cache()
point('after ' + mode)
