"""batch_sweep anomaly: PairEmbedder's bf16 hipGraph at B = 64 after ALL the training side blocks ran in the process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from witw_amd import cvig_fov, ops, synth, parallel
dev = torch.device('cuda:0')

def timed_fn(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def point(tag, mode='thread_local'):
    b = bench.StepBench('fov', 'infer', 'bf16', 64, 360, 0, 1, dev, pair=True)
    with torch.no_grad():
        s, p = b.preprocess(b.ground_raw, b.ov_raw)
        t_eager = timed_fn(lambda: b.pair._plain(s, p))
        g = parallel.CapturedStep(lambda a_, b_: b.pair._plain(a_, b_), [s, p], warmup=1, capture_error_mode=mode)
        t_replay = timed_fn(lambda: g.graph.replay())
        t_call = timed_fn(lambda: g(s, p))
        t_step = timed_fn(b.step)
    print('%-28s encoders eager %.3f ms, graph replay %.3f ms, replay + copies %.3f ms, whole step (PairEmbedder) %.3f ms  mem reserved %.1f GB'
          % (tag, t_eager, t_replay, t_call, t_step, torch.cuda.memory_reserved() / 1e9), flush=True)
    del b, g; torch.cuda.empty_cache()

point('fresh process')
for spec in (('fov', 'train', 'fp32'), ('semantic', 'infer', 'bf16'), ('semantic', 'train', 'bf16'), ('fov', 'train', 'bf16')):
    t = bench.StepBench(spec[0], spec[1], spec[2], 128, 360, 0, 1, dev).run(3, 2); del t; torch.cuda.empty_cache()
bench.baseline_bench(type('A', (), {'no_cpu_baseline': True})(), dev, full=False); torch.cuda.empty_cache()
point('after the training blocks')
point('... global capture mode', 'global')
base = bench.StepBench('fov', 'infer', 'fp32', 128, 360, 0, 1, dev, pair=False).run(3, 1)
b128 = bench.StepBench('fov', 'infer', 'bf16', 128, 360, 0, 1, dev, share=base, pair=False).run(5, 2); del b128
point('... with the fp32 base alive')
