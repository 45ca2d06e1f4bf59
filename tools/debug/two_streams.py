"""Experiment: the two encoders of a pair (independent until the match) on two HIP streams, so that one kernel's drain overlaps the
other encoder's kernels. Prints ms per step for one / two streams, bf16 and fp32 (B = 128)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from witw_amd import cvig_fov, cvig_semantic, ops, synth
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for name, mod, C, prec in (('semantic bf16', cvig_semantic, 5, 'bf16'), ('fov bf16', cvig_fov, 3, 'bf16'), ('fov fp32', cvig_fov, 3, 'fp32')):
    w = synth.fov_dsm_weights(5, in_channels=C)
    xs = torch.from_numpy(synth.normalized_images(5, C, (B, C, 128, 512))).to(dev)
    xo = torch.from_numpy(synth.normalized_images(6, C, (B, C, 128, 512))).to(dev)
    se = mod.FOV_DSM(False, weights=w).to(dev).eval()
    oe = mod.FOV_DSM(True, weights=w).to(dev).eval()
    f = (lambda e, x: e.forward_bf16(x)) if prec == 'bf16' else (lambda e, x: e(x))
    s2 = torch.cuda.Stream()

    def one():
        with torch.no_grad():
            return f(se, xs), f(oe, xo)

    def two():
        main = torch.cuda.current_stream()
        s2.wait_stream(main)
        with torch.no_grad():
            with torch.cuda.stream(s2):
                b = f(oe, xo)
            a = f(se, xs)
        main.wait_stream(s2)
        b.record_stream(main)
        return a, b
    n = 20 if prec == 'bf16' else 5
    res = {}
    for label, fn in (('one', one), ('two', two), ('one', one), ('two', two)):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        res.setdefault(label, []).append((time.perf_counter() - t0) / n * 1e3)
    a1, b1 = one()
    a2, b2 = two()
    torch.cuda.synchronize()
    print('%-14s one stream %s ms | two streams %s ms | same bits %s' % (name, ['%.3f' % v for v in res['one']], ['%.3f' % v for v in res['two']],
                                                                        bool(torch.equal(a1, a2) and torch.equal(b1, b2))))
