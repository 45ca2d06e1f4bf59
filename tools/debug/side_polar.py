"""Experiment: the overhead side's preprocessing (polar_from_raw) on a side stream under the surface encoder."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from witw_amd import cvig_fov, cvig_semantic, ops, synth
from oracle import cvig_fov_oracle as O


def main():
    dev = torch.device('cuda:0')
    B = 128
    for name, mod, C, prec in (('semantic bf16', cvig_semantic, 5, 'bf16'), ('fov bf16', cvig_fov, 3, 'bf16'), ('fov fp32', cvig_fov, 3, 'fp32')):
        w = synth.fov_dsm_weights(5, in_channels=C)
        g = torch.from_numpy(synth.images_u8(3, C, (B, C, 224, 224))).to(dev)
        o = torch.from_numpy(synth.images_u8(4, C, (B, C, 512, 512))).to(dev)
        mean, std = (list(O.SEM_MEAN), list(O.SEM_STD)) if C == 5 else (list(O.IMG_MEAN), list(O.IMG_STD))
        se = mod.FOV_DSM(False, weights=w).to(dev).eval()
        oe = mod.FOV_DSM(True, weights=w).to(dev).eval()
        f = (lambda e, x: e.forward_bf16(x)) if prec == 'bf16' else (lambda e, x: e(x))
        s2 = torch.cuda.Stream()

        def one():
            with torch.no_grad():
                s = ops.resize_bilinear(g, (128, 512), mean, std, 3)
                p = ops.polar_from_raw(o, mean=mean, std=std, n_div255=3)
                return f(se, s), f(oe, p)

        def two():
            main_s = torch.cuda.current_stream()
            with torch.no_grad():
                s2.wait_stream(main_s)
                with torch.cuda.stream(s2):
                    p = ops.polar_from_raw(o, mean=mean, std=std, n_div255=3)
                s = ops.resize_bilinear(g, (128, 512), mean, std, 3)
                a = f(se, s)
                main_s.wait_stream(s2)
                p.record_stream(main_s)
                return a, f(oe, p)
        n = 20 if prec == 'bf16' else 5
        res = {}
        for label, fn in (('one', one), ('two', two), ('one', one), ('two', two)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                out = fn()
            torch.cuda.synchronize()
            res.setdefault(label, []).append((time.perf_counter() - t0) / n * 1e3)
        a1, b1 = one()
        a2, b2 = two()
        torch.cuda.synchronize()
        print('%-14s in order %s ms | polar on a side stream %s ms | same bits %s' % (name, ['%.3f' % v for v in res['one']], ['%.3f' % v for v in res['two']],
                                                                                 bool(torch.equal(a1, a2) and torch.equal(b1, b2))), flush=True)


if __name__ == '__main__':
    main()
