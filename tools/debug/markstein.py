import numpy as np
rng = np.random.default_rng(1)
def check(b, vals):
    b32 = np.float32(b)
    y = np.float32(1.0 / np.float64(b32))            # RN(1/b) (double rounding risk checked below)
    # verify y is the correctly rounded reciprocal: |1/b - y| minimal among neighbours
    cands = np.array([np.nextafter(y, np.float32(0)), y, np.nextafter(y, np.float32(10))], dtype=np.float32)
    best = cands[np.argmin(np.abs(1.0 / np.float64(b32) - cands.astype(np.float64)))]
    assert best == y
    a = vals.astype(np.float32)
    q0 = (a * y).astype(np.float32)
    r = (a.astype(np.float64) - np.float64(b32) * q0.astype(np.float64)).astype(np.float32)   # exact in double, fma rounds once
    q1 = (q0.astype(np.float64) + r.astype(np.float64) * np.float64(y)).astype(np.float32)   # fma: one rounding (double has room)
    ref = (a / b32).astype(np.float32)
    bad = q1 != ref
    return int(bad.sum()), a[bad][:5], q1[bad][:5], ref[bad][:5]
divs = [255.0, 0.229, 0.224, 0.225, 0.22]
for b in divs:
    tot = 0
    for rep in range(8):
        # values as they occur: v in [0,255] with arbitrary mantissas; (v/255 - mean) in [-0.5, 0.6]; plus wide-exponent sweep
        vals = np.concatenate([rng.random(1 << 23) * 255, rng.random(1 << 23) - 0.49, np.exp(rng.uniform(-60, 20, 1 << 22)) * rng.choice([-1, 1], 1 << 22),
                               np.arange(0, 256, dtype=np.float64), np.arange(0, 1 << 16) / 256.0])
        n, a, q, r = check(b, vals)
        tot += n
        if n: print(b, 'MISMATCH', n, a, q, r)
    print(b, 'mismatches', tot)
