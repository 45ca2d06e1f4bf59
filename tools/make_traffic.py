#!/usr/bin/env python
"""profiles/traffic.json from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs as
MI355X_MICROARCH.md prescribes) over `bench.py --steps 2 --warmup 1 --no-cpu-baseline` in one or more modes.

    python tools/make_traffic.py [batch] TAG:<fetch counter_collection.csv>:<write counter_collection.csv> ... > profiles/traffic.json

TAG names the bench mode of the run ('infer', 'train', 'bf16', 'bf16_train', ...). bench.py looks up
'<dominant kernel>_bytes_per_launch_B<batch>' (infer / bf16) or '..._B<batch>_<tag>' (other modes).
"""
import csv
import json
import sys
from collections import defaultdict

DOMINANT = ('conv3x3_nhwc_f32_kernel<128,1,false,8', 'conv3x3_nhwc_bf16_kernel<128,1,false,8', 'conv3x3_nhwc_f16x3_kernel<128,1,false,8',
            'conv3x3_bf16_s16_kernel<false,false>', 'conv3x3_bf16_s16_kernel<false,true>')


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].replace(' ', '')
        acc[k].append(float(r['Counter_Value']))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


def main():
    args = sys.argv[1:]
    batch = 128
    if args and args[0].isdigit():
        batch = int(args.pop(0))
    out = {'_note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes per bench mode over `bench.py --steps 2 '
                    '--warmup 1 --no-cpu-baseline` (B=%d, fov 360); bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024: gfx950 FETCH_SIZE '
                    'reports half the bytes of 16-B/lane coalesced reads (MI355X_MICROARCH.md, HBM section); averaged over the '
                    'launches of each kernel in that mode (train modes: forward and dgrad launches together); FETCH_SIZE counts '
                    'L2 misses served by the fabric (Infinity Cache hits included)' % batch}
    import hashlib
    import os
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'witw_amd', 'csrc')
    # bench.py quotes these figures only while the kernel sources are the ones the counters were taken on
    out['_kernel_sources_sha16'] = {f: hashlib.sha256(open(os.path.join(csrc, f), 'rb').read()).hexdigest()[:16]
                                    for f in ('conv3x3.hip', 'conv3x3_bf16.hip', 'conv3x3_f16x3.hip', 'common.h')}
    for spec in args:
        tag, fpath, wpath = spec.split(':')
        fetch, write = per_kernel(fpath, 'FETCH_SIZE'), per_kernel(wpath, 'WRITE_SIZE')
        kernels = {}
        for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
            if k not in write or 'at::' in k or 'rocclr' in k:
                continue
            n, f = fetch[k]
            w = write[k][1]
            kernels[k] = {'launches': n, 'FETCH_SIZE_KB_avg': round(f, 1), 'WRITE_SIZE_KB_avg': round(w, 1),
                          'hbm_bytes_per_launch_corrected': int((2 * f + w) * 1024)}
            compact = k.replace(' ', '')
            if compact.startswith(DOMINANT):
                suffix = '' if tag in ('infer', 'bf16', 'fp16x3') else '_' + tag
                out['%s_bytes_per_launch_B%d%s' % (compact, batch, suffix)] = kernels[k]['hbm_bytes_per_launch_corrected']
        out['kernels_%s' % tag] = kernels
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
