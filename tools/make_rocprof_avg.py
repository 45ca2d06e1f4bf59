#!/usr/bin/env python
"""profiles/rocprof_kernel_avg.json: the average duration rocprofv3 --kernel-trace --stats reports for every kernel of a bench mode,
so that bench.py can quote the dominant kernel's roofline on BOTH clocks (HIP events live: `frac_events`; the committed rocprof
table of the same kernel sources: `frac_rocprof`) -- VERDICT r05: the two differed by 5 % on the bf16 kernels and the line named one.

    python tools/make_rocprof_avg.py TAG:<p_kernel_stats.csv>[:<tracked copy under profiles/>] ... > profiles/rocprof_kernel_avg.json

TAG = bench mode of the profiled command ('infer', 'train', 'bf16', 'sem_bf16', 'bf16_train', 'sem_bf16_train')."""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def norm(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].replace(' ', '')


def main():
    csrc = os.path.join(ROOT, 'witw_amd', 'csrc')
    out = {'_note': 'rocprofv3 --kernel-trace --stats over `bench.py [mode flags] --steps 5 --warmup 2 --no-cpu-baseline --no-side-blocks`: '
                    'AverageNs per kernel instantiation; bench.py quotes a figure only while the kernel sources hash as below',
           # the sources of the convolution kernels (the dominant kernels of every StepBench mode): a change to the matching / JPEG /
           # preprocessing sources does not touch them
           '_kernel_sources_sha16': {f: hashlib.sha256(open(os.path.join(csrc, f), 'rb').read()).hexdigest()[:16]
                                     for f in ('common.h', 'lds_frag.h', 'conv3x3.hip', 'conv3x3_bf16.hip', 'conv3x3_bf16_wres.hip',
                                               'conv3x3_f16x3.hip', 'conv3x3_wgrad.hip', 'conv_first.hip', 'conv_first2_bf16.hip',
                                               'wgrad_bf16.hip', 'wgrad_f16x3.hip')}}
    for spec in sys.argv[1:]:
        parts = spec.split(':')
        tag, path = parts[0], parts[1]
        tracked = parts[2] if len(parts) > 2 else path
        rows = {}
        for r in csv.DictReader(open(path)):
            k = norm(r['Name'])
            if 'at::' in k or 'rocclr' in k or 'rocprim' in k:
                continue
            rows[k] = {'calls': int(r['Calls']), 'avg_ms': round(float(r['AverageNs']) / 1e6, 5)}
        out[tag] = {'csv': tracked, 'kernels': rows}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
