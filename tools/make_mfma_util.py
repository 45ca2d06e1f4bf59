#!/usr/bin/env python
"""profiles/mfma_util.json: matrix-pipe utilisation per kernel from one rocprofv3 PMC pass per bench mode
(`--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`, no trace domains besides --kernel-trace).

    python tools/make_mfma_util.py TAG:<counter_collection.csv> ... > profiles/mfma_util.json

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums
the 8 XCDs; MI355X_MICROARCH.md 'DVFS give-back'). effective_clock_GHz = kernel cycles / wall time of the dispatch.
"""
import csv
import json
import sys
from collections import defaultdict

SIMDS = 256 * 4


def main():
    out = {'_note': 'rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over `bench.py --steps 2 '
                    '--warmup 1 --no-cpu-baseline` per mode (B=128, fov 360); per kernel: averages over its dispatches; mfma_util = '
                    'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); effective clock = (GRBM_GUI_ACTIVE/8) / dispatch '
                    'wall time (reads high on dispatches well under 0.3 ms)'}
    for spec in sys.argv[1:]:
        tag, path = spec.split(':')
        acc = defaultdict(lambda: defaultdict(list))
        dur = defaultdict(dict)
        for r in csv.DictReader(open(path)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
            dur[k][r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        kernels = {}
        for k in sorted(acc, key=lambda k: -sum(dur[k].values())):
            if 'at::' in k or 'rocclr' in k:
                continue
            c = {n: sum(v) / len(v) for n, v in acc[k].items()}
            d = list(dur[k].values())
            ns = sum(d) / len(d)
            cyc = c.get('GRBM_GUI_ACTIVE', 0.) / 8.
            row = {'launches': len(d), 'avg_us': round(ns / 1e3, 1)}
            row.update({n: round(v, 1) for n, v in c.items()})
            if cyc > 0:
                row['mfma_util'] = round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.) / (cyc * SIMDS), 4)
                row['effective_clock_GHz'] = round(cyc / ns, 3)
            kernels[k] = row
            if len(kernels) >= 12:
                break
        out['kernels_%s' % tag] = kernels
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
