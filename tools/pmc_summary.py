#!/usr/bin/env python
"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV (one row per dispatch and counter)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else ''
agg = defaultdict(lambda: defaultdict(list))
dur = defaultdict(dict)
for r in csv.DictReader(open(path)):
    k = r['Kernel_Name']
    if match and match not in k:
        continue
    k = k.replace('(anonymous namespace)::', '').split('(')[0]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[k][r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k in sorted(agg, key=lambda k: -sum(dur[k].values())):
    d = list(dur[k].values())
    print('%s  launches %d  avg %.1f us' % (k[:100], len(d), sum(d) / len(d) / 1e3))
    for c, v in sorted(agg[k].items()):
        print('    %-32s %16.1f' % (c, sum(v) / len(v)))
