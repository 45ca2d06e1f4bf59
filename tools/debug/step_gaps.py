"""Busy time vs wall span of the kernels of one bench step from a rocprofv3 kernel trace (tools/debug/kstats.sh leaves it under
gpurun_out/kstats): python tools/debug/step_gaps.py <kernel_trace.csv> <marker kernel substring>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mark = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
a, b = mark[-3], mark[-2]
t0 = int(rows[a]['Start_Timestamp'])
busy, prev_end, gaps = 0, None, []
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    if prev_end is not None and s - prev_end > 3000:
        gaps.append(((s - prev_end) / 1e3, r['Kernel_Name'][:60]))
    prev_end = max(prev_end or 0, e)
span = int(rows[b]['Start_Timestamp']) - t0
print('kernels %d  busy %.1f us  span %.1f us  idle %.1f us' % (b - a, busy / 1e3, span / 1e3, (span - busy) / 1e3))
for g in sorted(gaps, reverse=True)[:12]:
    print('  gap %.1f us before %s' % g)
