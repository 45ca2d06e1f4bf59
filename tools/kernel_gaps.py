#!/usr/bin/env python
"""Where a pass's time goes BETWEEN kernels: rocprofv3 --kernel-trace CSV (p_kernel_trace.csv) -> busy time, idle time, and the
largest gaps with the kernels on either side (host synchronisations and launch-bound stretches show up here).

    python tools/kernel_gaps.py <p_kernel_trace.csv> [min_gap_us=50] [skip_first_ms=0]"""
import csv
import sys


def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]


def main():
    path = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    if not rows:
        print('no kernels')
        return
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, end = 0, rows[0][0]
    gaps = []
    for s, e, n in rows:
        if s > end:
            gaps.append((s - end, end, n))
            busy += e - s
        else:
            busy += max(0, e - max(s, end))
        end = max(end, e)
    prev = {}
    last_end, last_name = rows[0][1], rows[0][2]
    named = []
    end = rows[0][0]
    for s, e, n in rows:
        if s > end:
            named.append((s - end, last_name, n, (end - t0) / 1e6))
        if e >= end:
            end, last_name = e, n
    print('span %.2f ms, busy %.2f ms, idle %.2f ms in %d gaps (%d kernels)' % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(named), len(rows)))
    big = [g for g in named if g[0] / 1e3 >= min_gap]
    print('gaps >= %.0f us: %d, total %.2f ms' % (min_gap, len(big), sum(g[0] for g in big) / 1e6))
    for g, a, b, at in sorted(big, key=lambda x: -x[0])[:60]:
        print('  %9.1f us at %9.2f ms  after %-60s before %s' % (g / 1e3, at, a, b))
    # histogram of small gaps
    small = [g[0] / 1e3 for g in named if g[0] / 1e3 < min_gap]
    print('gaps < %.0f us: %d, total %.2f ms' % (min_gap, len(small), sum(small) / 1e3))


if __name__ == '__main__':
    main()
