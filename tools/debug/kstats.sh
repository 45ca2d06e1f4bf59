#!/bin/bash
# per-kernel average durations of a python script (run through gpurun): bash tools/debug/kstats.sh <script.py> [args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
d=$R/gpurun_out/kstats; rm -rf $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/"$@" > /dev/null 2>&1
f=$(find $d -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print('%-90s calls %6s avg %9.1f us  total %8.2f ms' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
