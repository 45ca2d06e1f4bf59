#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r05b
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_bf16_train_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
rocprofv3 --kernel-trace --stats -d $O/prof_t -o p --output-format csv -- python3 bench.py --mode train --precision bf16 --steps 5 --warmup 2 > $O/t.json 2> $O/prof_t.log
python3 - <<PY
import csv, json
d=json.load(open('$O/t.json')); print(d['value'], d['ms_per_step'])
for r in csv.DictReader(open('$O/prof_t/p_kernel_stats.csv')):
    if 'pack_weights' in r['Name'] or 'reduce' in r['Name']: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3)
PY
