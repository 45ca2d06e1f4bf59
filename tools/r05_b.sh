#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r05b
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_bf16_gpu.py tests/test_bf16_train_gpu.py tests/test_drivers2_gpu.py tests/test_large_grid_parity_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python3 bench.py --model semantic --mode train --precision bf16 --steps 10 --warmup 3 > $O/bench_sem_bf16_train.json 2> $O/bench_sem_bf16_train.err
python3 - <<PY
import json
for f in ('bench_sem_bf16_train',):
    d=json.load(open('$O/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline'].get('wgrad_bf16_tflops_incl_layout_passes'), d['roofline'].get('whole_step_frac'), d['loss'])
PY
rocprofv3 --kernel-trace --stats -d $O/prof_sem -o p --output-format csv -- python3 bench.py --model semantic --mode train --precision bf16 --steps 5 --warmup 2 > /dev/null 2> $O/prof_sem.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/prof_sem/p_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:64]
    print('%-64s calls %5s avg %8.1f us tot/step %7.3f ms'%(n,r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/7e6))
print('total', tot/7e6)
PY
