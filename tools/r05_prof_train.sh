#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r05c
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_bf16_train -o p --output-format csv -- python3 bench.py --mode train --precision bf16 --steps 5 --warmup 2 > $O/bf16_train_under_rocprof.json 2> $O/prof_bf16_train.log &&
rocprofv3 --kernel-trace --stats -d $O/prof_sem_bf16_train -o p --output-format csv -- python3 bench.py --model semantic --mode train --precision bf16 --steps 5 --warmup 2 > $O/sem_bf16_train_under_rocprof.json 2> $O/prof_sem_bf16_train.log
rm -f $O/prof*/p_kernel_trace.csv
python3 - <<PY
import csv
for d in ('prof_bf16_train','prof_sem_bf16_train'):
    print('=====',d)
    rows=list(csv.DictReader(open('$O/%s/p_kernel_stats.csv'%d)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows[:26]:
        n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:64]
        print('%-64s calls %5s avg %8.1f us tot/step %7.3f ms %5.1f%%'%(n,r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/7e6,100*float(r['TotalDurationNs'])/tot))
    print('total ms/step', tot/7e6)
PY
