#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r05b
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_bf16_train_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
rocprofv3 --kernel-trace --stats -d $O/prof_wg -o p --output-format csv -- python3 tools/bench_wgrad_bf16.py --layers "L17,L19,L21,L23,L25,L27" > $O/wg.txt 2> $O/prof_wg.log
grep -v amdgpu $O/wg.txt
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$O/prof_wg/p_kernel_stats.csv')))[:12]:
    print('%-90s calls %4s avg %8.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
