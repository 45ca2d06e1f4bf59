#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r05b
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_bf16_train_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python3 tools/bench_wgrad_bf16.py --layers "L19,L25,L27,sem L0,sem L5" 2>&1 | grep -v amdgpu.ids
python3 bench.py --mode train --precision bf16 > $O/bench_bf16_train.json 2> $O/bench_bf16_train.err
python3 bench.py --model semantic --mode train --precision bf16 > $O/bench_sem_bf16_train.json 2> $O/bench_sem_bf16_train.err
python3 - <<PY
import json
for f in ('bench_bf16_train','bench_sem_bf16_train'):
    d=json.load(open('$O/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline'].get('wgrad_bf16_tflops_incl_layout_passes'), d['roofline'].get('whole_step_frac'), d['loss'])
PY
