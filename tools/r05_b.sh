#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r05b
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_bf16_train_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python3 tools/bench_wgrad_bf16.py --layers "sem L0,L25,L27" 2>&1 | grep -v amdgpu | cut -c1-120
python3 bench.py --model semantic --mode train --precision bf16 --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
