#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r05b
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_fullsize_properties_gpu.py tests/test_drivers_gpu.py tests/test_drivers2_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 600 python3 bench.py --mode sweep > $O/sweep.json 2> $O/sweep.err || { tail -20 $O/sweep.err; exit 1; }
python3 - <<PY
import json
d=json.load(open('bench_detail.json'))
for p in d['points']:
    print(p['precision'], p['pairs_per_gpu'], p['value'], p['ms_per_step'], p.get('plain_one_stream_eager'), p.get('graph_replay'), p.get('pair_embedder'))
PY
