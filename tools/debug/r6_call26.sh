cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call26; mkdir -p $O
( while true; do date >> $O/heartbeat.txt; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
t0=$(date +%s)
timeout -k 10 900 python3 bench.py > $O/bench_line.json 2> $O/bench.err; rc=$?
echo "bench rc=$rc wall $(( $(date +%s) - t0 )) s, line $(wc -c < $O/bench_line.json) bytes"
cp bench_detail.json $O/ 2>/dev/null
python3 - <<PY
import json
d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('frac'))
for k,v in d.items():
    if k.startswith('e2e') or k=='sides': print(k, json.dumps(v)[:1500])
PY
