cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call41; mkdir -p $O
python3 - <<PY
import ctypes, torch
hip = ctypes.CDLL('libamdhip64.so'); lo = ctypes.c_int(); hi = ctypes.c_int()
torch.cuda.init()
print('hipDeviceGetStreamPriorityRange rc', hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), 'least', lo.value, 'greatest', hi.value)
for p in (-1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p); print('priority', p, '->', s.priority)
    except Exception as e:
        print('priority', p, 'refused:', e)
PY
D=$(mktemp -d /tmp/witw_e2e_XXXX)
for pr in 0 1 -1 0 1; do
WITW_STAGING_PRIORITY=$pr timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 4 --e2e-pairs 8192 --e2e-dir $D --device-entropy all --detail-out $O/e2e_all_p$pr.json > /dev/null 2> $O/e2e_all_p$pr.err
python3 -c "
import json; d=json.load(open('$O/e2e_all_p$pr.json')); print('e2e bf16 all, 4 workers, staging priority $pr:', d['value'], d['steady_state_pairs_per_s'], {k[:12]: v for k, v in d['stage_pairs_per_s'].items()})"
done
rm -rf $D
