cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call42; mkdir -p $O
D=$(mktemp -d /tmp/witw_e2e_XXXX)
for pr in none -1 none -1; do
if [ $pr = none ]; then unset WITW_COMPUTE_PRIORITY; else export WITW_COMPUTE_PRIORITY=$pr; fi
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 4 --e2e-pairs 8192 --e2e-dir $D --device-entropy all --detail-out $O/e2e_all.json > /dev/null 2> $O/e2e_all.err
python3 -c "
import json; d=json.load(open('$O/e2e_all.json')); print('e2e bf16 all, 4 workers, compute priority $pr:', d['value'], d['steady_state_pairs_per_s'], {k[:12]: v for k, v in d['stage_pairs_per_s'].items()})"
done
rm -rf $D
