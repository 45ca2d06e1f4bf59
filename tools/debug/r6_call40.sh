cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call40; mkdir -p $O
D=$(mktemp -d /tmp/witw_e2e_XXXX)
for t in 512 1024 256 512; do
WITW_SELFSYNC_THREADS=$t timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 4 --e2e-pairs 8192 --e2e-dir $D --device-entropy all --detail-out $O/e2e_all_t$t.json > /dev/null 2> $O/e2e_all_t$t.err
python3 -c "
import json; d=json.load(open('$O/e2e_all_t$t.json')); print('e2e bf16 all, 4 workers, $t threads per file:', d['value'], d['steady_state_pairs_per_s'], {k[:12]: v for k, v in d['stage_pairs_per_s'].items()})"
done
rm -rf $D
