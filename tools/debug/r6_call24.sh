cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call24; mkdir -p $O
D=$(mktemp -d /tmp/witw_e2e_XXXX)
for wk in 4 8; do
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers $wk --e2e-pairs 8192 --e2e-dir $D --device-entropy all --detail-out $O/e2e_all_w$wk.json > /dev/null 2> $O/e2e_all_w$wk.err
python3 -c "
import json; d=json.load(open('$O/e2e_all_w$wk.json')); print('e2e bf16 all, $wk workers:', d['value'], d['steady_state_pairs_per_s'], {k[:30]: v for k, v in d['stage_pairs_per_s'].items()})"
done
timeout -k 10 400 python3 bench.py --mode e2e --workers 4 --e2e-pairs 2048 --e2e-dir $D --device-entropy all --detail-out $O/e2e_all_fp32.json > /dev/null 2> $O/e2e_all_fp32.err
python3 -c "
import json; d=json.load(open('$O/e2e_all_fp32.json')); print('e2e fp32 all, 4 workers:', d['value'], d['steady_state_pairs_per_s'], {k[:30]: v for k, v in d['stage_pairs_per_s'].items()})"
timeout -k 10 400 python3 bench.py --mode e2e --workers 16 --e2e-pairs 2048 --e2e-dir $D --device-entropy off --detail-out $O/e2e_host_fp32.json > /dev/null 2> $O/e2e_host_fp32.err
python3 -c "
import json; d=json.load(open('$O/e2e_host_fp32.json')); print('e2e fp32 host, 16 workers:', d['value'], d['steady_state_pairs_per_s'], {k[:30]: v for k, v in d['stage_pairs_per_s'].items()})"
rm -rf $D
