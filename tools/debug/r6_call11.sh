cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call11; mkdir -p $O
for nb in 2 4; do
D=$(mktemp -d /tmp/witw_rst_XXXX)
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 4 --e2e-pairs 8192 --jpeg-restart-blocks $nb --e2e-dir $D --detail-out $O/e2e_rst${nb}_w4.json > $O/e2e_rst${nb}_w4.line 2> $O/e2e_rst${nb}_w4.err
echo "rst blocks=$nb w4 rc=$?"; python3 -c "
import json; d=json.load(open('$O/e2e_rst${nb}_w4.json')); print(d['value'], d['steady_state_pairs_per_s'], {k[:40]: v for k, v in d['stage_pairs_per_s'].items()}, d['pcie_bytes_per_pair'])"
rm -rf $D
done
D=$(mktemp -d /tmp/witw_rst_XXXX)
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 16 --e2e-pairs 8192 --e2e-dir $D --no-decode-scaling --detail-out $O/e2e_host_w16.json > $O/e2e_host_w16.line 2> $O/e2e_host_w16.err
python3 -c "
import json; d=json.load(open('$O/e2e_host_w16.json')); print('host entropy w16', d['value'], d['steady_state_pairs_per_s'], {k[:40]: v for k, v in d['stage_pairs_per_s'].items()}, d['pcie_bytes_per_pair'])"
rm -rf $D
