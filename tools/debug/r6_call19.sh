cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call19; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_jpeg_gpu.py -x -q > $O/pytest.log 2>&1; echo "staged pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
python3 tools/debug/selfsync_bench.py 2>&1 | grep -v amdgpu | sed 's/threads\/file 256/staged launches/' | tee -a $O/selfsync_bench.txt
WITW_SELFSYNC_THREADS=512 python3 tools/debug/selfsync_bench.py 2>&1 | grep -v amdgpu | tee -a $O/selfsync_bench.txt
D=$(mktemp -d /tmp/witw_e2e_XXXX)
for wk in 4 8; do
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers $wk --e2e-pairs 8192 --e2e-dir $D --device-entropy all --detail-out $O/e2e_all_w$wk.json > /dev/null 2> $O/e2e_all_w$wk.err
python3 -c "
import json; d=json.load(open('$O/e2e_all_w$wk.json')); print('e2e bf16 all (staged), $wk workers:', d['value'], d['steady_state_pairs_per_s'], {k[:30]: v for k, v in d['stage_pairs_per_s'].items()})"
done
timeout -k 10 400 python3 bench.py --mode e2e --workers 4 --e2e-pairs 2048 --e2e-dir $D --device-entropy all --detail-out $O/e2e_all_fp32.json > /dev/null 2> $O/e2e_all_fp32.err
python3 -c "
import json; d=json.load(open('$O/e2e_all_fp32.json')); print('e2e fp32 all (staged), 4 workers:', d['value'], d['steady_state_pairs_per_s'], {k[:30]: v for k, v in d['stage_pairs_per_s'].items()})"
rm -rf $D
