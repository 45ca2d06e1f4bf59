cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call34; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_jpeg_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for t in 512 1024; do WITW_SELFSYNC_THREADS=$t python3 tools/debug/selfsync_bench.py 2>&1 | grep -v amdgpu | tee -a $O/selfsync_bench.txt; done
D=$(mktemp -d /tmp/witw_e2e_XXXX)
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 4 --e2e-pairs 8192 --e2e-dir $D --device-entropy all --detail-out $O/e2e_all_w4.json > /dev/null 2> $O/e2e_all_w4.err
python3 -c "
import json; d=json.load(open('$O/e2e_all_w4.json')); print('e2e bf16 all, 4 workers:', d['value'], d['steady_state_pairs_per_s'], {k[:30]: v for k, v in d['stage_pairs_per_s'].items()})"
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 16 --e2e-pairs 8192 --e2e-dir $D --device-entropy off --no-decode-scaling --detail-out $O/e2e_host_w16.json > /dev/null 2> $O/e2e_host_w16.err
python3 -c "
import json; d=json.load(open('$O/e2e_host_w16.json')); print('e2e bf16 host, 16 workers:', d['value'], d['steady_state_pairs_per_s'], {k[:30]: v for k, v in d['stage_pairs_per_s'].items()})"
rm -rf $D
