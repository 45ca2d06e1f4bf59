cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call18; mkdir -p $O
for t in 256 512 1024; do
WITW_SELFSYNC_THREADS=$t timeout -k 10 300 python3 -m pytest tests/test_jpeg_gpu.py -x -q -k "selfsync or damaged or mixed" > $O/pytest_$t.log 2>&1; echo "threads $t pytest rc=$?"; tail -1 $O/pytest_$t.log
WITW_SELFSYNC_THREADS=$t python3 tools/debug/selfsync_bench.py 2>&1 | grep -v amdgpu | tee -a $O/selfsync_bench.txt
done
D=$(mktemp -d /tmp/witw_e2e_XXXX)
for t in 256 512; do
WITW_SELFSYNC_THREADS=$t timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 4 --e2e-pairs 8192 --e2e-dir $D --device-entropy all --detail-out $O/e2e_all_$t.json > /dev/null 2> $O/e2e_all_$t.err
python3 -c "
import json; d=json.load(open('$O/e2e_all_$t.json')); print('e2e bf16 all, threads $t, 4 workers:', d['value'], d['steady_state_pairs_per_s'], {k[:30]: v for k, v in d['stage_pairs_per_s'].items()})"
done
WITW_SELFSYNC_THREADS=256 timeout -k 10 400 python3 bench.py --mode e2e --workers 4 --e2e-pairs 2048 --e2e-dir $D --device-entropy all --detail-out $O/e2e_all_fp32.json > /dev/null 2> $O/e2e_all_fp32.err
python3 -c "
import json; d=json.load(open('$O/e2e_all_fp32.json')); print('e2e fp32 all, 4 workers:', d['value'], d['steady_state_pairs_per_s'], {k[:30]: v for k, v in d['stage_pairs_per_s'].items()})"
rm -rf $D
