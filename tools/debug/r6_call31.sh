cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call31; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_jpeg_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for t in 256 512 1024; do WITW_SELFSYNC_THREADS=$t python3 tools/debug/selfsync_bench.py 2>&1 | grep -v amdgpu | tee -a $O/selfsync_bench.txt; done
for bl in 2 1; do
rocprofv3 --kernel-trace --stats -d $O/prof$bl -o huff -- python3 tools/debug/jpeg_huff_bench.py 0 $bl > $O/prof$bl.log 2>&1
grep "device decode" $O/prof$bl.log
python3 tools/debug/rocprof_db.py $O/prof$bl jpeg_ | grep avg
done
D=$(mktemp -d /tmp/witw_e2e_XXXX)
for wk in 4; do
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers $wk --e2e-pairs 8192 --e2e-dir $D --device-entropy all --detail-out $O/e2e_all_w$wk.json > /dev/null 2> $O/e2e_all_w$wk.err
python3 -c "
import json; d=json.load(open('$O/e2e_all_w$wk.json')); print('e2e bf16 all, $wk workers:', d['value'], d['steady_state_pairs_per_s'], {k[:30]: v for k, v in d['stage_pairs_per_s'].items()})"
done
rm -rf $D
