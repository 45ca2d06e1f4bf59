cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call22; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_jpeg_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for bl in 4 2 1; do python3 tools/debug/jpeg_huff_bench.py 0 $bl 2>&1 | grep -v amdgpu | tee -a $O/huff_bench.txt; done
for bl in 2 1; do
rocprofv3 --kernel-trace --stats -d $O/prof$bl -o huff -- python3 tools/debug/jpeg_huff_bench.py 0 $bl > $O/prof$bl.log 2>&1
python3 tools/debug/rocprof_db.py $O/prof$bl jpeg_ | grep avg
done
for bl in 1; do
D=$(mktemp -d /tmp/witw_e2e_XXXX)
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 4 --e2e-pairs 8192 --e2e-dir $D --jpeg-restart-blocks $bl --detail-out $O/e2e_rst${bl}_w4.json > /dev/null 2> $O/e2e_rst${bl}_w4.err
python3 -c "
import json; d=json.load(open('$O/e2e_rst${bl}_w4.json')); print('e2e bf16 restart blocks $bl, 4 workers:', d['value'], d['steady_state_pairs_per_s'], {k[:30]: v for k, v in d['stage_pairs_per_s'].items()})"
rm -rf $D
done
