cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call21; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_jpeg_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for bl in 8 4 2 1; do python3 tools/debug/jpeg_huff_bench.py 0 $bl 2>&1 | grep -v amdgpu | tee -a $O/huff_bench.txt; done
python3 tools/debug/jpeg_huff_bench.py 1 0 2>&1 | grep -v amdgpu | tee -a $O/huff_bench.txt
rocprofv3 --kernel-trace --stats -d $O/prof -o huff -- python3 tools/debug/jpeg_huff_bench.py 0 2 > $O/prof.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob('$O/prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print('%-90s %5s %9.3f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
for bl in 2 1; do
D=$(mktemp -d /tmp/witw_e2e_XXXX)
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 4 --e2e-pairs 8192 --e2e-dir $D --jpeg-restart-blocks $bl --detail-out $O/e2e_rst${bl}_w4.json > /dev/null 2> $O/e2e_rst${bl}_w4.err
python3 -c "
import json; d=json.load(open('$O/e2e_rst${bl}_w4.json')); print('e2e bf16 restart blocks $bl, 4 workers:', d['value'], d['steady_state_pairs_per_s'], {k[:30]: v for k, v in d['stage_pairs_per_s'].items()})"
rm -rf $D
done
