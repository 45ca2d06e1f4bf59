cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call49; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_retrieval_fullsize_gpu.py tests/test_match_dft_gpu.py tests/test_match_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
python3 bench.py --mode retrieval --match dft --steps 2 --warmup 1 --detail-out $O/d.json > $O/bench_retrieval_dft.json 2> $O/bench_retrieval_dft.err
python3 -c "
import json; d=json.loads(open('$O/bench_retrieval_dft.json').read().strip().splitlines()[-1]); print('retrieval dft:', d['value'], d['ms_per_step'])"
rocprofv3 --kernel-trace --stats -o p --output-format csv -d $O/prof -- python3 bench.py --mode retrieval --match dft --steps 2 --warmup 1 --detail-out $O/d.json > /dev/null 2> $O/prof.log
head -8 $O/prof/p_kernel_stats.csv | cut -c1-150
