cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_match_dft_gpu.py tests/test_retrieval_fullsize_gpu.py tests/test_match_gpu.py tests/test_parallel_world8_gpu.py tests/test_parallel_gpu.py -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for i in 1 2 3; do
python3 bench.py --mode retrieval --match dft --steps 2 --warmup 1 --detail-out $O/d.json 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['roofline']['frac'], d['index_exact'])"
done
rocprofv3 --kernel-trace --stats -o p --output-format csv -d $O/prof_retr -- python3 bench.py --mode retrieval --match dft --steps 2 --warmup 1 --detail-out $O/d.json > $O/retr.json 2> $O/retr.log
python3 tools/kernel_gaps.py $O/prof_retr/p_kernel_trace.csv 30 > $O/retr_gaps.txt 2>&1
rm -f $O/prof_retr/p_kernel_trace.csv
head -30 $O/retr_gaps.txt
