cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call12; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_bench_launch_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
S=$(date +%s)
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "default run rc=$? $(( $(date +%s) - S )) s wall, line $(wc -c < $O/bench.json) bytes"
grep "^\[bench" $O/bench.err
cp bench_detail.json $O/bench_detail.json
cut -c1-3000 $O/bench.json
