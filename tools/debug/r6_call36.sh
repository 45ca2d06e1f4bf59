cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call36; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_jpeg_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python3 tools/debug/selfsync_bench.py > $O/log.txt 2>&1
grep "device decode" $O/log.txt
python3 tools/debug/rocprof_db.py $O/prof selfsync | grep "us grid" | awk '{print $1}' | tr '\n' ' '; echo
