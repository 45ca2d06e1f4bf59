cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call37; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_jpeg_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for bl in 2 1; do
rocprofv3 --kernel-trace --stats -d $O/prof$bl -o p -- python3 tools/debug/jpeg_huff_lockstep.py $bl > $O/log$bl.txt 2>&1
grep "decode_packed" $O/log$bl.txt
python3 tools/debug/rocprof_db.py $O/prof$bl jpeg_huffman | grep "us grid" | awk '{print $1}' | tr '\n' ' '; echo
done
python3 tools/debug/jpeg_huff_bench.py 1 0 2>&1 | grep -v amdgpu
python3 tools/debug/jpeg_huff_bench.py 0 8 2>&1 | grep -v amdgpu
