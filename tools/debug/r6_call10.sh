cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call10; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_jpeg_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for a in "1 0" "0 16" "0 8" "0 4" "0 2"; do python3 tools/debug/jpeg_huff_bench.py $a 2>&1 | grep -v amdgpu.ids | tee -a $O/huff_bench.txt; done
