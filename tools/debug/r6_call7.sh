cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call7; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_jpeg_gpu.py tests/test_bf16_gpu.py -x -q -k "jpeg or entropy or device or data_path or training_form" > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -30 $O/pytest.log
