#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r05b
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_match_dft_gpu.py tests/test_retrieval_fullsize_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2 3; do
  echo "base: $(WITW_LIB=tools/bin/lib_dft_base.so python3 tools/time_match_dft.py 2>&1 | grep 'dft    match')"
  echo "new : $(python3 tools/time_match_dft.py 2>&1 | grep 'dft    match')"
done
