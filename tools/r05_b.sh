#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O=gpurun_out/r05b
mkdir -p $O
FUZZ_KINDS=4 timeout -k 10 400 python3 tools/fuzz_parity.py 150 51 2>&1 | tail -6
python3 - <<'PY'
# the same sweep on the 16x16x32 form (off by default)
import os, sys
sys.path.insert(0, '.')
from witw_amd import _lib
_lib.load().witw_conv3x3_wgrad_bf16_mfma16(1)
sys.argv = ['fuzz', '60', '52']
os.environ['FUZZ_KINDS'] = '4'
import runpy
try:
    runpy.run_path('tools/fuzz_parity.py', run_name='__main__')
except SystemExit as e:
    print('exit', e.code)
PY
