#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/fuzz
( while true; do date >> gpurun_out/fuzz/heartbeat.txt; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 500 python3 tools/fuzz_parity.py 400 505 2>&1 | grep -v amdgpu | tail -8
