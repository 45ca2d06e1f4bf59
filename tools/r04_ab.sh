#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r04_ab; mkdir -p $O
for rep in 1 2; do
  for v in head "$@"; do
    lib=""; [ $v != head ] && lib=tools/bin/lib_$v.so
    echo -n "$rep $v: "; WITW_LIB=$lib timeout -k 10 120 python3 tools/bench_f2_wres.py 2>/dev/null | tail -1
  done
done | tee $O/ab.txt
