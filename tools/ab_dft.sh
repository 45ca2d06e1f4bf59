#!/bin/bash
# Same-box A/B of the spectral match kernel: the committed library against alternatives (WITW_LIB), alternating, 32768 x 4096.
#   bash tools/ab_dft.sh <name>=<path to alternative libwitw_hip.so> ...
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for v in new= "$@"; do
    name=${v%%=*}; lib=${v#*=}
    out=$(WITW_LIB=$lib timeout -k 10 120 python3 tools/time_match_dft.py 32768 4096 2>&1 | grep "dft    match\|orientations")
    echo "$rep $name $out"
  done
done
