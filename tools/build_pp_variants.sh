#!/bin/bash
# Variant builds of the two two-team weight-resident bf16 kernels: tools/bin/lib_p<PF>r<PRIO>.so (operand read-ahead in steps, s_setprio scheme:
# 0 none, 1 the M phase high, 2 the V phase high). Same-box A/B: WITW_LIB=<lib> python3 tools/bench_f2_wres.py
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
python3 -c "from witw_amd import build; build.build(verbose=False)" >/dev/null 2>&1
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function"
others=$(ls witw_amd/build/*.o | grep -v "/conv_first2_bf16.o" | grep -v "/conv3x3_bf16_wres.o")
for v in "$@"; do
  pf=${v%%:*}; pr=${v#*:}
  /opt/rocm/bin/hipcc $F -DWITW_F2_PF=$pf -DWITW_F2_PRIO=$pr $EXTRA -c witw_amd/csrc/conv_first2_bf16.hip -o tools/bin/f2_$pf$pr.o 2>/dev/null &
  /opt/rocm/bin/hipcc $F -DWITW_WRES_PF=$pf -DWITW_WRES_PRIO=$pr $EXTRA -c witw_amd/csrc/conv3x3_bf16_wres.hip -o tools/bin/wres_$pf$pr.o 2>/dev/null &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/lib_p${pf}r${pr}.so $others tools/bin/f2_$pf$pr.o tools/bin/wres_$pf$pr.o
  echo tools/bin/lib_p${pf}r${pr}.so
done
