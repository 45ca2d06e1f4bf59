#!/bin/bash
# Same-box A/B of library builds (device-to-device spread on a pool is larger than most kernel changes): run through gpurun as
#   bash tools/ab_lib.sh "<bench.py flags>" <name>=<path to an alternative libwitw_hip.so> [<name>=<path> ...]
# Alternates the committed library ("head") and every alternative (WITW_LIB) for two rounds, prints value and roofline.frac.
# An alternative is built by compiling the changed .hip and linking it with the other objects of witw_amd/build/, e.g.
#   hipcc <flags of witw_amd/build.py> -c variant.hip -o /tmp/v.o && hipcc --offload-arch=gfx950 -shared -fPIC -o alt.so <other .o> /tmp/v.o
FLAGS=${1:---precision bf16 --model semantic}
shift
for rep in 1 2; do
  for v in head=  "$@"; do
    name=${v%%=*}; lib=${v#*=}
    out=$(WITW_LIB=$lib timeout -k 10 180 python3 bench.py $FLAGS --no-cpu-baseline --no-side-blocks --steps 10 --warmup 3 2>/dev/null |
          python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])")
    echo "$rep $name $out"
  done
done
