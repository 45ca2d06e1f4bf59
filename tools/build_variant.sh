#!/bin/bash
# Build an alternative libwitw_hip.so with one source compiled under extra -D flags (switch-off / diagnostic builds for
# same-box A/B runs, see tools/ab_lib.sh):   bash tools/build_variant.sh <name> <file.hip> [-DFLAG ...]  ->  tools/bin/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p tools/bin
python3 -c "from witw_amd import build; build.build(verbose=False)" >/dev/null 2>&1
obj=tools/bin/${name}_$(basename ${src%.hip}).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function "$@" -c witw_amd/csrc/$src -o $obj 2>/dev/null
others=$(ls witw_amd/build/*.o | grep -v "/$(basename ${src%.hip}).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/lib_${name}.so $others $obj
echo tools/bin/lib_${name}.so
