#!/bin/bash
# Round-4 first look at the box: phase stamps of the two weight-resident bf16 kernels, per-layer bf16 times, the config-4 line.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r04_probe
mkdir -p $O
WITW_F2_STAMPS=1 timeout -k 10 120 python3 tools/f2_stamps.py > $O/f2_stamps.txt 2>&1
WITW_WRES_STAMPS=1 timeout -k 10 120 python3 tools/debug/wres_stamps.py > $O/wres_stamps.txt 2>&1
timeout -k 10 180 python3 tools/bench_layers.py --bf16 --iters 20 > $O/bf16_layers.txt 2>&1
timeout -k 10 180 python3 bench.py --model semantic --precision bf16 --no-cpu-baseline --no-side-blocks --steps 20 --warmup 5 > $O/sem_bf16.json 2> $O/sem_bf16.err
tail -n 3 $O/f2_stamps.txt $O/wres_stamps.txt; cat $O/bf16_layers.txt; cat $O/sem_bf16.json
