#!/bin/bash
# Regenerates the judged artefacts under profiles/ on a GPU box (run through gpurun from the repo root):
#   plain bench lines, rocprofv3 --kernel-trace --stats summaries of the same commands, and the two PMC passes
#   (FETCH_SIZE, WRITE_SIZE; separate runs, no trace domains) that tools/make_traffic.py turns into traffic.json.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/final
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --mode train > $O/bench_train.json 2> $O/bench_train.err
python3 bench.py --precision bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
python3 bench.py --mode retrieval --steps 2 --warmup 1 > $O/bench_retrieval.json 2> $O/bench_retrieval.err
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof.log
rocprofv3 --kernel-trace --stats -d $O/prof_train -o p --output-format csv -- python3 bench.py --mode train --steps 5 --warmup 2 > $O/train_under_rocprof.json 2> $O/prof_train.log
rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o p --output-format csv -- python3 bench.py --precision bf16 --steps 5 --warmup 2 > $O/bf16_under_rocprof.json 2> $O/prof_bf16.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_bf16 -o p --output-format csv -- python3 bench.py --precision bf16 --steps 2 --warmup 1 > $O/pmc_fetch_bf16.json 2> $O/pmc_fetch_bf16.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_bf16 -o p --output-format csv -- python3 bench.py --precision bf16 --steps 2 --warmup 1 > $O/pmc_write_bf16.json 2> $O/pmc_write_bf16.log
ls -R $O | head -60
tail -c 600 $O/bench.json
