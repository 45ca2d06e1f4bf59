#!/bin/bash
# round 5, first GPU call: the changed tests, the default bench line as the driver runs it (size + wall time), and the kernel
# tables of the two bf16 training steps (VERDICT r04 missing #3)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r05a
mkdir -p $O
( while true; do date >> $O/heartbeat.txt; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 900 python3 -m pytest tests/test_bench_launch_gpu.py tests/test_conv_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
S=$(date +%s)
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
echo "default bench wall: $(( $(date +%s) - S )) s; line bytes: $(wc -c < $O/bench.json)"
grep "^\[bench" $O/bench.err
cp bench_detail.json $O/bench_detail.json
rocprofv3 --kernel-trace --stats -d $O/prof_bf16_train -o p --output-format csv -- python3 bench.py --mode train --precision bf16 --steps 5 --warmup 2 > $O/bf16_train_under_rocprof.json 2> $O/prof_bf16_train.log &&
rocprofv3 --kernel-trace --stats -d $O/prof_sem_bf16_train -o p --output-format csv -- python3 bench.py --model semantic --mode train --precision bf16 --steps 5 --warmup 2 > $O/sem_bf16_train_under_rocprof.json 2> $O/prof_sem_bf16_train.log
python3 bench.py --mode train --precision bf16 > $O/bench_bf16_train.json 2> $O/bench_bf16_train.err
python3 bench.py --model semantic --mode train --precision bf16 > $O/bench_sem_bf16_train.json 2> $O/bench_sem_bf16_train.err
rm -f $O/prof*/p_kernel_trace.csv
ls $O $O/prof_bf16_train | head -40
head -c 600 $O/bench.json
