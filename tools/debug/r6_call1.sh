cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call1; mkdir -p $O
tools/bin/hbm_yardstick 2048 > $O/yardstick.txt 2>&1
python3 tools/bench_hbm.py > $O/hbm_kernels.txt 2>&1
python3 tools/debug/hbm_calib.py >> $O/hbm_kernels.txt 2>&1
bash tools/pmc_wgrad_r06.sh > $O/pmc_wgrad.log 2>&1
P="rocprofv3 --kernel-trace --stats -o p --output-format csv"
$P -d $O/prof_sem_bf16_train -- python3 bench.py --model semantic --mode train --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-side-blocks --detail-out $O/d.json > $O/sem_bf16_train.json 2> $O/sem_bf16_train.log
$P -d $O/prof_sem_bf16 -- python3 bench.py --model semantic --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-side-blocks --detail-out $O/d.json > $O/sem_bf16.json 2> $O/sem_bf16.log
$P -d $O/prof_retr -- python3 bench.py --mode retrieval --match dft --steps 2 --warmup 1 --detail-out $O/d.json > $O/retr.json 2> $O/retr.log
rm -f $O/prof*/p_kernel_trace.csv
tail -3 $O/yardstick.txt
