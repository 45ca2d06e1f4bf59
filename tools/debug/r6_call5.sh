cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call5; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_bf16_gpu.py tests/test_bf16_train_gpu.py tests/test_match_gpu.py tests/test_match_dft_gpu.py tests/test_retrieval_fullsize_gpu.py tests/test_large_grid_parity_gpu.py tests/test_drivers_gpu.py tests/test_drivers2_gpu.py -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for i in 1 2; do
python3 bench.py --model semantic --mode train --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-side-blocks --detail-out $O/d.json 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sem train fused', d['ms_per_step'], d['roofline']['frac'])"
WITW_F2=0 python3 - <<'PY'
import subprocess, sys, json, os
PY
done
python3 bench.py --mode retrieval --match dft --steps 2 --warmup 1 --detail-out $O/d.json 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('retr', d['ms_per_step'], d['roofline']['frac'])"
P="rocprofv3 --kernel-trace --stats -o p --output-format csv"
$P -d $O/prof_sem_bf16_train -- python3 bench.py --model semantic --mode train --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-side-blocks --detail-out $O/d.json > $O/sem_bf16_train.json 2> $O/sem_bf16_train.log
$P -d $O/prof_retr -- python3 bench.py --mode retrieval --match dft --steps 2 --warmup 1 --detail-out $O/d.json > $O/retr.json 2> $O/retr.log
rm -f $O/prof*/p_kernel_trace.csv
