cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call2; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_bf16_gpu.py tests/test_large_grid_parity_gpu.py tests/test_bf16_train_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
bash tools/ab_lib.sh "--precision bf16 --model semantic" old=tools/bin/lib_s16_oldplane.so > $O/ab_infer.txt 2>&1
bash tools/ab_lib.sh "--precision bf16 --model semantic --mode train" old=tools/bin/lib_s16_oldplane.so > $O/ab_train.txt 2>&1
cat $O/ab_infer.txt $O/ab_train.txt
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_lds -o p --output-format csv -- python3 bench.py --model semantic --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-side-blocks --detail-out $O/d.json > /dev/null 2> $O/pmc_lds.log
python3 tools/pmc_summary.py $O/pmc_lds/p_counter_collection.csv > $O/sem_bf16_lds_pmc.txt 2>&1
rm -f $O/pmc_lds/p_counter_collection.csv $O/pmc_lds/p_kernel_trace.csv
head -30 $O/sem_bf16_lds_pmc.txt
