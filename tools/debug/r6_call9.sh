cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call9; mkdir -p $O
python3 tools/debug/jpeg_huff_bench.py 1 2>&1 | grep -v amdgpu.ids | tee $O/huff_bench.txt
rocprofv3 --kernel-trace --stats -o p --output-format csv -d $O/prof -- python3 tools/debug/jpeg_huff_bench.py 1 > /dev/null 2> $O/prof.log
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM -o p --output-format csv -d $O/pmc -- python3 tools/debug/jpeg_huff_bench.py 1 > /dev/null 2> $O/pmc.log
python3 tools/pmc_summary.py $O/pmc/p_counter_collection.csv huffman > $O/pmc.txt 2>&1
rm -f $O/prof/p_kernel_trace.csv $O/pmc/p_counter_collection.csv $O/pmc/p_kernel_trace.csv
head -6 $O/prof/p_kernel_stats.csv | cut -c1-160; cat $O/pmc.txt
