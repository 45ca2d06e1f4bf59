cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call39; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $O/pmc_a -o p --output-format csv -- python3 tools/debug/jpeg_huff_bench.py 0 2 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/pmc_b -o p --output-format csv -- python3 tools/debug/jpeg_huff_bench.py 0 2 > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $O/pmc_c -o p --output-format csv -- python3 tools/debug/selfsync_bench.py > $O/c.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/pmc_d -o p --output-format csv -- python3 tools/debug/selfsync_bench.py > $O/d.log 2>&1
for x in a b c d; do python3 tools/pmc_summary.py $O/pmc_$x/p_counter_collection.csv 2>&1 | grep -A8 "jpeg_huffman_kernel\|jpeg_selfsync" ; done | tee $O/jpeg_pmc.txt
