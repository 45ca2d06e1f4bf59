#!/bin/bash
# Round 6: what does conv3x3_wgrad_bf16_nhwc_kernel<1,8,2,2,4> wait on? rocprofv3 --pmc passes (kernel-trace
# only, program directly after `--`) over tools/bench_wgrad_bf16.py --layers L19, counters filtered against `rocprofv3 -L`.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/r6_wgrad_pmc
mkdir -p $O
rocprofv3 -L > $O/counters_all.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $O/counters_all.txt | sort -u > $O/counters_sq.txt
have() { for c in "$@"; do grep -qx "$c" $O/counters_sq.txt && printf "%s " "$c"; done; }
G1=$(have SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_VMEM SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES)
G2=$(have SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_EXP_GDS)
G3=$(have SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD)
G4=$(have SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_BARRIER SQ_INSTS_BARRIER SQ_INSTS_WAVE32_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES)
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  [ -z "$G" ] && continue
  echo "pass $i: $G" | tee -a $O/passes.txt
  rocprofv3 --kernel-trace --pmc $G -d $O/p$i -o p --output-format csv -- python3 tools/bench_wgrad_bf16.py --layers L19,L17 --iters 4 > $O/p$i.out 2> $O/p$i.log
  python3 tools/pmc_summary.py $O/p$i/p_counter_collection.csv wgrad >> $O/summary.txt 2>&1
done
grep -i "wait\|barrier" $O/counters_sq.txt > $O/counters_wait.txt
rm -f $O/p*/p_counter_collection.csv $O/p*/p_kernel_trace.csv
cat $O/summary.txt
