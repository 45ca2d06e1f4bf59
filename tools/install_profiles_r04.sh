#!/bin/bash
# Copies the summaries tools/collect_profiles_r04.sh left under gpurun_out/final4 into profiles/ (tracked), named per round.
set -eu
R=r04
O=gpurun_out/final4
for m in "" _train _bf16 _bf16_train _semantic_bf16 _retrieval_dft _retrieval _e2e _e2e_bf16 _e2e_bf16_host_torch _baseline; do
  cp $O/bench$m.json profiles/${R}_bench$m.json
done
cp $O/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp $O/prof/p_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp $O/prof_all/p_kernel_stats.csv profiles/${R}_bench_all_blocks_kernel_stats.csv
cp $O/prof_train/p_kernel_stats.csv profiles/${R}_train_kernel_stats.csv
cp $O/prof_sem_bf16/p_kernel_stats.csv profiles/${R}_semantic_bf16_kernel_stats.csv
cp $O/prof_baseline/p_kernel_stats.csv profiles/${R}_baseline_kernel_stats.csv
cp $O/prof_retr_dft/p_kernel_stats.csv profiles/${R}_retrieval_dft_kernel_stats.csv
cp $O/match_dft_pmc.json profiles/${R}_match_dft_pmc.json
cp $O/traffic.json profiles/traffic.json
cp $O/bf16_layers.txt profiles/${R}_bf16_layers.txt
cp $O/f32_layers.txt profiles/${R}_f32_layers.txt
cp $O/polar_from_raw_pmc.txt profiles/${R}_polar_from_raw_pmc.txt
cp $O/prof_e2e_bf16/p_kernel_stats.csv profiles/${R}_e2e_bf16_kernel_stats.csv
cp $O/mfma_util.json profiles/${R}_mfma_util.json
cp $O/bench_sweep.json profiles/${R}_bench_sweep.json
cp $O/hbm_kernels.txt profiles/${R}_hbm_kernels.txt
cp $O/weight_resident_kernels.txt profiles/${R}_weight_resident_kernels.txt
cp $O/weight_resident_pmc.txt profiles/${R}_weight_resident_pmc.txt
ls profiles
