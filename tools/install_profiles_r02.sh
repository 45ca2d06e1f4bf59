#!/bin/bash
# Copies the summaries tools/collect_profiles_r02.sh left under gpurun_out/final2 into profiles/ (tracked), named per round.
set -eu
R=r02
O=gpurun_out/final2
for m in "" _train _bf16 _semantic_bf16 _retrieval_dft _retrieval _e2e _baseline; do
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
cp $O/mfma_util.json profiles/${R}_mfma_util.json
ls profiles
