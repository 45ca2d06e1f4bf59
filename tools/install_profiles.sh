#!/bin/bash
# Copies the summaries tools/collect_profiles.sh left under gpurun_out/final into profiles/ (tracked), named per round.
set -eu
R=${1:-r01}
O=gpurun_out/final
for m in "" _train _bf16 _bf16_train _fp16x3 _fp16x3_fov70 _fp16x3_train _semantic _semantic_train _semantic_bf16 _semantic_bf16_train _fov70 _retrieval; do
  cp $O/bench$m.json profiles/${R}_bench$m.json
done
cp $O/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp $O/train_under_rocprof.json profiles/${R}_train_under_rocprof.json
cp $O/bf16_under_rocprof.json profiles/${R}_bf16_under_rocprof.json
cp $O/bf16_train_under_rocprof.json profiles/${R}_bf16_train_under_rocprof.json
cp $O/prof/p_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp $O/prof_train/p_kernel_stats.csv profiles/${R}_train_kernel_stats.csv
cp $O/prof_bf16/p_kernel_stats.csv profiles/${R}_bench_bf16_kernel_stats.csv
cp $O/prof_bf16_train/p_kernel_stats.csv profiles/${R}_bf16_train_kernel_stats.csv
cp $O/prof_fp16x3/p_kernel_stats.csv profiles/${R}_fp16x3_kernel_stats.csv
cp $O/fp16x3_under_rocprof.json profiles/${R}_fp16x3_under_rocprof.json
cp $O/prof_fp16x3_train/p_kernel_stats.csv profiles/${R}_fp16x3_train_kernel_stats.csv
cp $O/traffic.json profiles/traffic.json
cp $O/mfma_util.json profiles/mfma_util.json
ls profiles
