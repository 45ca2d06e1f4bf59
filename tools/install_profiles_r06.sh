#!/bin/bash
# Copies the summaries tools/collect_profiles_r06.sh left under gpurun_out/final6 into profiles/ (tracked), named per round.
# Parts that have not been collected are skipped (bench / stats / pmc are separate gpurun calls).
set -u
R=r06
O=gpurun_out/final6
c() { [ -f "$1" ] && cp "$1" "$2" || echo "skip $1"; }
for m in "" _train _bf16 _bf16_train _semantic_bf16 _semantic_bf16_train _retrieval_dft _retrieval _e2e _e2e_bf16 _baseline _sweep; do
  c $O/bench$m.json profiles/${R}_bench$m.json
done
c $O/bench_detail.json profiles/${R}_bench_detail.json
c $O/bench_sweep_detail.json profiles/${R}_bench_sweep_detail.json
c $O/bench_default_run.txt profiles/${R}_bench_default_run.txt
c $O/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
c $O/prof/p_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
c $O/prof_all/p_kernel_stats.csv profiles/${R}_bench_all_blocks_kernel_stats.csv
c $O/prof_train/p_kernel_stats.csv profiles/${R}_train_kernel_stats.csv
c $O/prof_bf16_train/p_kernel_stats.csv profiles/${R}_bf16_train_kernel_stats.csv
c $O/prof_sem_bf16_train/p_kernel_stats.csv profiles/${R}_semantic_bf16_train_kernel_stats.csv
c $O/prof_sem_bf16/p_kernel_stats.csv profiles/${R}_semantic_bf16_kernel_stats.csv
c $O/prof_baseline/p_kernel_stats.csv profiles/${R}_baseline_kernel_stats.csv
c $O/prof_retr_dft/p_kernel_stats.csv profiles/${R}_retrieval_dft_kernel_stats.csv
c $O/prof_e2e_bf16/p_kernel_stats.csv profiles/${R}_e2e_bf16_kernel_stats.csv
c $O/match_dft_pmc.json profiles/${R}_match_dft_pmc.json
c $O/traffic.json profiles/traffic.json
c $O/mfma_util.json profiles/${R}_mfma_util.json
c $O/bf16_train_lds_pmc.txt profiles/${R}_bf16_train_lds_pmc.txt
c $O/bf16_layers.txt profiles/${R}_bf16_layers.txt
c $O/f32_layers.txt profiles/${R}_f32_layers.txt
c $O/wgrad_bf16_layers.txt profiles/${R}_wgrad_bf16_layers.txt
c $O/polar_from_raw_pmc.txt profiles/${R}_polar_from_raw_pmc.txt
c $O/hbm_kernels.txt profiles/${R}_hbm_kernels.txt
c $O/weight_resident_kernels.txt profiles/${R}_weight_resident_kernels.txt
c $O/weight_resident_pmc.txt profiles/${R}_weight_resident_pmc.txt
c $O/hbm_yardstick.txt profiles/${R}_hbm_yardstick.txt
c $O/jpeg_huffman_intervals.txt profiles/${R}_jpeg_huffman_intervals.txt
c $O/bench_e2e_bf16_device_entropy.json profiles/${R}_bench_e2e_bf16_device_entropy.json
c $O/bench_e2e_bf16_device_entropy_all.json profiles/${R}_bench_e2e_bf16_device_entropy_all.json
c $O/e2e_device_entropy_dev.json profiles/${R}_e2e_device_entropy_dev.json
c $O/jpeg_selfsync.txt profiles/${R}_jpeg_selfsync.txt
c $O/prof_huff2/p_kernel_stats.csv profiles/${R}_jpeg_huffman_rst2_kernel_stats.csv
c $O/prof_huff1/p_kernel_stats.csv profiles/${R}_jpeg_huffman_rst1_kernel_stats.csv
c $O/prof_selfsync/p_kernel_stats.csv profiles/${R}_jpeg_selfsync_kernel_stats.csv
# both clocks of the dominant kernels: the rocprof tables of this round, keyed by the kernel sources' hashes (bench.py: frac_rocprof)
python3 tools/make_rocprof_avg.py infer:$O/prof/p_kernel_stats.csv:profiles/${R}_bench_kernel_stats.csv \
  train:$O/prof_train/p_kernel_stats.csv:profiles/${R}_train_kernel_stats.csv \
  bf16_train:$O/prof_bf16_train/p_kernel_stats.csv:profiles/${R}_bf16_train_kernel_stats.csv \
  sem_bf16:$O/prof_sem_bf16/p_kernel_stats.csv:profiles/${R}_semantic_bf16_kernel_stats.csv \
  sem_bf16_train:$O/prof_sem_bf16_train/p_kernel_stats.csv:profiles/${R}_semantic_bf16_train_kernel_stats.csv > profiles/rocprof_kernel_avg.json
ls profiles | grep $R
