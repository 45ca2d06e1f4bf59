#!/bin/bash
# Regenerates the judged artefacts under profiles/ on a GPU box (run through gpurun from the repo root):
#   plain bench lines, rocprofv3 --kernel-trace --stats summaries of the same commands, and the PMC passes
#   (FETCH_SIZE, WRITE_SIZE; separate runs, no trace domains) that tools/make_traffic.py turns into traffic.json.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/final
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --mode train > $O/bench_train.json 2> $O/bench_train.err
python3 bench.py --precision bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
python3 bench.py --precision fp16x3 > $O/bench_fp16x3.json 2> $O/bench_fp16x3.err
python3 bench.py --precision fp16x3 --fov 70 > $O/bench_fp16x3_fov70.json 2> $O/bench_fp16x3_fov70.err
python3 bench.py --precision fp16x3 --mode train > $O/bench_fp16x3_train.json 2> $O/bench_fp16x3_train.err
python3 bench.py --mode train --precision bf16 > $O/bench_bf16_train.json 2> $O/bench_bf16_train.err
python3 bench.py --model semantic --precision bf16 > $O/bench_semantic_bf16.json 2> $O/bench_semantic_bf16.err
python3 bench.py --model semantic --mode train --precision bf16 > $O/bench_semantic_bf16_train.json 2> $O/bench_semantic_bf16_train.err
python3 bench.py --model semantic > $O/bench_semantic.json 2> $O/bench_semantic.err
python3 bench.py --model semantic --mode train > $O/bench_semantic_train.json 2> $O/bench_semantic_train.err
python3 bench.py --fov 70 > $O/bench_fov70.json 2> $O/bench_fov70.err
python3 bench.py --mode retrieval --steps 2 --warmup 1 > $O/bench_retrieval.json 2> $O/bench_retrieval.err
echo benches done
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof.log
rocprofv3 --kernel-trace --stats -d $O/prof_train -o p --output-format csv -- python3 bench.py --mode train --steps 5 --warmup 2 > $O/train_under_rocprof.json 2> $O/prof_train.log
rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o p --output-format csv -- python3 bench.py --precision bf16 --steps 5 --warmup 2 > $O/bf16_under_rocprof.json 2> $O/prof_bf16.log
rocprofv3 --kernel-trace --stats -d $O/prof_fp16x3 -o p --output-format csv -- python3 bench.py --precision fp16x3 --steps 5 --warmup 2 > $O/fp16x3_under_rocprof.json 2> $O/prof_fp16x3.log
rocprofv3 --kernel-trace --stats -d $O/prof_fp16x3_train -o p --output-format csv -- python3 bench.py --precision fp16x3 --mode train --steps 5 --warmup 2 > $O/fp16x3_train_under_rocprof.json 2> $O/prof_fp16x3_train.log
rocprofv3 --kernel-trace --stats -d $O/prof_bf16_train -o p --output-format csv -- python3 bench.py --mode train --precision bf16 --steps 5 --warmup 2 > $O/bf16_train_under_rocprof.json 2> $O/prof_bf16_train.log
echo stats done
for m in "infer:" "train:--mode train" "bf16:--precision bf16" "bf16_train:--mode train --precision bf16" "fp16x3:--precision fp16x3"; do
  tag=${m%%:*}; flags=${m#*:}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$tag -o p --output-format csv -- python3 bench.py $flags --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_$tag.json 2> $O/pmc_fetch_$tag.log
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$tag -o p --output-format csv -- python3 bench.py $flags --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_$tag.json 2> $O/pmc_write_$tag.log
  echo pmc $tag done
done
python3 tools/make_traffic.py 128 infer:$O/pmc_fetch_infer/p_counter_collection.csv:$O/pmc_write_infer/p_counter_collection.csv \
  train:$O/pmc_fetch_train/p_counter_collection.csv:$O/pmc_write_train/p_counter_collection.csv \
  bf16:$O/pmc_fetch_bf16/p_counter_collection.csv:$O/pmc_write_bf16/p_counter_collection.csv \
  bf16_train:$O/pmc_fetch_bf16_train/p_counter_collection.csv:$O/pmc_write_bf16_train/p_counter_collection.csv \
  fp16x3:$O/pmc_fetch_fp16x3/p_counter_collection.csv:$O/pmc_write_fp16x3/p_counter_collection.csv > $O/traffic.json
for m in "infer:" "train:--mode train" "bf16:--precision bf16" "bf16_train:--mode train --precision bf16" "fp16x3:--precision fp16x3"; do
  tag=${m%%:*}; flags=${m#*:}
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma_$tag -o p --output-format csv -- python3 bench.py $flags --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_mfma_$tag.json 2> $O/pmc_mfma_$tag.log
done
python3 tools/make_mfma_util.py infer:$O/pmc_mfma_infer/p_counter_collection.csv train:$O/pmc_mfma_train/p_counter_collection.csv \
  bf16:$O/pmc_mfma_bf16/p_counter_collection.csv bf16_train:$O/pmc_mfma_bf16_train/p_counter_collection.csv \
  fp16x3:$O/pmc_mfma_fp16x3/p_counter_collection.csv > $O/mfma_util.json
# the big per-dispatch CSVs stay on the box; only summaries come back
rm -f $O/pmc_*/p_counter_collection.csv $O/pmc_*/p_kernel_trace.csv $O/prof*/p_kernel_trace.csv
ls -R $O | head -80
tail -c 600 $O/bench.json
