#!/bin/bash
# same-box A/B of bf16 conv kernel variants (scratch)
for rep in 1 2; do
for v in head persist persist_off swap0; do
  case $v in
    head) L=""; E="";;
    persist) L="tools/scratch/ab/libwitw_persist.so"; E="";;
    persist_off) L="tools/scratch/ab/libwitw_persist.so"; E="0";;
    swap0) L="tools/scratch/ab/libwitw_swap0.so"; E="";;
  esac
  out=$(WITW_LIB=$L WITW_BF_PERSIST=$E timeout -k 10 120 python3 bench.py --model semantic --precision bf16 --no-cpu-baseline --no-side-blocks --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])")
  echo "$rep $v $out"
done
done
