#!/bin/bash
# tools/run_variants_text.sh FRAMES WORDS NAME...: expand times on text-like blocks (LA_EXP_TEXT=WORDS distinct words) for the shipped library and variants
frames=$1; words=$2; shift 2
echo "== shipped"; LA_EXP_TEXT=$words python tools/exp_expand_time.py $frames 2 2>/dev/null | tail -n +2
for v in "$@"; do
  echo "== $v"
  LA_EXP_TEXT=$words LA_GPU_LIB=$PWD/libarchive_amd/csrc/libla_gpu_$v.so timeout -k 5 200 python tools/exp_expand_time.py $frames 2 2>/dev/null | tail -n +2
done
