#!/bin/bash
# tools/run_variants.sh FRAMES OPTS NAME...: exp_expand_time.py for the shipped library and for each libla_gpu_NAME.so (on the GPU box)
frames=$1; opts=$2; shift 2
echo "== shipped"; python tools/exp_expand_time.py $frames $opts 2>/dev/null | tail -n +2
for v in "$@"; do
  echo "== $v"
  LA_GPU_LIB=$PWD/libarchive_amd/csrc/libla_gpu_$v.so timeout -k 5 200 python tools/exp_expand_time.py $frames $opts 2>/dev/null | tail -n +2
done
