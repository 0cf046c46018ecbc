#!/bin/bash
# time the kernels of several builds (libla_gpu_<name>.so) on a 4 GiB workload (diagnostic);
# the shipped library is restored afterwards
# builds are selected through LA_GPU_LIB (libarchive_amd/_native.py): the shipped libla_gpu.so is never overwritten
for v in "$@"; do
  export LA_GPU_LIB=$PWD/libarchive_amd/csrc/libla_gpu_$v.so
  echo "== $v"
  timeout -k 5 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['bit_exact'], d['ms_per_step'], d['phases_ms'])"
done
