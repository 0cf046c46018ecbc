#!/bin/bash
# time the gzip kernels of several builds (libla_gpu_<name>.so) on the 4 GiB C3 shape (diagnostic)
cp libarchive_amd/csrc/libla_gpu.so /tmp/libla_gpu_keep.so
for v in "$@"; do
  cp libarchive_amd/csrc/libla_gpu_$v.so libarchive_amd/csrc/libla_gpu.so
  echo "== $v"
  timeout -k 5 200 python bench.py --workload gzip --gib ${GIB:-4} --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['bit_exact'], d['ms_per_step'], d['phases_ms'])"
done
cp /tmp/libla_gpu_keep.so libarchive_amd/csrc/libla_gpu.so
