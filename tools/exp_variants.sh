#!/bin/bash
# time the expand kernel of several builds on a 2 GiB workload (diagnostic)
for v in "$@"; do
  cp libarchive_amd/csrc/libla_gpu_$v.so libarchive_amd/csrc/libla_gpu.so
  echo "== $v"
  timeout -k 5 120 python bench.py --gib 2 --unique-mib 256 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['bit_exact'], d['phases_ms'])"
done
