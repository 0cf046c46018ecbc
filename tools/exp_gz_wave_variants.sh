#!/bin/bash
# wave-per-member inflate kernel of several builds (libla_gpu_<name>.so), forced with option 2 (diagnostic)
# builds are selected through LA_GPU_LIB (libarchive_amd/_native.py): the shipped libla_gpu.so is never overwritten
for v in "$@"; do
  export LA_GPU_LIB=$PWD/libarchive_amd/csrc/libla_gpu_$v.so
  for mib in 16 256 1024; do
    gib=$(python -c "print($mib/1024)")
    echo -n "$v decoded $mib MiB ($((mib*16)) members): "
    timeout -k 5 120 python bench.py --workload gzip --gib $gib --unique-mib $mib --steps 5 --warmup 2 --no-cpu-baseline --extra-options 2 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['bit_exact'], d['ms_per_step'], d['phases_ms'])"
  done
done
