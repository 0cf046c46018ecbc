#!/bin/bash
# tools/run_gz_variants.sh GIB NAME...: gzip C3 phase times for the shipped library and for each libla_gpu_NAME.so (on the GPU box)
gib=$1; shift
one() { python bench.py --workload gzip --gib $gib --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms'], d['bit_exact'])"; }
echo "== shipped"; one
for v in "$@"; do echo "== $v"; LA_GPU_LIB=$PWD/libarchive_amd/csrc/libla_gpu_$v.so one; done
