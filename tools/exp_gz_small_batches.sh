#!/bin/bash
# gzip decode time per batch for small batches: lane-per-member two-phase (default from 512 members), forced wave-per-member (2),
# forced two-phase (8).  diagnostic; usage: exp_gz_small_batches.sh > out
for mib in 64 128 192 256 384 512 1024; do
  for opt in 0 2 8; do
    gib=$(python -c "print($mib/1024)")
    echo -n "decoded $mib MiB ($((mib*16)) members) options $opt: "
    timeout -k 5 120 python bench.py --workload gzip --gib $gib --unique-mib $mib --steps 5 --warmup 2 --no-cpu-baseline --extra-options $opt 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['bit_exact'], d['ms_per_step'], d['phases_ms'])"
  done
done
