#!/bin/bash
# diagnostic: cost of the parts of the staged parse kernel (full / no checksums / no table)
for o in "--extra-options 0" "--extra-options 2" "--general-only" "--general-only --extra-options 2" "--extra-options 4"; do
  echo "== $o"
  timeout -k 5 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['bit_exact'], d['ms_per_step'], d['phases_ms'])"
done
