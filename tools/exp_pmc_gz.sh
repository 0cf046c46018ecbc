#!/bin/bash
# SQ counters of the gzip kernels on a 2 GiB C3-shaped workload (diagnostic).  usage (GPU box): exp_pmc_gz.sh <outdir>
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/p1 -- python bench.py --workload gzip --gib 2 --steps 2 --warmup 1 --no-cpu-baseline > $out.p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/p2 -- python bench.py --workload gzip --gib 2 --steps 2 --warmup 1 --no-cpu-baseline > $out.p2.log 2>&1
python - <<PY
import csv,glob,collections
for p in ("p1","p2"):
    for f in glob.glob("$out/%s/*/*counter_collection.csv"%p):
        acc=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,d in acc.items():
            if "inflate" in k or "lz4_expand" in k or "gz_" in k: print(p,k,{a:int(b) for a,b in d.items()})
PY
