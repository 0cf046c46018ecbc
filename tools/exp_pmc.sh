#!/bin/bash
# PMC counters of the lz4 kernels on a 2 GiB workload (diagnostic).  usage: exp_pmc.sh <variant> <outdir>
# The variant is selected through LA_GPU_LIB (libarchive_amd/_native.py): the shipped library is never overwritten.
# <variant> "shipped" = libla_gpu.so; LA_EXTRA = extra LA_LZ4_OPT_* bits (8 = the queue-generation expand kernel).
v=$1; out=$2
if [ "$v" != "shipped" ]; then export LA_GPU_LIB=$PWD/libarchive_amd/csrc/libla_gpu_$v.so; fi
EXTRA=${LA_EXTRA:-0}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/p1 -- python bench.py --gib 2 --unique-mib 256 --steps 2 --warmup 1 --no-cpu-baseline --extra-options $EXTRA > $out.p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/p2 -- python bench.py --gib 2 --unique-mib 256 --steps 2 --warmup 1 --no-cpu-baseline --extra-options $EXTRA > $out.p2.log 2>&1
python - <<PY
import csv,glob,collections
for p in ("p1","p2"):
    for f in glob.glob("$out/%s/*/*counter_collection.csv"%p):
        acc=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,d in acc.items():
            if "lz4" in k and "sums" not in k: print(p,k,{a:int(b) for a,b in d.items()})
PY
