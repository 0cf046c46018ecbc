#!/bin/bash
# counters of the global-window expand kernel on a 2 GiB workload (diagnostic).  usage: exp_pmc_wide.sh <outdir>
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --gib 2 --unique-mib 256 --steps 2 --warmup 1 --no-cpu-baseline --api-mib 0 --no-secondary --extra-options 32"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/p1 -- $B > $out.p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum --kernel-trace --output-format csv -d $out/p2 -- $B > $out.p2.log 2>&1
python - <<PY
import csv,glob,collections
    for f in glob.glob("$out/%s/*/*counter_collection.csv"%p):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k]+=1
        for k,d in acc.items():
            if "expand_wide" in k: print(p,k,"rows",cnt[k],{a:int(b) for a,b in d.items()})
    import subprocess
    print(subprocess.run("tail -2 $out.%s.log"%p, shell=True, capture_output=True, text=True).stdout[-300:])
PY
