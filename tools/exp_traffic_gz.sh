#!/bin/bash
# HBM traffic counters of the deflate kernels at the full C3 size (separate --pmc passes, MI355X_MICROARCH.md HBM section)
# usage (GPU box): bash tools/exp_traffic_gz.sh gpurun_out/traffic_gz > profiles/r03_traffic_gzip.json
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python bench.py --workload gzip --gib 16 --steps 2 --warmup 1 --no-cpu-baseline > $out.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python bench.py --workload gzip --gib 16 --steps 2 --warmup 1 --no-cpu-baseline > $out.write.log 2>&1
python - <<PY
import csv,glob,collections,json,hashlib,os
res=collections.defaultdict(dict)
for p in ("fetch","write"):
    for f in glob.glob("$out/%s/*/*counter_collection.csv"%p):
        acc=collections.defaultdict(float); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")[:48]
            acc[k]+=float(r["Counter_Value"]); cnt[k]+=1
        for k in acc:
            if "inflate" in k or "lz4_expand" in k or "gz_" in k or "crc32" in k:
                res[k][p+"_kb_per_launch"]=acc[k]/cnt[k]; res[k]["launches"]=cnt[k]
lib=os.environ.get("LA_GPU_LIB") or "libarchive_amd/csrc/libla_gpu.so"
res["library_sha16"]=hashlib.sha256(open(lib,"rb").read()).hexdigest()[:16]
# one step = one launch of each kernel; FETCH_SIZE counts half the bytes of wide coalesced reads on gfx950: doubled; units KiB.
# (The entropy decoder's loads are 8 bytes per lane and its stores 16: outside the calibrated pattern, see MI355X_MICROARCH.md -- ratios, not absolutes.)
tot=0.0
for k,v in res.items():
    if isinstance(v,dict) and ("inflate_lanes_kernel" in k or "lz4_expand_fast_kernel" in k):
        tot+=(2*v.get("fetch_kb_per_launch",0)+v.get("write_kb_per_launch",0))*1024
res["inflate_symbols_plus_expand"]={"hbm_bytes_per_step": tot}
print(json.dumps(res, indent=1))
PY
