#!/bin/bash
# Collects the artifacts that go under profiles/ for one round: bench line (with CPU baseline),
# rocprofv3 kernel stats of the same command, HBM traffic counters (separate --pmc passes).
# usage (on the GPU box): bash tools/collect_profiles.sh gpurun_out/prof_<tag>
out=$1
mkdir -p $out
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --no-cpu-baseline > $out/stats.log 2>&1 || exit 1
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv
bash tools/exp_traffic.sh $out/traffic > $out/traffic.json 2> $out/traffic.err || exit 1
timeout -k 10 300 python bench.py --workload gzip --gib 16 > $out/bench_gzip.json 2> $out/bench_gzip.err || exit 1
echo done
