#!/bin/bash
# dev tool (GPU box): refresh profiles/ for one bench variant: rocprofv3 kernel stats of the bench command, the bench line
# printed under the profiler, the per-launch timeline of the last step, and the PMC HBM-traffic summary (separate passes).
set -e
variant=${1:-p2}; tag=${2:-r01_final}
R=/root/repo; O=$R/gpurun_out/prof_$variant
cd /tmp && export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
extra=""; [ "$variant" = "p1" ] && extra="--steps 5 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --variant $variant --no-cpu-baseline $extra > $O/bench_under_rocprof.json 2> $O/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --variant $variant --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/bench.py --variant $variant --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/write.log 2>&1
cd $R
mkdir -p profiles
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) profiles/${tag}_kernel_stats_cfg3$variant.csv
grep '^{"metric"' $O/bench_under_rocprof.json > profiles/${tag}_bench_under_rocprof_cfg3$variant.json
python3 tools_trace.py $O/stats 1 > profiles/${tag}_step_timeline_cfg3$variant.txt
python3 tools_pmc.py $O/fetch $O/write profiles/r01_pmc_traffic_cfg3$variant.json > $O/pmc_summary.txt
cp profiles/${tag}_* profiles/r01_pmc_traffic_cfg3$variant.json gpurun_out/ 2>/dev/null || true
mkdir -p gpurun_out/profiles_new && cp profiles/${tag}_*cfg3$variant* profiles/r01_pmc_traffic_cfg3$variant.json gpurun_out/profiles_new/
