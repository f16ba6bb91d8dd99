#!/bin/bash
# dev tool (GPU box): refresh profiles/ for one bench variant: rocprofv3 kernel stats of the bench command, the bench line
# printed under the profiler, the per-launch timeline of the last step, and the PMC HBM-traffic summary (separate passes).
#   tools/profile.sh [p2|p1] [tag]
set -e
variant=${1:-p2}; tag=${2:-r02}
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/prof_$variant
export TMPDIR=${TMPDIR:-/tmp}; cd "$TMPDIR"
rm -rf $O && mkdir -p $O
extra=""; [ "$variant" = "p1" ] && extra="--steps 5 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --variant $variant --no-cpu-baseline --no-variants --no-dp-probe $extra > $O/bench_under_rocprof.json 2> $O/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --variant $variant --steps 3 --warmup 1 --steady-steps 0 --no-dp-probe --no-cpu-baseline --no-roofline --no-variants > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/bench.py --variant $variant --steps 3 --warmup 1 --steady-steps 0 --no-dp-probe --no-cpu-baseline --no-roofline --no-variants > $O/write.log 2>&1
cd $R
mkdir -p profiles gpurun_out/profiles_new
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) profiles/${tag}_kernel_stats_cfg3$variant.csv
grep '^{"metric"' $O/bench_under_rocprof.json > profiles/${tag}_bench_under_rocprof_cfg3$variant.json
python3 tools/trace.py $O/stats 1 > profiles/${tag}_step_timeline_cfg3$variant.txt
python3 tools/pmc.py $O/fetch $O/write profiles/${tag}_pmc_traffic_cfg3$variant.json > $O/pmc_summary.txt
cp profiles/${tag}_*cfg3$variant* gpurun_out/profiles_new/
