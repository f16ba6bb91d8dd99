#!/bin/bash
# dev tool (GPU box): rocprofv3 kernel stats + HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the spatial-mode step
#   tools/profile_spatial.sh [tag]
set -e
tag=${1:-r03}
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/prof_spatial
export TMPDIR=${TMPDIR:-/tmp}; cd "$TMPDIR"
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/tools/spatial.py > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/tools/spatial.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/tools/spatial.py > $O/write.log 2>&1
cd $R
mkdir -p profiles gpurun_out/profiles_new
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) profiles/${tag}_kernel_stats_spatial.csv
python3 tools/pmc.py $O/fetch $O/write profiles/${tag}_pmc_traffic_spatial.json > $O/pmc_summary.txt
cp profiles/${tag}_*spatial* gpurun_out/profiles_new/
cat $O/pmc_summary.txt
