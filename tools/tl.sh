#!/bin/bash
# dev tool: rocprofv3 kernel trace of a short bench run + per-launch timeline of the last step -> gpurun_out/$1_tl.txt
tag=${1:-tl}; shift
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=${TMPDIR:-/tmp}; cd "$TMPDIR"
rm -rf $R/gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -o g -- python3 $R/bench.py --steps 20 --warmup 2 --steady-steps 0 --no-dp-probe --no-cpu-baseline --no-roofline --no-variants "$@" > $R/gpurun_out/$tag.log 2>&1
python3 $R/tools/trace.py $R/gpurun_out/$tag 1 > $R/gpurun_out/${tag}_tl.txt
