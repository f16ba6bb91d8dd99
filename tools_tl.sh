#!/bin/bash
# dev tool: rocprofv3 kernel trace of a short bench run + per-launch timeline of the last step -> gpurun_out/$1_tl.txt
tag=${1:-tl}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$tag -o g -- python3 /root/repo/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline "$@" > /root/repo/gpurun_out/$tag.log 2>&1
python3 /root/repo/tools_trace.py /root/repo/gpurun_out/$tag 1 > /root/repo/gpurun_out/${tag}_tl.txt
