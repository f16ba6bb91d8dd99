#!/bin/bash
# dev tool: tools_sweep.py under rocprofv3 kernel trace -> gpurun_out/$1
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/$tag
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/$tag -o g -- python3 /root/repo/tools_sweep.py > /root/repo/gpurun_out/$tag.log 2>&1
cd /root/repo && python3 tools_sweep_parse.py gpurun_out/$tag > gpurun_out/${tag}_best.txt
