#!/bin/bash
# dev tool: rocprofv3 kernel stats of an arbitrary python tool -> gpurun_out/$1_stats.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$tag -o g -- python3 "$@" > /root/repo/gpurun_out/$tag.log 2>&1
python3 - <<PY > /root/repo/gpurun_out/${tag}_stats.txt
import csv, glob
f = glob.glob('/root/repo/gpurun_out/$tag/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'aefft' in r['Name']: print(r['Name'][:80].ljust(80), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.1f}us")
PY
