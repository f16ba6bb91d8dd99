#!/bin/bash
# dev tool (GPU box): cfg3-P1 kernel times of experiment builds build_x/libaefft_x*.so swapped in for libaefft.so (scratch copy only)
R=$(cd "$(dirname "$0")/.." && pwd)
cp $R/autoencoder-fft_amd/libaefft.so /tmp/libaefft_base.so
export TMPDIR=/tmp
for f in /tmp/libaefft_base.so $R/build_x/libaefft_x*.so; do
  n=$(basename $f .so); cp $f $R/autoencoder-fft_amd/libaefft.so
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/xp_$n -o s -- python3 $R/tools/cfgstep.py p1 3 > $R/gpurun_out/xp_$n.log 2>&1)
  echo "== $n: $(grep 'p1:' $R/gpurun_out/xp_$n.log)"; python3 $R/tools/trace.py $R/gpurun_out/xp_$n 1 | grep -E "kspec|tail" | head -4
done
cp /tmp/libaefft_base.so $R/autoencoder-fft_amd/libaefft.so
