#!/bin/bash
# dev tool (GPU box): time experiment builds build_x/libaefft_x*.so by swapping them in for libaefft.so (scratch copy only)
R=$(cd "$(dirname "$0")/.." && pwd)
cp $R/autoencoder-fft_amd/libaefft.so /tmp/libaefft_base.so
for f in /tmp/libaefft_base.so $R/build_x/libaefft_x*.so; do
  n=$(basename $f .so)
  cp $f $R/autoencoder-fft_amd/libaefft.so
  bash $R/tools/tl.sh xv_$n "$@" || exit 1
  echo "== $n"; python3 $R/tools/stats.py $R/gpurun_out/xv_$n/g_kernel_stats.csv | head -${XV_LINES:-6}
done
cp /tmp/libaefft_base.so $R/autoencoder-fft_amd/libaefft.so
