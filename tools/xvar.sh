#!/bin/bash
# dev tool (GPU box): time experiment builds build_x/libaefft_x*.so against the product library under rocprofv3 (AEFFT_LIB selects the
# build, autoencoder-fft_amd/__init__.py; the product's libaefft.so is never overwritten)
R=$(cd "$(dirname "$0")/.." && pwd)
for f in $R/autoencoder-fft_amd/libaefft.so $R/build_x/libaefft_x*.so; do
  n=$(basename $f .so); export AEFFT_LIB=$f
  bash $R/tools/tl.sh xv_$n "$@" || exit 1
  echo "== $n"; python3 $R/tools/stats.py $R/gpurun_out/xv_$n/g_kernel_stats.csv | head -${XV_LINES:-6}
done
