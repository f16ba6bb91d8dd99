#!/bin/bash
# dev tool (GPU box): plain bench (no profiler) of experiment builds build_x/libaefft_x*.so swapped in for libaefft.so
R=$(cd "$(dirname "$0")/.." && pwd)
cp $R/autoencoder-fft_amd/libaefft.so /tmp/libaefft_base.so
for f in /tmp/libaefft_base.so $R/build_x/libaefft_x*.so; do
  cp $f $R/autoencoder-fft_amd/libaefft.so
  echo "== $(basename $f .so)"
  for i in 1 2; do python $R/bench.py --steps 200 --no-cpu-baseline --no-roofline --no-variants "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
done
cp /tmp/libaefft_base.so $R/autoencoder-fft_amd/libaefft.so
