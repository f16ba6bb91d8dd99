#!/bin/bash
# dev tool (GPU box): plain bench (no profiler) of experiment builds build_x/libaefft_x*.so against the product library (AEFFT_LIB selects
# the build; the product's libaefft.so is never overwritten)
R=$(cd "$(dirname "$0")/.." && pwd)
for f in $R/autoencoder-fft_amd/libaefft.so $R/build_x/libaefft_x*.so; do
  export AEFFT_LIB=$f
  echo "== $(basename $f .so)"
  for i in 1 2; do python $R/bench.py --steps 200 --steady-steps 0 --no-dp-probe --no-cpu-baseline --no-roofline --no-variants "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
done
