#!/bin/bash
# dev tool (GPU box): spatial-step kernel times of experiment builds build_x/libaefft_x*.so against the product library (AEFFT_LIB selects
# the build; the product's libaefft.so is never overwritten)
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
for f in $R/autoencoder-fft_amd/libaefft.so $R/build_x/libaefft_x*.so; do
  n=$(basename $f .so); export AEFFT_LIB=$f
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/xs_$n -o s -- python3 $R/tools/spatial.py > $R/gpurun_out/xs_$n.log 2>&1)
  echo "== $n: $(grep '^step' $R/gpurun_out/xs_$n.log)"; python3 $R/tools/stats.py $R/gpurun_out/xs_$n/s_kernel_stats.csv | grep -E "rcorr_kernel|mcorr"
done
