#!/bin/bash
# dev tool (GPU box): spatial-step kernel times of experiment builds build_x/libaefft_x*.so swapped in for libaefft.so (scratch copy only)
R=$(cd "$(dirname "$0")/.." && pwd)
cp $R/autoencoder-fft_amd/libaefft.so /tmp/libaefft_base.so
export TMPDIR=/tmp
for f in /tmp/libaefft_base.so $R/build_x/libaefft_x*.so; do
  n=$(basename $f .so); cp $f $R/autoencoder-fft_amd/libaefft.so
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/xs_$n -o s -- python3 $R/tools/spatial.py > $R/gpurun_out/xs_$n.log 2>&1)
  echo "== $n: $(grep '^step' $R/gpurun_out/xs_$n.log)"; python3 $R/tools/stats.py $R/gpurun_out/xs_$n/s_kernel_stats.csv | grep -E "rcorr_kernel|mcorr"
done
cp /tmp/libaefft_base.so $R/autoencoder-fft_amd/libaefft.so
