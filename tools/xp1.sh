#!/bin/bash
# dev tool (GPU box): cfg3-P1 kernel times of experiment builds build_x/libaefft_x*.so against the product library.  The binding is pointed
# at each build through AEFFT_LIB (autoencoder-fft_amd/__init__.py); the product's libaefft.so is never overwritten.
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
for f in $R/autoencoder-fft_amd/libaefft.so $R/build_x/libaefft_x*.so; do
  n=$(basename $f .so); export AEFFT_LIB=$f
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/xp_$n -o s -- python3 $R/tools/cfgstep.py p1 3 > $R/gpurun_out/xp_$n.log 2>&1)
  echo "== $n: $(grep 'p1:' $R/gpurun_out/xp_$n.log)"; python3 $R/tools/trace.py $R/gpurun_out/xp_$n 1 | grep -E "kspec|tail" | head -4
done
