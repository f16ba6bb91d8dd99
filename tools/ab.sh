#!/bin/bash
# dev tool (GPU box): A/B of experiment builds against the product library on ONE box -- tools/cfgstep.py <cfg> 400 with the per-kernel
# bracket times (PROF=1), each library twice, interleaved.   tools/ab.sh p2 [TAG ...]   (no tags: every build_x/libaefft_x*.so)
R=$(cd "$(dirname "$0")/.." && pwd)
cfg=${1:-p2}; shift
libs="$R/autoencoder-fft_amd/libaefft.so"
if [ $# -eq 0 ]; then libs="$libs $(ls $R/build_x/libaefft_x*.so | grep -v WGT)"; else for t in "$@"; do libs="$libs $R/build_x/libaefft_x$t.so"; done; fi
for i in 1 2; do
  for f in $libs; do
    echo "== $(basename $f .so)"
    AEFFT_LIB=$f PROF=1 python3 $R/tools/cfgstep.py $cfg 400 | head -${AB_LINES:-8}
  done
done
