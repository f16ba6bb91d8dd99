#!/bin/bash
# dev tool: an experiment build of the library -- build_x/libaefft_x<TAG>.so = the product objects with the named sources recompiled
# under extra compiler flags.   tools/mkx.sh TAG "-DAEFFT_X_FOO=1" fft_kernels [more sources]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/autoencoder-fft_amd/csrc
TAG=$1; FLAGS=$2; shift 2
make -s -j8 -C $C
mkdir -p $R/build_x/obj_$TAG
OBJS=""
for o in $C/build/*.o; do
  b=$(basename $o .o); use=$o
  for s in "$@"; do
    if [ "$s" = "$b" ]; then
      src=$C/$b.hip; [ -f $src ] || src=$C/$b.cpp
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize $FLAGS -x hip -c -o $R/build_x/obj_$TAG/$b.o $src
      use=$R/build_x/obj_$TAG/$b.o
    fi
  done
  OBJS="$OBJS $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_x/libaefft_x$TAG.so $OBJS
echo "built build_x/libaefft_x$TAG.so ($FLAGS: $*)"
