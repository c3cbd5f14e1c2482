#!/bin/bash
# Build a second copy of the library with extra hipcc flags (diagnostics / A-B variants) without touching the product
# build: tools/build_variant.sh <out.so> <extra hipcc flags...>     e.g.  tools/build_variant.sh gpurun_out/libs/libdbg.so -DNNTK_SPEC_DBG
set -e
R=$(cd $(dirname $0)/.. && pwd)
OUT=$1; shift
T=$(mktemp -d)
for f in runtime activation conv_1d recurrent dense spectrogram mel train; do
  gcc -O2 -fPIC -std=gnu11 -I$R/include -c $R/nntoolkitcore_amd/csrc/host/$f.c -o $T/$f.c.o
done
for f in runtime conv1d conv1d_s2 recurrent recurrent_rr spectrogram dist conv1d_grad train; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function "$@" -c $R/nntoolkitcore_amd/csrc/hip/$f.hip -o $T/$f.hip.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $T/*.o
rm -rf $T
echo built $OUT
