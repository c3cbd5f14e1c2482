#!/bin/bash
# Timing ablations of bptt_persistent_kernel: one library per compile-time mask (BPTT_DBG in train.hip).  Built HERE (no GPU
# needed) from the product build's objects into nntoolkitcore_amd/lib/variants/ (git-ignored, travels to the GPU box).
# usage: tools/bptt_ablate.sh build <mask>...   |   tools/bptt_ablate.sh run <mask>...   (run: on the GPU box)
set -e
R=$(cd $(dirname $0)/.. && pwd)
V=$R/nntoolkitcore_amd/lib/variants
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p $V
  for m in "$@"; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -fvisibility=default -DNNTK_BPTT_DBG=$m -c $R/nntoolkitcore_amd/csrc/hip/train.hip -o $V/train_$m.o &
  done
  wait
  for m in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libbptt_$m.so $(ls $R/nntoolkitcore_amd/lib/obj/*.o | grep -v "/train.hip.o") $V/train_$m.o
    rm -f $V/train_$m.o
  done
  ls -la $V
else
  for m in "$@"; do
    echo "== mask $m"
    NNTK_LIB=$V/libbptt_$m.so timeout -k 10 120 python $R/tools/lstm_train_prof.py 2>&1 | grep "^fwd"
  done
fi
