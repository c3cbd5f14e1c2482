#!/bin/bash
# Timing ablations of lstm_rr_kernel: one library per compile-time mask (see RR_DBG in recurrent_rr.hip), built on the GPU box.
# usage: tools/rr_ablate.sh <outdir> <mask> [<mask> ...]      extra hipcc flags for recurrent_rr.hip via RR_EXTRA
set -e
R=$(cd $(dirname $0)/.. && pwd)
OUT=$1; shift
T=$(mktemp -d)
for f in runtime activation conv_1d recurrent dense spectrogram mel train; do
  gcc -O2 -fPIC -std=gnu11 -I$R/include -c $R/nntoolkitcore_amd/csrc/host/$f.c -o $T/$f.c.o &
done
for f in runtime conv1d conv1d_s2 recurrent spectrogram dist conv1d_grad train; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -c $R/nntoolkitcore_amd/csrc/hip/$f.hip -o $T/$f.hip.o &
done
for m in "$@"; do
  # a mask may carry extra defines after a colon:  0:-DRR_NPRE=4
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -DNNTK_RR_DBG=${m%%:*} $(echo "${m#*:}" | sed "s/^${m%%:*}\$//; s/,/ /g") $RR_EXTRA -c $R/nntoolkitcore_amd/csrc/hip/recurrent_rr.hip -o $T/rr_$(echo $m | tr -c "A-Za-z0-9\n" _).o &
done
wait
mkdir -p $OUT
for m in "$@"; do
  tag=$(echo $m | tr -c "A-Za-z0-9\n" _)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/librr_$tag.so $T/*.c.o $T/*.hip.o $T/rr_$tag.o
  echo "== $m" | tee -a $OUT/ablate.log
  NNTK_LIB=/tmp/librr_$tag.so timeout -k 10 120 python $R/tools/rr_ablate.py ${m%%:*} 2>&1 | grep "^dbg" | tee -a $OUT/ablate.log
done
rm -rf $T
