#!/bin/bash
# Compile-time timing ablations of the full-K recurrent kernels (WRONG results; one library per flag set, see recurrent_rr_common.hpp
# for the NNTK_RR_DBG masks).   usage (inside gpurun): bash tools/fk_ablate.sh <outdir-under-gpurun_out> "<flags>" ["<flags>" ...]
# e.g.  bash tools/fk_ablate.sh ab1 "" "-DNNTK_RR_DBG=1" "-DFK_ND_4=6"
O=gpurun_out/$1; shift; mkdir -p $O/libs
i=0
for f in "$@"; do
  i=$((i+1))
  if [ -z "$f" ]; then L=nntoolkitcore_amd/lib/libnntoolkitcore_hip.so; else
    L=$O/libs/libv$i.so; python tools/build_variant.py $L recurrent_fk.hip $f > $O/build_$i.log 2>&1 || { echo "build '$f' failed"; tail -3 $O/build_$i.log; continue; }; fi
  echo "[$f] $(NNTK_LIB=$L timeout -k 10 200 python tools/fk_time.py ${FK_SPECS:-gru:128:256 gru:256:256} 2>&1 | grep -v amdgpu.ids | tail -1)" | tee -a $O/ablate.log
done
