#!/bin/bash
# A/B of compile-time variants of ONE translation unit on one box, alternating processes (guide rule 24).
# usage (inside gpurun): bash tools/ab_variants.sh <outdir-under-gpurun_out> <unit.hip> <workload> <rounds> "<flags A>" "<flags B>" ...
# ("" = the product library).  Prints ms/step and the phases of bench.py for every variant and round.
O=gpurun_out/$1; U=$2; W=$3; R=$4; shift 4; mkdir -p $O/libs
i=0; LIBS=()
for f in "$@"; do
  i=$((i+1))
  if [ -z "$f" ]; then LIBS+=("nntoolkitcore_amd/lib/libnntoolkitcore_hip.so"); else
    L=$O/libs/libv$i.so; python tools/build_variant.py $L $U $f > $O/build_$i.log 2>&1 || { echo "build '$f' failed"; tail -3 $O/build_$i.log; LIBS+=(""); continue; }; LIBS+=("$L"); fi
done
for r in $(seq 1 $R); do
  i=0
  for f in "$@"; do
    L=${LIBS[$i]}; i=$((i+1)); [ -z "$L" ] && continue
    NNTK_LIB=$L timeout -k 10 200 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$f]', round(d['ms_per_step'],3), {k: round(v,3) for k,v in (d.get('phase_ms') or {}).items()})" | tee -a $O/ab.log
  done
done
