#!/bin/bash
# A/B of library variants (tools/build_variant.py) on one box, alternating processes: bash tools/ab_lib.sh <workload> <rounds> <lib> <lib> ...
W=$1; R=$2; shift 2
for r in $(seq 1 $R); do
  for L in "$@"; do
    NNTK_LIB=$L timeout -k 10 200 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],3), d['phase_ms'], round(d['roofline']['ms_per_launch'],3))"
  done
done
