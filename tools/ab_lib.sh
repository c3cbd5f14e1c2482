#!/bin/bash
# Compare two builds of the library on the SAME box: tools/ab_lib.sh <workload> <libA.so> <libB.so> [rounds]
# (each library runs in its own process; interleaved A B A B to average drift)
W=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq $R); do
  for L in $A $B; do
    NNTK_LIB=$L python bench.py --workload $W --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', 'ms=%.3f'%d['ms_per_step'], d['phase_ms'])"
  done
done
