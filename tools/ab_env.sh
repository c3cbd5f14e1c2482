#!/bin/bash
# A/B of bench.py under environment settings, alternating processes on one box.
# usage (inside gpurun): bash tools/ab_env.sh <out.log> <rounds> <bench args...> -- "ENV1=.. ENV2=.." "ENVA=.." ...      ("" = defaults)
OUT=$1; ROUNDS=$2; shift 2; mkdir -p $(dirname $OUT)
ARGS=()
while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
for r in $(seq $ROUNDS); do
  for e in "$@"; do
    env $e timeout -k 10 300 python bench.py "${ARGS[@]}" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[%s]' % sys.argv[1], round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['phase_ms'].items()})" "$e" >> $OUT || exit 1
  done
done
cat $OUT
