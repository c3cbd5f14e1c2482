#!/bin/bash
# Samples rocm-smi (power, clocks) while bench.py runs a long timed region: is the step power-limited?
# usage (inside gpurun): bash tools/power_probe.sh <out.log> [bench args...]
OUT=$1; shift
mkdir -p $(dirname $OUT)
rocm-smi --showpower --showclocks --showmaxpower > $OUT.idle 2>&1
python bench.py --steps 3000 --warmup 5 --no-cpu-baseline "$@" > $OUT.json 2>/dev/null &
BP=$!
sleep 14     # (import + weights + warm-up)
for i in 1 2 3 4 5 6 7 8; do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk" | head -6 >> $OUT
  echo "--" >> $OUT
  sleep 1
done
wait $BP
tail -c 400 $OUT.json; echo; cat $OUT; grep -i "max" $OUT.idle | head -3
