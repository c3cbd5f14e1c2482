mkdir -p gpurun_out/r04d
for v in ring direct f32 ring2 direct2; do
  case $v in
    ring*) env="NNTK_DENSE_FRAG3=1";;
    direct*) env="NNTK_DENSE_FRAG3=2";;
    f32) env="NNTK_BENCH_STACK_F32=1";;
  esac
  env $env timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04d/bench_stack_$v.json 2> gpurun_out/r04d/bench_stack_$v.err
  echo "$v rc=$?"; python -c "
import json,sys
d=json.load(open('gpurun_out/r04d/bench_stack_$v.json')); print(d['ms_per_step'], d['phase_ms'], d['roofline']['ms_per_launch'] if d.get('roofline') else None)"
done
timeout -k 10 200 python bench.py --workload gru --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04d/bench_gru.json 2> gpurun_out/r04d/bench_gru.err
python -c "
import json
d=json.load(open('gpurun_out/r04d/bench_gru.json')); print(d['ms_per_step'], d['phase_ms'])"
