#!/bin/bash
# Copies the round-end measurement pass (tools/final_profile.sh -> gpurun_out/<dir>) into profiles/ under a round prefix.
# usage: bash tools/collect_profiles.sh <dir-under-gpurun_out> <prefix>        e.g.  r03_final r03
R=$(cd $(dirname $0)/.. && pwd)
S=$R/gpurun_out/$1; P=$R/profiles/$2
for w in stack gru conv spectrogram conv_exact stack_exact gru_fused gru_fk0 gru_fk1 stack_b4096 stack_f32route stack_frag3route conv_chunked; do [ -s $S/bench_$w.json ] && cp $S/bench_$w.json ${P}_bench_$w.json; done
for w in stack gru conv spectrogram elementwise lstm_train; do
  f=$(ls -t $S/prof_$w/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f ${P}_${w}_kernel_stats.csv
done
for f in pmc_traffic pmc_traffic_gru pmc_traffic_conv pmc_traffic_spectrogram pmc_sq pmc_sq_gru; do [ -s $S/$f.json ] && cp $S/$f.json ${P}_$f.json; done
[ -s $S/split_error.log ] && cp $S/split_error.log ${P}_split_error.log
[ -s $S/conv_probe_ab.log ] && cp $S/conv_probe_ab.log ${P}_conv_probe_split_vs_exact.log
[ -s $S/train_bench.log ] && cp $S/train_bench.log ${P}_train_bench.log
[ -s $S/rec_ab.log ] && cp $S/rec_ab.log ${P}_rec_ab.log
[ -s $S/rr_repeat_check.log ] && cp $S/rr_repeat_check.log ${P}_rr_repeat_check.log
[ -s $S/hf_soak.log ] && cp $S/hf_soak.log ${P}_hf_soak.log
[ -s $S/hf_wide_time.log ] && cp $S/hf_wide_time.log ${P}_hf_wide_time.log
[ -s $S/elementwise.log ] && grep -v "rocprofv3\|amdgpu.ids\|output_stream\|tool.cpp" $S/elementwise.log > ${P}_elementwise.log
ls $R/profiles | grep "^$2_" | wc -l
