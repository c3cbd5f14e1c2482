#!/bin/bash
# Round-end measurement pass on the GPU box: benches, rocprofv3 kernel stats, PMC passes.
# usage (from the repo root, inside gpurun): bash tools/final_profile.sh <outdir-under-gpurun_out> [bench|prof]
# (two gpurun calls: "bench" = the bench lines and logs, "prof" = rocprofv3 kernel stats and PMC passes; default both)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-final}
mkdir -p $O
cd $R
PART=${2:-all}
if [ "$PART" != prof ]; then
for w in stack gru conv spectrogram; do
  timeout -k 10 300 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; tail -c 300 $O/bench_$w.json; echo
done
NNTK_GEMM_SPLIT_BF16=0 timeout -k 10 300 python bench.py --workload conv --no-cpu-baseline > $O/bench_conv_exact.json 2> $O/bench_conv_exact.err; tail -c 300 $O/bench_conv_exact.json; echo
NNTK_GEMM_SPLIT_BF16=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_stack_exact.json 2> $O/bench_stack_exact.err; tail -c 300 $O/bench_stack_exact.json; echo
timeout -k 10 300 python tools/split_error.py 2>&1 | grep -v amdgpu.ids > $O/split_error.log; timeout -k 10 200 python tools/split_error.py --stress 2>&1 | grep -v amdgpu.ids >> $O/split_error.log; echo split_error $?
timeout -k 10 300 python tools/conv_probe.py gemm_split_bf16=0,1 2>&1 | grep -v amdgpu.ids > $O/conv_probe_ab.log; cat $O/conv_probe_ab.log
timeout -k 10 300 python tools/train_bench.py 2>&1 | grep -v amdgpu.ids > $O/train_bench.log; cat $O/train_bench.log
timeout -k 10 300 python tools/rr_repeat_check.py 40 2>&1 | grep -v amdgpu.ids > $O/rr_repeat_check.log; tail -1 $O/rr_repeat_check.log
timeout -k 10 300 python tools/rec_ab.py 1024 500 2>&1 | grep -v amdgpu.ids > $O/rec_ab.log; cat $O/rec_ab.log
NNTK_REC_FUSED2=1 timeout -k 10 300 python bench.py --workload gru --no-cpu-baseline > $O/bench_gru_fused.json 2> /dev/null; tail -c 200 $O/bench_gru_fused.json; echo
# the two-f16-image kernels: race detector at the full grid, the 256-wide-input instantiation against the exact kernels, the contraction microbenchmark
timeout -k 10 300 python tools/hf_soak.py 8 2>&1 | grep -v amdgpu.ids > $O/hf_soak.log; tail -1 $O/hf_soak.log
timeout -k 10 200 python tools/hf_wide_time.py 2>&1 | grep -v amdgpu.ids > $O/hf_wide_time.log; cat $O/hf_wide_time.log
# north_star's own batch on ONE GPU (4096 utterances = 8 back-to-back launches of the 256-workgroup LSTM kernel), and the round-3 route
timeout -k 10 300 python bench.py --batch-per-gpu 4096 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_stack_b4096.json 2> $O/bench_stack_b4096.err; tail -c 300 $O/bench_stack_b4096.json; echo
# the LSTM -> dense seam on the frag3 form (six products; bit-identical to the f32 route) instead of the default FRAG2H form (three products)
NNTK_DENSE_F16X2=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_stack_frag3route.json 2> /dev/null; tail -c 200 $O/bench_stack_frag3route.json; echo
NNTK_BENCH_STACK_F32=1 NNTK_REC_XF=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_stack_f32route.json 2> /dev/null; tail -c 200 $O/bench_stack_f32route.json; echo
NNTK_CONV_FLATK=0 timeout -k 10 300 python bench.py --workload conv --no-cpu-baseline > $O/bench_conv_chunked.json 2> /dev/null; tail -c 200 $O/bench_conv_chunked.json; echo
# the GRU pair with the full-K family off (both layers split-K: the round-4 default) and on for every shape it takes (layer 1 too)
NNTK_REC_FK=0 timeout -k 10 300 python bench.py --workload gru --no-cpu-baseline > $O/bench_gru_fk0.json 2> /dev/null; tail -c 200 $O/bench_gru_fk0.json; echo
NNTK_REC_FK=1 timeout -k 10 300 python bench.py --workload gru --no-cpu-baseline > $O/bench_gru_fk1.json 2> /dev/null; tail -c 200 $O/bench_gru_fk1.json; echo
fi
[ "$PART" = bench ] && { ls $O; exit 0; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lstm_train -- python3 $R/tools/lstm_train_prof.py > $O/lstm_train_prof.log 2>&1; grep "^fwd" $O/lstm_train_prof.log
for f in $(ls $O/prof_lstm_train/*/*kernel_stats.csv 2>/dev/null); do cut -c1-150 $f | head -8; done
for w in stack gru conv spectrogram; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline --steps 5 > $O/prof_$w.log 2>&1
  for f in $(ls $O/prof_$w/*/*kernel_stats.csv 2>/dev/null); do cut -c1-150 $f | head -7; done
done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1; echo fetch $?
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1; echo write $?
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1; echo sq $?
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq_gru -- python3 $R/bench.py --workload gru --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq_gru.log 2>&1; echo sq_gru $?
for wb in gru:1024 conv:1024 spectrogram:256; do
  w=${wb%%:*}
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$w -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_$w.log 2>&1; echo fetch_$w $?
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$w -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_$w.log 2>&1; echo write_$w $?
done
# memory-path counters (L1 / L2 requests, hit rates) of the stack kernels; summarised by hand into profiles/r04_pmc_mem.json this round.
# (TA_* / TCP_PENDING_STALL counters in one pass made rocprofv3 run into the 300 s limit on this pool: left out.)
timeout -k 10 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $O/pmc_mem -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_mem.log 2>&1; echo mem $?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_elementwise -- python3 $R/tools/elementwise_bench.py > $O/elementwise.log 2>&1; cat $O/elementwise.log | grep -v amdgpu.ids
for f in $(ls $O/prof_elementwise/*/*kernel_stats.csv 2>/dev/null); do cut -c1-150 $f | head -7; done
cd $R
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json 512 stack > /dev/null && echo traffic ok
for wb in gru:1024 conv:1024 spectrogram:256; do
  w=${wb%%:*}; b=${wb##*:}
  python3 tools/pmc_summary.py $O/pmc_fetch_$w $O/pmc_write_$w $O/pmc_traffic_$w.json $b $w > /dev/null && echo traffic_$w ok
done
python3 tools/pmc_sq_summary.py $O/pmc_sq $O/pmc_sq.json > /dev/null && echo sq ok
python3 tools/pmc_sq_summary.py $O/pmc_sq_gru $O/pmc_sq_gru.json > /dev/null && echo sq_gru ok
ls $O
