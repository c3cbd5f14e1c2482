#!/bin/bash
# Round-end measurement pass on the GPU box: tests, benches, rocprofv3 kernel stats, PMC passes.
# usage (from the repo root, inside gpurun): bash tools/final_profile.sh <outdir-under-gpurun_out>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-final}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for w in stack gru conv spectrogram; do
  timeout -k 10 300 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; tail -c 400 $O/bench_$w.json; echo
done
cd /tmp && export TMPDIR=/tmp
for w in stack gru; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline --steps 3 > $O/prof_$w.log 2>&1
  for f in $(ls $O/prof_$w/*/*kernel_stats.csv 2>/dev/null); do cut -c1-140 $f | head -6; done
done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1; echo fetch $?
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1; echo write $?
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1; echo sq $?
cd $R
ls $O
