#!/bin/bash
# PMC passes over the conv / GEMM kernel alone (tools/conv_probe.py): tools/conv_pmc.sh <outdir-under-gpurun_out> [probe args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-conv_pmc}; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAILED=0
run() { n=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 $R/tools/conv_probe.py --reps 2 $PROBE_ARGS > $O/$n.log 2>&1; rc=$?; echo $n $rc
        [ $rc -ne 0 ] && { FAILED=1; echo "pass $n FAILED (rc $rc): $(grep -m1 -i "exceeds\|error" $O/$n.log)"; }; }
PROBE_ARGS="$*"
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run b SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
run c GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
# the cache counters in two passes: TCP_* and TCC_* together exceed what the hardware collects at once (round 2's pass d aborted)
run d TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
run e TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run f TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd $R
python3 - <<PY
import csv, glob, collections
for p in ("a","b","c","d","e","f"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob("$O/%s/*/*counter_collection.csv" % p):
        for row in csv.DictReader(open(path)):
            if "conv1d_mfma" not in row["Kernel_Name"]: continue
            key = row["Kernel_Name"][:40] + " grid=%s" % row["Grid_Size"]
            vals[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in sorted(vals.items()):
        print(p, k, {n: round(sum(v)/len(v)) for n, v in c.items()})
PY
[ $FAILED -eq 0 ] || { echo "conv_pmc: at least one pass failed"; exit 1; }
