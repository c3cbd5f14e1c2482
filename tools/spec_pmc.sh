#!/bin/bash
# PMC passes over K1 alone (tools/spec_probe.py): tools/spec_pmc.sh <outdir-under-gpurun_out>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-spec_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/spec_probe.py --reps 3 > $O/a.log 2>&1; echo a $?
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/b -- python3 $R/tools/spec_probe.py --reps 3 > $O/b.log 2>&1; echo b $?
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $O/c -- python3 $R/tools/spec_probe.py --reps 3 > $O/c.log 2>&1; echo c $?
cd $R
python3 - <<PY
import csv, glob, collections
for p in ("a","b","c"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob("$O/%s/*/*counter_collection.csv" % p):
        for row in csv.DictReader(open(path)):
            if "spectrogram512" not in row["Kernel_Name"]: continue
            key = "grid=%s" % row["Grid_Size"]
            vals[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in vals.items():
        print(p, k, {n: round(sum(v)/len(v)) for n, v in c.items()})
PY
