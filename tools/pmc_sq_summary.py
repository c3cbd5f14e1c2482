#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ pass (with --kernel-trace) of bench.py into profiles/*.json.
usage: pmc_sq_summary.py <pmc_dir> <out_json>
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); clock = GRBM_GUI_ACTIVE/8/duration
(MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
import collections
import csv
import glob
import json
import sys


def main():
    d, out_json = sys.argv[1], sys.argv[2]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/*/*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if not any(t in name for t in ("rec_", "gru2_", "lstm_rr", "gru_rr", "lstm_fk", "gru_fk", "conv1d", "spectrogram", "bptt_", "outer_mfma", "dense_frag3", "frag3_pack")):
                continue
            key = "%s grid=%d" % (name, int(row["Grid_Size"]))
            vals[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if "Start_Timestamp" in row and "End_Timestamp" in row:
                vals[key]["_dur_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    out = {"_how": "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                   "SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline (stack, 512 utterances x 1000 frames). mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
                   "(GRBM_GUI_ACTIVE/8 * 1024 SIMDs); clock_GHz = GRBM_GUI_ACTIVE/8/duration.", "kernels": {}}
    for k, c in vals.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        e = {}
        if "_dur_ns" in m:
            e["duration_us"] = round(m["_dur_ns"] / 1e3, 1)
            if gui:
                e["clock_GHz"] = round(gui / m["_dur_ns"], 3)
        if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            e["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024), 3)
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            if "SQ_WAIT_ANY" in m: e["wait_any_frac_of_wave_cycles"] = round(m["SQ_WAIT_ANY"] / wc, 3)
            if "SQ_LDS_BANK_CONFLICT" in m: e["lds_bank_conflict_frac_of_wave_cycles"] = round(m["SQ_LDS_BANK_CONFLICT"] / wc, 4)
        if m.get("SQ_WAVES"):
            e["valu_insts_per_wave"] = round(m.get("SQ_INSTS_VALU", 0.0) / m["SQ_WAVES"], 1)
        out["kernels"][k] = e
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
