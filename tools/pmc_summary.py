#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (FETCH_SIZE pass, WRITE_SIZE pass, optional SQ pass) of a bench.py
run into profiles/*.json.  usage: pmc_summary.py <fetch_dir> <write_dir> <out_json> <utterances_per_gpu> [workload]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nntoolkitcore_amd import capi  # noqa: E402


def source_hash():
    """the hash the BUILT library carries (the profile belongs to the binary that ran, not to the source tree)"""
    return (capi.load().nntk_build_source_hash() or b"").decode()


def agg(d, counter):
    out = collections.defaultdict(list)
    for path in glob.glob(d + "/*/*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                out["%s grid=%d" % (name, int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
    return out


def main():
    fetch_dir, write_dir, out_json, B = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    workload = sys.argv[5] if len(sys.argv) > 5 else "stack"
    f, w = agg(fetch_dir, "FETCH_SIZE"), agg(write_dir, "WRITE_SIZE")
    out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes (each with --kernel-trace only) of "
                   "`python bench.py --workload <workload> --steps 2 --warmup 1 --no-cpu-baseline`. FETCH_SIZE / WRITE_SIZE are KiB per dispatch. "
                   "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE "
                   "reports half of a wide coalesced read, so it is doubled; the factor is calibrated for 16-B/lane streams, "
                   "for narrower reads it is an upper bound on the read side.",
           "workload": workload, "utterances_per_gpu": B, "source_hash": source_hash(), "kernels": {}}
    for k, v in f.items():
        if not any(t in k for t in ("rec_", "gru2_", "lstm_rr", "gru_rr", "lstm_fk", "gru_fk", "conv1d", "spectrogram", "bptt_", "outer_mfma", "dense_frag3", "frag3_pack")) or k not in w:
            continue
        fv, wv = sum(v) / len(v), sum(w[k]) / len(w[k])
        out["kernels"][k] = {"FETCH_SIZE_KiB": round(fv, 1), "WRITE_SIZE_KiB": round(wv, 1),
                             "hbm_bytes_per_launch": int((2 * fv + wv) * 1024), "dispatches": len(v)}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
