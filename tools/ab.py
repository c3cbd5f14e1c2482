#!/usr/bin/env python3
"""A/B kernel variants in ONE process, interleaved rounds (guide rule 24).
usage: python tools/ab.py <workload> OPTION=v1,v2,... [--rounds N] [--batch B] [--frames F]
Prints the median / min per-phase milliseconds for each variant."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from nntoolkitcore_amd import capi, layers as NL
    wl_name = sys.argv[1]
    var, vals = sys.argv[2].split("=")
    vals = vals.split(",")
    opt = var[5:].lower() if var.startswith("NNTK_") else var      # option name of nntk_hip_set_option
    rounds, batch, frames = 7, 0, 1000
    for i, a in enumerate(sys.argv):
        if a == "--rounds": rounds = int(sys.argv[i + 1])
        if a == "--batch": batch = int(sys.argv[i + 1])
        if a == "--frames": frames = int(sys.argv[i + 1])
    torch.cuda.set_device(0)
    capi.load()
    NL.use_torch_stream()
    defaults = {"stack": 512, "spectrogram": 256, "conv": 1024, "gru": 1024}
    B = batch or defaults[wl_name]
    weights = bench.make_weights(wl_name, 3)
    wl = bench.Workload(wl_name, B, frames, weights, torch, NL)
    res = {v: {} for v in vals}
    for v in vals:                      # warm-up each variant
        capi.set_option(opt, v)
        wl.step()
    torch.cuda.synchronize()
    for r in range(rounds):
        for v in vals:
            capi.set_option(opt, v)
            ev = wl.step(timed=True)
            torch.cuda.synchronize()
            for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
                res[v].setdefault(n1, []).append(e0.elapsed_time(e1))
    for v in vals:
        print(var, "=", v, {k: "med %.3f min %.3f ms" % (np.median(x), np.min(x)) for k, x in res[v].items()})
    wl.destroy()


if __name__ == "__main__":
    main()
