#!/usr/bin/env python3
"""Time the conv / GEMM kernel alone on the BASELINE shapes for each value of a library option.
usage: [NNTK_LIB=variant.so] python tools/conv_probe.py [OPTION=v1,v2,...] [--fixed OPTION=v] [--reps N] [--cases a,b]
(diagnostics build -DNNTK_CONV_DBG, option conv_dbg: 1 no stores, 2 no MFMAs, 4 no weight loads in the loop, 8 no LDS
reads, 16 no window split/staging, 32 no window loads; sums allowed)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {  # name: B, T, Cin, Cout, k
    "conv3": (1024, 1000, 40, 128, 5),
    "stackconv": (512, 1000, 257, 128, 5),
    "xw": (512, 996, 128, 2048, 1),
    "tdd": (512, 996, 512, 1000, 1),
}


def main():
    import torch
    from nntoolkitcore_amd import capi, layers as NL
    opt, vals, reps, cases = None, [None], 10, list(CASES)
    args = sys.argv[1:]
    fixed = []
    i = 0
    while i < len(args):
        a = args[i]
        if a == "--reps": reps = int(args[i + 1]); i += 2; continue
        if a == "--cases": cases = args[i + 1].split(","); i += 2; continue
        if a == "--fixed": fixed.append(args[i + 1].split("=")); i += 2; continue
        if "=" in a: opt, v = a.split("="); vals = v.split(",")
        i += 1
    torch.cuda.set_device(0); capi.load(); NL.use_torch_stream()
    for k_, v_ in fixed: capi.set_option(k_, v_)
    g = torch.Generator(device="cuda").manual_seed(1)
    for name in cases:
        B, T, Cin, Cout, k = CASES[name]
        x = torch.rand(B, T, Cin, device="cuda", generator=g) - 0.5
        conv = NL.Conv1d(Cin, Cout, k, 1, T)
        conv.set_weights((np.random.default_rng(0).uniform(-1, 1, (Cout, Cin, k)) * (Cin * k) ** -0.5).astype(np.float32),
                         np.zeros(Cout, np.float32))
        out = torch.empty(B, T - k + 1, Cout, device="cuda")
        flops = 2.0 * B * (T - k + 1) * Cout * Cin * k
        nbytes = 4.0 * (x.numel() + out.numel())
        res = {v: [] for v in vals}
        for v in vals:
            if opt: capi.set_option(opt, v)
            for _ in range(2): conv.apply_device(x, out=out)
        torch.cuda.synchronize()
        for _ in range(5):
            for v in vals:
                if opt: capi.set_option(opt, v)
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps): conv.apply_device(x, out=out)
                e1.record(); torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) / reps)
        for v in vals:
            ms = float(np.median(res[v]))
            print("%-10s %s=%-4s %8.1f us  %6.1f TFLOP/s  %5.2f TB/s" % (name, opt, v, ms * 1e3, flops / ms / 1e9, nbytes / ms / 1e9), flush=True)
        conv.destroy()
        del x, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
