#!/usr/bin/env python3
"""Time K1 (spectrogram512_kernel) alone at the stack's size and at config 2, for each value of a library option.
usage: [NNTK_LIB=variant.so] python tools/spec_probe.py [OPTION=v1,v2,...] [--reps N]
(diagnostics build -DNNTK_SPEC_DBG: conv_dbg=1 no stores, 2 no sample loads, 4 no LDS passes, 8 no shuffles; sums allowed)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nntoolkitcore_amd import capi, layers as NL
    opt, vals = None, [None]
    reps = 20
    for i, a in enumerate(sys.argv[1:]):
        if "=" in a:
            opt, v = a.split("="); vals = v.split(",")
        if a == "--reps": reps = int(sys.argv[i + 2])
    torch.cuda.set_device(0); capi.load(); NL.use_torch_stream()
    for name, B, N in (("stack-size", 512, 240 + 160 * 1000), ("config2", 256, 16000)):
        spec = NL.Spectrogram(512, 400, 240, N)
        x = (0.1 * torch.randn(B, N, device="cuda")).clamp_(-1, 1)
        out = torch.empty((B,) + spec.out_shape, device="cuda")
        nbytes = B * (N * 4 + spec.out_shape[0] * spec.out_shape[1] * 4)
        res = {v: [] for v in vals}
        for v in vals:
            if opt: capi.set_option(opt, v)
            for _ in range(3): spec.apply_device(x, out=out)
        torch.cuda.synchronize()
        for _ in range(5):
            for v in vals:
                if opt: capi.set_option(opt, v)
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps): spec.apply_device(x, out=out)
                e1.record(); torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) / reps)
        for v in vals:
            ms = float(np.median(res[v]))
            print("%-10s %s=%-4s  %8.2f us/launch  %6.2f TB/s  frac %.3f" % (name, opt, v, ms * 1e3, nbytes / ms / 1e9, nbytes / ms / 1e9 / 8.0), flush=True)
        spec.destroy()


if __name__ == "__main__":
    main()
