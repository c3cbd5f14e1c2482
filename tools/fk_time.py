#!/usr/bin/env python3
"""Times single recurrent layers on the GPU (device pointers, median of N launches, frag3 pack included unless --f3).
usage: python tools/fk_time.py [--reps N] cell:in:H[:B[:T]] ...      e.g.  gru:128:256 gru:256:256 lstm:128:512:512:996"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nntoolkitcore_amd import capi, layers as NL
    torch.cuda.set_device(0); L = capi.load(); NL.use_torch_stream()
    capi.set_option("rec_fk", int(os.environ.get("NNTK_REC_FK", "1")))
    reps = 7
    args = sys.argv[1:]
    if args and args[0] == "--reps":
        reps = int(args[1]); args = args[2:]
    r = np.random.default_rng(3)
    u = lambda *sh, sc=1.0: r.uniform(-sc, sc, sh).astype(np.float32)
    out = []
    for spec in args or ["gru:128:256", "gru:256:256"]:
        f = spec.split(":")
        kind, I, H = f[0], int(f[1]), int(f[2])
        B = int(f[3]) if len(f) > 3 else 1024
        T = int(f[4]) if len(f) > 4 else 1000
        G = 4 if kind == "lstm" else 3
        lay = NL.LSTM(I, H, True, T, v2=True) if kind == "lstm" else NL.GRU(I, H, True, T)
        lay.set_weights(u(I, G * H, sc=I ** -0.5), u(H, G * H, sc=H ** -0.5), u(G * H, sc=0.1), u(G * H, sc=0.1))
        x = torch.randn(B, T, I, device="cuda"); h = torch.empty(B, T, H, device="cuda")
        for _ in range(2):
            lay.apply_device(x, out=h)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lay.apply_device(x, out=h); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out.append("%s %s med %.3f min %.3f ms" % (spec, L.nntk_hip_last_recurrent_kernel().decode(), np.median(ts), np.min(ts)))
        lay.destroy()
    print(" | ".join(out), flush=True)


if __name__ == "__main__":
    main()
