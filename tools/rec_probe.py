#!/usr/bin/env python3
"""Time one recurrent layer at a given shape for each value of a library option (nntk_hip_set_option): rec_probe.py lstm|gru|rnn H B T VAR=v1,v2"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    import torch
    from nntoolkitcore_amd import capi, layers as NL
    kind, H, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    var, vals = sys.argv[5].split("="); vals = vals.split(",")
    opt = var[5:].lower() if var.startswith("NNTK_") else var      # option name of nntk_hip_set_option
    torch.cuda.set_device(0); capi.load(); NL.use_torch_stream()
    G = {"lstm": 4, "gru": 3, "rnn": 1}[kind]
    r = np.random.default_rng(0)
    seq = os.environ.get("REC_PROBE_SEQ", "1") == "1"      # REC_PROBE_SEQ=0: return_sequences = false (no per-step output store)
    mk = {"lstm": lambda: NL.LSTM(128, H, seq, T), "gru": lambda: NL.GRU(128, H, seq, T), "rnn": lambda: NL.RNN(128, H, seq, T)}[kind]
    l = mk()
    l.set_weights(r.standard_normal((128, G * H)).astype(np.float32) * 0.05, r.standard_normal((H, G * H)).astype(np.float32) * H ** -0.5,
                  np.zeros(G * H, np.float32), np.zeros(G * H, np.float32))
    x = torch.randn(B, T, 128, device="cuda"); out = torch.empty((B, T, H) if seq else (B, H), device="cuda")
    res = {v: [] for v in vals}
    for v in vals:
        capi.set_option(opt, v); l.apply_device(x, out=out)
    torch.cuda.synchronize()
    for _ in range(5):
        for v in vals:
            capi.set_option(opt, v)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); l.apply_device(x, out=out); e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1))
    for v in vals: print(kind, "H=%d B=%d T=%d" % (H, B, T), var, "=", v, "med %.3f ms" % np.median(res[v]))
    l.destroy()

if __name__ == "__main__":
    main()
