#!/usr/bin/env python3
"""What the one-gate RNN (layers/rnn.c:144-166) costs per timestep on the exact-f32 kernels it runs on, next to a GRU of the same shape on the
register-resident kernels: python tools/rnn_time.py [B] [T]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nntoolkitcore_amd import capi, layers as NL
torch.cuda.set_device(0); L = capi.load(); NL.use_torch_stream()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
r = np.random.default_rng(0)
u = lambda *s, sc=1.0: (sc * r.uniform(-1, 1, s)).astype(np.float32)
for I, H in ((128, 256), (256, 256), (128, 512)):
    x = torch.from_numpy(u(B, T, I)).cuda()
    rows = []
    for name in ("rnn", "gru"):
        G = 1 if name == "rnn" else 3
        lay = NL.RNN(I, H, True, T) if name == "rnn" else NL.GRU(I, H, True, T)
        lay.set_weights(u(I, G * H, sc=I ** -0.5), u(H, G * H, sc=H ** -0.5), u(G * H, sc=0.1), u(G * H, sc=0.1))
        out = torch.empty((B, T, H), device="cuda")
        lay.apply_device(x, out=out); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); lay.apply_device(x, out=out); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        rows.append("%s %.2f ms = %.2f us/step (%s)" % (name, best * 1e3, best * 1e6 / T, L.nntk_hip_last_recurrent_kernel().decode()))
        lay.destroy()
    print("B=%d T=%d in=%d H=%d: " % (B, T, I, H) + " | ".join(rows))
