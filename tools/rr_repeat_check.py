#!/usr/bin/env python3
"""Repeat-run check of the register-resident recurrent kernels' hand-off at full size, in ONE process: N launches of the two-layer GRU-256
stack (pending-pattern hand-off, 256 workgroups), of LSTM-512 at the stack's size (flag protocol) and of an LSTM-256 with ragged tiles,
every result compared bit for bit with the first.   usage: python tools/rr_repeat_check.py [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nntoolkitcore_amd import capi, layers as NL
torch.cuda.set_device(0); L = capi.load(); NL.use_torch_stream()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
r = np.random.default_rng(5)
u = lambda *s, sc=1.0: (sc * r.uniform(-1, 1, s)).astype(np.float32)
def gru(i, h, t):
    g = NL.GRU(i, h, True, t); g.set_weights(u(i, 3 * h, sc=i ** -0.5), u(h, 3 * h, sc=h ** -0.5), u(3 * h, sc=0.1), u(3 * h, sc=0.1)); return g
def lstm(i, h, t):
    m = NL.LSTM(i, h, True, t, v2=True); m.set_weights(u(i, 4 * h, sc=i ** -0.5), u(h, 4 * h, sc=h ** -0.5), u(4 * h, sc=0.1), u(4 * h, sc=0.1)); return m
cases = []
g1, g2 = gru(128, 256, 1000), gru(256, 256, 1000)
x = torch.randn(1024, 1000, 128, device="cuda")
cases.append(("2 x GRU-256, B = 1024, T = 1000", lambda: NL.gru_stack2_apply_device(g1, g2, x)))
l5 = lstm(128, 512, 996); x5 = torch.randn(512, 996, 128, device="cuda")
cases.append(("LSTM-512, B = 512, T = 996", lambda: l5.apply_device(x5)))
l2 = lstm(72, 256, 300); x2 = torch.randn(333, 300, 72, device="cuda")
cases.append(("LSTM-256, in = 72, B = 333 (ragged tiles), T = 300", lambda: l2.apply_device(x2)))
for name, fn in cases:
    first = fn().clone(); kern = L.nntk_hip_last_recurrent_kernel().decode()
    bad = sum(0 if torch.equal(fn(), first) else 1 for _ in range(N))
    torch.cuda.synchronize()
    print("%-52s %-22s %d launches, %d differ from the first, device status %d" % (name, kern, N, bad, L.nntk_hip_device_status()))
    assert bad == 0 and L.nntk_hip_device_status() == 0
print("ok")
