#!/usr/bin/env python3
"""Where do the streaming and the batch recurrent paths differ?  (debug helper)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nntoolkitcore_amd import capi, layers as NL
torch.cuda.set_device(0); capi.load()
r = np.random.default_rng(0)
for cell, I, H in (("rnn", 40, 64), ("gru", 128, 256), ("lstm", 128, 512)):
    G = {"gru": 3, "lstm": 4, "rnn": 1}[cell]
    W, U = (r.uniform(-1, 1, (I, G * H)) * I ** -0.5).astype(np.float32), (r.uniform(-1, 1, (H, G * H)) * H ** -0.5).astype(np.float32)
    bi, bh = r.uniform(-.1, .1, G * H).astype(np.float32), r.uniform(-.1, .1, G * H).astype(np.float32)
    for T, zero_u, zero_w in ((1, False, False), (2, False, False), (2, True, False), (2, False, True)):
        x = r.uniform(-1, 1, (T, I)).astype(np.float32)
        outs = {}
        for mode in ("auto", "0"):
            capi.set_option("rec_stream", mode)
            mk = {"gru": lambda: NL.GRU(I, H, True, T), "lstm": lambda: NL.LSTM(I, H, True, T, v2=True), "rnn": lambda: NL.RNN(I, H, True, T)}[cell]
            l = mk(); l.set_weights(W * (0 if zero_w else 1), U * (0 if zero_u else 1), bi, bh)
            outs[mode] = l.apply(x); l.destroy()
        d = np.abs(outs["auto"] - outs["0"])
        print(cell, "T=%d zeroU=%d zeroW=%d" % (T, zero_u, zero_w), "max diff %.3e" % d.max(), "per step", [float("%.2e" % v) for v in d.max(axis=1)],
              "ndiff", int((d > 0).sum()), "of", d.size)
