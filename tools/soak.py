#!/usr/bin/env python3
"""Random-shape soak of the conv / dense / recurrent kernels against the oracle (440 cases incl. spectrogram geometries and the RNN, ~30 s on the GPU):
python tools/soak.py [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle as O
from nntoolkitcore_amd import capi, layers as NL
torch.cuda.set_device(0); capi.load()
r = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 99)
def u(*s, sc=1.0): return (sc * r.uniform(-1, 1, s)).astype(np.float32)
def close(a, b, tol=2e-5):
    err = np.abs(a - b).max() / (1e-6 + np.abs(b).max())
    assert err < tol, err
n = 0
for _ in range(150):
    cin, cout = int(r.integers(1, 300)), int(r.integers(1, 300))
    k, stride = int(r.integers(1, 10)), int(r.integers(1, 4))
    T = int(r.integers(k, 700)); B = int(r.integers(1, 6))
    x, W, b = u(B, T, cin), u(cout, cin, k, sc=(cin * k) ** -0.5), u(cout, sc=0.2)
    conv = NL.Conv1d(cin, cout, k, stride, T); conv.set_weights(W, b)
    close(conv.apply(x), O.conv1d(x, W, b, stride)); conv.destroy(); n += 1
for _ in range(100):
    ts, I, Ov = int(r.integers(1, 400)), int(r.integers(1, 600)), int(r.integers(1, 1100))
    Bb = int(r.integers(1, 5))
    W, b, x = u(I, Ov, sc=I ** -0.5), u(Ov, sc=0.2), u(Bb, ts, I)
    tdd = NL.TimeDistributedDense(ts, I, Ov); tdd.set_weights(W, b)
    close(tdd.apply(x), O.time_distributed_dense(x, W, b)); tdd.destroy(); n += 1
for _ in range(60):
    I, H = int(r.integers(1, 200)), int(r.integers(1, 140)) * 4
    T, B = int(r.integers(1, 10)), int(r.integers(1, 200))
    x = u(B, T, I)
    W, U, bi, bh = u(I, 3 * H, sc=I ** -0.5), u(H, 3 * H, sc=H ** -0.5), u(3 * H, sc=0.1), u(3 * H, sc=0.1)
    g = NL.GRU(I, H, True, T); g.set_weights(W, U, bi, bh)
    close(g.apply(x), O.gru(x, W, U, bi, bh), 1e-4); g.destroy()
    W, U, bi, bh = u(I, 4 * H, sc=I ** -0.5), u(H, 4 * H, sc=H ** -0.5), u(4 * H, sc=0.1), u(4 * H, sc=0.1)
    l = NL.LSTM(I, H, False, T); l.set_weights(W, U, bi, bh)
    close(l.apply(x), O.lstm(x, W, U, bi, bh, return_sequences=False), 1e-4); l.destroy(); n += 2
for _ in range(40):
    win = int(r.integers(8, 513)); nov = int(r.integers(0, win)); nfft = 512
    N = int(r.integers(win, 6000)); B = int(r.integers(1, 5))
    x = u(B, N, sc=0.1)
    mode = "psd" if r.integers(0, 2) else "magnitude"
    sp = NL.Spectrogram(nfft, win, nov, N, mode=mode, fs=16000, window_name="hamming_window")
    ref = O.spectrogram(x, O.window("hamming", win), nfft, nov, mode=mode, fs=16000)
    got = sp.apply(x)
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-5 * (1e-6 + np.abs(ref).max()), (win, nov, N)
    sp.destroy(); n += 1
for _ in range(30):
    I, H = int(r.integers(1, 100)), int(r.integers(1, 140)) * 4
    T, B = int(r.integers(1, 8)), int(r.integers(1, 150))
    x = u(B, T, I)
    W, U, bi, bh = u(I, H, sc=I ** -0.5), u(H, H, sc=H ** -0.5), u(H, sc=0.1), u(H, sc=0.1)
    v2 = bool(r.integers(0, 2))
    m = NL.RNN(I, H, True, T, v2=v2); m.set_weights(W, U, bi, bh)
    close(m.apply(x), O.rnn(x, W, U, bi, bh, v2=v2), 1e-4); m.destroy(); n += 1
print("soak ok:", n, "cases")
