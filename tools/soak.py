#!/usr/bin/env python3
"""Random-shape soak of the conv / dense / recurrent kernels against the oracle (~915 cases incl. round 5's FRAG2H routes and HF kernels, round 5's conv -> frag3 epilogue, dense-on-frag3 and full-K shapes, spectrogram geometries, the RNN, and round 2's streaming / fused-GRU / mixed-radix / log-mel paths and the training path, ~1.5 min on the GPU):
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
# ---- round 2 paths: streaming single-sequence calls, fused GRU stack, mixed-radix FFT, fused log-mel ----
for _ in range(60):
    I, H = int(r.integers(1, 150)), int(r.integers(1, 130)) * 4
    T = int(r.integers(1, 33))
    kind = ("gru", "lstm", "rnn")[int(r.integers(0, 3))]
    G = {"gru": 3, "lstm": 4, "rnn": 1}[kind]
    W, U, bi, bh = u(I, G * H, sc=I ** -0.5), u(H, G * H, sc=H ** -0.5), u(G * H, sc=0.1), u(G * H, sc=0.1)
    x = u(2 * T, I)
    seq = bool(r.integers(0, 2))
    l = {"gru": lambda: NL.GRU(I, H, seq, T), "lstm": lambda: NL.LSTM(I, H, seq, T, v2=bool(r.integers(0, 2))), "rnn": lambda: NL.RNN(I, H, seq, T)}[kind]()
    l.set_weights(W, U, bi, bh)
    a, b2 = l.apply(x[:T]), l.apply(x[T:])                      # two streaming calls, carried state
    if kind == "lstm":
        full = O.lstm(x, W, U, bi, bh, v2=bool(l.cfg.v2))[0]
    elif kind == "gru":
        full = O.gru(x, W, U, bi, bh)[0]
    else:
        full = O.rnn(x, W, U, bi, bh)[0]
    ra, rb = (full[:T], full[T:]) if seq else (full[T - 1], full[2 * T - 1])
    close(a, ra, 1e-4); close(b2, rb, 1e-4); l.destroy(); n += 1
for _ in range(30):
    I, H = int(r.integers(1, 150)), int(r.integers(1, 65)) * 4
    T, B = int(r.integers(1, 12)), int(r.integers(1, 200))
    x = u(B, T, I)
    W1, U1, bi1, bh1 = u(I, 3 * H, sc=I ** -0.5), u(H, 3 * H, sc=H ** -0.5), u(3 * H, sc=0.1), u(3 * H, sc=0.1)
    W2, U2, bi2, bh2 = u(H, 3 * H, sc=H ** -0.5), u(H, 3 * H, sc=H ** -0.5), u(3 * H, sc=0.1), u(3 * H, sc=0.1)
    seq = bool(r.integers(0, 2))
    g1, g2 = NL.GRU(I, H, True, T), NL.GRU(H, H, seq, T)
    g1.set_weights(W1, U1, bi1, bh1); g2.set_weights(W2, U2, bi2, bh2)
    close(NL.gru_stack2_apply(g1, g2, x), O.gru(O.gru(x, W1, U1, bi1, bh1), W2, U2, bi2, bh2, return_sequences=seq), 1e-4)
    g1.destroy(); g2.destroy(); n += 1
for _ in range(40):
    nfft = int((2 ** r.integers(1, 9)) * (3 ** r.integers(0, 3)) * (5 ** r.integers(0, 2)))
    if nfft > 4096 or nfft < 4: continue
    win = int(r.integers(max(2, nfft // 4), nfft + 1)); nov = int(r.integers(0, win))
    N = int(r.integers(win, win + 40 * (win - nov) + 1)); B = int(r.integers(1, 4))
    x = u(B, N, sc=0.1)
    sp = NL.Spectrogram(nfft, win, nov, N, window_name="hamming_window")
    ref = O.spectrogram(x, O.window("hamming", win), nfft, nov)
    got = sp.apply(x)
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 3e-5 * (1e-6 + np.abs(ref).max()), (nfft, win, nov, N)
    sp.destroy(); n += 1
import ctypes as C
L = capi.load()
for _ in range(15):
    n_mels = int(r.integers(8, 129)); N = 240 + 160 * int(r.integers(1, 40)); B = int(r.integers(1, 4))
    x = u(B, N, sc=0.1)
    sp = NL.Spectrogram(512, 400, 240, N)
    T = sp.out_shape[0]
    cfg = L.MelFilterBankConfigCreate(n_mels, 512, 16000, C.c_float(20.0), C.c_float(7600.0))
    w = O.mel_filterbank_weights(n_mels, 512, 16000, 20.0, 7600.0)
    spec = O.spectrogram(x, O.window("hann", 400), 512, 240)
    ref = np.stack([O.log_mel(spec[i], w) for i in range(B)])
    lm = L.LogMelSpectrogramCreate(sp.h, cfg)
    out = np.empty((B, T, n_mels), np.float32)
    assert L.LogMelSpectrogramApplyBatch(lm, x.ctypes.data_as(capi.fp), out.ctypes.data_as(capi.fp), B) == 0
    assert np.abs(out - ref).max() < 5e-5, (n_mels, N)
    L.LogMelSpectrogramDestroy(lm); sp.destroy(); n += 1
# ---- training path: random small shapes (incl. T = 1, mini-batch 1, return_sequences = False) against the oracle ----
P = lambda a: a.ctypes.data_as(capi.fp)
def fill(dst, src): C.memmove(dst, src.ctypes.data, src.nbytes)
def rel(a, b): return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
for _ in range(40):
    G = int(r.choice([1, 3, 4]))
    B, T, I, H = int(r.integers(1, 7)), int(r.integers(1, 12)), int(r.integers(1, 40)), int(r.integers(1, 48))
    seq, v2 = bool(r.integers(0, 2)), bool(r.integers(0, 2))
    x, dout = u(B, T, I), (u(B, T, H) if seq else u(B, H))
    W, U, bi, bh = u(I, G * H, sc=I ** -0.5), u(H, G * H, sc=H ** -0.5), u(G * H, sc=0.1), u(G * H, sc=0.1)
    tc = capi.ConvTrainingConfig(B)
    if G == 3:
        acts = L.GRUActivationsCreateDefault(H); cfg = L.GRUConfigCreate(I, H, seq, T, acts)
        h = L.GRUCreateForTraining(cfg, tc); w = L.GRUGetWeights(h).contents; g = L.GRUGradientCreate(cfg, tc)
        fw, bw, de = L.GRUApplyTrainingBatch, L.GRUCalculateGradient, L.GRUDestroy
        oh, ref = O.gru_training(x, W, U, bi, bh, dout, return_sequences=seq)
    elif G == 4:
        acts = L.LSTMActivationsCreateDefault(H); cfg = L.LSTMConfigCreate(I, H, seq, T, v2, acts)
        h = L.LSTMCreateForTraining(cfg, tc); w = L.LSTMGetWeights(h).contents; g = L.LSTMGradientCreate(cfg, tc)
        fw, bw, de = L.LSTMApplyTrainingBatch, L.LSTMCalculateGradient, L.LSTMDestroy
        oh, ref = O.lstm_training(x, W, U, bi, bh, dout, return_sequences=seq, v2=v2)
    else:
        act = L.ActivationFunctionCreateTanh(H); cfg = L.RNNConfigCreate(I, H, seq, T, v2, act)
        h = L.RNNCreateForTraining(cfg, tc); w = L.RNNGetWeights(h).contents; g = L.RNNGradientCreate(cfg, tc)
        fw, bw, de = L.RNNApplyTrainingBatch, L.RNNCalculateGradient, L.RNNDestroy
        oh, ref = O.rnn_training(x, W, U, bi, bh, dout, return_sequences=seq, v2=v2)
    for dst, src in ((w.W, W), (w.U, U), (w.b_i, bi), (w.b_h, bh)): fill(dst, src)
    y = np.empty((B, T, H) if seq else (B, H), np.float32)
    assert fw(h, P(x), P(y)) == 0, capi.last_error()
    assert rel(y, oh if seq else oh[:, -1]) < 2e-5
    bw(h, g, P(dout)); assert capi.last_error() == ""
    gc = g.contents
    for ptr, rf in zip((gc.d_W, gc.d_U, gc.d_b_i, gc.d_b_h, gc.d_X), ref):
        assert rel(np.ctypeslib.as_array(ptr, shape=rf.shape), rf) < 3e-5, (G, B, T, I, H, seq, v2)
    L.RecurrentGradientDestroy(g); de(h); n += 1
for _ in range(30):
    B, n_in, n_out = int(r.integers(1, 40)), int(r.integers(1, 70)), int(r.integers(1, 60))
    kind = int(r.choice([-1, O.ACT_SIGMOID, O.ACT_TANH, O.ACT_RELU, O.ACT_SOFTMAX]))
    ah = {-1: None, O.ACT_SIGMOID: L.ActivationFunctionCreateSigmoid(n_out), O.ACT_TANH: L.ActivationFunctionCreateTanh(n_out),
          O.ACT_RELU: L.ActivationFunctionCreateReLU(n_out, 1.0), O.ACT_SOFTMAX: L.ActivationFunctionCreateSoftmax(1, n_out)}[kind]
    x, W, b, dout = u(B, n_in), u(n_in, n_out, sc=n_in ** -0.5), u(n_out, sc=0.1), u(B, n_out)
    cfg = L.DenseConfigCreate(n_in, n_out, ah); h = L.DenseCreateForTraining(cfg, capi.ConvTrainingConfig(B))
    w = L.DenseGetWeights(h).contents; fill(w.W, W); fill(w.b, b)
    y = np.empty((B, n_out), np.float32)
    assert L.DenseApplyTrainingBatch(h, P(x), P(y)) == 0, capi.last_error()
    act = None if kind < 0 else kind
    z, a = O.dense_forward_training(x, W, b, act=act, softmax_vector_size=n_out)
    assert rel(y, a) < 2e-5
    g = L.DenseGradientCreateFromFilter(h); L.DenseCalculateGradient(h, g, P(dout)); assert capi.last_error() == ""
    for ptr, rf in zip((g.contents.d_W, g.contents.d_b, g.contents.d_X), O.dense_gradient(x, W, z, a, dout, act=act, softmax_vector_size=n_out)):
        assert rel(np.ctypeslib.as_array(ptr, shape=rf.shape), rf) < 2e-5, (B, n_in, n_out, kind)
    L.DenseGradientDestroy(g); L.DenseDestroy(h); n += 1
for _ in range(20):
    count, mb, F = int(r.integers(1, 30)), int(r.integers(1, 9)), int(r.integers(1, 200))
    N = count * mb
    if N < 2: continue
    x, dout = u(N, F, sc=2.0), u(N, F)
    gam, be, mm0, mv0 = 1 + u(F, sc=0.5), u(F, sc=0.5), u(F, sc=0.2), 1 + u(F, sc=0.3)
    cfg = L.BatchNormConfigCreate(F, 1e-3, count); tc = L.BatchNormTrainingConfigCreate(0.9, mb)
    h = L.BatchNormCreateForTraining(cfg, tc); w = L.BatchNormGetWeights(h).contents
    for dst, src in ((w.gamma, gam), (w.beta, be), (w.moving_mean, mm0), (w.moving_variance, mv0)): fill(dst, src)
    y = np.empty((N, F), np.float32)
    assert L.BatchNormApplyTrainingBatch(h, P(x), P(y)) == 0, capi.last_error()
    oy, om, ov, omm, omv = O.batch_norm_training_forward(x, gam, be, 1e-3, 0.9, mm0, mv0)
    assert rel(y, oy) < 3e-5 and rel(np.ctypeslib.as_array(w.moving_variance, shape=(F,)), omv) < 1e-5
    g = L.BatchNormGradientCreate(cfg, tc); L.BatchNormCalculateGradient(h, g, P(dout)); assert capi.last_error() == ""
    for ptr, rf in zip((g.contents.d_beta, g.contents.d_gamma, g.contents.d_x), O.batch_norm_gradient(x, dout, gam, om, ov, 1e-3)):
        assert rel(np.ctypeslib.as_array(ptr, shape=rf.shape), rf) < 1e-4, (count, mb, F)
    L.BatchNormGradientDestroy(g); L.BatchNormDestroy(h); n += 1
# ---- round 5: the conv epilogue that writes frag3 (random shapes, bit-identity with conv -> pack; ragged utterance groups and timestep blocks,
#      padding channels, every activation), the dense GEMM's LDS epilogue on frag3 inputs, and the full-K recurrent family on wide inputs ----
for _ in range(60):
    cin, cout = int(r.integers(4, 300)), int(r.integers(32, 300))
    k, stride = int(r.integers(1, 12)), (1 if r.random() < 0.8 else 2)
    T = int(r.integers(k + 1, 120)); B = int(r.integers(1, 80))
    conv = NL.Conv1d(cin, cout, k, stride, T); conv.set_weights(u(cout, cin, k, sc=(cin * k) ** -0.5), u(cout, sc=0.2))
    Tc = conv.out_shape[0]
    bn = act = None
    if r.random() < 0.6:
        bn = NL.BatchNorm(cout, 1e-3, B * Tc); bn.set_weights(1 + u(cout, sc=0.4), u(cout, sc=0.5), u(cout, sc=0.2), 1 + u(cout, sc=0.4))
    kind = ["relu", "sigmoid", "tanh", "identity", None][int(r.integers(0, 5))]
    if kind: act = NL.Activation(kind, B * Tc * cout, 0.5)
    xd = torch.from_numpy(u(B, T, cin)).cuda()
    y = conv.apply_device(xd, bn=bn, act=act)
    ref3 = NL.frag3_pack_device(y)
    out3 = torch.zeros_like(ref3)
    conv.apply_device_frag3(xd, out_f3=out3, bn=bn, act=act)
    assert torch.equal(out3.view(torch.int32), ref3.view(torch.int32)), (B, T, cin, cout, k, stride, kind)
    for h in (conv, bn, act):
        if h is not None: h.destroy()
    n += 1
for _ in range(40):
    ts, I, Ov = int(r.integers(1, 60)), int(r.integers(1, 40)) * 16, int(r.integers(1, 280)) * 4
    Bb = int(r.integers(1, 150))
    W, b, x = u(I, Ov, sc=I ** -0.5), u(Ov, sc=0.2), u(Bb, ts, I)
    tdd = NL.TimeDistributedDense(ts, I, Ov); tdd.set_weights(W, b)
    xd = torch.from_numpy(x).cuda()
    a = tdd.apply_device(xd)
    b3 = NL.tdd_apply_device_frag3(tdd, NL.frag3_pack_device(xd), Bb)
    assert torch.equal(a, b3), (Bb, ts, I, Ov)
    close(a.cpu().numpy(), O.time_distributed_dense(x, W, b)); tdd.destroy(); n += 1
for _ in range(30):
    I, H = int(r.integers(129, 257)), int(r.integers(9, 17)) * 16
    T, B = int(r.integers(1, 12)), int(r.integers(1, 150))
    x = u(B, T, I)
    W, U, bi, bh = u(I, 3 * H, sc=I ** -0.5), u(H, 3 * H, sc=H ** -0.5), u(3 * H, sc=0.1), u(3 * H, sc=0.1)
    g = NL.GRU(I, H, True, T); g.set_weights(W, U, bi, bh)
    close(g.apply(x), O.gru(x, W, U, bi, bh), 1e-4)
    assert capi.load().nntk_hip_last_recurrent_kernel().decode().startswith("gru_fk_kernel"), (I, H)
    g.destroy()
    W, U, bi, bh = u(I, 4 * H, sc=I ** -0.5), u(H, 4 * H, sc=H ** -0.5), u(4 * H, sc=0.1), u(4 * H, sc=0.1)
    l = NL.LSTM(I, H, False, T); l.set_weights(W, U, bi, bh)
    close(l.apply(x), O.lstm(x, W, U, bi, bh, return_sequences=False), 1e-4); l.destroy(); n += 2
# ---- round 5, late: FRAG2H -- the fused LSTM -> TimeDistributedDense call on random shapes (the HF instantiations for 256 < H <= 512 and inputs up to
#      256 channels, the six-product kernels' output wave for H <= 256, the pack pass behind every other kernel; dense shapes the f16 kernel takes and
#      does not take) against the oracle, and the frag2h LSTM output against the oracle ----
for _ in range(40):
    H = int(r.integers(4, 33)) * 16 if r.integers(0, 4) else int(r.integers(1, 100)) * 4
    I = int(r.integers(1, 257))
    T, B, N = int(r.integers(1, 14)), int(r.integers(1, 200)), int(r.integers(1, 40)) * (32 if r.integers(0, 3) else 1)
    x = u(B, T, I)
    W, U, bi, bh = u(I, 4 * H, sc=I ** -0.5), u(H, 4 * H, sc=H ** -0.5), u(4 * H, sc=0.1), u(4 * H, sc=0.1)
    Wd, bd = u(H, N, sc=H ** -0.5), u(N, sc=0.1)
    l = NL.LSTM(I, H, True, T); l.set_weights(W, U, bi, bh)
    tdd = NL.TimeDistributedDense(T, H, N); tdd.set_weights(Wd, bd)
    xd = torch.from_numpy(x).cuda()
    h = O.lstm(x, W, U, bi, bh)
    h = h[0] if isinstance(h, tuple) else h
    try:
        close(NL.lstm_tdd_apply_device(l, tdd, xd).cpu().numpy(), O.time_distributed_dense(h, Wd, bd), 1e-4)
        close(NL.frag2h_unpack_device(NL.lstm_apply_device_frag2h(l, x=xd), B, T, H).cpu().numpy(), h, 1e-4)
    except AssertionError:
        print("FRAG2H soak case failed: B=%d T=%d in=%d H=%d N=%d, kernel %s" % (B, T, I, H, N, capi.load().nntk_hip_last_recurrent_kernel().decode()))
        raise
    if H % 16 == 0 and 256 < H <= 512:
        k = capi.load().nntk_hip_last_recurrent_kernel().decode()
        assert k.startswith("lstm_rr_kernel<8,") and k.endswith(",hf>"), (I, H, k)
    l.destroy(); tdd.destroy(); n += 2
print("soak ok:", n, "cases")
