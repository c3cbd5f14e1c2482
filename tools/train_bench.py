#!/usr/bin/env python3
"""Wall time of the training entry points through the C boundary (host pointers in and out, so uploads / downloads are
inside): forward + gradient per mini-batch.  usage: python tools/train_bench.py [--reps N]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = lambda a: a.ctypes.data_as(__import__("nntoolkitcore_amd").capi.fp)


def main():
    from nntoolkitcore_amd import capi
    L = capi.load()
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
    r = np.random.default_rng(0)
    u = lambda *s, sc=1.0: r.uniform(-sc, sc, s).astype(np.float32)

    def timed(name, fwd, bwd, work):
        fwd(); bwd()
        capi.load().nntk_hip_synchronize()
        tf, tb = [], []
        for _ in range(reps):
            t0 = time.perf_counter(); fwd(); t1 = time.perf_counter(); bwd(); t2 = time.perf_counter()
            tf.append(t1 - t0); tb.append(t2 - t1)
        print("%-44s forward %8.2f ms   gradient %8.2f ms   (%s)" % (name, 1e3 * min(tf), 1e3 * min(tb), work), flush=True)

    # Conv1d config-3 shape, mini-batch 64
    B, T, Cin, Cout, k = 64, 1000, 40, 128, 5
    cfg = L.Conv1dConfigCreate(Cin, Cout, k, 1, T)
    tc = capi.ConvTrainingConfig(B)
    h = L.Conv1dCreateForTraining(cfg, tc)
    x, y, d = u(B, T, Cin), np.empty((B, T - k + 1, Cout), np.float32), u(B, T - k + 1, Cout)
    g = L.Conv1dCreateGradient(cfg, tc)
    timed("Conv1d(40->128,k5) B=64 T=1000", lambda: L.Conv1dApplyTrainingBatch(h, P(x), P(y)), lambda: L.Conv1dCalculateGradient(h, g, P(d)),
          "%.1f GFLOP fwd" % (2e-9 * B * (T - k + 1) * Cin * k * Cout))
    import torch
    torch.cuda.set_device(0)
    dpc = lambda t: C.c_void_p(t.data_ptr())
    xd, dd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(d).cuda(), torch.empty(B, T - k + 1, Cout, device="cuda")
    gd, gx = torch.zeros(Cout * Cin * k + Cout, device="cuda"), torch.empty(B, T, Cin, device="cuda")
    timed("Conv1d(40->128,k5) B=64 T=1000, device pointers", lambda: (L.Conv1dApplyTrainingBatchDevice(h, dpc(xd), dpc(yd)), L.nntk_hip_synchronize()),
          lambda: (L.Conv1dCalculateGradientDevice(h, dpc(gd), dpc(gx), dpc(dd)), L.nntk_hip_synchronize()), "same")
    L.ConvGradientDestroy(g); L.Conv1dDestroy(h)
    # BatchNorm over the conv output
    F, count = 128, 996
    bcfg = L.BatchNormConfigCreate(F, 1e-3, count)
    btc = L.BatchNormTrainingConfigCreate(0.9, B)
    h = L.BatchNormCreateForTraining(bcfg, btc)
    w = L.BatchNormGetWeights(h).contents
    gam = np.ones(F, np.float32); C.memmove(w.gamma, gam.ctypes.data, gam.nbytes)
    x, y, d = u(B * count, F), np.empty((B * count, F), np.float32), u(B * count, F)
    g = L.BatchNormGradientCreate(bcfg, btc)
    timed("BatchNorm(128) N=%d rows" % (B * count), lambda: L.BatchNormApplyTrainingBatch(h, P(x), P(y)), lambda: L.BatchNormCalculateGradient(h, g, P(d)),
          "%.0f MB tensor" % (4e-6 * B * count * F))
    L.BatchNormGradientDestroy(g); L.BatchNormDestroy(h)
    # recurrent layers
    for name, G, mk in (("GRU(128->256)", 3, "GRU"), ("LSTM(128->512)", 4, "LSTM")):
        H = 256 if G == 3 else 512
        B, T, n_in = 64, 200, 128
        if G == 3:
            acts = L.GRUActivationsCreateDefault(H)
            cfg = L.GRUConfigCreate(n_in, H, True, T, acts)
            h = L.GRUCreateForTraining(cfg, capi.ConvTrainingConfig(B)); w = L.GRUGetWeights(h).contents
            g = L.GRUGradientCreate(cfg, capi.ConvTrainingConfig(B)); fw, bw, de = L.GRUApplyTrainingBatch, L.GRUCalculateGradient, L.GRUDestroy
        else:
            acts = L.LSTMActivationsCreateDefault(H)
            cfg = L.LSTMConfigCreate(n_in, H, True, T, True, acts)
            h = L.LSTMCreateForTraining(cfg, capi.ConvTrainingConfig(B)); w = L.LSTMGetWeights(h).contents
            g = L.LSTMGradientCreate(cfg, capi.ConvTrainingConfig(B)); fw, bw, de = L.LSTMApplyTrainingBatch, L.LSTMCalculateGradient, L.LSTMDestroy
        W = u(n_in * G * H + H * G * H + 2 * G * H, sc=0.05); C.memmove(w.W, W.ctypes.data, W.nbytes)
        x, y, d = u(B, T, n_in), np.empty((B, T, H), np.float32), u(B, T, H)
        timed("%s B=64 T=200" % name, lambda: fw(h, P(x), P(y)), lambda: bw(h, g, P(d)),
              "%.1f GFLOP fwd" % (2e-9 * B * T * G * H * (n_in + H)))
        import torch
        torch.cuda.set_device(0)
        dp_ = lambda t: C.c_void_p(t.data_ptr())
        xd, dd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(d).cuda(), torch.empty(B, T, H, device="cuda")
        gd, gx = torch.zeros(W.size, device="cuda"), torch.empty(B, T, n_in, device="cuda")
        fwd_, bwd_ = (L.GRUApplyTrainingBatchDevice, L.GRUCalculateGradientDevice) if G == 3 else (L.LSTMApplyTrainingBatchDevice, L.LSTMCalculateGradientDevice)
        timed("%s B=64 T=200, device pointers" % name, lambda: (fwd_(h, dp_(xd), dp_(yd)), L.nntk_hip_synchronize()),
              lambda: (bwd_(h, dp_(gd), dp_(gx), dp_(dd)), L.nntk_hip_synchronize()), "same")
        L.RecurrentGradientDestroy(g); de(h)
    # dense head
    B, n_in, n_out = 64 * 996, 512, 1000
    cfg = L.DenseConfigCreate(n_in, n_out, None)
    h = L.DenseCreateForTraining(cfg, capi.ConvTrainingConfig(B))
    x, y, d = u(B, n_in), np.empty((B, n_out), np.float32), u(B, n_out)
    g = L.DenseGradientCreateFromFilter(h)
    timed("Dense(512->1000) rows=%d" % B, lambda: L.DenseApplyTrainingBatch(h, P(x), P(y)), lambda: L.DenseCalculateGradient(h, g, P(d)),
          "%.1f GFLOP fwd" % (2e-9 * B * n_in * n_out))
    # the same mini-batch through the device-pointer forms (tensors stay in HBM: no PCIe in the timed region)
    import torch
    torch.cuda.set_device(0)
    dp = lambda t: C.c_void_p(t.data_ptr())
    xd, dd = torch.from_numpy(x).cuda(), torch.from_numpy(d).cuda()
    yd = torch.empty(B, n_out, device="cuda"); gWd = torch.zeros(n_in * n_out + n_out, device="cuda"); gXd = torch.empty(B, n_in, device="cuda")
    sync = capi.load().nntk_hip_synchronize
    timed("Dense(512->1000) rows=%d, device pointers" % B,
          lambda: (L.DenseApplyTrainingBatchDevice(h, dp(xd), dp(yd)), sync()),
          lambda: (L.DenseCalculateGradientDevice(h, dp(gWd), dp(gXd), dp(dd)), sync()), "same")
    L.DenseGradientDestroy(g); L.DenseDestroy(h)
    F, count, Bm = 128, 996, 64
    bcfg = L.BatchNormConfigCreate(F, 1e-3, count)
    btc = L.BatchNormTrainingConfigCreate(0.9, Bm)
    hb = L.BatchNormCreateForTraining(bcfg, btc)
    wb = L.BatchNormGetWeights(hb).contents
    gam = np.ones(F, np.float32); C.memmove(wb.gamma, gam.ctypes.data, gam.nbytes)
    N = Bm * count
    xd, dd, yd = torch.randn(N, F, device="cuda"), torch.randn(N, F, device="cuda"), torch.empty(N, F, device="cuda")
    dbe, dga, dxx = torch.empty(F, device="cuda"), torch.empty(F, device="cuda"), torch.empty(N, F, device="cuda")
    timed("BatchNorm(128) N=%d rows, device pointers" % N,
          lambda: (L.BatchNormApplyTrainingBatchDevice(hb, dp(xd), dp(yd)), sync()),
          lambda: (L.BatchNormCalculateGradientDevice(hb, dp(dbe), dp(dga), dp(dxx), dp(dd)), sync()), "33 MB tensor")
    L.BatchNormDestroy(hb)


if __name__ == "__main__":
    main()
