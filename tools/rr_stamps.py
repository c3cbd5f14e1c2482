#!/usr/bin/env python3
"""Diagnostics build only (python tools/build_variant.py build/libs/libstamps.so recurrent_rr.hip -DNNTK_REC_STAMPS; NNTK_LIB=<lib>): run the stack's LSTM on
lstm_rr_kernel once and print where workgroup 0 / wave 0 spends a half-step (s_memtime cycles).
usage: NNTK_LIB=<lib built with -DNNTK_REC_STAMPS> python tools/rr_stamps.py [B] [T] [lstm|gru] [in] [H]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch, bench
    from nntoolkitcore_amd import capi, layers as NL
    torch.cuda.set_device(0); capi.load(); NL.use_torch_stream()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 996
    kind = sys.argv[3] if len(sys.argv) > 3 else "lstm"
    I = int(sys.argv[4]) if len(sys.argv) > 4 else 128
    H = int(sys.argv[5]) if len(sys.argv) > 5 else 512
    G = 4 if kind == "lstm" else 3
    r = np.random.default_rng(3)
    u = lambda *sh, sc=1.0: r.uniform(-sc, sc, sh).astype(np.float32)
    lstm = NL.LSTM(I, H, True, T, v2=True) if kind == "lstm" else NL.GRU(I, H, True, T)
    lstm.set_weights(u(I, G * H, sc=I ** -0.5), u(H, G * H, sc=H ** -0.5), u(G * H, sc=0.1), u(G * H, sc=0.1))
    capi.set_option("rec_rr", 1)
    capi.set_option("rec_xf", 1)
    x = torch.randn(B, T, I, device="cuda"); h = torch.empty(B, T, H, device="cuda")
    for _ in range(3):
        lstm.apply_device(x, out=h)
    torch.cuda.synchronize()
    path = os.path.join(ROOT, "gpurun_out", "rr_stamps.bin")
    os.environ["NNTK_REC_STAMP_FILE"] = path
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lstm.apply_device(x, out=h); e1.record(); torch.cuda.synchronize()
    del os.environ["NNTK_REC_STAMP_FILE"]
    s = np.fromfile(path, dtype=np.uint64).astype(np.int64).reshape(T + 1, 4, 16)
    lo, hi = T // 10, T - T // 10
    nst = (4 if H <= 256 else 8) + (1 if I <= 64 else 2 if I <= 128 else 4)
    kname = capi.load().nntk_hip_last_recurrent_kernel().decode()
    ns = 2
    print("%s in=%d H=%d: %s, %d k steps per half-step, %d streams" % (kind, I, H, kname, nst, ns))
    print("launch %.3f ms incl. stamping = %.2f us/step" % (e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / T))
    for half in range(ns):
        d = s[lo:hi, half]
        nxt = s[lo:hi, half + 1] if half < ns - 1 else s[lo + 1:hi + 1, 0]
        dur = [(d[:, i + 1] - d[:, i]) for i in range(nst)]
        print("half %d: half-step %.0f cyc; k steps (mean): %s | end->next start %.0f" % (
            half, (nxt[:, 0] - d[:, 0]).mean(), " ".join("%.0f" % x.mean() for x in dur), (nxt[:, 0] - d[:, nst]).mean()))
        print("          k steps (p90):  %s" % " ".join("%.0f" % np.percentile(x, 90) for x in dur))
    per = (s[hi, 0, 0] - s[lo, 0, 0]) / (hi - lo)
    print("step period %.0f cycles" % per)
    lstm.destroy()


if __name__ == "__main__":
    main()
