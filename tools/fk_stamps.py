#!/usr/bin/env python3
"""Diagnostics build only (python tools/build_variant.py gpurun_out/libs/libstamps.so recurrent_rr.hip,recurrent_fk.hip -DNNTK_REC_STAMPS;
NNTK_LIB=<lib>): run one GRU / LSTM layer on the full-K kernel and print where workgroup 0 / wave 0 spends a step (s_memtime cycles).
usage: NNTK_LIB=<lib> python tools/fk_stamps.py [B] [T] [gru|lstm] [in] [H]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nntoolkitcore_amd import capi, layers as NL
    torch.cuda.set_device(0); capi.load(); NL.use_torch_stream()
    capi.set_option("rec_fk", int(os.environ.get("NNTK_REC_FK", "1")))
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    kind = sys.argv[3] if len(sys.argv) > 3 else "gru"
    I = int(sys.argv[4]) if len(sys.argv) > 4 else 128
    H = int(sys.argv[5]) if len(sys.argv) > 5 else 256
    G = 4 if kind == "lstm" else 3
    r = np.random.default_rng(3)
    u = lambda *sh, sc=1.0: r.uniform(-sc, sc, sh).astype(np.float32)
    lay = NL.LSTM(I, H, True, T, v2=True) if kind == "lstm" else NL.GRU(I, H, True, T)
    lay.set_weights(u(I, G * H, sc=I ** -0.5), u(H, G * H, sc=H ** -0.5), u(G * H, sc=0.1), u(G * H, sc=0.1))
    x = torch.randn(B, T, I, device="cuda"); h = torch.empty(B, T, H, device="cuda")
    for _ in range(3):
        lay.apply_device(x, out=h)
    torch.cuda.synchronize()
    path = os.path.join(ROOT, "gpurun_out", "fk_stamps.bin")
    os.environ["NNTK_REC_STAMP_FILE"] = path
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lay.apply_device(x, out=h); e1.record(); torch.cuda.synchronize()
    del os.environ["NNTK_REC_STAMP_FILE"]
    kname = capi.load().nntk_hip_last_recurrent_kernel().decode()
    s = np.fromfile(path, dtype=np.uint64).astype(np.int64).reshape(T + 1, 64)
    import re
    nkh, nkx, nw = [int(v) for v in re.search(r"<(\d+),(\d+),(\d+)>", kname).groups()]
    NP, NPH = (nkh + nkx) // nw, nkh // nw
    lo, hi = T // 10, T - T // 10
    print("%s in=%d H=%d B=%d: %s; launch %.3f ms incl. pack + stamping = %.2f us/step" % (kind, I, H, B, kname, e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / T))
    d = s[lo:hi]
    nxt = s[lo + 1:hi + 1]
    seq = [d[:, i] for i in range(NP)] + [nxt[:, 0]]
    names = ["h%d" % i for i in range(NPH)] + ["x%d" % i for i in range(NP - NPH)]
    print("step period %.0f cycles" % (nxt[:, 0] - d[:, 0]).mean())
    print("  ".join("%s %.0f" % (names[i], (seq[i + 1] - seq[i]).mean()) for i in range(NP)))
    print("p90: " + "  ".join("%s %.0f" % (names[i], np.percentile(seq[i + 1] - seq[i], 90)) for i in range(NP)))
    print("finish stores issued %.0f cycles after the end of the h part" % (d[:, 33] - d[:, NPH]).mean())
    lab = ["own0-2+ring_get+look", "branch", "own3-5+ring_put"] + ["partner%d" % j for j in range(1, nw)]
    f = d[:, 48:48 + len(lab) + 1]
    print("inside group 2 (s_memtime, no waits): " + "  ".join("%s %.0f" % (lab[i], (f[:, i + 1] - f[:, i]).mean()) for i in range(len(lab))))
    if os.environ.get("FK_COARSE_GS"):
        g = int(os.environ["FK_COARSE_GS"])
        pts = [g, 40, 41] + [41 + j for j in range(1, nw)]
        lab = ["block a + branch", "block b"] + ["partner%d" % j for j in range(1, nw)]
        print("inside group %d (stamps with a store each, ~70 cycles apiece): " % g + "  ".join("%s %.0f" % (lab[i], (d[:, pts[i + 1]] - d[:, pts[i]]).mean()) for i in range(len(lab))))
    lay.destroy()


if __name__ == "__main__":
    main()
