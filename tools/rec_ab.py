"""A/B of the recurrent kernels on one layer: register-resident split-bf16 (rec_rr = 1) against the exact-f32 persistent kernel +
projection GEMM (rec_rr = 0), device-pointer calls, B x T x in -> H.   usage: python tools/rec_ab.py [B] [T]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from nntoolkitcore_amd import capi, layers as NL

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
L = capi.load()
r = np.random.default_rng(0)
u = lambda *s, sc=1.0: r.uniform(-sc, sc, s).astype(np.float32)
for kind, I, H in (("gru", 128, 256), ("gru", 256, 256), ("gru", 128, 512), ("gru", 128, 384), ("lstm", 128, 256), ("lstm", 128, 128), ("lstm", 128, 384), ("lstm", 128, 512)):
    G = 3 if kind == "gru" else 4
    layer = NL.GRU(I, H, True, T) if kind == "gru" else NL.LSTM(I, H, True, T, v2=True)
    layer.set_weights(u(I, G * H, sc=I ** -0.5), u(H, G * H, sc=H ** -0.5), u(G * H, sc=0.1), u(G * H, sc=0.1))
    x = torch.from_numpy(u(B, T, I)).cuda()
    out = torch.empty(B, T, H, device="cuda")
    res = []
    for mode in (1, 0):
        capi.set_option("rec_rr", mode)
        for _ in range(2): layer.apply_device(x, out)
        L.nntk_hip_synchronize()
        t0 = time.perf_counter()
        for _ in range(3): layer.apply_device(x, out)
        L.nntk_hip_synchronize()
        res.append(((time.perf_counter() - t0) / 3 * 1e3, L.nntk_hip_last_recurrent_kernel().decode()))
    capi.set_option("rec_rr", "auto")
    print("%-4s in=%3d H=%3d B=%d T=%d:  rr %7.2f ms (%s)   exact %7.2f ms (%s)   -> %.2f us/step vs %.2f" % (
        kind, I, H, B, T, res[0][0], res[0][1], res[1][0], res[1][1], res[0][0] * 1e3 / T, res[1][0] * 1e3 / T))
    layer.destroy(); del x, out
