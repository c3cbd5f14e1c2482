#!/usr/bin/env python3
"""Quick GPU check of the full-K recurrent kernels (recurrent_fk.hip) against the oracle, then the config-4 timing with / without them.
usage: python tools/fk_check.py [--no-bench]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import oracle as O
    from nntoolkitcore_amd import capi, layers as NL
    torch.cuda.set_device(0); L = capi.load(); NL.use_torch_stream()
    capi.set_option("rec_fk", int(os.environ.get("NNTK_REC_FK", "1")))
    r = np.random.default_rng(11)
    u = lambda *sh, sc=1.0: r.uniform(-sc, sc, sh).astype(np.float32)
    bad = 0
    for cell, B, I, H, T, seq in [("gru", 64, 128, 256, 20, True), ("gru", 64, 256, 256, 20, True), ("lstm", 64, 128, 256, 7, True),
                                  ("gru", 33, 200, 192, 11, True), ("lstm", 70, 256, 160, 3, False), ("gru", 32, 128, 256, 1, True),
                                  ("gru", 130, 72, 256, 2, False), ("gru", 1, 128, 256, 40, True), ("lstm", 1100, 100, 144, 5, True)]:
        G = 4 if cell == "lstm" else 3
        x = u(B, T, I)
        W, U, bi, bh = u(I, G * H, sc=I ** -0.5), u(H, G * H, sc=H ** -0.5), u(G * H, sc=0.1), u(G * H, sc=0.1)
        lay = NL.LSTM(I, H, seq, T, v2=True) if cell == "lstm" else NL.GRU(I, H, seq, T)
        lay.set_weights(W, U, bi, bh)
        xd = torch.from_numpy(x).cuda()
        t0 = time.time()
        got = lay.apply_device(xd).clone(); torch.cuda.synchronize()
        name = L.nntk_hip_last_recurrent_kernel().decode()
        st = L.nntk_hip_device_status()
        ref = O.lstm(x, W, U, bi, bh, v2=True, return_sequences=seq) if cell == "lstm" else O.gru(x, W, U, bi, bh, return_sequences=seq)
        err = float(np.abs(got.cpu().numpy() - ref).max())
        got2 = lay.apply_device(xd).clone(); torch.cuda.synchronize()
        rep = bool(torch.equal(got, got2))
        ok = err < 1e-5 and st == 0 and rep
        bad += not ok
        print("%s B=%d I=%d H=%d T=%d seq=%d: %s err %.2e status %d repeat-equal %s %.2fs %s" % (cell, B, I, H, T, seq, name, err, st, rep, time.time() - t0, "ok" if ok else "FAIL"), flush=True)
        lay.destroy()
    print("fk_check: %d failure(s)" % bad, flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
