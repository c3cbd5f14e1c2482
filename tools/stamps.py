#!/usr/bin/env python3
"""Diagnostics build only (NNTK_EXTRA_HIPFLAGS=-DNNTK_REC_STAMPS): run the stack LSTM once and print the
per-phase s_memtime anatomy of workgroup 0's two leader waves.  usage: stamps.py <pingpong 0|1|2> [wavemap]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    import torch, bench
    from nntoolkitcore_amd import capi, layers as NL
    capi.set_option("rec_pingpong", sys.argv[1])
    if len(sys.argv) > 2: os.environ["NNTK_REC_WAVEMAP"] = sys.argv[2]
    torch.cuda.set_device(0); capi.load(); NL.use_torch_stream()
    B, T = 512, 996
    w = bench.make_weights("stack", 3)
    lstm = NL.LSTM(128, 512, True, T, v2=True)
    lstm.set_weights(w["lstm_W"], w["lstm_U"], w["lstm_bi"], w["lstm_bh"])
    x = torch.randn(B, T, 128, device="cuda"); h = torch.empty(B, T, 512, device="cuda")
    lstm.apply_device(x, out=h); torch.cuda.synchronize()
    path = os.path.join(ROOT, "gpurun_out", "stamps_pp%s.bin" % sys.argv[1])
    os.environ["NNTK_REC_STAMP_FILE"] = path
    lstm.apply_device(x, out=h); torch.cuda.synchronize()
    del os.environ["NNTK_REC_STAMP_FILE"]
    raw = np.fromfile(path, dtype=np.uint64).astype(np.int64)
    s = raw[:2 * T * 8].reshape(2, T, 8)
    if raw.size >= 2 * T * 8 + 8 * T * 2:
        kw = raw[2 * T * 8:2 * T * 8 + 8 * T * 2].reshape(8, T, 2)[:, 100:900]
        base = kw[0, :, 0]
        print("per-wave K loop (start offset vs wave 0's start, duration), mean over steps:")
        for w in range(8):
            print("  wave %d: start %+7.0f  dur %6.0f  (min %6.0f max %6.0f)" % (
                w, (kw[w, :, 0] - base).mean(), (kw[w, :, 1] - kw[w, :, 0]).mean(),
                (kw[w, :, 1] - kw[w, :, 0]).min(), (kw[w, :, 1] - kw[w, :, 0]).max()))
    names = ["top", "poll", "kstart", "kend", "xchg", "gates+st", "drain", "arrive"]
    for half in range(2):
        d = s[half, 100:900]
        per = np.diff(d[:, 0]).mean()
        print("half/grp %d: period %.0f cyc" % (half, per), " phases:",
              " ".join("%s %.0f" % (names[i + 1], (d[:, i + 1] - d[:, i]).mean()) for i in range(7)),
              " arrive->next top %.0f" % (d[1:, 0] - d[:-1, 7]).mean())
    # relative phase of the two halves: kstart of half 1 relative to kstart of half 0 (same step)
    rel = (s[1, 100:900, 2] - s[0, 100:900, 2])
    print("half1.kstart - half0.kstart: mean %.0f  min %.0f max %.0f" % (rel.mean(), rel.min(), rel.max()))
    rel2 = (s[1, 100:900, 2] - s[0, 100:900, 3])
    print("half1.kstart - half0.kend:   mean %.0f  min %.0f max %.0f" % (rel2.mean(), rel2.min(), rel2.max()))
    for t in range(500, 504):
        print(t, [int(v - s[0, 500, 0]) for v in s[0, t]], [int(v - s[0, 500, 0]) for v in s[1, t]])
    lstm.destroy()

if __name__ == "__main__":
    main()
