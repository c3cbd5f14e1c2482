#!/usr/bin/env python3
"""Latency of the reference-shaped single-sequence calls (host pointers, carried state): SURVEY 8(f) rank 2."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    import torch
    import bench
    from nntoolkitcore_amd import capi, layers as NL
    torch.cuda.set_device(0); capi.load()
    r = np.random.default_rng(0)
    for name, mk in (("GRU-256", lambda T: NL.GRU(128, 256, True, T)), ("LSTM-512", lambda T: NL.LSTM(128, 512, True, T))):
        for T in (1, 2, 5, 10, 20, 50, 100):
            l = mk(T)
            G = 3 if "GRU" in name else 4
            H = 256 if "GRU" in name else 512
            l.set_weights(r.standard_normal((128, G * H)).astype(np.float32) * 0.05, r.standard_normal((H, G * H)).astype(np.float32) * 0.05,
                          np.zeros(G * H, np.float32), np.zeros(G * H, np.float32))
            x = r.standard_normal((T, 128)).astype(np.float32)
            for _ in range(5): l.apply(x)
            t0 = time.perf_counter()
            n = 200
            for _ in range(n): l.apply(x)
            dt = (time.perf_counter() - t0) / n
            print("%s  T=%3d  %.1f us per call  (%.1f us per frame)" % (name, T, dt * 1e6, dt * 1e6 / T))
            l.destroy()

if __name__ == "__main__":
    main()
