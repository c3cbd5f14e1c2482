#!/usr/bin/env python3
"""Two independent GRU-256 layers (config-4 shapes) on two HIP streams: does the second persistent kernel
co-reside with the first (LDS 63 KB + 63 KB per CU) and what does the pair cost?"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from nntoolkitcore_amd import capi, layers as NL
    torch.cuda.set_device(0)
    lib = capi.load()
    B, T = 1024, 1000
    w = bench.make_weights("gru", 3)
    g1 = NL.GRU(256, 256, True, T)
    g2 = NL.GRU(256, 256, True, T)
    g1.set_weights(w["g2_W"], w["g2_U"], w["g2_bi"], w["g2_bh"])
    g2.set_weights(w["g2_W"], w["g2_U"], w["g2_bi"], w["g2_bh"])
    x1 = torch.randn(B, T, 256, device="cuda")
    x2 = torch.randn(B, T, 256, device="cuda")
    h1 = torch.empty(B, T, 256, device="cuda")
    h2 = torch.empty(B, T, 256, device="cuda")
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def on(stream):
        lib.nntk_hip_set_stream(C.c_void_p(stream.cuda_stream))

    def timed(fn, streams):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.default_stream())
        for s in streams:
            s.wait_event(e0)
        fn()
        ends = []
        for s in streams:
            e = torch.cuda.Event(enable_timing=True)
            e.record(s)
            ends.append(e)
        torch.cuda.synchronize()
        return ["%.2f" % e0.elapsed_time(e) for e in ends]

    def one():
        on(sA); g1.apply_device(x1, out=h1)

    def seq():
        on(sA); g1.apply_device(x1, out=h1); g2.apply_device(x2, out=h2)

    def par():
        on(sA); g1.apply_device(x1, out=h1)
        on(sB); g2.apply_device(x2, out=h2)

    one(); seq(); par()
    torch.cuda.synchronize()
    for r in range(3):
        print("one layer            ", timed(one, [sA]))
        print("two, one stream      ", timed(seq, [sA]))
        print("two, two streams     ", timed(par, [sA, sB]))
        sys.stdout.flush()
    g1.destroy(); g2.destroy()


if __name__ == "__main__":
    main()
