#!/usr/bin/env python3
"""Does the (normal) dense GEMM co-run with the persistent GRU-256 kernel?  (config-4 shapes)"""
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
    g1 = NL.GRU(128, 256, True, T)
    g1.set_weights(w["g1_W"], w["g1_U"], w["g1_bi"], w["g1_bh"])
    tdd = NL.TimeDistributedDense(T, 256, 768)
    tdd.set_weights(w["g2_W"], w["g2_bi"])
    x = torch.randn(B, T, 128, device="cuda")
    h = torch.empty(B, T, 256, device="cuda")
    h2 = torch.randn(B, T, 256, device="cuda")
    y = torch.empty(B, T, 768, device="cuda")
    sA = torch.cuda.Stream(priority=-1)
    sB = torch.cuda.Stream(priority=0)

    def on(stream):
        lib.nntk_hip_set_stream(C.c_void_p(stream.cuda_stream))

    def run_gru():
        on(sA)
        g1.apply_device(x, out=h)

    def run_gemm(n=1):
        on(sB)
        for _ in range(n):
            tdd.apply_device(h2, out=y)

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

    run_gru(); run_gemm()
    torch.cuda.synchronize()
    for r in range(2):
        print("gru alone          ", timed(run_gru, [sA]))
        print("gemm alone         ", timed(lambda: run_gemm(1), [sB]))
        print("gemm x2 alone      ", timed(lambda: run_gemm(2), [sB]))
        for cyc in (8000000,):
            for n in (1, 2, 3):
                def delayed():
                    run_gru()
                    with torch.cuda.stream(sB):
                        torch.cuda._sleep(cyc)
                    run_gemm(n)
                print("gru + sleep(%d) + gemm x%d" % (cyc, n), timed(delayed, [sA, sB]))
        sys.stdout.flush()
    g1.destroy(); tdd.destroy()


if __name__ == "__main__":
    main()
