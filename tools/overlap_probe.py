#!/usr/bin/env python3
"""Does a dense GEMM co-run with the persistent LSTM kernel?  (stack shapes, one GPU)

Times, with torch events on two HIP streams of different priority:
  lstm alone | tdd alone (normal kernel) | tdd alone (low-LDS kernel) | lstm (hi prio) + tdd low-LDS (lo prio)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from nntoolkitcore_amd import capi, layers as NL
    torch.cuda.set_device(0)
    lib = capi.load()
    B, T = 512, 996
    w = bench.make_weights("stack", 3)
    lstm = NL.LSTM(128, 512, True, T, v2=True)
    lstm.set_weights(w["lstm_W"], w["lstm_U"], w["lstm_bi"], w["lstm_bh"])
    tdd = NL.TimeDistributedDense(T, 512, 1000)
    tdd.set_weights(w["tdd_W"], w["tdd_b"])
    x = torch.randn(B, T, 128, device="cuda")
    h = torch.empty(B, T, 512, device="cuda")
    h2 = torch.randn(B, T, 512, device="cuda")
    y = torch.empty(B, T, 1000, device="cuda")
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    sA = torch.cuda.Stream(priority=-1)
    sB = torch.cuda.Stream(priority=0)

    def on(stream):
        lib.nntk_hip_set_stream(C.c_void_p(stream.cuda_stream))

    def run_lstm():
        on(sA)
        os.environ["NNTK_CONV_LOWLDS"] = "0"
        lstm.apply_device(x, out=h)

    def run_tdd(low, n=1):
        on(sB)
        os.environ["NNTK_CONV_LOWLDS"] = "1" if low else "0"
        for _ in range(n):
            tdd.apply_device(h2, out=y)
        os.environ["NNTK_CONV_LOWLDS"] = "0"

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
        return [e0.elapsed_time(e) for e in ends]

    run_lstm(); run_tdd(False)
    torch.cuda.synchronize()
    y_ref = y.clone()
    y.zero_()
    run_tdd(True)
    torch.cuda.synchronize()
    print("low-LDS kernel bitwise equal to the normal one:", bool(torch.equal(y, y_ref)))
    for r in range(3):
        print("lstm alone            ", timed(run_lstm, [sA]))
        print("tdd alone             ", timed(lambda: run_tdd(False), [sB]))
        print("tdd low-LDS alone     ", timed(lambda: run_tdd(True), [sB]))
        print("tdd low-LDS x2 alone  ", timed(lambda: run_tdd(True, 2), [sB]))

        def both(n):
            run_lstm()
            run_tdd(True, n)
        print("lstm + tdd lowLDS x1  ", timed(lambda: both(1), [sA, sB]))
        print("lstm + tdd lowLDS x2  ", timed(lambda: both(2), [sA, sB]))

        def delayed(n, cyc):
            run_lstm()
            with torch.cuda.stream(sB):
                torch.cuda._sleep(cyc)
            run_tdd(True, n)
        with torch.cuda.stream(sB):
            print("sleep 1e6 cycles ms   ", timed(lambda: torch.cuda._sleep(1000000), [sB]))
        for cyc in (12000000, 20000000):
            print("lstm + sleep(%d) + tdd lowLDS x1" % cyc, timed(lambda: delayed(1, cyc), [sA, sB]))
            print("lstm + sleep(%d) + tdd lowLDS x2" % cyc, timed(lambda: delayed(2, cyc), [sA, sB]))

        def both_norm():
            run_lstm()
            run_tdd(False, 1)
        print("lstm + tdd normal x1  ", timed(both_norm, [sA, sB]))
        sys.stdout.flush()
    prof_names = ["rec_persistent_kernel", "conv1d_mfma_kernel"]
    lstm.destroy(); tdd.destroy()


if __name__ == "__main__":
    main()
