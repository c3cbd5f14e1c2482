#!/usr/bin/env python3
"""The standalone elementwise kernels at BASELINE config 3's size ([1024 x 996, 128] f32): BatchNorm (runtime.hip
bn_kernel_vec4_rows; batch_norm.c:140-163), ReLU and sigmoid (act_kernel; activation_default.c:28-33, :123-129), softmax over
128-vectors (softmax_short_kernel; activation_default.c:149-167).  In the stack they are fused into the conv / GEMM epilogues; this
is the memory-bound path north_star asks a GB/s figure for.  Prints achieved GB/s from HIP events (algorithmic bytes = one
read + one write of the tensor); run it under `rocprofv3 --kernel-trace --stats` for the per-kernel durations
(tools/final_profile.sh does, profiles/r03_elementwise_kernel_stats.csv)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import ctypes as C
    import torch
    from nntoolkitcore_amd import capi, layers as NL
    torch.cuda.set_device(0); L = capi.load(); NL.use_torch_stream()
    rows, Cc = 1024 * 996, 128
    r = np.random.default_rng(0)
    x = torch.randn(rows, Cc, device="cuda")
    y = torch.empty_like(x)
    nbytes = 2.0 * x.numel() * 4
    bn = NL.BatchNorm(Cc, 1e-3, rows)
    bn.set_weights(*(r.uniform(0.5, 1.5, Cc).astype(np.float32) for _ in range(4)))
    relu, sig = NL.Activation("relu", rows * Cc, 1.0), NL.Activation("sigmoid", rows * Cc)
    sm = NL.Activation("softmax", rows, 1.0, Cc)
    cases = [("BatchNormApplyDevice  (bn_kernel_vec4_rows)", lambda: bn.apply_device(x, out=y)),
             ("ReLU                  (act_kernel)", lambda: relu.apply_device(x, out=y)),
             ("sigmoid               (act_kernel)", lambda: sig.apply_device(x, out=y)),
             ("softmax over 128      (softmax_short_kernel<32>)", lambda: sm.apply_device(x, out=y))]
    for name, fn in cases:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%-42s %8.1f us  %6.2f TB/s = %.2f of 8 TB/s   (%.0f MB read + written)" % (name, ms * 1e3, nbytes / ms / 1e9, nbytes / ms / 1e9 / 8.0, nbytes / 1e6), flush=True)
    for o in (bn, relu, sig, sm):
        o.destroy()


if __name__ == "__main__":
    main()
