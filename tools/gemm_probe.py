#!/usr/bin/env python3
"""Time the dense GEMM kernel at the stack's shapes (diagnostics build: NNTK_EXTRA_HIPFLAGS=-DNNTK_CONV_DBG lets
NNTK_CONV_DBG=1|2|4 drop the stores / MFMAs / in-loop global loads)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    import torch
    from nntoolkitcore_amd import capi, layers as NL
    torch.cuda.set_device(0); capi.load(); NL.use_torch_stream()
    B, T = 512, 996
    shapes = [("xW  [510k,128]x[128,2048]", 128, 2048), ("TDD [510k,512]x[512,1000]", 512, 1000), ("    [510k,256]x[256,768]", 256, 768)]
    r = np.random.default_rng(0)
    for name, K, N in shapes:
        tdd = NL.TimeDistributedDense(T, K, N)
        tdd.set_weights(r.standard_normal((K, N)).astype(np.float32), r.standard_normal(N).astype(np.float32))
        x = torch.randn(B, T, K, device="cuda"); y = torch.empty(B, T, N, device="cuda")
        res = []
        for dbg in ("0", "1", "8", "2", "4", "7"):
            capi.set_option("conv_dbg", dbg)
            tdd.apply_device(x, out=y); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); tdd.apply_device(x, out=y); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res.append("dbg%s %.3f" % (dbg, np.median(ts)))
        flops = 2.0 * B * T * K * N
        print(name, " ".join(res), " ms | MFMA floor %.3f ms" % (flops / 157.3e12 * 1e3))
        tdd.destroy()
    capi.set_option("conv_dbg", "0")

if __name__ == "__main__":
    main()
