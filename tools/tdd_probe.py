"""Where the time of the frag3 dense GEMM goes: the stack's TimeDistributedDense (rows = B x T, 512 -> 1000) on the f32-input kernel, the
register-direct frag3 kernel and the LDS-ring frag3 kernel, the latter with pieces switched off (option conv_dbg: 1 no output stores,
2 operand fetch of k step 0 only, 4 no MFMAs).   usage: python tools/tdd_probe.py [B] [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nntoolkitcore_amd import capi, layers as NL

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 996
K, N = 512, 1000
torch.cuda.set_device(0); capi.load(); NL.use_torch_stream()
r = np.random.default_rng(1)
tdd = NL.TimeDistributedDense(T, K, N)
tdd.set_weights(r.uniform(-0.04, 0.04, (K, N)).astype(np.float32), r.uniform(-0.1, 0.1, N).astype(np.float32))
x = torch.rand(B, T, K, device="cuda") - 0.5
x3 = NL.frag3_pack_device(x)
out = torch.empty(B, T, N, device="cuda")

def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

flops = 2.0 * B * T * K * N
for rnd in range(2):
    print("f32 input (LDS-staged, splits A in the kernel): %.3f ms" % timeit(lambda: tdd.apply_device(x, out=out)))
    for mode, name in ((1, "frag3 register-direct"), (3, "frag3 LDS ring")):
        capi.set_option("dense_frag3", mode)
        for dbg in ((0, 1, 2, 4, 3, 7) if mode == 3 and os.environ.get('TDD_DBG') else (0,)):
            capi.set_option("conv_dbg", dbg)
            ms = timeit(lambda: NL.tdd_apply_device_frag3(tdd, x3, B, out=out))
            print("%s dbg=%d: %.3f ms  (%.0f TFLOP/s)" % (name, dbg, ms, flops / ms / 1e9))
        capi.set_option("conv_dbg", 0)
capi.set_option("dense_frag3", "auto")
