#!/usr/bin/env python3
"""Error and time of the opt-in split-bf16 contraction (option gemm_split_bf16) beside the exact-f32 MFMA path, on the
BASELINE GEMM shapes: conv config 3, the stack's conv / LSTM input projection / TimeDistributedDense.  Errors are max and
rms |y - y64| against a float64 contraction of the same f32 inputs (torch, on the GPU), relative to rms(y64).
usage: python tools/split_error.py [--rounds N]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nntoolkitcore_amd import capi, layers as NL
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 5
    torch.cuda.set_device(0)
    capi.load()
    NL.use_torch_stream()
    g = torch.Generator(device="cuda").manual_seed(5)
    u = lambda *s, sc=1.0: (torch.rand(*s, generator=g, device="cuda") * 2 - 1) * sc
    cases = [  # name, B, T, Cin, Cout, k
        ("conv config 3 (40->128, k5)", 1024, 1000, 40, 128, 5),
        ("stack conv (257->128, k5)", 256, 1000, 257, 128, 5),
        ("LSTM-512 input projection (128->2048)", 256, 996, 128, 2048, 1),
        ("TimeDistributedDense (512->1000)", 256, 996, 512, 1000, 1),
        ("GRU-256 input projection (128->768)", 256, 1000, 128, 768, 1),
    ]
    # stress distributions on one shape: error measured against the float64 contraction and normalised by the
    # condition-free scale sum_k |x_k| |w_k| (the quantity any f32 summation's error bound is proportional to)
    def lognormal(*s):
        return torch.exp(4.0 * torch.randn(*s, generator=g, device="cuda")) * torch.sign(u(*s))
    stress = [
        ("uniform", lambda *s: u(*s), lambda *s: u(*s)),
        ("lognormal sigma 4 (13 decades)", lognormal, lambda *s: u(*s)),
        ("both lognormal", lognormal, lognormal),
        ("cancellation: x = 1000 + noise, w alternating", lambda *s: 1000.0 + u(*s), None),
        ("tiny 1e-30", lambda *s: 1e-30 * u(*s), lambda *s: u(*s)),
        ("huge 1e30 x 1e-3", lambda *s: 1e30 * u(*s), lambda *s: 1e-3 * u(*s)),
        ("denormal inputs 1e-40", lambda *s: 1e-40 * u(*s), lambda *s: u(*s)),
    ]
    if "--stress" in sys.argv:
        B, T, Cin, Cout, k = 64, 500, 40, 128, 5
        for name, fx, fw in stress:
            x = fx(B, T, Cin)
            if fw is None:
                W = torch.ones(Cout, Cin, k, device="cuda")
                W.view(Cout, -1)[:, 1::2] = -1.0
                W = W * (1 + 1e-3 * u(Cout, Cin, k))
            else:
                W = fw(Cout, Cin, k)
            bias = torch.zeros(Cout, device="cuda")
            conv = NL.Conv1d(Cin, Cout, k, 1, T)
            conv.set_weights(W.cpu().numpy(), bias.cpu().numpy())
            xs = x.double().transpose(1, 2)
            y64 = torch.nn.functional.conv1d(xs, W.double()).transpose(1, 2)
            mag = torch.nn.functional.conv1d(xs.abs(), W.double().abs()).transpose(1, 2) + 1e-300
            res = {}
            for mode in ("0", "1"):
                capi.set_option("gemm_split_bf16", mode)
                out = torch.empty(B, T - k + 1, Cout, device="cuda")
                conv.apply_device(x, out=out)
                torch.cuda.synchronize()
                e = (out.double() - y64).abs() / mag
                res[mode] = dict(max=float(e.max()), mean=float(e.mean()), finite=bool(torch.isfinite(out).all()))
            capi.set_option("gemm_split_bf16", "0")
            conv.destroy()
            print(json.dumps(dict(stress=name, units="error / sum|x||w|  (2^-24 = 5.96e-8)", exact=res["0"], split=res["1"])))
        return 0
    rows = []
    for name, B, T, Cin, Cout, k in cases:
        x = u(B, T, Cin)
        W = u(Cout, Cin, k, sc=(Cin * k) ** -0.5)
        bias = u(Cout, sc=0.1)
        conv = NL.Conv1d(Cin, Cout, k, 1, T)
        conv.set_weights(W.cpu().numpy(), bias.cpu().numpy())
        Tout = T - k + 1
        # float64 reference in slices of the batch (memory)
        y64 = torch.empty(B, Tout, Cout, dtype=torch.float64, device="cuda")
        W64 = W.double()
        for b0 in range(0, B, 32):
            xs = x[b0:b0 + 32].double().transpose(1, 2)                   # [b, Cin, T]
            y64[b0:b0 + 32] = torch.nn.functional.conv1d(xs, W64, bias.double()).transpose(1, 2)
        scale = float(y64.pow(2).mean().sqrt())
        res = {}
        for mode in ("0", "1"):
            capi.set_option("gemm_split_bf16", mode)
            out = torch.empty(B, Tout, Cout, device="cuda")
            conv.apply_device(x, out=out)
            torch.cuda.synchronize()
            ts = []
            for _ in range(rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                conv.apply_device(x, out=out)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            d = (out.double() - y64)
            res[mode] = dict(ms=float(np.median(ts)), max_rel=float(d.abs().max()) / scale, rms_rel=float(d.pow(2).mean().sqrt()) / scale,
                             max_abs=float(d.abs().max()))
            if mode == "0":
                exact = out.clone()
            else:
                res[mode]["max_abs_vs_exact"] = float((out - exact).abs().max())
        capi.set_option("gemm_split_bf16", "0")
        conv.destroy()
        rows.append(dict(case=name, shape=[B, T, Cin, Cout, k], rms_ref=scale, exact=res["0"], split=res["1"]))
        print(json.dumps(rows[-1]))
        del y64, x
        torch.cuda.empty_cache()
    return 0


if __name__ == "__main__":
    sys.exit(main())
