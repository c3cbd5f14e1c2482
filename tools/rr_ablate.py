#!/usr/bin/env python3
"""Diagnostics build only (tools/build_variant.sh <lib> -DNNTK_RR_DBG; NNTK_LIB=<lib>, one library per compile-time mask: tools/rr_ablate.sh): time lstm_rr_kernel at the stack's
LSTM shape with parts of the half-step removed (WRONG results, timing only).  usage: rr_ablate.py [B] [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch, bench
    from nntoolkitcore_amd import capi, layers as NL
    torch.cuda.set_device(0); capi.load(); NL.use_torch_stream()
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    B, T = 512, 996
    w = bench.make_weights("stack", 3)
    lstm = NL.LSTM(128, 512, True, T, v2=True)
    lstm.set_weights(w["lstm_W"], w["lstm_U"], w["lstm_bi"], w["lstm_bh"])
    x = torch.randn(B, T, 128, device="cuda"); h = torch.empty(B, T, 512, device="cuda")
    names = {1: "no operand loads", 2: "no finish", 4: "no arrive/poll", 8: "no x", 16: "no MFMA", 32: "no partial writes"}
    for _ in range(3):
        lstm.apply_device(x, out=h)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lstm.apply_device(x, out=h)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("dbg %3d: %.3f ms = %.2f us/step   [%s]" % (m, ms, ms * 1e3 / T, ", ".join(v for k, v in names.items() if m & k) or "everything"), flush=True)
    lstm.destroy()


if __name__ == "__main__":
    main()
