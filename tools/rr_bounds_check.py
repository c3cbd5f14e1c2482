#!/usr/bin/env python3
"""Do the register-resident recurrent kernels ever ask for a byte outside the tensors they were given?

Builds a diagnostics copy of the library with recurrent_rr.hip and recurrent_fk.hip compiled -DNNTK_RR_BOUNDS (every request towards a caller-visible
tensor records the last byte it really touches; lanes the buffer range check drops are skipped, as the hardware skips them), runs
GRU / LSTM layers whose last batch tile is ragged (B = 33, 65, 130: a tile with one row in its second half, one row in its first
half, a half-empty tile) through every input / output form, and compares the recorded extents with the tensors' sizes.
Round 3 found such a read by reading the code (the half-tile rode in the scalar offset, which the range check does not see);
this makes it a test.   usage: python tools/rr_bounds_check.py [--keep]      (needs a GPU and hipcc; ~2 minutes)"""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "rr_bounds")
LIB = os.path.join(OUT, "libnntoolkitcore_hip_bounds.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    from nntoolkitcore_amd import _build
    _build.build()                                                   # the product objects (up to date on the GPU box: they travel)
    objs, procs = [], []
    for unit in ("recurrent_rr.hip", "recurrent_fk.hip"):            # both register-resident families record their requests
        obj = os.path.join(OUT, unit[:-4] + "_bounds.o")
        objs.append(obj)
        procs.append(subprocess.Popen([_build.HIPCC, "-O3", "--offload-arch=" + _build.ARCH, "-fPIC", "-std=c++17", "-Wno-unused-function",
                                       "-DNNTK_RR_BOUNDS", "-c", os.path.join(_build.CSRC, "hip", unit), "-o", obj]))
    assert all(p.wait() == 0 for p in procs)
    others = [o for o in glob.glob(os.path.join(_build.OBJ, "*.o")) if not o.endswith(("recurrent_rr.hip.o", "recurrent_fk.hip.o"))]
    subprocess.check_call([_build.HIPCC, "--offload-arch=" + _build.ARCH, "-shared", "-fPIC", "-o", LIB] + objs + others)


def run():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from nntoolkitcore_amd import capi, layers as NL
    assert os.path.abspath(capi.LIB_PATH) == os.path.abspath(LIB), capi.LIB_PATH
    L = capi.load()
    fetch = L.nntk_shim_rr_bounds_fetch
    fetch.restype, fetch.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
    torch.cuda.set_device(0)
    NL.use_torch_stream()
    r = np.random.default_rng(3)
    u = lambda *s, sc=1.0: r.uniform(-sc, sc, s).astype(np.float32)
    bad = 0
    names = ["x f32 rows", "x frag3", "out f32", "hseq frag3", "h0 slot"]
    for cell in ("lstm", "gru"):
        for (B, I, H, T) in ((33, 128, 512, 5), (65, 64, 256, 4), (130, 40, 128, 3), (130, 256, 256, 3), (1, 8, 64, 6)):
            if cell == "lstm" and H == 512 and I > 128:
                continue
            G = 4 if cell == "lstm" else 3
            layer = NL.LSTM(I, H, True, T, v2=True) if cell == "lstm" else NL.GRU(I, H, True, T)
            layer.set_weights(u(I, G * H, sc=I ** -0.5), u(H, G * H, sc=H ** -0.5), u(G * H, sc=0.1), u(G * H, sc=0.1))
            x = torch.from_numpy(u(B, T, I)).cuda()
            x3 = NL.frag3_pack_device(x)
            f3_in, f3_out = 4 * L.nntk_frag3_floats(B, T, I), 4 * L.nntk_frag3_floats(B, T, H)
            limits = [4 * B * T * I, f3_in, 4 * B * T * H, f3_out, f3_out // T]
            for route in ("f32->f32", "frag3->f32", "f32->frag3", "frag3->frag3"):
                capi.set_option("rec_xf", 0)                         # "f32" input really runs the f32-row form of the kernel
                src, dst = route.split("->")
                NL.recurrent_apply_device_frag3(layer, x=x if src == "f32" else None, x_f3=x3 if src == "frag3" else None, batch=B,
                                                want_f32=dst == "f32", want_f3=dst == "frag3")
                kern = L.nntk_hip_last_recurrent_kernel().decode()
                assert kern.startswith((cell + "_rr_kernel", cell + "_fk_kernel")), kern
                fk = "_fk_kernel" in kern                              # (the full-K family reads frag3 only: an f32 input is packed first)
                got = (C.c_ulonglong * 8)()
                assert fetch(got) == 0
                got = list(got)[:5]
                ok = all(g <= lim for g, lim in zip(got, limits))
                # the instrument is alive: the forms in use reach exactly the end of their tensors (the last row's last bytes)
                live = (got[0] == limits[0]) if (src == "f32" and not fk) else (got[1] > 0 and got[0] == 0)
                live = live and ((got[2] == limits[2]) if dst == "f32" else got[2] == 0) and got[3] > 0 and got[4] > 0
                print("%s B=%d in=%d H=%d T=%d %-12s %-22s %s%s" % (cell, B, I, H, T, route, kern,
                      "  ".join("%s %d/%d" % (n, g, lim) for n, g, lim in zip(names, got, limits)),
                      "" if ok and live else "   <-- %s" % ("OUT OF BOUNDS" if not ok else "instrument silent")))
                bad += (not ok) or (not live)
            if cell == "lstm":                                          # the FRAG2H form of the output (the output wave's two 1 KB stores per block)
                NL.lstm_apply_device_frag2h(layer, x_f3=x3, batch=B)
                kern = L.nntk_hip_last_recurrent_kernel().decode()
                got = (C.c_ulonglong * 8)()
                assert fetch(got) == 0
                lim = 4 * L.nntk_frag2h_floats(B, T, H)
                fk = "_fk_kernel" in kern                                # (the full-K family writes f32 rows and the pack pass makes the form)
                hfk = kern.endswith(",hf>")                             # (the HF instantiation's hand-off IS the FRAG2H tensor: word 3)
                ok = got[5] <= lim and got[3] <= (lim if hfk else f3_out) and (got[2] <= limits[2] if fk else got[2] == 0)
                live = fk or (got[3] == lim and got[5] == 0 if hfk else got[5] == lim)
                print("%s B=%d in=%d H=%d T=%d %-12s %-22s out frag2h %d/%d%s" % (cell, B, I, H, T, "frag3->frag2h", kern, got[5], lim,
                      "" if ok and live else "   <-- %s" % ("OUT OF BOUNDS" if not ok else "instrument silent")))
                bad += (not ok) or (not live)
            capi.set_option("rec_xf", "auto")
            layer.destroy()
    print("rr bounds: %d violation(s)" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    if "--run" in sys.argv:
        sys.exit(run())
    build()
    env = dict(os.environ, NNTK_LIB=LIB)
    rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--run"], env=env)
    if "--keep" not in sys.argv and os.path.exists(LIB):
        os.remove(LIB)
    sys.exit(rc)
