#!/usr/bin/env python3
"""LSTM(256 -> 512) through LSTMApplyDeviceFrag2h: the HF instantiation lstm_rr_kernel<8,4,hf> (in > 128 at H = 512 fits once U has no LDS image)
against the exact kernels the shape takes otherwise (option rec_hf = 0).   usage: python tools/hf_wide_time.py [B T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nntoolkitcore_amd import capi, layers as NL

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 500)
torch.cuda.set_device(0)
NL.use_torch_stream()
L = capi.load()
r = np.random.default_rng(5)
for I, H in ((256, 512), (200, 384), (128, 512)):
    uw = lambda fan, *s: r.uniform(-fan ** -0.5, fan ** -0.5, s).astype(np.float32)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(uw(I, I, 4 * H), uw(H, H, 4 * H), uw(H, 4 * H), uw(H, 4 * H))
    x = torch.randn(B, T, I, device="cuda")
    out = torch.empty(L.nntk_frag2h_floats(B, T, H), device="cuda")
    row = []
    for opt in ("auto", 0):
        capi.set_option("rec_hf", opt)
        for _ in range(2):
            NL.lstm_apply_device_frag2h(lstm, x=x, out_h2=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            NL.lstm_apply_device_frag2h(lstm, x=x, out_h2=out)
        e1.record(); torch.cuda.synchronize()
        row.append("%s %.3f ms (%s)" % ("rec_hf auto" if opt == "auto" else "rec_hf 0", e0.elapsed_time(e1) / 5, L.nntk_hip_last_recurrent_kernel().decode()))
    capi.set_option("rec_hf", "auto")
    print("LSTM(%d -> %d) B=%d T=%d frag2h output incl. packing of the f32 input: %s" % (I, H, B, T, " | ".join(row)))
    lstm.destroy()
