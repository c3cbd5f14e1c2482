#!/usr/bin/env python3
"""Race detector for the HF instantiation of lstm_rr_kernel at the bench's own launch (512 x 996, 128 -> 512: all 256 workgroups), whose flag protocol
runs on a shorter half-step, an earlier arrival and two stores per publication: two DIFFERENT inputs alternate through ONE output buffer (a stale block
of the launch before would be the other input's h), the buffer is filled with garbage in between, and every launch must equal its input's reference
(taken from a fresh buffer) bit for bit over the whole tensor.   usage: python tools/hf_soak.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nntoolkitcore_amd import capi, layers as NL

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
torch.cuda.set_device(0)
NL.use_torch_stream()
L = capi.load()
B, I, H, T = 512, 128, 512, 996
r = np.random.default_rng(77)
uw = lambda fan, *s: r.uniform(-fan ** -0.5, fan ** -0.5, s).astype(np.float32)
lstm = NL.LSTM(I, H, True, T, v2=True)
lstm.set_weights(uw(I, I, 4 * H), uw(H, H, 4 * H), uw(H, 4 * H), uw(H, 4 * H))
g = torch.Generator(device="cuda").manual_seed(11)
xs = [torch.randn(B, T, I, device="cuda", generator=g) for _ in range(2)]
x3 = [NL.frag3_pack_device(x) for x in xs]
refs = [NL.lstm_apply_device_frag2h(lstm, x_f3=x3[k], batch=B).clone() for k in range(2)]
assert L.nntk_hip_last_recurrent_kernel().decode() == "lstm_rr_kernel<8,2,hf>", L.nntk_hip_last_recurrent_kernel()
assert not torch.equal(refs[0], refs[1])
buf = torch.empty_like(refs[0])
bad = 0
for it in range(rounds):
    for k in (0, 1):
        if it % 3 == 0:
            buf.copy_(torch.rand_like(buf))                 # finite garbage (neither input's h, no pending pattern)
        NL.lstm_apply_device_frag2h(lstm, x_f3=x3[k], batch=B, out_h2=buf)
        same = torch.equal(buf, refs[k])
        bad += not same
        print("round %d input %d: %s" % (it, k, "equal" if same else "DIFFERENT (max |d| of the raw words: %d)" % int((buf.view(torch.int32) - refs[k].view(torch.int32)).abs().max())), flush=True)
print("hf soak: %d launches, %d different, device status %d" % (2 * rounds, bad, L.nntk_hip_device_status()))
sys.exit(1 if bad or L.nntk_hip_device_status() else 0)
