"""GPU tests of Conv1dBatchNormActivationApplyDeviceFrag3 (csrc/hip/conv1d_kernels.hpp conv_epilogue_frag3): the fused Conv1d -> BatchNorm
-> activation whose epilogue writes the NEXT layer's operand form -- a frag3 tensor -- instead of f32 [B][Tout][Cout] followed by the pack
pass.

Reference semantics: layers/conv_1d.c:122-147 (valid cross-correlation, channels-last), layers/batch_norm.c:140-163,
layers/activation_default.c; the seam is conv_1d.c's output feeding layers/lstm.c:201.  The contract: the tensor equals
Conv1dBatchNormActivationApplyDevice -> nntk_frag3_pack_device BIT FOR BIT (all three bf16 images, zeros in the padding channels), for the
shapes the frag3 epilogue takes (tile = 8 utterances x 16 timesteps) and for the ones it leaves to those two calls.
"""
import numpy as np
import pytest

import oracle as O
from nntoolkitcore_amd import capi, layers as NL

pytestmark = pytest.mark.gpu

F3 = "conv1d_mfma_bf16x3_kernel<frag3>"


def u(r, *shape, sc=1.0):
    return r.uniform(-sc, sc, shape).astype(np.float32)


def make(r, B, T, cin, cout, k, stride, bn, act, a=1.0):
    conv = NL.Conv1d(cin, cout, k, stride, T)
    W, b = u(r, cout, cin, k, sc=(cin * k) ** -0.5), u(r, cout, sc=0.5)
    conv.set_weights(W, b)
    Tc = conv.out_shape[0]
    bnl = bnw = None
    if bn:
        bnl = NL.BatchNorm(cout, 1e-3, B * Tc)
        bnw = (r.uniform(0.5, 1.5, cout).astype(np.float32), u(r, cout, sc=0.5), u(r, cout, sc=0.1), r.uniform(0.5, 1.5, cout).astype(np.float32))
        bnl.set_weights(*bnw)
    actl = NL.Activation(act, B * Tc * cout, a) if act else None
    return conv, bnl, actl, (W, b, bnw)


def oracle_of(x, W, b, bnw, act, a, stride):
    y = O.conv1d(x, W, b, stride)
    if bnw is not None:
        y = O.batch_norm(y, *bnw, 1e-3)
    if act:
        y = O.activation({"relu": O.ACT_RELU, "sigmoid": O.ACT_SIGMOID, "tanh": O.ACT_TANH, "identity": O.ACT_IDENTITY}[act], y, a)
    return y


@pytest.mark.parametrize("B,T,cin,cout,k,stride,bn,act,want", [
    (64, 100, 257, 128, 5, 1, True, "relu", F3),             # the stack's layer (configs[4]): Cin % 4 != 0, BN + ReLU, <2,2,2,2>
    (70, 37, 64, 100, 3, 1, True, "relu", F3),               # ragged utterance group, ragged last timestep block, Cout = 100: padding channels
    (5, 24, 48, 64, 9, 1, False, "sigmoid", F3),             # widest window (8 x 24 rows), <4,1,1,2>, no BatchNorm: padding channels must stay 0, not 0.5
    (130, 19, 32, 32, 2, 1, True, "tanh", F3),               # <4,1,1,1>, three row blocks
    (8, 16, 16, 40, 1, 1, False, None, F3),                  # k = 1 (a dense GEMM), plain Conv1dApplyDevice semantics, Cout = 40: 3 k steps
    (1, 40, 128, 128, 5, 1, True, "relu", F3),               # one utterance
    (9, 50, 40, 128, 5, 1, True, "relu", "conv1d_flatk_bf16x3_kernel"),     # flat-K layer (configs[2]): keeps its kernel, packed afterwards
    (16, 64, 32, 64, 3, 2, True, "relu", "conv1d_mfma_bf16x3_kernel"),      # stride 2: the two calls
    (16, 40, 32, 64, 11, 1, False, "relu", "conv1d_mfma_bf16x3_kernel"),    # k = 11: window past the staging budget
])
def test_conv_frag3_equals_conv_then_pack_bit_for_bit(gpu, B, T, cin, cout, k, stride, bn, act, want):
    import torch
    L = capi.load()
    r = np.random.default_rng(B * 7 + T + cin + cout + k)
    conv, bnl, actl, (W, b, bnw) = make(r, B, T, cin, cout, k, stride, bn, act, 0.5 if act == "relu" else 1.0)
    x = u(r, B, T, cin)
    xd = torch.from_numpy(x).cuda()
    Tc = conv.out_shape[0]
    y = conv.apply_device(xd, bn=bnl, act=actl)                          # f32 route (oracle-checked below and in test_gpu_parity.py)
    ref3 = NL.frag3_pack_device(y)                                       # zeros in padding rows and channels
    out3 = torch.zeros_like(ref3)                                        # (the frag3 epilogue does not write padding rows)
    conv.apply_device_frag3(xd, out_f3=out3, bn=bnl, act=actl)
    assert L.nntk_hip_last_conv_kernel().decode() == want
    assert torch.equal(out3.view(torch.int32), ref3.view(torch.int32))   # every image of every block, padding included
    assert torch.equal(NL.frag3_unpack_device(out3, B, Tc, cout), y)
    capi.set_option("conv_frag3_out", 0)                                 # the two calls, always
    alt = torch.zeros_like(ref3)
    conv.apply_device_frag3(xd, out_f3=alt, bn=bnl, act=actl)
    assert L.nntk_hip_last_conv_kernel().decode() != F3 and torch.equal(alt.view(torch.int32), ref3.view(torch.int32))
    capi.set_option("conv_frag3_out", "auto")
    np.testing.assert_allclose(y.cpu().numpy(), oracle_of(x, W, b, bnw, act, 0.5 if act == "relu" else 1.0, stride), rtol=2e-5, atol=2e-5)
    for h in (conv, bnl, actl):
        if h is not None:
            h.destroy()


def test_conv_frag3_garbage_in_the_buffer_and_exact_only_weights(gpu):
    """Stale contents of the output buffer never show through in a valid row; a weight the bf16 split cannot hold (inf) sends the layer to the
    exact-f32 kernel + pack, like the f32 call."""
    import torch
    L = capi.load()
    r = np.random.default_rng(5)
    B, T, cin, cout, k = 24, 33, 64, 128, 5
    conv, bnl, actl, (W0, b0, _) = make(r, B, T, cin, cout, k, 1, True, "relu")
    xd = torch.from_numpy(u(r, B, T, cin)).cuda()
    Tc = conv.out_shape[0]
    y = conv.apply_device(xd, bn=bnl, act=actl)
    out3 = torch.full((L.nntk_frag3_floats(B, Tc, cout),), float("nan"), device="cuda")
    conv.apply_device_frag3(xd, out_f3=out3, bn=bnl, act=actl)
    assert L.nntk_hip_last_conv_kernel().decode() == F3
    assert torch.equal(NL.frag3_unpack_device(out3, B, Tc, cout), y)
    W = np.array(W0, copy=True)
    W[3, 2, 1] = np.inf
    conv.set_weights(W, b0)
    conv.sync_weights()
    y2 = conv.apply_device(xd, bn=bnl, act=actl)
    assert L.nntk_hip_last_conv_kernel().decode() == "conv1d_mfma_kernel"
    o2 = conv.apply_device_frag3(xd, bn=bnl, act=actl)
    assert L.nntk_hip_last_conv_kernel().decode() == "conv1d_mfma_kernel"
    a, bq = NL.frag3_unpack_device(o2, B, Tc, cout), y2
    fin = torch.isfinite(bq)                                             # (inf / NaN outputs are outside the format: INTEGRATION.md section 6)
    assert torch.equal(a[fin], bq[fin])
    for h in (conv, bnl, actl):
        h.destroy()


@pytest.mark.parametrize("cell,H", [("lstm", 512), ("gru", 256)])
def test_conv_frag3_feeds_the_recurrent_layer_like_the_f32_tensor(gpu, cell, H):
    """The seam the format exists for: conv (frag3 epilogue) -> register-resident recurrence, against conv (f32) -> the same layer."""
    import torch
    L = capi.load()
    r = np.random.default_rng(H)
    B, T, cin, cout, k = 96, 44, 257, 128, 5
    conv, bnl, actl, _ = make(r, B, T, cin, cout, k, 1, True, "relu")
    Tc = conv.out_shape[0]
    G = 4 if cell == "lstm" else 3
    lay = NL.LSTM(cout, H, True, Tc, v2=True) if cell == "lstm" else NL.GRU(cout, H, True, Tc)
    lay.set_weights(u(r, cout, G * H, sc=cout ** -0.5), u(r, H, G * H, sc=H ** -0.5), u(r, G * H, sc=0.1), u(r, G * H, sc=0.1))
    xd = torch.from_numpy(u(r, B, T, cin)).cuda()
    base = lay.apply_device(conv.apply_device(xd, bn=bnl, act=actl)).clone()
    c3 = conv.apply_device_frag3(xd, bn=bnl, act=actl)
    assert L.nntk_hip_last_conv_kernel().decode() == F3
    got, _ = NL.recurrent_apply_device_frag3(lay, x_f3=c3, batch=B, want_f32=True)
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith(cell + "_rr_kernel")
    assert torch.equal(got, base)
    for h in (conv, bnl, actl, lay):
        h.destroy()


def test_conv_frag3_at_the_stack_size(gpu):
    """configs[4]'s conv at its real size (512 utterances x 1000 frames x 257 bins): the frag3 tensor against the f32 route over the whole
    batch, a second run, and a 64-utterance shard (other tiles, same bits)."""
    import torch
    L = capi.load()
    r = np.random.default_rng(11)
    B, T, cin, cout, k = 512, 1000, 257, 128, 5
    conv, bnl, actl, _ = make(r, B, T, cin, cout, k, 1, True, "relu")
    Tc = conv.out_shape[0]
    xd = torch.rand((B, T, cin), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 4.0
    y = conv.apply_device(xd, bn=bnl, act=actl)
    o3 = conv.apply_device_frag3(xd, bn=bnl, act=actl)
    assert L.nntk_hip_last_conv_kernel().decode() == F3
    assert torch.equal(o3.view(torch.int32), NL.frag3_pack_device(y).view(torch.int32))
    assert torch.equal(conv.apply_device_frag3(xd, bn=bnl, act=actl).view(torch.int32), o3.view(torch.int32))
    sh = conv.apply_device_frag3(xd[448:].contiguous(), bn=bnl, act=actl)
    assert torch.equal(NL.frag3_unpack_device(sh, 64, Tc, cout), y[448:])
    for h in (conv, bnl, actl):
        h.destroy()
