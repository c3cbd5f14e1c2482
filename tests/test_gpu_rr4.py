"""GPU tests of the four-stream register-resident recurrent kernels (csrc/hip/recurrent_rr4.hip: gru_rr4_kernel / lstm_rr4_kernel).

Reference semantics: layers/gru.c:129-204, :246-293; layers/lstm.c:185-239, :426-475.  The kernels sum the same products in the same
order as the two-stream family (recurrent_rr.hip), so they must equal it BIT FOR BIT -- the host chooses between the families by
speed -- and both are compared with the oracle.
"""
import numpy as np
import pytest

import oracle as O
from nntoolkitcore_amd import capi, layers as NL

pytestmark = pytest.mark.gpu


def rng(seed):
    return np.random.default_rng(seed)


def u(r, *shape, sc=1.0):
    return r.uniform(-sc, sc, shape).astype(np.float32)


def make(cell, I, H, T, seq=True):
    return NL.LSTM(I, H, seq, T, v2=True) if cell == "lstm" else NL.GRU(I, H, seq, T)


def weights(r, cell, I, H):
    G = 4 if cell == "lstm" else 3
    return u(r, I, G * H, sc=I ** -0.5), u(r, H, G * H, sc=H ** -0.5), u(r, G * H, sc=0.1), u(r, G * H, sc=0.1)


def oracle(cell, x, W, U, bi, bh, **kw):
    ref = O.lstm(x, W, U, bi, bh, v2=True, **kw) if cell == "lstm" else O.gru(x, W, U, bi, bh, **kw)
    return ref[0] if isinstance(ref, tuple) else ref


@pytest.mark.parametrize("cell", ["gru", "lstm"])
@pytest.mark.parametrize("B,I,H,T", [
    (128, 128, 256, 9),      # configs[3] layer 1: KH = 4 / KX = 2, one full 128-row tile
    (130, 40, 128, 7),       # ragged: the second tile holds one half-stream with two rows, three empty streams
    (33, 256, 256, 5),       # configs[3] layer 2: KX = 4; a stream with one row
    (300, 64, 64, 6),        # KX = 1, three tiles, 8 column tiles
    (70, 128, 512, 5),       # KH = 8: 64 column tiles
    (1, 8, 64, 4),
    (96, 72, 192, 3),        # H between the compiled depths
    (64, 100, 256, 1),       # T = 1; in % 8 != 0 (frag3 input only)
])
def test_rr4_equals_the_two_stream_family_bit_for_bit_and_the_oracle(gpu, cell, B, I, H, T):
    import torch
    r = rng(B + I + H + T)
    x = u(r, B, T, I)
    W, U, bi, bh = weights(r, cell, I, H)
    layer = make(cell, I, H, T)
    layer.set_weights(W, U, bi, bh)
    L = capi.load()
    xd = torch.from_numpy(x).cuda()
    capi.set_option("rec_xf", 1)                      # frag3 input (the four-stream kernels take nothing else)
    capi.set_option("rec_rr4", 0)
    two, two3 = NL.recurrent_apply_device_frag3(layer, x=xd, want_f32=True, want_f3=True)
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith(cell + "_rr_kernel")
    capi.set_option("rec_rr4", 1)
    four, four3 = NL.recurrent_apply_device_frag3(layer, x=xd, want_f32=True, want_f3=True)
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith(cell + "_rr4_kernel"), L.nntk_hip_last_recurrent_kernel()
    assert L.nntk_hip_device_status() == 0
    assert torch.equal(four, two)
    assert torch.equal(NL.frag3_unpack_device(four3, B, T, H), two)
    _, only3 = NL.recurrent_apply_device_frag3(layer, x=xd, want_f32=False, want_f3=True)       # no f32 output at all
    assert torch.equal(NL.frag3_unpack_device(only3, B, T, H), two)
    assert torch.equal(layer.apply_device(xd), two)                                             # the plain device call
    np.testing.assert_allclose(four.cpu().numpy(), oracle(cell, x, W, U, bi, bh), rtol=1e-5, atol=1e-5)
    layer.destroy()


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_rr4_last_state_only_and_carried_state(gpu, cell):
    """return_sequences = false (only the last step leaves), and the stateful single-sequence call forced onto the kernels
    (rec_rr = 1): three calls with carried h (and c) equal one long oracle run."""
    import torch
    r = rng(17)
    B, I, H, T = 150, 64, 128, 8
    x = u(r, B, T, I)
    W, U, bi, bh = weights(r, cell, I, H)
    capi.set_option("rec_xf", 1); capi.set_option("rec_rr4", 1)
    layer = make(cell, I, H, T, seq=False)
    layer.set_weights(W, U, bi, bh)
    got = layer.apply_device(torch.from_numpy(x).cuda()).cpu().numpy()
    assert capi.load().nntk_hip_last_recurrent_kernel().decode().startswith(cell + "_rr4_kernel")
    np.testing.assert_allclose(got, oracle(cell, x, W, U, bi, bh, return_sequences=False), rtol=1e-5, atol=1e-5)
    layer.destroy()
    capi.set_option("rec_rr", 1); capi.set_option("rec_stream", 0)
    x1 = u(r, 3 * T, I)
    layer = make(cell, I, H, T)
    layer.set_weights(W, U, bi, bh)
    outs = np.concatenate([layer.apply(x1[i * T:(i + 1) * T]) for i in range(3)])
    assert capi.load().nntk_hip_last_recurrent_kernel().decode().startswith(cell + "_rr4_kernel")
    ref = O.lstm(x1, W, U, bi, bh, v2=True) if cell == "lstm" else O.gru(x1, W, U, bi, bh)
    np.testing.assert_allclose(outs, ref[0], rtol=1e-5, atol=1e-5)
    st = layer.state()
    np.testing.assert_allclose(st[0] if cell == "lstm" else st, ref[1], rtol=1e-5, atol=1e-5)
    if cell == "lstm":
        np.testing.assert_allclose(st[1], ref[2], rtol=1e-5, atol=1e-5)
    layer.destroy()


def test_gru_stack_on_the_four_stream_kernels(gpu):
    """GRUStack2ApplyDevice with both layers on gru_rr4_kernel (<4,2> then <4,4>, layer 1's frag3 hand-off is layer 2's x operand):
    equal to the two-stream pair bit for bit, within tolerance of the oracle; B = 200: a full tile and a ragged one."""
    import torch
    r = rng(23)
    B, I, H, T = 200, 128, 256, 12
    x = u(r, B, T, I)
    w1, w2 = weights(r, "gru", I, H), weights(r, "gru", H, H)
    g1, g2 = NL.GRU(I, H, True, T), NL.GRU(H, H, True, T)
    g1.set_weights(*w1); g2.set_weights(*w2)
    xd = torch.from_numpy(x).cuda()
    L = capi.load()
    capi.set_option("rec_rr4", 0)
    two = NL.gru_stack2_apply_device(g1, g2, xd).clone()
    assert L.nntk_hip_last_recurrent_kernel().decode() == "gru_rr_kernel<4,4>"
    capi.set_option("rec_rr4", 1)
    four = NL.gru_stack2_apply_device(g1, g2, xd).clone()
    assert L.nntk_hip_last_recurrent_kernel().decode() == "gru_rr4_kernel<4,4>"
    assert torch.equal(four, two)
    assert torch.equal(NL.gru_stack2_apply_device(g1, g2, xd), four)          # reproducible
    ref = O.gru(O.gru(x, *w1), *w2)
    np.testing.assert_allclose(four.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    assert L.nntk_hip_device_status() == 0
    g1.destroy(); g2.destroy()


def test_rr4_fault_is_reported(gpu):
    """A poll that runs out of budget (forced: rec_spin_us = 0) raises the sticky fault word, like every persistent kernel here."""
    import torch
    L = capi.load()
    r = rng(29)
    B, I, H, T = 140, 64, 128, 6
    x = u(r, B, T, I)
    W, U, bi, bh = weights(r, "gru", I, H)
    capi.set_option("rec_xf", 1); capi.set_option("rec_rr4", 1)
    gru = NL.GRU(I, H, True, T)
    gru.set_weights(W, U, bi, bh)
    xd = torch.from_numpy(x).cuda()
    good = gru.apply_device(xd).cpu().numpy()
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith("gru_rr4_kernel")
    capi.set_option("rec_spin_us", 0)
    gru.apply_device(xd)
    torch.cuda.synchronize()
    assert L.nntk_hip_device_status() == 1
    assert L.nntk_hip_synchronize() == -1 and "timed out" in capi.last_error()
    capi.set_option("rec_spin_us", "auto"); capi.set_option("rec_persistent", 1)      # re-arm
    np.testing.assert_array_equal(gru.apply_device(xd).cpu().numpy(), good)
    assert L.nntk_hip_synchronize() == 0
    gru.destroy()
