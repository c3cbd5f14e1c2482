"""GPU tests of the frag3 tensor path (csrc/hip/frag3.hip, the XF instantiations of recurrent_rr.hip): activations that travel between
layers already split into the three bf16 images of the split-bf16 x 3 contraction, in MFMA fragment order.

Reference semantics: layers/lstm.c:185-239, :426-475; layers/gru.c:129-204, :246-293; layers/dense.c:122-133;
layers/time_distributed_dense.c:52-58.  The format is exact (x = hi + mid + lo), so every frag3 route must equal its f32 route BIT FOR BIT;
the f32 routes are the ones the rest of the suite compares with the oracle -- and the new kernels are compared with the oracle here too.
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


def bf16_to_f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


def decode_frag3(buf, B, T, C):
    """Host model of the documented layout: [T][2 ceil(B/64)][ceil(C/16)][3 images] blocks of 1 KB; lane 32 kh + n of a block holds
    channels 16 ks + 8 kh .. + 7 of batch row 32 ht + n as 8 consecutive bf16."""
    NHT, NKS = (B + 63) // 64 * 2, (C + 15) // 16
    raw = buf.view(np.uint16).reshape(T, NHT, NKS, 3, 2, 32, 8)          # [t][ht][ks][m][kh][n][q]
    imgs = bf16_to_f32(raw)
    val = (imgs[:, :, :, 0] + imgs[:, :, :, 1]) + imgs[:, :, :, 2]        # [t][ht][ks][kh][n][q]
    full = val.transpose(1, 4, 0, 2, 3, 5).reshape(NHT * 32, T, NKS * 16)  # [b][t][c]
    return full, imgs


@pytest.mark.parametrize("B,T,C", [(70, 5, 40), (1, 3, 257), (130, 2, 128), (64, 4, 16), (33, 7, 100)])
def test_frag3_pack_layout_and_exact_round_trip(gpu, B, T, C):
    import torch
    r = rng(B + T + C)
    x = u(r, B, T, C) * np.float32(10.0) ** r.integers(-6, 6, (B, T, C)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    f3 = NL.frag3_pack_device(xd)
    assert f3.numel() == capi.load().nntk_frag3_floats(B, T, C)
    full, imgs = decode_frag3(f3.cpu().numpy(), B, T, C)
    assert np.array_equal(full[:B, :, :C], x)                     # hi + mid + lo is the f32 value exactly
    assert not full[B:].any() and not full[:, :, C:].any()        # padding rows and channels are zeros
    back = NL.frag3_unpack_device(f3, B, T, C)
    assert torch.equal(back, xd)


@pytest.mark.parametrize("cell", ["lstm", "gru"])
@pytest.mark.parametrize("B,I,H,T", [
    (64, 128, 512, 12),      # the stack's LSTM shape, KH = 8 / KX = 2
    (33, 40, 128, 9),        # ragged second half-tile; in padded to 48 of 64
    (130, 100, 256, 7),      # in % 8 != 0: only the frag3 input form can feed the register-resident kernels
    (65, 256, 256, 6),       # 256-wide input: the full-K kernel (gru_fk / lstm_fk <16,16,4>), which reads frag3 only
    (96, 72, 320, 5),        # H between the compiled depths: k steps past H / 16 read as zeros
    (5, 8, 64, 11),          # fewer rows than one half-tile
])
def test_recurrent_frag3_routes_equal_the_f32_route_bit_for_bit(gpu, cell, B, I, H, T):
    import torch
    if cell == "lstm" and H == 512 and I > 128:
        pytest.skip("shape not compiled")
    r = rng(B * 3 + I + H + T)
    G = 4 if cell == "lstm" else 3
    x = u(r, B, T, I)
    W, U, bi, bh = u(r, I, G * H, sc=I ** -0.5), u(r, H, G * H, sc=H ** -0.5), u(r, G * H, sc=0.1), u(r, G * H, sc=0.1)
    layer = NL.LSTM(I, H, True, T, v2=True) if cell == "lstm" else NL.GRU(I, H, True, T)
    layer.set_weights(W, U, bi, bh)
    L = capi.load()
    xd = torch.from_numpy(x).cuda()
    base = layer.apply_device(xd).clone()                               # the f32 route (oracle-checked below)
    x3 = NL.frag3_pack_device(xd)
    o_a, f_a = NL.recurrent_apply_device_frag3(layer, x=xd, want_f32=True, want_f3=True)
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith(cell + ("_fk_kernel" if (I, H) == (256, 256) else "_rr_kernel"))
    o_b, f_b = NL.recurrent_apply_device_frag3(layer, x_f3=x3, batch=B, want_f32=True, want_f3=True)
    _, f_c = NL.recurrent_apply_device_frag3(layer, x_f3=x3, batch=B, want_f32=False, want_f3=True)
    assert torch.equal(o_a, base) and torch.equal(o_b, base)
    for f in (f_a, f_b, f_c):
        assert torch.equal(NL.frag3_unpack_device(f, B, T, H), base)    # the frag3 output holds the f32 output exactly
    capi.set_option("rec_xf", 1)                                        # the plain device call, packing x itself
    assert torch.equal(layer.apply_device(xd), base)
    capi.set_option("rec_xf", "auto")
    ofn = O.lstm if cell == "lstm" else O.gru
    ref = ofn(x, W, U, bi, bh, **({"v2": True} if cell == "lstm" else {}))
    ref = ref[0] if isinstance(ref, tuple) else ref
    np.testing.assert_allclose(base.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    assert L.nntk_hip_device_status() == 0
    layer.destroy()


def test_frag3_routes_of_shapes_the_register_resident_kernels_do_not_take(gpu):
    """H % 16 != 0 and non-default activations run the exact kernels through f32 scratch: same bits as the f32 call."""
    import torch
    r = rng(5)
    B, I, H, T = 9, 24, 40, 6
    x = u(r, B, T, I)
    W, U, bi, bh = u(r, I, 4 * H, sc=0.2), u(r, H, 4 * H, sc=0.15), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    xd = torch.from_numpy(x).cuda()
    base = lstm.apply_device(xd).clone()
    o, f = NL.recurrent_apply_device_frag3(lstm, x_f3=NL.frag3_pack_device(xd), batch=B, want_f32=True, want_f3=True)
    assert torch.equal(o, base) and torch.equal(NL.frag3_unpack_device(f, B, T, H), base)
    np.testing.assert_allclose(base.cpu().numpy(), O.lstm(x, W, U, bi, bh, v2=True), rtol=1e-5, atol=1e-5)
    lstm.destroy()


@pytest.mark.parametrize("B,T,K,N,act", [
    (64, 9, 512, 1000, None),          # the stack's TimeDistributedDense: 256-wide tiles, N padded to 1024
    (70, 5, 256, 128, "relu"),         # 128-wide tiles, ragged row blocks
    (3, 4, 64, 384, "sigmoid"),
    (130, 3, 40, 96, None),            # N_p = 96: not taken by the register-direct kernel -> unpack + the LDS-staged GEMM
    (33, 2, 48, 130, "tanh"),          # N % 4 != 0: fallback
    (40, 6, 128, 256, "softmax"),
])
def test_dense_with_a_frag3_input_equals_the_f32_call_bit_for_bit(gpu, B, T, K, N, act):
    import torch
    r = rng(B + T + K + N)
    x = u(r, B, T, K)
    W, b = u(r, K, N, sc=K ** -0.5), u(r, N, sc=0.1)
    a = None
    if act == "softmax":
        a = NL.Activation("softmax", N // 64, vector_size=64)
    elif act:
        a = NL.Activation(act, N, a=1.0)
    tdd = NL.TimeDistributedDense(T, K, N, act=a)
    tdd.set_weights(W, b)
    xd = torch.from_numpy(x).cuda()
    base = tdd.apply_device(xd).clone()
    got = NL.tdd_apply_device_frag3(tdd, NL.frag3_pack_device(xd), B)
    assert torch.equal(got, base)
    kind = {None: O.ACT_NONE, "relu": O.ACT_RELU, "sigmoid": O.ACT_SIGMOID, "tanh": O.ACT_TANH, "softmax": O.ACT_SOFTMAX}[act]
    ref = O.time_distributed_dense(x, W, b, act=kind, **({"softmax_vector_size": 64, "act_size": N // 64} if act == "softmax" else {}))
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    capi.set_option("dense_frag3", 0)         # the fallback route on its own
    assert torch.equal(NL.tdd_apply_device_frag3(tdd, NL.frag3_pack_device(xd), B), base)
    capi.set_option("dense_frag3", "auto")
    tdd.destroy()
    if a:
        a.destroy()


@pytest.mark.parametrize("B,I,H,T,N", [(64, 128, 512, 10, 1000), (33, 40, 128, 7, 256), (130, 128, 256, 5, 96), (7, 24, 40, 6, 64),
                                       (106, 83, 300, 8, 4), (184, 130, 116, 3, 192)])      # H % 16 != 0: ceil(H / 16) k steps in the frag3 scratch
def test_fused_lstm_tdd_equals_the_two_calls_and_the_oracle(gpu, B, I, H, T, N):
    import torch
    r = rng(B + I + H + T + N)
    x = u(r, B, T, I)
    W, U, bi, bh = u(r, I, 4 * H, sc=I ** -0.5), u(r, H, 4 * H, sc=H ** -0.5), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1)
    Wd, bd = u(r, H, N, sc=H ** -0.5), u(r, N, sc=0.1)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    tdd = NL.TimeDistributedDense(T, H, N)
    tdd.set_weights(Wd, bd)
    xd = torch.from_numpy(x).cuda()
    two = tdd.apply_device(lstm.apply_device(xd)).clone()
    capi.set_option("dense_f16x2", 0)          # the frag3 route of the fused call (its default route, FRAG2H, is not bit-identical: tests/test_gpu_frag2h.py)
    one = NL.lstm_tdd_apply_device(lstm, tdd, xd)
    assert torch.equal(one, two)
    capi.set_option("rec_xf", 1)
    assert torch.equal(NL.lstm_tdd_apply_device(lstm, tdd, xd), two)
    capi.set_option("rec_xf", "auto")
    capi.set_option("dense_f16x2", "auto")
    ref = O.time_distributed_dense(O.lstm(x, W, U, bi, bh, v2=True), Wd, bd)
    np.testing.assert_allclose(one.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
    # the fused call leaves the handles as it found them (round 5's soak: for H % 16 != 0 its frag3 scratch was one k step per row block short,
    # and the pack pass wrote over what lay behind it -- the NEXT calls on the handle were wrong)
    assert torch.equal(tdd.apply_device(lstm.apply_device(xd)), two)
    assert capi.load().nntk_hip_device_status() == 0
    lstm.destroy(); tdd.destroy()


def test_kernel_plan_names_the_kernel_family_and_the_reason(gpu):
    """GRUKernelPlan / LSTMKernelPlan: the family a layer's batch forms take is a property of the layer, and the query says why a layer
    misses the register-resident kernels (make the perf cliff visible)."""
    import torch
    L = capi.load()
    r = rng(3)
    cases = [("lstm", 128, 512, "lstm_rr_kernel<8,2>"), ("gru", 256, 256, "gru_fk_kernel<16,16,4>"), ("gru", 128, 256, "gru_rr_kernel<4,2>"), ("lstm", 100, 128, "frag3"),
             ("lstm", 24, 40, "H % 16"), ("gru", 300, 512, "wider"), ("lstm", 8, 640, "outside")]
    for cell, I, H, want in cases:
        layer = NL.LSTM(I, H, True, 4, v2=True) if cell == "lstm" else NL.GRU(I, H, True, 4)
        G = 4 if cell == "lstm" else 3
        layer.set_weights(u(r, I, G * H, sc=0.1), u(r, H, G * H, sc=0.1), u(r, G * H, sc=0.1), u(r, G * H, sc=0.1))
        plan = (L.LSTMKernelPlan if cell == "lstm" else L.GRUKernelPlan)(layer.h).decode()
        assert want in plan, plan
        layer.apply_device(torch.from_numpy(u(r, 3, 4, I)).cuda())
        ran = L.nntk_hip_last_recurrent_kernel().decode()
        assert ran.startswith((cell + "_rr_kernel", cell + "_fk_kernel")) == ("_rr_kernel" in plan or "_fk_kernel" in plan), (plan, ran)
        if "_kernel<" in want:
            assert ran == want, (ran, want)
        layer.destroy()


@pytest.mark.parametrize("bad", [np.inf, -np.inf, 3.4e38, 1e-40])
def test_dense_frag3_weights_the_split_cannot_hold_take_the_exact_gemm(gpu, bad):
    """nntk_shim_dense_frag3 always ran the split-bf16 GEMM, so a Dense weight the split cannot represent (hi = inf, rest =
    inf - inf) gave NaN through TimeDistributedDenseApplyDeviceFrag3 / LSTMTimeDistributedDenseApplyDevice while the f32 call, which
    keeps the exact-f32 kernel for such a block, stayed finite.  The frag3 route now makes the same decision: same bits as the f32 call,
    finite wherever the oracle is."""
    import torch
    r = rng(77)
    B, T, K, N = 70, 4, 128, 256
    x = u(r, B, T, K)
    W, b = u(r, K, N, sc=K ** -0.5), u(r, N, sc=0.1)
    W[5, 9] = bad
    tdd = NL.TimeDistributedDense(T, K, N)
    tdd.set_weights(W, b)
    xd = torch.from_numpy(x).cuda()
    base = tdd.apply_device(xd).clone()
    got = NL.tdd_apply_device_frag3(tdd, NL.frag3_pack_device(xd), B)
    assert torch.equal(got, base)
    capi.set_option("gemm_split_bf16", 0)
    exact = tdd.apply_device(xd).clone()
    capi.set_option("gemm_split_bf16", "auto")
    assert torch.equal(base, exact)
    with np.errstate(all="ignore"):
        ref = O.time_distributed_dense(x, W, b)
    fin = np.isfinite(ref)
    g = got.cpu().numpy()
    assert np.array_equal(np.isfinite(g), fin)
    np.testing.assert_allclose(g[fin], ref[fin], rtol=1e-5, atol=1e-5)
    tdd.destroy()
