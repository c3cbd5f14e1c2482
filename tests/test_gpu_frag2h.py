"""GPU tests of the FRAG2H path (csrc/hip/frag3.hip, the output wave of recurrent_rr.hip): a bounded activation tensor (|x| < 2) as two f16
images of x * 2^15, and the dense GEMM that sums three products per k step on it (W as two f16 images of W * 2^q).

Reference semantics: layers/lstm.c:185-239 (|h| = |o tanh(c)| < 1 with the standard activations), layers/dense.c:122-133,
layers/time_distributed_dense.c:52-58.  Unlike frag3 the form is not exact -- operands are rounded to 2^-23 relative at worst (one f32 ulp) -- so these tests state
tolerances: against the oracle (the reference's f32 accumulation order), against an f64 contraction of the same f32 operands, and against
the frag3 / f32 routes of the same call.
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


def decode_frag2h(buf, B, T, C):
    """Host model of the documented layout: [T][2 ceil(B/64)][ceil(C/16)][2 images] blocks of 1 KB; lane 32 kh + n of a block holds channels
    16 ks + 8 kh .. + 7 of batch row 32 ht + n as 8 consecutive f16 of x * 2^15."""
    NHT, NKS = (B + 63) // 64 * 2, (C + 15) // 16
    raw = buf.view(np.float16).reshape(T, NHT, NKS, 2, 2, 32, 8).astype(np.float64)        # [t][ht][ks][m][kh][n][q]
    val = (raw[:, :, :, 0] + raw[:, :, :, 1]) / 32768.0
    return val.transpose(1, 4, 0, 2, 3, 5).reshape(NHT * 32, T, NKS * 16), raw


@pytest.mark.parametrize("B,T,C", [(70, 5, 40), (1, 3, 257), (130, 2, 128), (33, 7, 100)])
def test_frag2h_pack_layout_and_rounding(gpu, B, T, C):
    import torch
    r = rng(B + T + C)
    x = u(r, B, T, C) * np.float32(10.0) ** r.integers(-7, 1, (B, T, C)).astype(np.float32)       # |x| < 1, down to 1e-7
    x[0, 0, :4] = [1.0, -1.0, 1.9990234375, 0.0]
    xd = torch.from_numpy(x).cuda()
    h2 = NL.frag2h_pack_device(xd)
    assert h2.numel() == capi.load().nntk_frag2h_floats(B, T, C)
    full, raw = decode_frag2h(h2.cpu().numpy(), B, T, C)
    err = np.abs(full[:B, :, :C] - x.astype(np.float64))
    assert (err <= np.abs(x) * 2.0 ** -23 + 2.0 ** -40).all()       # hi + lo is the value to one f32 ulp at worst (absolute 2^-40 in f16's subnormals)
    assert np.sqrt(((err / np.maximum(np.abs(x), 1e-30))[np.abs(x) > 1e-4] ** 2).mean()) < 2.0 ** -24          # (rms: a third of an ulp)
    assert np.isfinite(raw).all()
    assert not full[B:].any() and not full[:, :, C:].any()          # padding rows and channels are zeros
    back = NL.frag2h_unpack_device(h2, B, T, C).cpu().numpy()
    assert (np.abs(back.astype(np.float64) - x) <= np.abs(x) * 2.0 ** -23 + 2.0 ** -40).all()


@pytest.mark.parametrize("B,T,K,N,act", [
    (40, 7, 512, 1000, None),          # the stack's dense layer: 256-wide tiles, N padded to 1024
    (70, 5, 256, 128, "relu"),         # 128-wide tiles, ragged row blocks
    (3, 4, 64, 384, "sigmoid"),
    (130, 3, 40, 96, None),            # N_p = 96: not taken by the register-direct kernel -> unpack + the LDS-staged GEMM
    (40, 6, 128, 256, "softmax"),
])
def test_dense_with_a_frag2h_input_against_f64_the_oracle_and_the_frag3_route(gpu, B, T, K, N, act):
    import torch
    r = rng(B + T + K + N)
    x = u(r, B, T, K) ** 3                                            # an LSTM output: |x| < 1, most of the mass near 0
    W, b = u(r, K, N, sc=K ** -0.5), u(r, N, sc=0.1)
    a = None
    if act == "softmax":
        a = NL.Activation("softmax", N // 64, vector_size=64)
    elif act:
        a = NL.Activation(act, N, a=1.0)
    tdd = NL.TimeDistributedDense(T, K, N, act=a)
    tdd.set_weights(W, b)
    xd = torch.from_numpy(x).cuda()
    got = NL.tdd_apply_device_frag2h(tdd, NL.frag2h_pack_device(xd), B).cpu().numpy()
    f3 = NL.tdd_apply_device_frag3(tdd, NL.frag3_pack_device(xd), B).cpu().numpy()
    kind = {None: O.ACT_NONE, "relu": O.ACT_RELU, "sigmoid": O.ACT_SIGMOID, "tanh": O.ACT_TANH, "softmax": O.ACT_SOFTMAX}[act]
    ref = O.time_distributed_dense(x, W, b, act=kind, **({"softmax_vector_size": 64, "act_size": N // 64} if act == "softmax" else {}))
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    if act is None:
        # the contraction itself against f64 on the same f32 operands: not worse than the six-product route (whose operands are exact)
        z = x.astype(np.float64).reshape(B * T, K) @ W.astype(np.float64) + b.astype(np.float64)
        e2, e3 = np.abs(got.reshape(B * T, N) - z), np.abs(f3.reshape(B * T, N) - z)
        print("dense %dx%dx%d: max / rms error vs f64: frag2h %.2e / %.2e, frag3 %.2e / %.2e"
              % (B * T, K, N, e2.max(), np.sqrt((e2 ** 2).mean()), e3.max(), np.sqrt((e3 ** 2).mean())))
        assert e2.max() < 2e-6 and np.sqrt((e2 ** 2).mean()) <= 1.25 * np.sqrt((e3 ** 2).mean()) + 1e-9
    assert np.abs(got - f3).max() < 3e-6
    tdd.destroy()
    if a:
        a.destroy()


@pytest.mark.parametrize("scale", [1e-20, 3e4, 1e18])
def test_dense_frag2h_weight_scale_follows_the_weights_magnitude(gpu, scale):
    """The weights' power-of-two scale is chosen from max |W| at upload: tiny and huge (finite) weight blocks keep their relative accuracy."""
    import torch
    r = rng(5)
    B, T, K, N = 64, 3, 128, 256
    x = u(r, B, T, K)
    W, b = (u(r, K, N) * np.float32(scale)).astype(np.float32), np.zeros(N, np.float32)
    tdd = NL.TimeDistributedDense(T, K, N)
    tdd.set_weights(W, b)
    xd = torch.from_numpy(x).cuda()
    got = NL.tdd_apply_device_frag2h(tdd, NL.frag2h_pack_device(xd), B).cpu().numpy().astype(np.float64)
    z = (x.astype(np.float64).reshape(B * T, K) @ W.astype(np.float64)).reshape(B, T, N)
    assert np.isfinite(got).all()
    assert np.abs(got - z).max() <= 2e-6 * np.abs(z).max()
    tdd.destroy()


@pytest.mark.parametrize("B,I,H,T", [
    (64, 128, 512, 12),      # the stack's LSTM shape: the HF instantiation (and, with rec_hf = 0, lstm_rr_kernel<8,2>'s output wave)
    (130, 64, 384, 40),      # KH = 8 / KX = 1, ragged tiles, 40 steps of recurrence
    (64, 128, 512, 1),       # a single step: the peeled prologue / drain only
    (33, 100, 320, 2),       # two steps, in % 8 != 0, H between the compiled depths
    (70, 256, 512, 9),       # in > 128 at H = 512: only the HF instantiation holds W's images next to U (KX = 4); rec_hf = 0: the exact kernels
    (64, 200, 384, 5),
    (33, 40, 128, 9),        # KH = 4 (pending-pattern hand-off), ragged second half-tile
    (130, 100, 256, 7),      # in % 8 != 0: frag3 input form inside the call
    (65, 256, 256, 6),       # the full-K family's shape: f32 scratch, then the pack pass
    (7, 24, 40, 6),          # no register-resident kernel at all
])
def test_lstm_frag2h_output_is_the_pack_of_the_f32_output(gpu, B, I, H, T):
    """The form is a function of the f32 value, so whoever writes it -- the rr kernel's output wave or the pack pass behind another kernel -- writes
    the same bits for the rows of the batch."""
    import torch
    r = rng(B * 3 + I + H + T)
    x = u(r, B, T, I)
    W, U, bi, bh = u(r, I, 4 * H, sc=I ** -0.5), u(r, H, 4 * H, sc=H ** -0.5), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    xd = torch.from_numpy(x).cuda()
    base = lstm.apply_device(xd).clone()
    want = NL.frag2h_unpack_device(NL.frag2h_pack_device(base), B, T, H)
    if H > 256:
        # H > 256: the HF instantiation -- the recurrence itself on two f16 images of h (three products per k step), its hand-off IS the tensor.
        # Another contraction than the bf16 x 3 kernel's: same tolerance against the oracle, not the same bits
        got = NL.frag2h_unpack_device(NL.lstm_apply_device_frag2h(lstm, x=xd), B, T, H)
        kern = capi.load().nntk_hip_last_recurrent_kernel().decode()
        assert kern.startswith("lstm_rr_kernel<8,") and kern.endswith(",hf>"), kern
        got3 = NL.frag2h_unpack_device(NL.lstm_apply_device_frag2h(lstm, x_f3=NL.frag3_pack_device(xd), batch=B), B, T, H)
        assert torch.equal(got3, got)                  # the form of x does not matter, and a second run repeats
        d = float((got - base).abs().max())
        print("LSTM-%d T=%d: HF kernel vs bf16 x 3 kernel %.2e" % (H, T, d))
        assert d < 2e-6
        np.testing.assert_allclose(got.cpu().numpy(), O.lstm(x, W, U, bi, bh, v2=True), rtol=2e-5, atol=2e-5)
        capi.set_option("rec_hf", 0)                   # ... from here on the bf16 x 3 kernel, whose output wave writes the form
    got = NL.frag2h_unpack_device(NL.lstm_apply_device_frag2h(lstm, x=xd), B, T, H)
    assert torch.equal(got, want)
    got3 = NL.frag2h_unpack_device(NL.lstm_apply_device_frag2h(lstm, x_f3=NL.frag3_pack_device(xd), batch=B), B, T, H)
    assert torch.equal(got3, want)
    capi.set_option("rec_hf", "auto")
    assert float((got - base).abs().max()) <= 2.0 ** -23
    ref = O.lstm(x, W, U, bi, bh, v2=True)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
    assert capi.load().nntk_hip_device_status() == 0
    lstm.destroy()


def test_lstm_frag2h_needs_the_standard_activations(gpu):
    """An output activation other than tanh leaves |h| unbounded: the form (|x| < 2) is refused, and the fused call takes the frag3 route."""
    import ctypes as C
    import torch
    L = capi.load()
    r = rng(9)
    B, I, H, T, N = 8, 16, 64, 3, 128
    acts = L.LSTMActivationsCreate(L.ActivationFunctionCreateSigmoid(H), L.ActivationFunctionCreateSigmoid(H),
                                   L.ActivationFunctionCreateTanh(H), L.ActivationFunctionCreateSigmoid(H),
                                   L.ActivationFunctionCreateReLU(H, C.c_float(40.0)))
    lstm = NL.LSTM(I, H, True, T, v2=True, acts=acts)
    lstm.set_weights(u(r, I, 4 * H, sc=0.2), u(r, H, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1))
    xd = torch.from_numpy(u(r, B, T, I)).cuda()
    with pytest.raises(capi.NNTKError):
        NL.lstm_apply_device_frag2h(lstm, x=xd)
    tdd = NL.TimeDistributedDense(T, H, N)
    tdd.set_weights(u(r, H, N, sc=0.1), u(r, N, sc=0.1))
    assert torch.equal(NL.lstm_tdd_apply_device(lstm, tdd, xd), tdd.apply_device(lstm.apply_device(xd)))
    lstm.destroy(); tdd.destroy()


@pytest.mark.parametrize("B,I,H,T,N", [(64, 128, 512, 10, 1000), (33, 40, 128, 7, 256), (130, 128, 256, 5, 96), (7, 24, 40, 6, 64)])
def test_fused_lstm_tdd_takes_the_frag2h_route_by_default(gpu, B, I, H, T, N):
    """LSTMTimeDistributedDenseApplyDevice: the default route (FRAG2H where it applies) against the oracle and against the two f32 calls;
    option dense_f16x2 = 0 is the frag3 route, which equals the two calls bit for bit."""
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
    one = NL.lstm_tdd_apply_device(lstm, tdd, xd).clone()
    assert torch.equal(NL.lstm_tdd_apply_device(lstm, tdd, xd), one)          # repeatable
    d = float((one - two).abs().max())
    assert d < 3e-6, d
    if N_takes(N):
        assert d > 0.0                                                        # (it IS another contraction: the default route was taken)
    ref = O.time_distributed_dense(O.lstm(x, W, U, bi, bh, v2=True), Wd, bd)
    np.testing.assert_allclose(one.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
    capi.set_option("dense_f16x2", 0)
    assert torch.equal(NL.lstm_tdd_apply_device(lstm, tdd, xd), two)
    capi.set_option("dense_f16x2", "auto")
    assert capi.load().nntk_hip_device_status() == 0
    lstm.destroy(); tdd.destroy()


def N_takes(N):
    return ((N + 31) // 32 * 32) % 128 == 0 and N % 4 == 0


@pytest.mark.parametrize("bad", [np.inf, 3.4e38])
def test_fused_call_with_weights_the_f16_form_cannot_hold_takes_the_frag3_route(gpu, bad):
    import torch
    r = rng(78)
    B, I, H, T, N = 40, 32, 64, 4, 256
    x = u(r, B, T, I)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(u(r, I, 4 * H, sc=I ** -0.5), u(r, H, 4 * H, sc=H ** -0.5), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1))
    Wd, bd = u(r, H, N, sc=H ** -0.5), u(r, N, sc=0.1)
    Wd[5, 9] = bad
    tdd = NL.TimeDistributedDense(T, H, N)
    tdd.set_weights(Wd, bd)
    xd = torch.from_numpy(x).cuda()
    two = tdd.apply_device(lstm.apply_device(xd)).clone()
    one = NL.lstm_tdd_apply_device(lstm, tdd, xd)
    assert torch.equal(one, two)
    lstm.destroy(); tdd.destroy()


def test_achieved_error_lstm_hf_512_T996_and_full_grid_repeat(gpu):
    """The stack's LSTM(128 -> 512, v2) over 996 steps on the HF instantiation (recurrence on two f16 images of h): the deviation from the
    oracle (libm gates, scalar k order) and from torch float64 as NUMBERS, asserted at the bar of the bf16 x 3 kernel's own test
    (tests/test_gpu_lstm_rr.py); and, at the bench's own launch (512 rows: all 256 workgroups), a second run and a 64-row shard equal to the
    first run bit for bit over the whole batch -- the race detector for the flag protocol's counted drains with two stores per publication."""
    import torch
    r = rng(502)
    B, I, H, T = 512, 128, 512, 996
    uw = lambda fan, *s: r.uniform(-fan ** -0.5, fan ** -0.5, s).astype(np.float32)
    W, U, bi, bh = uw(I, I, 4 * H), uw(H, H, 4 * H), uw(H, 4 * H), uw(H, 4 * H)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    xd = torch.randn(B, T, I, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    h2 = NL.lstm_apply_device_frag2h(lstm, x=xd)
    assert capi.load().nntk_hip_last_recurrent_kernel().decode() == "lstm_rr_kernel<8,2,hf>"
    got = NL.frag2h_unpack_device(h2, B, T, H)
    again = NL.frag2h_unpack_device(NL.lstm_apply_device_frag2h(lstm, x=xd), B, T, H)
    assert torch.equal(again, got)
    del again
    shard = NL.frag2h_unpack_device(NL.lstm_apply_device_frag2h(lstm, x=xd[448:].contiguous()), 64, T, H)
    assert torch.equal(shard, got[448:])
    del shard
    rows = [0, 31, 32, 63, 511]
    x = xd[rows].cpu().numpy()
    ref = O.lstm(x, W, U, bi, bh, v2=True)
    ref = ref[0] if isinstance(ref, tuple) else ref
    import torch as t
    m = t.nn.LSTM(I, H, batch_first=True).double()
    with t.no_grad():
        m.weight_ih_l0.copy_(t.tensor(W).double().T); m.weight_hh_l0.copy_(t.tensor(U).double().T)
        m.bias_ih_l0.copy_(t.tensor(bi).double()); m.bias_hh_l0.copy_(t.tensor(bh).double())
        r64 = m(t.tensor(x).double())[0].numpy()
    g = got[rows].cpu().numpy()
    base = lstm.apply_device(xd[:64].contiguous())[[0, 31, 32, 63]].cpu().numpy()       # the bf16 x 3 kernel on the same rows
    e_or, e_64 = float(np.abs(g - ref).max()), float(np.abs(g - r64).max())
    b_64 = float(np.abs(base - r64[:4]).max())
    print("lstm HF LSTM(128->512, v2) T=996: max abs err vs oracle %.2e, vs torch float64 %.2e (bf16 x 3 kernel vs float64 %.2e; oracle vs float64 %.2e)"
          % (e_or, e_64, b_64, float(np.abs(ref - r64).max())))
    assert e_or < 3e-6 and e_64 < 3e-6
    assert capi.load().nntk_hip_device_status() == 0
    lstm.destroy()
