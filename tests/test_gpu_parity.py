"""GPU parity: every hot-path layer, called through the C boundary
(libnntoolkitcore_hip.so), against the CPU oracle on the same seeded inputs.

Tolerances (fp32 everywhere; SURVEY 4 / BASELINE.md 2): the oracle accumulates
left-to-right in fp32, the kernels accumulate in MFMA k-order, so results differ by
rounding only.  ATOL 1e-5 / RTOL 1e-5 for single layers, 1e-4 for T=1000 recurrences
and the chained stack (hardware exp/tanh rounding compounds over time).
"""
import numpy as np
import pytest

import oracle as O
from nntoolkitcore_amd import capi, layers as NL

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-5, 1e-5


def rng(seed):
    return np.random.default_rng(seed)


def u(r, *shape, sc=1.0):
    return r.uniform(-sc, sc, shape).astype(np.float32)


def close(a, b, atol=ATOL, rtol=RTOL):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.isfinite(a).all()
    err = np.abs(a - b)
    bad = err > atol + rtol * np.abs(b)
    assert not bad.any(), "max abs err %.3e (rel %.3e) at %d/%d elements" % (
        err.max(), (err / (np.abs(b) + 1e-30)).max(), bad.sum(), bad.size)


# ------------------------------------------------------------------ conv1d ---

@pytest.mark.parametrize("cin,cout,k,stride,T", [
    (1, 16, 9, 1, 16000),      # BASELINE config 1 shape (VALU path: K = 9)
    (40, 128, 5, 1, 1000),     # config 3 shape, one utterance (MFMA 128x128 tile)
    (3, 4, 5, 2, 23),          # stride 2, tiny, ragged
    (40, 64, 5, 1, 300),       # BN = 64 tile
    (33, 32, 3, 1, 130),       # odd Cin (padded K), BN = 32 tile, partial second x tile
    (7, 100, 4, 1, 257),       # Cout not a multiple of 32
    (257, 128, 5, 1, 200),     # config 5 conv shape: many channel chunks
    (16, 48, 3, 2, 99),        # stride 2 with enough K for the MFMA path selection logic
    (5, 7, 1, 1, 11),          # k = 1
])
def test_conv1d_single_sequence(gpu, cin, cout, k, stride, T):
    r = rng(cin * 1000 + cout)
    x, W, b = u(r, T, cin), u(r, cout, cin, k, sc=(cin * k) ** -0.5), u(r, cout, sc=0.2)
    conv = NL.Conv1d(cin, cout, k, stride, T)
    conv.set_weights(W, b)
    close(conv.apply(x), O.conv1d(x, W, b, stride))
    conv.destroy()


@pytest.mark.parametrize("cin,cout,k,stride,T", [
    (40, 128, 5, 1, 300),      # config 3 shape, 16-byte window loads
    (33, 32, 3, 1, 130),       # odd Cin: 4-byte window loads, ragged last chunk, BN = 32
    (257, 128, 5, 1, 200),     # stack conv: 17 channel chunks, 4-byte loads
    (64, 192, 1, 1, 257),      # dense GEMM, BN = 64 tile, partial row tile, two chunks per barrier
    (48, 96, 1, 1, 200),       # dense GEMM with an odd number of channel chunks, BN = 32
    (128, 1000, 1, 1, 140),    # TimeDistributedDense-like: Cout padded to 1024
])
def test_split_bf16_contraction_matches_oracle(gpu, cin, cout, k, stride, T):
    """Option gemm_split_bf16 (auto: on for conv / dense): the 3-way split-bf16 MFMA contraction (conv1d.hip).  It is an f32-accuracy
    contraction but not the exact k-ordered chain, so the check is the oracle tolerance, plus: it must differ from the
    exact path by no more than a few f32 roundings of the row's magnitude."""
    r = rng(cin * 7 + cout)
    B = 3
    x, W, b = u(r, B, T, cin), u(r, cout, cin, k, sc=(cin * k) ** -0.5), u(r, cout, sc=0.2)
    conv = NL.Conv1d(cin, cout, k, stride, T)
    conv.set_weights(W, b)
    capi.set_option("gemm_split_bf16", "0")
    exact = conv.apply(x)
    capi.set_option("gemm_split_bf16", "1")
    split = conv.apply(x)
    capi.set_option("gemm_split_bf16", "auto")
    assert np.array_equal(conv.apply(x), split)                 # auto = split for conv / dense
    ref = O.conv1d(x, W, b, stride)
    close(split, ref)
    assert not np.array_equal(split, exact)                     # the option really selected the other kernel
    assert np.abs(split - exact).max() < 4e-6
    conv.destroy()


@pytest.mark.parametrize("cin,cout,k", [(33, 32, 3), (257, 128, 5), (7, 64, 4), (5, 40, 1), (129, 96, 1), (18, 32, 2)])
@pytest.mark.parametrize("split", ["0", "1"])
def test_odd_channel_counts_take_16_byte_window_loads(gpu, cin, cout, k, split):
    """Rows whose length is not a multiple of 16 bytes are fetched with 16-byte loads all the same (conv_a4, default on);
    the pieces that run into the next row are masked.  Same bits as the 4-byte-load form, the very last row of the tensor
    included, and a NaN at the start of the FOLLOWING row must not reach a row that does not own it."""
    r = rng(cin * 31 + k)
    B, T = 2, 150
    x, W, b = u(r, B, T, cin), u(r, cout, cin, k, sc=(cin * k) ** -0.5), u(r, cout, sc=0.2)
    conv = NL.Conv1d(cin, cout, k, 1, T)
    conv.set_weights(W, b)
    capi.set_option("gemm_split_bf16", split)
    capi.set_option("conv_a4", "0")
    ref4 = conv.apply(x)
    capi.set_option("conv_a4", "1")
    got = conv.apply(x)
    np.testing.assert_array_equal(got, ref4)
    close(got, O.conv1d(x, W, b, 1))
    xn = x.copy()
    xn[1, 40, :3] = np.nan                                   # row 40 of sequence 1: windows of outputs 40-k+1 .. 40 own it
    gn = conv.apply(xn)
    own = np.zeros(gn.shape[:2], bool)
    own[1, max(0, 40 - k + 1):41] = True
    assert np.isnan(gn[own]).all() and np.isfinite(gn[~own]).all()
    np.testing.assert_array_equal(gn[~own], got[~own])
    conv.destroy()


@pytest.mark.parametrize("cin,cout,k,T", [(40, 128, 5, 300), (257, 128, 5, 200), (64, 192, 1, 257), (128, 1000, 1, 140), (33, 32, 3, 130)])
def test_gemm_option_matrix_is_bit_identical_where_promised(gpu, cin, cout, k, T):
    """conv_store (accumulator orientation / store width) and conv_a4 (window load width) change how the kernel moves
    data, never what it computes: for each contraction (exact f32, split bf16x3) all four combinations must agree bit
    for bit, fused BatchNorm + ReLU included; the two contractions agree to a few f32 roundings."""
    import torch
    r = rng(cin + cout + k)
    B = 3
    x, W, b = u(r, B, T, cin), u(r, cout, cin, k, sc=(cin * k) ** -0.5), u(r, cout, sc=0.2)
    g, be, mu, var = 1 + u(r, cout, sc=0.5), u(r, cout, sc=0.5), u(r, cout, sc=0.1), 1 + u(r, cout, sc=0.5)
    conv = NL.Conv1d(cin, cout, k, 1, T)
    conv.set_weights(W, b)
    Tc = conv.out_shape[0]
    bn, relu = NL.BatchNorm(cout, 1e-3, B * Tc), NL.Activation("relu", B * Tc * cout, 1.0)
    bn.set_weights(g, be, mu, var)
    xd = torch.from_numpy(x).cuda()
    res = {}
    for split in ("0", "1"):
        capi.set_option("gemm_split_bf16", split)
        outs = []
        for store in ("0", "1"):
            for a4 in ("0", "1"):
                capi.set_option("conv_store", store)
                capi.set_option("conv_a4", a4)
                outs.append((conv.apply_device(xd).cpu().numpy(), conv.apply_device(xd, bn=bn, act=relu).cpu().numpy()))
        for plain, fused in outs[1:]:
            np.testing.assert_array_equal(plain, outs[0][0])
            np.testing.assert_array_equal(fused, outs[0][1])
        res[split] = outs[0]
    assert np.abs(res["0"][0] - res["1"][0]).max() < 4e-6
    close(res["1"][1], O.activation(O.ACT_RELU, O.batch_norm(O.conv1d(x, W, b, 1), g, be, mu, var, 1e-3)))
    for o in (conv, bn, relu):
        o.destroy()


def test_conv1d_output_shorter_than_kernel_is_empty(gpu):
    conv = NL.Conv1d(2, 3, 5, 1, 4)        # output_size = 0 (conv_1d.c:84)
    assert conv.cfg.output_size == 0
    out = conv.apply(np.zeros((4, 2), np.float32))
    assert out.shape == (0, 3)
    conv.destroy()


def test_conv1d_batch_and_weight_edit_detection(gpu):
    r = rng(5)
    B, T, cin, cout, k = 5, 140, 40, 128, 5
    x, W, b = u(r, B, T, cin), u(r, cout, cin, k, sc=0.1), u(r, cout, sc=0.2)
    conv = NL.Conv1d(cin, cout, k, 1, T)
    # weights are zero-initialised like f_malloc (weights_private.c:18): output == 0
    assert np.all(conv.apply(x) == 0.0)
    conv.set_weights(W, b)                 # in-place edit, no notification (like a C caller's memcpy)
    close(conv.apply(x), O.conv1d(x, W, b, 1))
    conv.destroy()


def test_fused_conv_bn_relu_matches_chain_and_oracle(gpu):
    import torch
    r = rng(11)
    B, T, cin, cout, k = 4, 333, 40, 128, 5
    x, W, b = u(r, B, T, cin), u(r, cout, cin, k, sc=0.1), u(r, cout, sc=0.2)
    g, be, mu, var = 1 + u(r, cout, sc=0.5), u(r, cout, sc=0.5), u(r, cout, sc=0.1), 1 + u(r, cout, sc=0.5)
    conv = NL.Conv1d(cin, cout, k, 1, T)
    conv.set_weights(W, b)
    Tc = conv.out_shape[0]
    bn = NL.BatchNorm(cout, 1e-3, B * Tc)
    bn.set_weights(g, be, mu, var)
    relu = NL.Activation("relu", B * Tc * cout, 0.5)      # a is an OUTPUT scale (activation_default.c:123-129)
    xd = torch.from_numpy(x).cuda()
    fused = conv.apply_device(xd, bn=bn, act=relu).cpu().numpy()
    chain = relu.apply_device(bn.apply_device(conv.apply_device(xd))).cpu().numpy()
    ref = O.activation(O.ACT_RELU, O.batch_norm(O.conv1d(x, W, b, 1), g, be, mu, var, 1e-3), relu_a=0.5)
    close(fused, ref)
    close(chain, ref)
    # the fused epilogue and the three-kernel chain round every operation alike: equal up to the rare last-bit case of
    # the epilogue's refined reciprocal quotient against the chain's IEEE division
    assert (fused != chain).mean() < 1e-4 and np.abs(fused - chain).max() <= 2.4e-7 * max(1.0, float(np.abs(chain).max()))
    for o in (conv, bn, relu):
        o.destroy()


# -------------------------------------------------------- batchnorm / acts ---

@pytest.mark.parametrize("C,rows", [(128, 996), (7, 13), (4, 1)])
def test_batch_norm(gpu, C, rows):
    r = rng(C)
    x = u(r, rows, C, sc=3)
    g, be, mu, var = 1 + u(r, C, sc=0.5), u(r, C, sc=0.5), u(r, C, sc=0.3), 1 + u(r, C, sc=0.9)
    bn = NL.BatchNorm(C, 1e-3, rows)
    # default weights are all zero, gamma included (batch_norm.c:79): 0/sqrt(eps)*0+0
    assert np.all(bn.apply(x) == O.batch_norm(x, *[np.zeros(C, np.float32)] * 4, 1e-3))
    bn.set_weights(g, be, mu, var)
    got, ref = bn.apply(x), O.batch_norm(x, g, be, mu, var, 1e-3)
    np.testing.assert_array_equal(got, ref)       # every operation separately rounded, as batch_norm.c:140-163
    bn.destroy()


@pytest.mark.parametrize("kind,okind", [("sigmoid", O.ACT_SIGMOID), ("tanh", O.ACT_TANH),
                                        ("identity", O.ACT_IDENTITY), ("relu", O.ACT_RELU)])
@pytest.mark.parametrize("n", [1, 257, 4096])
def test_elementwise_activations(gpu, kind, okind, n):
    x = u(rng(n), n, sc=6)
    act = NL.Activation(kind, n, a=0.25)
    close(act.apply(x), O.activation(okind, x, relu_a=0.25), atol=1e-6, rtol=1e-6)
    act.destroy()


def test_identity_activation_copies_exactly_its_created_size(gpu):
    """Pinned by the REAL reference (tests/golden/ref_probe.json "identity_size5_on_8"): ActivationFunctionApply on an
    identity handle created with size 5 copies 5 floats and leaves the rest of the output alone."""
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_probe.json")))
    L = capi.load()
    x = np.arange(1, 9, dtype=np.float32)
    y = np.full(8, -1, np.float32)
    h = L.ActivationFunctionCreateIdentity(5)
    L.ActivationFunctionApply(h, x.ctypes.data_as(capi.fp), y.ctypes.data_as(capi.fp))
    L.ActivationFunctionDestroy(h)
    assert y.tolist() == gold["identity_size5_on_8"]


def test_softmax_no_max_subtraction(gpu):
    x = u(rng(3), 6, 1000, sc=4)
    act = NL.Activation("softmax", 6, vector_size=1000)
    got = act.apply(x)
    close(got, O.activation(O.ACT_SOFTMAX, x, softmax_vector_size=1000), atol=1e-7, rtol=1e-5)
    # like the reference (activation_default.c:149-154) large logits overflow: exp(100) = inf -> nan/0
    big = np.full((6, 1000), 100.0, np.float32)
    assert np.isnan(act.apply(big)).all()
    act.destroy()


def test_custom_activation_is_called_on_host_and_rejected_in_layers(gpu):
    import ctypes as C
    L = capi.load()
    calls = []

    @capi.ACT_IMPL_FN
    def twice(impl, inp, out, n):
        calls.append(n)
        for i in range(n):
            out[i] = 2.0 * inp[i]

    h = L.ActivationFunctionCreate(4, None, None, C.cast(twice, C.c_void_p), None, None)
    x = np.arange(4, dtype=np.float32)
    out = np.empty(4, np.float32)
    L.ActivationFunctionApply(h, x.ctypes.data_as(capi.fp), out.ctypes.data_as(capi.fp))
    assert calls == [4] and np.all(out == 2 * x)
    # a GRU configured with a host-callback gate cannot run on the device: -1, like a wrong-mode handle
    sig = L.ActivationFunctionCreateSigmoid(4)
    acts = L.GRUActivationsCreate(h, sig, sig)
    gru = NL.GRU(3, 4, True, 5, acts=acts)
    rc = L.GRUApplyInference(gru.h, np.zeros(15, np.float32).ctypes.data_as(capi.fp),
                             np.zeros(20, np.float32).ctypes.data_as(capi.fp))
    assert rc == -1 and "activation" in capi.last_error()
    L.GRUDestroy(gru.h)
    L.ActivationFunctionDestroy(h)
    L.ActivationFunctionDestroy(sig)


# --------------------------------------------------------------- recurrent ---

def gru_weights(r, I, H):
    return u(r, I, 3 * H, sc=I ** -0.5), u(r, H, 3 * H, sc=H ** -0.5), u(r, 3 * H, sc=0.1), u(r, 3 * H, sc=0.1)


def lstm_weights(r, I, H):
    return u(r, I, 4 * H, sc=I ** -0.5), u(r, H, 4 * H, sc=H ** -0.5), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1)


@pytest.mark.parametrize("I,H,T", [(5, 7, 11), (128, 256, 40), (33, 48, 17), (16, 20, 9)])
@pytest.mark.parametrize("seq", [True, False])
def test_gru_single_sequence_stateful(gpu, I, H, T, seq):
    r = rng(I + H)
    W, U, bi, bh = gru_weights(r, I, H)
    x1, x2 = u(r, T, I), u(r, T, I)
    gru = NL.GRU(I, H, seq, T)
    gru.set_weights(W, U, bi, bh)
    o1, h1 = O.gru(x1, W, U, bi, bh, return_sequences=seq)
    o2, h2 = O.gru(x2, W, U, bi, bh, h0=h1, return_sequences=seq)
    close(gru.apply(x1), o1)
    close(gru.state(), h1)
    close(gru.apply(x2), o2)          # second call continues from the carried state (gru.c:201)
    gru.reset_state()
    close(gru.apply(x1), o1)
    gru.destroy()


@pytest.mark.parametrize("B,I,H,T,seq", [(3, 5, 7, 11, True), (70, 128, 256, 25, True), (65, 40, 32, 12, False),
                                         (1, 8, 16, 5, True)])
def test_gru_batch_zero_state(gpu, B, I, H, T, seq):
    r = rng(B + H)
    W, U, bi, bh = gru_weights(r, I, H)
    x = u(r, B, T, I)
    gru = NL.GRU(I, H, seq, T)
    gru.set_weights(W, U, bi, bh)
    close(gru.apply(x), O.gru(x, W, U, bi, bh, return_sequences=seq))
    gru.destroy()


@pytest.mark.parametrize("v2", [True, False])
@pytest.mark.parametrize("I,H,T,seq", [(5, 7, 11, True), (128, 512, 20, True), (24, 40, 13, False)])
def test_lstm_single_sequence_stateful(gpu, v2, I, H, T, seq):
    r = rng(I * H + int(v2))
    W, U, bi, bh = lstm_weights(r, I, H)
    x1, x2 = u(r, T, I), u(r, T, I)
    lstm = NL.LSTM(I, H, seq, T, v2=v2)
    lstm.set_weights(W, U, bi, bh)
    o1, h1, c1 = O.lstm(x1, W, U, bi, bh, return_sequences=seq, v2=v2)
    o2, h2, c2 = O.lstm(x2, W, U, bi, bh, h0=h1, c0=c1, return_sequences=seq, v2=v2)
    close(lstm.apply(x1), o1)
    h, c = lstm.state()
    close(h, h1)
    close(c, c1)
    close(lstm.apply(x2), o2)
    lstm.destroy()


@pytest.mark.parametrize("B,I,H,T,seq,v2", [(4, 5, 7, 11, True, True), (66, 128, 512, 12, True, True),
                                            (9, 20, 36, 10, False, False)])
def test_lstm_batch_zero_state(gpu, B, I, H, T, seq, v2):
    r = rng(B * 7 + H)
    W, U, bi, bh = lstm_weights(r, I, H)
    x = u(r, B, T, I)
    lstm = NL.LSTM(I, H, seq, T, v2=v2)
    lstm.set_weights(W, U, bi, bh)
    close(lstm.apply(x), O.lstm(x, W, U, bi, bh, return_sequences=seq, v2=v2))
    lstm.destroy()


def test_gru_nondefault_gate_activations(gpu):
    L = capi.load()
    I, H, T = 6, 10, 8
    r = rng(77)
    W, U, bi, bh = gru_weights(r, I, H)
    x = u(r, T, I)
    # GRUActivationsCreate argument order is (z, h, r) (gru.c:220-230)
    acts = L.GRUActivationsCreate(L.ActivationFunctionCreateTanh(H), L.ActivationFunctionCreateReLU(H, 1.0),
                                  L.ActivationFunctionCreateSigmoid(H))
    gru = NL.GRU(I, H, True, T, acts=acts)
    gru.set_weights(W, U, bi, bh)
    ref, _ = O.gru(x, W, U, bi, bh, acts=(O.ACT_TANH, O.ACT_RELU, O.ACT_SIGMOID))
    close(gru.apply(x), ref)
    gru.destroy()


def test_recurrent_long_sequence_T1000(gpu):
    """BASELINE config 4 geometry on a few utterances: 2-layer GRU(128->256->256), T=1000."""
    r = rng(1000)
    B, T, I, H = 2, 1000, 128, 256
    x = u(r, B, T, I)
    W1, U1, bi1, bh1 = gru_weights(r, I, H)
    W2, U2, bi2, bh2 = gru_weights(r, H, H)
    g1, g2 = NL.GRU(I, H, True, T), NL.GRU(H, H, True, T)
    g1.set_weights(W1, U1, bi1, bh1)
    g2.set_weights(W2, U2, bi2, bh2)
    got = g2.apply(g1.apply(x))
    ref = O.gru(O.gru(x, W1, U1, bi1, bh1), W2, U2, bi2, bh2)
    close(got, ref, atol=1e-4, rtol=1e-4)
    g1.destroy()
    g2.destroy()


# ------------------------------------------------------------------- dense ---

@pytest.mark.parametrize("ts,I,Ov,act", [(9, 5, 7, None), (40, 512, 1000, None), (13, 64, 96, "relu"),
                                         (6, 32, 50, "softmax"), (5, 20, 33, "sigmoid")])
def test_time_distributed_dense(gpu, ts, I, Ov, act):
    r = rng(ts * I)
    W, b, x = u(r, I, Ov, sc=I ** -0.5), u(r, Ov, sc=0.2), u(r, ts, I)
    a, okind, kw = None, O.ACT_NONE, {}
    if act == "softmax":
        a, okind, kw = NL.Activation("softmax", 1, vector_size=Ov), O.ACT_SOFTMAX, dict(softmax_vector_size=Ov)
    elif act is not None:
        a = NL.Activation(act, Ov, a=0.5)
        okind, kw = {"relu": O.ACT_RELU, "sigmoid": O.ACT_SIGMOID}[act], dict(relu_a=0.5)
    tdd = NL.TimeDistributedDense(ts, I, Ov, act=a)
    tdd.set_weights(W, b)
    close(tdd.apply(x), O.time_distributed_dense(x, W, b, act=okind, **kw))
    xb = u(r, 3, ts, I)
    close(tdd.apply(xb), O.time_distributed_dense(xb, W, b, act=okind, **kw))
    tdd.destroy()


def test_dense_single_vector(gpu):
    r = rng(8)
    W, b, x = u(r, 12, 5, sc=0.3), u(r, 5), u(r, 12)
    d = NL.Dense(12, 5)
    d.set_weights(W, b)
    close(d.apply(x), O.time_distributed_dense(x[None], W, b)[0])
    d.destroy()


# ------------------------------------------------------------- spectrogram ---

WIN = {"ones": "ones", "hann_window": "hann", "hamming_window": "hamming", "periodic_hann_window": "periodic_hann",
       "periodic_hamming_window": "periodic_hamming", "blackman_window": "blackman"}


@pytest.mark.parametrize("wname", list(WIN))
@pytest.mark.parametrize("mode", ["magnitude", "psd"])
def test_spectrogram_512_all_windows(gpu, wname, mode):
    r = rng(len(wname))
    x = (0.1 * r.standard_normal(16000)).astype(np.float32)
    sp = NL.Spectrogram(512, 400, 240, 16000, mode=mode, fs=16000, window_name=wname)
    assert sp.out_shape == (98, 257)
    ref = O.spectrogram(x, O.window(WIN[wname], 400), 512, 240, mode=mode, fs=16000)
    close(sp.apply(x), ref, atol=1e-6 * float(np.abs(ref).max()), rtol=2e-5)
    sp.destroy()


@pytest.mark.parametrize("nfft,win,nov,N,B", [(512, 400, 240, 16000, 5), (512, 512, 0, 5120, 2), (512, 400, 399, 900, 3),
                                              (256, 200, 120, 8000, 2), (64, 48, 16, 1000, 3), (60, 45, 15, 777, 2),
                                              (16, 16, 8, 40, 1),
                                              # every count of 64-sample blocks the 512-point kernel is specialised for
                                              (512, 40, 8, 2000, 2), (512, 64, 0, 1300, 1), (512, 100, 20, 3000, 3),
                                              (512, 192, 64, 2500, 2), (512, 256, 128, 3000, 2), (512, 300, 37, 4000, 2),
                                              (512, 384, 100, 5000, 2), (512, 385, 0, 5000, 3), (512, 448, 200, 6000, 2),
                                              (512, 449, 10, 6000, 2),
                                              # the mixed-radix path: 2^a 3^b 5^c (256 / 1024 / 2048 are ordinary speech settings)
                                              (1024, 800, 480, 16000, 3), (2048, 2048, 1024, 20000, 2), (4096, 3000, 0, 12288, 1),
                                              (480, 400, 240, 8000, 2), (100, 100, 50, 2000, 2), (250, 160, 80, 3000, 3),
                                              (4, 4, 0, 64, 1), (3, 3, 1, 50, 2), (1000, 999, 998, 1100, 1),
                                              # a prime factor above 5: the direct-DFT kernel
                                              (77, 70, 7, 1000, 2), (1022, 600, 100, 5000, 1)])
def test_spectrogram_batch_geometries(gpu, nfft, win, nov, N, B):
    r = rng(nfft + N)
    x = (0.1 * r.standard_normal((B, N))).astype(np.float32)
    sp = NL.Spectrogram(nfft, win, nov, N, fft_norm=0.5)
    ref = O.spectrogram(x, O.window("hann", win), nfft, nov, fft_norm=0.5)
    close(sp.apply(x), ref, atol=1e-6 * float(np.abs(ref).max()), rtol=2e-5)
    sp.destroy()


def test_spectrogram_zero_padding_is_exact_next_to_inf_and_nan(gpu):
    """The reference zero-pads each frame to nfft (spectrogram.c:120-121): a non-finite sample just outside a frame's
    window must not leak into that frame (the kernel clears out-of-window samples with an AND, it does not multiply
    them by a zero tap)."""
    N = 4000
    x = (0.1 * rng(3).standard_normal(N)).astype(np.float32)
    x[1000] = np.inf
    x[2500] = np.nan
    sp = NL.Spectrogram(512, 400, 240, N)
    got = sp.apply(x)
    ref = O.spectrogram(x, O.window("hann", 400), 512, 240)
    bad = ~np.isfinite(ref).all(axis=1)
    assert bad.sum() in (5, 6)                       # exactly the frames whose window holds sample 1000 or 2500
    # the kernel transforms frames 2p, 2p+1 as ONE complex FFT, so a non-finite frame takes its pair partner with it --
    # but nothing else: every other frame is finite and equal to the reference
    pair_bad = bad.copy()
    pair_bad[0::2] |= bad[1::2] if len(bad) % 2 == 0 else np.append(bad[1::2], False)
    pair_bad[1::2] |= bad[0::2][:len(bad[1::2])]
    got_bad = ~np.isfinite(got).all(axis=1)
    assert not (got_bad & ~pair_bad).any() and (got_bad | ~bad).all()
    close(got[~pair_bad], ref[~pair_bad], atol=1e-6 * float(np.abs(ref[~pair_bad]).max()), rtol=2e-5)
    sp.destroy()


def test_spectrogram_default_window_is_ones_and_scale_override(gpu):
    L = capi.load()
    x = (0.1 * rng(2).standard_normal(4000)).astype(np.float32)
    sp = NL.Spectrogram(512, 400, 240, 4000, window_name=None)      # spectrogram.c:96 installs `ones`
    ref = O.spectrogram(x, O.window("ones", 400), 512, 240)
    close(sp.apply(x), ref, atol=1e-6 * float(np.abs(ref).max()), rtol=2e-5)
    L.SpectrogramSetScaleFactor(sp.h, 2.0)                            # spectrogram.c:100
    ref2 = O.spectrogram(x, O.window("ones", 400), 512, 240, scale=2.0)
    close(sp.apply(x), ref2, atol=1e-6 * float(np.abs(ref2).max()), rtol=2e-5)
    sp.destroy()


def test_config2_at_its_real_size_and_specialisation(gpu):
    """BASELINE configs[1] exactly: B = 256 x 16000 samples, win = 400 (hann), noverlap = 240, nfft = 512, magnitude -- the 7-block
    instantiation the bench times (the Parseval test below runs the 8-block `ones` / PSD one).  Rows 0 / 128 / 255 against the oracle."""
    import torch
    B, N = 256, 16000
    r = rng(2)
    x = (0.1 * r.standard_normal((B, N))).astype(np.float32)
    sp = NL.Spectrogram(512, 400, 240, N)
    out = sp.apply_device(torch.from_numpy(x).cuda()).cpu().numpy()
    assert out.shape == (B, 98, 257)
    rows = [0, 128, 255]
    ref = O.spectrogram(x[rows], O.window("hann", 400), 512, 240)
    close(out[rows], ref, atol=1e-6 * float(np.abs(ref).max()), rtol=2e-5)
    sp.destroy()


def test_parseval_property_full_size(gpu):
    """Size-independent property at BASELINE config 2 size (256 x 16000): with a `ones`
    window of nfft samples, sum_k |X_k|^2 over the full spectrum = nfft * sum x^2 per frame."""
    import torch
    B, N = 256, 16000
    x = torch.randn(B, N, device="cuda") * 0.1
    sp = NL.Spectrogram(512, 512, 352, N, mode="psd", fs=1, window_name=None)     # step 160
    out = sp.apply_device(x)                                        # psd = |X|^2 * (2 or 1) / (fs * sum w^2)
    nts = sp.out_shape[0]
    frames = x.unfold(1, 512, 160)[:, :nts]
    energy = (frames.double() ** 2).sum(-1)                         # = (1/nfft) sum_k |X_k|^2 = sum over one-sided psd
    one_sided = out.double().sum(-1)
    torch.testing.assert_close(one_sided, energy, rtol=1e-4, atol=1e-6)
    sp.destroy()


# -------------------------------------------------------------- full stack ---

def test_full_stack_config5_geometry_small_batch(gpu):
    """Spectrogram -> Conv1d(257->128,k=5)+BN+ReLU -> LSTM(512) -> TDD(1000) on 2 utterances of 100 frames."""
    import torch
    r = rng(55)
    B, frames = 2, 100
    N = 240 + 160 * frames
    audio = (0.1 * r.standard_normal((B, N))).astype(np.float32)
    spec = NL.Spectrogram(512, 400, 240, N)
    T, F = spec.out_shape
    assert (T, F) == (frames, 257)
    conv = NL.Conv1d(F, 128, 5, 1, T)
    Tc = conv.out_shape[0]
    bn, relu = NL.BatchNorm(128, 1e-3, Tc), NL.Activation("relu", Tc * 128, 1.0)
    lstm, tdd = NL.LSTM(128, 512, True, Tc, v2=True), NL.TimeDistributedDense(Tc, 512, 1000)
    Wc, bc = u(r, 128, F, 5, sc=(F * 5) ** -0.5), u(r, 128, sc=0.1)
    g, be, mu, var = 1 + u(r, 128, sc=0.5), u(r, 128, sc=0.5), u(r, 128, sc=0.1), 1 + u(r, 128, sc=0.5)
    Wl, Ul, bi, bh = lstm_weights(r, 128, 512)
    Wd, bd = u(r, 512, 1000, sc=512 ** -0.5), u(r, 1000, sc=0.1)
    conv.set_weights(Wc, bc); bn.set_weights(g, be, mu, var); lstm.set_weights(Wl, Ul, bi, bh); tdd.set_weights(Wd, bd)
    xd = torch.from_numpy(audio).cuda()
    y = tdd.apply_device(lstm.apply_device(conv.apply_device(spec.apply_device(xd), bn=bn, act=relu))).cpu().numpy()
    rs = O.spectrogram(audio, O.window("hann", 400), 512, 240)
    rc = O.activation(O.ACT_RELU, O.batch_norm(O.conv1d(rs, Wc, bc, 1), g, be, mu, var, 1e-3))
    ry = O.time_distributed_dense(O.lstm(rc, Wl, Ul, bi, bh, v2=True), Wd, bd)
    close(y, ry, atol=1e-4, rtol=1e-4)
    for o in (spec, conv, bn, relu, lstm, tdd):
        o.destroy()


def test_sharding_is_bit_identical(gpu):
    """Utterances are independent: processing a batch in two shards gives bit-identical
    outputs to processing it whole (the multi-GPU correctness argument)."""
    import torch
    r = rng(99)
    B, T, I, H = 6, 30, 40, 64
    x = torch.from_numpy(u(r, B, T, I)).cuda()
    W, U, bi, bh = gru_weights(r, I, H)
    gru = NL.GRU(I, H, True, T)
    gru.set_weights(W, U, bi, bh)
    whole = gru.apply_device(x).clone()
    a = gru.apply_device(x[:3].contiguous()).clone()
    b = gru.apply_device(x[3:].contiguous()).clone()
    assert torch.equal(whole, torch.cat([a, b]))
    gru.destroy()


# ------------------------------------------------ BASELINE configs at full size ---
# Utterances are independent on this path, so a full-size batch is checked exactly on a few
# sampled utterances (first / middle / last: tile and shard edges) against the oracle, which
# only has to run those rows.

def test_full_size_config3_conv_bn_relu(gpu):
    import torch
    r = rng(303)
    B, T, cin, cout, k = 1024, 1000, 40, 128, 5
    x = torch.randn(B, T, cin, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    W, b = u(r, cout, cin, k, sc=(cin * k) ** -0.5), u(r, cout, sc=0.1)
    g, be, mu, var = 1 + u(r, cout, sc=0.5), u(r, cout, sc=0.5), u(r, cout, sc=0.1), 1 + u(r, cout, sc=0.5)
    conv = NL.Conv1d(cin, cout, k, 1, T)
    conv.set_weights(W, b)
    Tc = conv.out_shape[0]
    bn, relu = NL.BatchNorm(cout, 1e-3, Tc), NL.Activation("relu", Tc * cout, 1.0)
    bn.set_weights(g, be, mu, var)
    y = conv.apply_device(x, bn=bn, act=relu)
    assert y.shape == (B, 996, 128)
    for i in (0, 511, 1023):
        ref = O.activation(O.ACT_RELU, O.batch_norm(O.conv1d(x[i].cpu().numpy(), W, b, 1), g, be, mu, var, 1e-3))
        close(y[i].cpu().numpy(), ref)
    for o in (conv, bn, relu):
        o.destroy()


def test_full_size_config4_two_layer_gru(gpu):
    import torch
    r = rng(404)
    B, T, I, H = 1024, 1000, 128, 256
    x = torch.randn(B, T, I, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    W1, U1, bi1, bh1 = gru_weights(r, I, H)
    W2, U2, bi2, bh2 = gru_weights(r, H, H)
    g1, g2 = NL.GRU(I, H, True, T), NL.GRU(H, H, True, T)
    g1.set_weights(W1, U1, bi1, bh1)
    g2.set_weights(W2, U2, bi2, bh2)
    y = g2.apply_device(g1.apply_device(x))
    assert y.shape == (B, T, H)
    # the kernel behind bench.py's config-4 line: both layers in ONE persistent launch (gru2_persistent_kernel), at the
    # benchmark's own size -- 16 batch tiles x 16 column tiles = all 256 CUs, 1000 steps
    # ... and the stack call's default since round 3: two register-resident launches (the bench's config-4 kernels: gru_rr_kernel<4,2>
    # for the 128-wide layer 1, and since round 5 the full-K gru_fk_kernel<16,16,4> for the 256-wide layer 2)
    L = capi.load()
    yr = NL.gru_stack2_apply_device(g1, g2, x).clone()
    assert L.nntk_hip_last_recurrent_kernel().decode() == "gru_fk_kernel<16,16,4>" and torch.equal(yr, y)
    # race detector for the hand-off without flags (pending pattern, H <= 256) at the full grid of 256 workgroups: a second run and a
    # 512-row shard (another set of batch tiles, half the workgroups) equal the first run bit for bit over the whole batch
    assert torch.equal(NL.gru_stack2_apply_device(g1, g2, x), yr)
    assert torch.equal(NL.gru_stack2_apply_device(g1, g2, x[512:].contiguous()), yr[512:])
    capi.set_option("rec_fused2", 1)
    yf = NL.gru_stack2_apply_device(g1, g2, x)
    capi.set_option("rec_fused2", "auto")
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith("gru2_persistent_kernel")
    assert yf.shape == (B, T, H)
    assert L.nntk_hip_device_status() == 0
    for i in (0, 700, 1023):
        ref = O.gru(O.gru(x[i:i + 1].cpu().numpy(), W1, U1, bi1, bh1), W2, U2, bi2, bh2)[0]
        close(y[i].cpu().numpy(), ref, atol=1e-4, rtol=1e-4)
        close(yf[i].cpu().numpy(), ref, atol=1e-4, rtol=1e-4)
        e, er = float(np.abs(yf[i].cpu().numpy() - ref).max()), float(np.abs(y[i].cpu().numpy() - ref).max())
        print("2xGRU-256 B=1024 T=1000 row %d: max abs err vs oracle: fused exact kernel %.2e, register-resident pair %.2e"
              % (i, e, er))
        assert e < 3e-6 and er < 3e-6
    # the two forms differ in summation order only, over the WHOLE batch
    assert float((y - yf).abs().max()) < 1e-5
    g1.destroy()
    g2.destroy()


def test_full_size_config5_stack_one_gpu_shard(gpu):
    """The bench workload itself: 512 utterances x 1000 frames through the whole stack (the conv layer hands its output to the LSTM in frag3 form, the
    LSTM its output to the dense layer in FRAG2H form -- the default -- or in frag3 form).  Every stage of rows 0 and 511 against the oracle on
    both routes; and over the WHOLE batch: the f32 route (LSTMApplyDevice then TimeDistributedDenseApplyDevice) equal to the frag3 route bit
    for bit, the FRAG2H route within its stated rounding of them, and the exact kernels (rec_rr = 0, gemm_split_bf16 = 0) within the
    summation-order bound -- the race detector VERDICT r03 asked for."""
    import torch
    import bench
    w = bench.make_weights("stack", 3)
    wl = bench.Workload("stack", 512, 1000, w, torch, NL)
    assert wl.h2_route
    wl.step()
    torch.cuda.synchronize()
    y_h2 = wl.tdd_out.clone()
    h_h2 = NL.frag2h_unpack_device(wl.lstm_h2, 512, 996, 512)
    wl.step()
    torch.cuda.synchronize()
    assert torch.equal(y_h2, wl.tdd_out)                      # a second run of the default route: same bits
    wl.h2_route = False                                       # ... from here on the frag3 route (option dense_f16x2 = 0)
    wl.step()
    torch.cuda.synchronize()
    assert wl.tdd_out.shape == (512, 996, 1000) and not wl.f32_route and wl.conv_f3_route
    assert capi.load().nntk_hip_last_conv_kernel().decode() == "conv1d_mfma_bf16x3_kernel<frag3>"
    lstm_out = NL.frag3_unpack_device(wl.lstm_f3, 512, 996, 512)
    # the FRAG2H route's LSTM is the HF instantiation (its recurrence on two f16 images of h, three products per k step): the same h within
    # the contraction's noise over 996 steps and the whole batch, and the stack output likewise
    d_hh, d_yy = float((h_h2 - lstm_out).abs().max()), float((y_h2 - wl.tdd_out).abs().max())
    print("stack B=512: frag2h vs frag3 route, whole batch: LSTM output %.2e, stack output %.2e" % (d_hh, d_yy))
    assert 0.0 < d_hh < 3e-6 and 0.0 < d_yy < 5e-6
    del h_h2
    # the conv layer's output exists in frag3 form only (its epilogue wrote it): the f32 route of the same layer, same bits over the whole batch
    conv_out = wl.conv.apply_device(wl.spec_out, out=wl.conv_out, bn=wl.bn, act=wl.relu)
    assert torch.equal(NL.frag3_unpack_device(wl.conv_f3, 512, 996, 128), conv_out)
    win = O.window("hann", 400)
    for i in (0, 511):
        a = wl.x[i].cpu().numpy()
        s = O.spectrogram(a, win, 512, 240)
        close(wl.spec_out[i].cpu().numpy(), s, atol=1e-6 * float(np.abs(s).max()), rtol=2e-5)
        c = O.activation(O.ACT_RELU, O.batch_norm(O.conv1d(s, w["conv_W"], w["conv_b"], 1), w["bn_gamma"], w["bn_beta"],
                                                  w["bn_mean"], w["bn_var"], 1e-3))
        close(wl.conv_out[i].cpu().numpy(), c, atol=1e-5, rtol=1e-4)
        h, _, _ = O.lstm(c, w["lstm_W"], w["lstm_U"], w["lstm_bi"], w["lstm_bh"], v2=True)
        close(lstm_out[i].cpu().numpy(), h, atol=1e-4, rtol=1e-4)
        y = O.time_distributed_dense(h, w["tdd_W"], w["tdd_b"])
        close(wl.tdd_out[i].cpu().numpy(), y, atol=1e-4, rtol=1e-4)
        close(y_h2[i].cpu().numpy(), y, atol=1e-4, rtol=1e-4)
        e3, e2 = float(np.abs(wl.tdd_out[i].cpu().numpy() - y).max()), float(np.abs(y_h2[i].cpu().numpy() - y).max())
        print("stack row %d: max abs err of the stack output vs oracle: frag3 route %.2e, frag2h route %.2e" % (i, e3, e2))
        assert e2 < 1e-5 and e3 < 1e-5
    del y_h2
    # whole batch, the f32 route: same bits
    h32 = wl.lstm.apply_device(wl.conv_out)
    assert torch.equal(h32, lstm_out)
    del lstm_out
    y32 = wl.tdd.apply_device(h32)
    assert torch.equal(y32, wl.tdd_out)
    # whole batch, a second run of the frag3 route: same bits
    keep = wl.tdd_out.clone()
    wl.step()
    torch.cuda.synchronize()
    assert torch.equal(keep, wl.tdd_out)
    del keep
    # whole batch, the exact-f32 kernels
    capi.set_option("rec_rr", 0)
    hx = wl.lstm.apply_device(wl.conv_out)
    capi.set_option("rec_rr", "auto")
    d_h = float((hx - h32).abs().max())
    del h32
    capi.set_option("gemm_split_bf16", 0)
    yx = wl.tdd.apply_device(hx)
    capi.set_option("gemm_split_bf16", "auto")
    d_y = float((yx - y32).abs().max())
    print("stack B=512: register-resident vs exact LSTM %.2e, stack output split vs exact %.2e (whole batch)" % (d_h, d_y))
    assert d_h < 1e-5 and d_y < 2e-5
    assert capi.load().nntk_hip_device_status() == 0
    wl.destroy()


def test_persistent_and_per_step_recurrent_paths_agree_bitwise_in_sharding(gpu, monkeypatch):
    """Both recurrent code paths (persistent launch / one launch per timestep) match the oracle, and each
    is bit-identical between a whole batch and its shards."""
    import torch
    r = rng(77)
    B, T, I, H = 130, 20, 24, 64
    xs = u(r, B, T, I)
    x = torch.from_numpy(xs).cuda()
    W, U, bi, bh = lstm_weights(r, I, H)
    ref = O.lstm(xs, W, U, bi, bh, v2=True)
    capi.set_option("rec_rr", 0)                   # the exact-f32 kernels are the subject here
    for mode in ("1", "0"):
        capi.set_option("rec_persistent", mode)
        lstm = NL.LSTM(I, H, True, T, v2=True)
        lstm.set_weights(W, U, bi, bh)
        whole = lstm.apply_device(x).clone()
        close(whole.cpu().numpy(), ref)
        parts = torch.cat([lstm.apply_device(x[:70].contiguous()).clone(), lstm.apply_device(x[70:].contiguous()).clone()])
        assert torch.equal(whole, parts)
        lstm.destroy()


def test_lstm512_pingpong_and_classic_kernels_agree_bitwise(gpu, monkeypatch):
    """LSTM-512 runs the ping-pong variant of the persistent kernel by default (two 32-row halves per
    workgroup in alternation); it must equal the classic variant bit for bit -- ragged batch (a tile whose
    second half is empty, a tile with a partial half), carried state, last-step-only output."""
    import torch
    r = rng(512)
    I, H, T = 24, 512, 7
    W, U, bi, bh = lstm_weights(r, I, H)
    capi.set_option("rec_rr", 0)                   # the exact-f32 kernel's two variants are the subject here
    # 600 rows = 10 batch tiles: more than the 8 that fit the chip at once, i.e. two launches sharing the buffers
    for B, seq in ((1, True), (33, True), (130, True), (97, False), (600, True)):
        xs = u(r, B, T, I)
        x = torch.from_numpy(xs).cuda()
        ref = O.lstm(xs, W, U, bi, bh, return_sequences=seq, v2=True)
        outs = []
        for mode in ("1", "0"):
            capi.set_option("rec_pingpong", mode)
            lstm = NL.LSTM(I, H, seq, T, v2=True)
            lstm.set_weights(W, U, bi, bh)
            o = lstm.apply_device(x).clone()
            close(o.cpu().numpy(), ref)
            outs.append(o)
            lstm.destroy()
        assert torch.equal(outs[0], outs[1])
    # single-sequence API with carried (h, c): the second call starts from a non-zero tiled h_0
    x1, x2 = u(r, T, I), u(r, T, I)
    o1, h1, c1 = O.lstm(x1, W, U, bi, bh, v2=True)
    o2, h2, c2 = O.lstm(x2, W, U, bi, bh, h0=h1, c0=c1, v2=True)
    for mode in ("1", "0"):
        capi.set_option("rec_pingpong", mode)
        lstm = NL.LSTM(I, H, True, T, v2=True)
        lstm.set_weights(W, U, bi, bh)
        close(lstm.apply(x1), o1)
        close(lstm.apply(x2), o2)
        h, c = lstm.state()
        close(h, h2)
        close(c, c2)
        lstm.destroy()


@pytest.mark.parametrize("cell,H", [("gru", 384), ("gru", 512), ("lstm", 320), ("lstm", 448)])
def test_pingpong_default_shapes_match_classic_bitwise(gpu, monkeypatch, cell, H):
    """The ping-pong variant is selected wherever the K loop is long enough (LSTM H > 256, GRU H > 256);
    every such selection must equal the classic kernel bit for bit and the oracle within tolerance."""
    import torch
    r = rng(H + len(cell))
    B, T, I = 70, 5, 12
    xs = u(r, B, T, I)
    x = torch.from_numpy(xs).cuda()
    if cell == "gru":
        W, U, bi, bh = gru_weights(r, I, H)
        ref = O.gru(xs, W, U, bi, bh)
    else:
        W, U, bi, bh = lstm_weights(r, I, H)
        ref = O.lstm(xs, W, U, bi, bh, v2=True)
    outs = []
    capi.set_option("rec_rr", 0)
    for mode in ("1", "0"):
        capi.set_option("rec_pingpong", mode)
        l = NL.GRU(I, H, True, T) if cell == "gru" else NL.LSTM(I, H, True, T, v2=True)
        l.set_weights(W, U, bi, bh)
        outs.append(l.apply_device(x).clone())
        l.destroy()
    close(outs[0].cpu().numpy(), ref)
    assert torch.equal(outs[0], outs[1])


def test_hidden_sizes_beyond_the_persistent_kernel(gpu):
    """H > 512 does not fit the LDS-resident kernel: the per-timestep kernels take over (also for H % 4 != 0)."""
    r = rng(600)
    for (I, H, B, T) in ((20, 600, 5, 4), (8, 1024, 3, 3), (6, 258, 4, 5)):
        x = u(r, B, T, I)
        W, U, bi, bh = lstm_weights(r, I, H)
        l = NL.LSTM(I, H, True, T, v2=True)
        l.set_weights(W, U, bi, bh)
        close(l.apply(x), O.lstm(x, W, U, bi, bh, v2=True))
        l.destroy()
        W, U, bi, bh = gru_weights(r, I, H)
        g = NL.GRU(I, H, False, T)
        g.set_weights(W, U, bi, bh)
        close(g.apply(x), O.gru(x, W, U, bi, bh, return_sequences=False))
        g.destroy()


def test_lstm512_nondefault_gate_activations(gpu):
    """Non-standard gate activations take the generic (run-time dispatched) gate code of the persistent kernel."""
    L = capi.load()
    I, H, T = 8, 512, 5
    r = rng(5512)
    W, U, bi, bh = lstm_weights(r, I, H)
    x = u(r, T, I)
    # LSTMActivationsCreate argument order: input gate, forget gate, candidate, output gate, output (lstm.c:246-258)
    acts = L.LSTMActivationsCreate(L.ActivationFunctionCreateSigmoid(H), L.ActivationFunctionCreateSigmoid(H),
                                   L.ActivationFunctionCreateReLU(H, 1.0), L.ActivationFunctionCreateSigmoid(H),
                                   L.ActivationFunctionCreateTanh(H))
    lstm = NL.LSTM(I, H, True, T, v2=True, acts=acts)
    lstm.set_weights(W, U, bi, bh)
    ref, _, _ = O.lstm(x, W, U, bi, bh, v2=True, acts=(O.ACT_SIGMOID, O.ACT_SIGMOID, O.ACT_RELU, O.ACT_SIGMOID, O.ACT_TANH))
    close(lstm.apply(x), ref)
    lstm.destroy()


def test_time_major_projection_tiled_over_batch_is_bit_identical(gpu, monkeypatch):
    """The recurrent input projection writes a time-major tensor; by default its GEMM is tiled over the batch at a
    fixed timestep instead of over time within a sequence.  Same arithmetic per element: bit-identical layers."""
    import torch
    r = rng(2048)
    for (B, T, I, H) in ((130, 9, 24, 64), (3, 140, 20, 32), (257, 2, 36, 48)):
        xs = u(r, B, T, I)
        x = torch.from_numpy(xs).cuda()
        W, U, bi, bh = gru_weights(r, I, H)
        outs = []
        for mode in ("1", "0"):
            capi.set_option("gemm_tm_batch", mode)
            g = NL.GRU(I, H, True, T)
            g.set_weights(W, U, bi, bh)
            outs.append(g.apply_device(x).clone())
            g.destroy()
        close(outs[0].cpu().numpy(), O.gru(xs, W, U, bi, bh))
        assert torch.equal(outs[0], outs[1])


# ------------------------------------------- next row: RNN cell and bidirectional helpers ---

def rnn_weights(r, I, H):
    return u(r, I, H, sc=I ** -0.5), u(r, H, H, sc=H ** -0.5), u(r, H, sc=0.1), u(r, H, sc=0.1)


@pytest.mark.parametrize("I,H,T,seq,v2", [(5, 7, 11, True, True), (24, 40, 13, False, False), (64, 256, 9, True, True),
                                          (16, 512, 6, True, True)])
def test_rnn_single_sequence_stateful_and_batch(gpu, I, H, T, seq, v2):
    """layers/rnn.c forward: stateful single-sequence call, carried state, zero-state batch."""
    r = rng(I + H)
    W, U, bi, bh = rnn_weights(r, I, H)
    x1, x2 = u(r, T, I), u(r, T, I)
    rnn = NL.RNN(I, H, seq, T, v2=v2)
    rnn.set_weights(W, U, bi, bh)
    o1, h1 = O.rnn(x1, W, U, bi, bh, return_sequences=seq, v2=v2)
    o2, h2 = O.rnn(x2, W, U, bi, bh, h0=h1, return_sequences=seq, v2=v2)
    close(rnn.apply(x1), o1)
    close(rnn.state(), h1)
    close(rnn.apply(x2), o2)
    rnn.reset_state()
    close(rnn.apply(x1), o1)
    xb = u(r, 70, T, I)
    close(rnn.apply(xb), O.rnn(xb, W, U, bi, bh, return_sequences=seq, v2=v2))
    rnn.destroy()


def test_rnn_relu_and_both_recurrent_paths(gpu, monkeypatch):
    L = capi.load()
    r = rng(31)
    B, T, I, H = 9, 12, 10, 48
    W, U, bi, bh = rnn_weights(r, I, H)
    x = u(r, B, T, I)
    ref = O.rnn(x, W, U, bi, bh, act=O.ACT_RELU)
    for mode in ("1", "0"):
        capi.set_option("rec_persistent", mode)
        rnn = NL.RNN(I, H, True, T, act=L.ActivationFunctionCreateReLU(H, 1.0))
        rnn.set_weights(W, U, bi, bh)
        close(rnn.apply(x), ref)
        rnn.destroy()


def test_bidirectional_helpers_and_a_bidirectional_gru(gpu):
    """bidirectional.h forward helpers on host and device pointers, then the composition they exist for:
    forward GRU + backward GRU on the time-reversed input, backward output reversed back, concat / sum."""
    import torch
    import ctypes as C
    L = capi.load()
    r = rng(77)
    B, T, I, H = 5, 9, 6, 20
    x = u(r, B, T, I)
    cfg_in = capi.RecurrentConfig(I, H, True, T)
    out = np.empty_like(x)
    L.bd_reverse_input_batch(NL._p(x), NL._p(out), cfg_in, B)
    assert np.array_equal(out, O.bd_reverse(x))
    Wf, Uf, bif, bhf = gru_weights(r, I, H)
    Wb, Ub, bib, bhb = gru_weights(r, I, H)
    fwd_ref = O.gru(x, Wf, Uf, bif, bhf)
    bwd_ref = O.bd_reverse(O.gru(O.bd_reverse(x), Wb, Ub, bib, bhb))
    gf, gb = NL.GRU(I, H, True, T), NL.GRU(I, H, True, T)
    gf.set_weights(Wf, Uf, bif, bhf)
    gb.set_weights(Wb, Ub, bib, bhb)
    xd = torch.from_numpy(x).cuda()
    f = gf.apply_device(xd)
    bw = NL.bd_reverse_device(gb.apply_device(NL.bd_reverse_device(xd, "input")), "backward")
    close(NL.bd_merge_device(f, bw, "concat").cpu().numpy(), O.bd_merge(fwd_ref, bwd_ref, "concat"))
    close(NL.bd_merge_device(f, bw, "sum").cpu().numpy(), O.bd_merge(fwd_ref, bwd_ref, "sum"))
    # host-pointer merges (reference signatures), last-step-only layout
    cfg_last = capi.RecurrentConfig(I, H, False, T)
    a, b2 = fwd_ref[:, -1].copy(), bwd_ref[:, 0].copy()
    outc = np.empty((B, 2 * H), np.float32)
    buf = np.empty(L.bd_merge_concat_buffer_size(cfg_last), np.float32)
    L.bd_merge_concat(NL._p(a), NL._p(b2), NL._p(outc), cfg_last, B, NL._p(buf))
    assert np.array_equal(outc, np.concatenate([a, b2], axis=1))
    outs = np.empty((B, H), np.float32)
    L.bd_merge_sum(NL._p(a), NL._p(b2), NL._p(outs), cfg_last, B)
    assert np.array_equal(outs, a + b2)
    assert L.bd_merge_concat_buffer_size(cfg_in) == 2 * T * H
    gf.destroy(); gb.destroy()


# ------------------------------------------------ next row: mel filterbank / log-mel ---

def test_mel_filterbank_and_log_mel_spectrogram(gpu):
    import ctypes as C
    L = capi.load()
    r = rng(88)
    B, N = 3, 8240
    x = (0.1 * r.standard_normal((B, N))).astype(np.float32)
    sp = NL.Spectrogram(512, 400, 240, N)
    T, F = sp.out_shape
    cfg = L.MelFilterBankConfigCreate(40, 512, 16000, 20.0, 8000.0)
    w = O.mel_filterbank_weights(40, 512, 16000, 20.0, 8000.0)
    spec = O.spectrogram(x, O.window("hann", 400), 512, 240)
    # MelFilterBankApply: spec[T,257] x W[257,40]
    bank = L.MelFilterBankCreate(cfg)
    mel = np.full((T, 40), np.nan, np.float32)
    L.MelFilterBankApply(bank, spec[0].ctypes.data_as(capi.fp), mel.ctypes.data_as(capi.fp), T)
    assert capi.last_error() == ""
    close(mel, (spec[0].astype(np.float64) @ w.astype(np.float64)).astype(np.float32), atol=1e-6, rtol=1e-5)
    L.MelFilterBankDestroy(bank)
    # LogMelSpectrogramApply (one utterance) and the batched form
    lm = L.LogMelSpectrogramCreate(sp.h, cfg)
    out1 = np.full((T, 40), np.nan, np.float32)
    L.LogMelSpectrogramApply(lm, x[0].ctypes.data_as(capi.fp), out1.ctypes.data_as(capi.fp))
    ref = np.stack([O.log_mel(spec[i], w) for i in range(B)])
    close(out1, ref[0], atol=2e-5, rtol=1e-5)
    outb = np.empty((B, T, 40), np.float32)
    check = L.LogMelSpectrogramApplyBatch(lm, x.ctypes.data_as(capi.fp), outb.ctypes.data_as(capi.fp), B)
    assert check == 0, capi.last_error()
    close(outb, ref, atol=2e-5, rtol=1e-5)
    # the calls above ran the FUSED kernel (mel + log inside the STFT kernel's output stage, nfft = 512); the two-kernel
    # form (STFT, then the k = 1 GEMM with the log epilogue) must agree with it and with the oracle
    capi.set_option("spec_variant", 1)
    out2 = np.empty((B, T, 40), np.float32)
    assert L.LogMelSpectrogramApplyBatch(lm, x.ctypes.data_as(capi.fp), out2.ctypes.data_as(capi.fp), B) == 0
    capi.set_option("spec_variant", "auto")
    close(out2, ref, atol=2e-5, rtol=1e-5)
    close(out2, outb, atol=2e-5, rtol=1e-5)
    print("log-mel fused vs two-kernel: max abs diff %.2e; fused vs oracle %.2e" % (np.abs(out2 - outb).max(), np.abs(outb - ref).max()))
    L.LogMelSpectrogramDestroy(lm)
    sp.destroy()


@pytest.mark.parametrize("n_mels,nts_odd,mode", [(80, True, "magnitude"), (13, False, "psd"), (128, True, "magnitude")])
def test_fused_log_mel_other_banks(gpu, n_mels, nts_odd, mode):
    """Fused path with more filters than lanes per frame pair (2 x 80, 2 x 128 > 64), an odd frame count (the last pair has
    one frame), and PSD input."""
    L = capi.load()
    r = rng(n_mels)
    N = 240 + 160 * (21 if nts_odd else 20)
    x = (0.1 * r.standard_normal((2, N))).astype(np.float32)
    sp = NL.Spectrogram(512, 400, 240, N, mode=mode, fs=16000)
    T = sp.out_shape[0]
    assert T % 2 == (1 if nts_odd else 0)
    cfg = L.MelFilterBankConfigCreate(n_mels, 512, 16000, 0.0, 8000.0)
    w = O.mel_filterbank_weights(n_mels, 512, 16000, 0.0, 8000.0)
    spec = O.spectrogram(x, O.window("hann", 400), 512, 240, mode=mode, fs=16000)
    ref = np.stack([O.log_mel(spec[i], w) for i in range(2)])
    lm = L.LogMelSpectrogramCreate(sp.h, cfg)
    out = np.full((2, T, n_mels), np.nan, np.float32)
    assert L.LogMelSpectrogramApplyBatch(lm, x.ctypes.data_as(capi.fp), out.ctypes.data_as(capi.fp), 2) == 0, capi.last_error()
    close(out, ref, atol=3e-5, rtol=2e-5)
    L.LogMelSpectrogramDestroy(lm)
    sp.destroy()


# ------------------------------------------------------ randomized shape sweeps ---
# Seeded sweeps over ragged shapes: they reach every kernel variant (MFMA tiles of 32/64/128
# columns, the VALU conv, padded K, H not a multiple of 4 or 16, batch tails, both recurrent paths).

def test_random_conv_shapes(gpu):
    r = rng(2024)
    for _ in range(24):
        cin, cout = int(r.integers(1, 70)), int(r.integers(1, 140))
        k, stride = int(r.integers(1, 8)), int(r.integers(1, 4))
        T = int(r.integers(k, 300))
        B = int(r.integers(1, 4))
        x, W, b = u(r, B, T, cin), u(r, cout, cin, k, sc=(cin * k) ** -0.5), u(r, cout, sc=0.2)
        conv = NL.Conv1d(cin, cout, k, stride, T)
        conv.set_weights(W, b)
        close(conv.apply(x), O.conv1d(x, W, b, stride))
        conv.destroy()


def test_random_recurrent_shapes(gpu, monkeypatch):
    r = rng(4048)
    for i in range(16):
        I, H = int(r.integers(1, 40)), int(r.integers(1, 90))
        T, B = int(r.integers(1, 12)), int(r.integers(1, 80))
        seq, v2 = bool(r.integers(0, 2)), bool(r.integers(0, 2))
        capi.set_option("rec_persistent", "1" if i % 2 else "0")
        x = u(r, B, T, I)
        W, U, bi, bh = gru_weights(r, I, H)
        g = NL.GRU(I, H, seq, T)
        g.set_weights(W, U, bi, bh)
        close(g.apply(x), O.gru(x, W, U, bi, bh, return_sequences=seq))
        g.destroy()
        W, U, bi, bh = lstm_weights(r, I, H)
        l = NL.LSTM(I, H, seq, T, v2=v2)
        l.set_weights(W, U, bi, bh)
        close(l.apply(x), O.lstm(x, W, U, bi, bh, return_sequences=seq, v2=v2))
        l.destroy()


def test_random_dense_shapes(gpu):
    r = rng(777)
    for _ in range(12):
        ts, I, Ov = int(r.integers(1, 50)), int(r.integers(1, 100)), int(r.integers(1, 130))
        W, b, x = u(r, I, Ov, sc=I ** -0.5), u(r, Ov, sc=0.2), u(r, 2, ts, I)
        tdd = NL.TimeDistributedDense(ts, I, Ov)
        tdd.set_weights(W, b)
        close(tdd.apply(x), O.time_distributed_dense(x, W, b))
        tdd.destroy()


@pytest.mark.parametrize("n", [1, 2, 8, 60, 512, 1000, 4096])
@pytest.mark.parametrize("forward", [True, False])
def test_public_dft_api_matches_kissfft_restatement_and_numpy(gpu, n, forward):
    """signal/dft.h: DFTSetupCreate / DFTPerform / split_complex / join_complex_split on host split-complex buffers --
    against the oracle's kissfft restatement (the sizes it factors) and numpy's float64 FFT."""
    import ctypes as C
    L = capi.load()
    r = rng(n + forward)
    z = (u(r, n) + 1j * u(r, n)).astype(np.complex64)
    inter = np.ascontiguousarray(np.stack([z.real, z.imag], -1).astype(np.float32))
    re, im = np.empty(n, np.float32), np.empty(n, np.float32)
    sp_in = capi.ComplexFloatSplit(re.ctypes.data_as(capi.fp), im.ctypes.data_as(capi.fp))
    L.split_complex(inter.ctypes.data, C.byref(sp_in), n)
    np.testing.assert_array_equal(re, z.real); np.testing.assert_array_equal(im, z.imag)
    ore, oim = np.empty(n, np.float32), np.empty(n, np.float32)
    sp_out = capi.ComplexFloatSplit(ore.ctypes.data_as(capi.fp), oim.ctypes.data_as(capi.fp))
    s = L.DFTSetupCreate(L.DFTConfigCreate(n, forward, True))
    assert s, capi.last_error()
    L.DFTPerform(s, C.byref(sp_in), C.byref(sp_out))
    assert capi.last_error() == ""
    ref64 = np.fft.fft(z.astype(np.complex128)) if forward else np.fft.ifft(z.astype(np.complex128)) * n
    got = ore + 1j * oim
    scale = max(1.0, float(np.abs(ref64).max()))
    assert np.abs(got - ref64).max() <= 3e-7 * scale
    try:
        ok = O.kiss_fft(z, inverse=not forward)
    except Exception:
        ok = None                                   # sizes the restated kissfft does not take
    if ok is not None:
        assert np.abs(got - ok).max() <= 2e-6 * scale * max(1.0, np.log2(n))
    back = np.empty((n, 2), np.float32)
    L.join_complex_split(C.byref(sp_out), back.ctypes.data, n)
    np.testing.assert_array_equal(back[:, 0], ore); np.testing.assert_array_equal(back[:, 1], oim)
    L.DFTSetupDestroy(s)


@pytest.mark.parametrize("B,cin,cout,k,T,fused", [
    (3, 40, 128, 5, 300, True),       # BASELINE configs[2] shape: K = 200 -> 13 k steps instead of the chunked kernel's 15
    (2, 24, 64, 3, 131, False),       # BN = 64 tile, ragged last row tile, K = 72 -> 80: the last k step's second half is padding
    (5, 8, 128, 9, 140, True),        # one 8-channel group per tap: every k step straddles two taps
    (1, 56, 256, 2, 257, False),      # two column tiles
    (4, 104, 192, 4, 129, True),      # three 64-wide column tiles; 13 groups of 8 per tap
])
def test_flat_k_convolution_matches_oracle_and_the_chunked_kernel(gpu, B, cin, cout, k, T, fused):
    """conv1d_flatk.hip (conv_1d.c:122-147 semantics): stride-1 convolutions whose channel count is a multiple of 8 but not of 16 walk
    K = tap * Cin + channel without per-tap padding.  Against the oracle; against the chunked split kernel (conv_flatk = 0: same
    contraction, K in another order -- a few f32 roundings apart); shard-independent and reproducible; a NaN / inf input stays in
    exactly the outputs whose window holds it (the K padding's operand is a zero slot, not a neighbouring row)."""
    import torch
    r = rng(cin * 31 + cout + k)
    x, W, b = u(r, B, T, cin), u(r, cout, cin, k, sc=(cin * k) ** -0.5), u(r, cout, sc=0.2)
    conv = NL.Conv1d(cin, cout, k, 1, T)
    conv.set_weights(W, b)
    Tc = conv.out_shape[0]
    bn = relu = None
    ref = O.conv1d(x, W, b, 1)
    if fused:
        bn, relu = NL.BatchNorm(cout, 1e-3, Tc), NL.Activation("relu", Tc * cout, 1.0)
        g, be, mu, var = 1 + u(r, cout, sc=0.5), u(r, cout, sc=0.5), u(r, cout, sc=0.1), 1 + u(r, cout, sc=0.5)
        bn.set_weights(g, be, mu, var)
        ref = O.activation(O.ACT_RELU, O.batch_norm(ref, g, be, mu, var, 1e-3))
    xd = torch.from_numpy(x).cuda()
    flat = conv.apply_device(xd, bn=bn, act=relu).clone()
    capi.set_option("conv_flatk", 0)
    chunked = conv.apply_device(xd, bn=bn, act=relu).clone()
    capi.set_option("conv_flatk", "auto")
    close(flat.cpu().numpy(), ref)
    assert not torch.equal(flat, chunked)                       # the option really selected the other kernel
    assert float((flat - chunked).abs().max()) < 4e-6
    assert torch.equal(conv.apply_device(xd, bn=bn, act=relu), flat)
    assert torch.equal(conv.apply_device(xd[B - 1:].contiguous(), bn=bn, act=relu), flat[B - 1:])
    xp = xd.clone()
    xp[0, 50, 3] = float("nan")
    xp[B - 1, 100, cin - 1] = float("inf")
    bad = conv.apply_device(xp, bn=bn, act=relu)
    touched = torch.zeros(B, Tc, dtype=torch.bool, device="cuda")
    touched[0, max(0, 50 - k + 1):51] = True
    touched[B - 1, max(0, 100 - k + 1):101] = True
    assert torch.equal(bad[~touched], flat[~touched])
    assert not torch.isfinite(bad[touched]).all(dim=-1).any() or fused       # (ReLU turns -inf into 0: the unfused form must be all non-finite)
    if not fused:
        assert (~torch.isfinite(bad[touched])).any(dim=-1).all()
    for o in (conv, bn, relu):
        if o:
            o.destroy()
