"""Boundary robustness on the GPU (ADVICE r01 + VERDICT r01 "Next round" #4): outputs beyond 2 GiB, ReLU gate
scale, two host threads on two streams, and a persistent-kernel fault that heals itself."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle as O
from nntoolkitcore_amd import capi, layers as NL

pytestmark = pytest.mark.gpu


def rng(seed):
    return np.random.default_rng(seed)


def u(r, *shape, sc=1.0):
    return r.uniform(-sc, sc, shape).astype(np.float32)


def test_dense_output_beyond_2_gib_has_no_stray_writes(gpu):
    """padded output columns were masked with the vector offset 0x7ffffff0, which is IN range once
    the output tensor passes 2 GiB (TimeDistributedDense(1000) from ~537 k rows).  Descriptors are now tile-based and
    clamped; this runs 545 k rows x 1000 (2.18 GB) and checks rows on both sides of the 2 GiB line, every column."""
    import torch
    rows, cin, cout = 545_000, 32, 1000           # cout % 32 != 0: the last column tile has padded lanes
    r = rng(21)
    W, b = u(r, cin, cout, sc=cin ** -0.5), u(r, cout, sc=0.1)
    tdd = NL.TimeDistributedDense(rows, cin, cout)
    tdd.set_weights(W, b)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = torch.randn(1, rows, cin, device="cuda", generator=g)
    out = torch.full((1, rows, cout), float("nan"), device="cuda")
    tdd.apply_device(x, out=out)
    torch.cuda.synchronize()
    assert capi.load().nntk_hip_synchronize() == 0
    assert not bool(torch.isnan(out).any())                       # every element written exactly by its owner
    line = (2 ** 31) // (cout * 4)                                 # first row that crosses 2 GiB
    pick = np.unique(np.concatenate([np.arange(0, 300), np.arange(line - 400, line + 400), np.arange(rows - 300, rows),
                                     r.integers(0, rows, 1500)]))
    xs = x[0, pick].cpu().numpy()
    got = out[0, pick].cpu().numpy()
    ref = O.time_distributed_dense(xs, W, b)
    err = float(np.abs(got - ref).max())
    print("dense 545k x 1000 (2.18 GB out): max abs err on %d sampled rows %.3e" % (len(pick), err))
    assert err < 1e-5
    # the advisory's failure mode: column 908-ish of some row overwritten with act(0) = bias-free 0 -> caught above as
    # a mismatch; also make sure the whole tensor agrees with a device-side float64 product on a strided sample
    idx = torch.arange(0, rows, 97, device="cuda")
    ref64 = (x[0, idx].double() @ torch.from_numpy(W).cuda().double() + torch.from_numpy(b).cuda().double())
    assert float((out[0, idx].double() - ref64).abs().max()) < 1e-4
    tdd.destroy()


@pytest.mark.parametrize("persistent", ["1", "0"])
def test_relu_gate_activation_keeps_its_output_scale(gpu, persistent):
    """a gate configured with ActivationFunctionCreateReLU(H, a != 1) multiplies by a
    (activation_default.c:123-129), in both recurrent code paths."""
    capi.set_option("rec_persistent", persistent)
    L = capi.load()
    r = rng(31)
    B, I, H, T = 5, 12, 32, 9
    x = u(r, B, T, I)
    # GRU: h gate = ReLU * 0.5
    W, U, bi, bh = u(r, I, 3 * H, sc=0.3), u(r, H, 3 * H, sc=0.2), u(r, 3 * H, sc=0.1), u(r, 3 * H, sc=0.1)
    z, h, rr = L.ActivationFunctionCreateSigmoid(H), L.ActivationFunctionCreateReLU(H, C.c_float(0.5)), L.ActivationFunctionCreateSigmoid(H)
    gru = NL.GRU(I, H, True, T, acts=L.GRUActivationsCreate(z, h, rr))
    gru.set_weights(W, U, bi, bh)
    ref = O.gru(x, W, U, bi, bh, acts=(O.ACT_SIGMOID, O.ACT_RELU, O.ACT_SIGMOID), relu_a=(1.0, 0.5, 1.0))
    unscaled = O.gru(x, W, U, bi, bh, acts=(O.ACT_SIGMOID, O.ACT_RELU, O.ACT_SIGMOID))
    got = gru.apply(x)
    assert np.abs(ref - unscaled).max() > 1e-2                     # the scale matters in this case
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    gru.destroy()
    # LSTM: output activation = ReLU * 2, candidate = ReLU * 0.25
    W, U, bi, bh = u(r, I, 4 * H, sc=0.3), u(r, H, 4 * H, sc=0.2), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1)
    acts = L.LSTMActivationsCreate(L.ActivationFunctionCreateSigmoid(H), L.ActivationFunctionCreateSigmoid(H),
                                   L.ActivationFunctionCreateReLU(H, C.c_float(0.25)), L.ActivationFunctionCreateSigmoid(H),
                                   L.ActivationFunctionCreateReLU(H, C.c_float(2.0)))
    lstm = NL.LSTM(I, H, True, T, v2=True, acts=acts)
    lstm.set_weights(W, U, bi, bh)
    ka = (O.ACT_SIGMOID, O.ACT_SIGMOID, O.ACT_RELU, O.ACT_SIGMOID, O.ACT_RELU)
    ref = O.lstm(x, W, U, bi, bh, v2=True, acts=ka, relu_a=(1, 1, 0.25, 1, 2.0))
    np.testing.assert_allclose(lstm.apply(x), ref, rtol=1e-5, atol=1e-5)
    lstm.destroy()
    # RNN: ReLU * 0.25, stateful single sequence over two calls
    W, U, bi, bh = u(r, I, H, sc=0.3), u(r, H, H, sc=0.2), u(r, H, sc=0.1), u(r, H, sc=0.1)
    rnn = NL.RNN(I, H, True, T, v2=True, act=L.ActivationFunctionCreateReLU(H, C.c_float(0.25)))
    rnn.set_weights(W, U, bi, bh)
    o1, h1 = O.rnn(x[0], W, U, bi, bh, act=O.ACT_RELU, relu_a=0.25)
    o2, _ = O.rnn(x[1], W, U, bi, bh, h0=h1, act=O.ACT_RELU, relu_a=0.25)
    np.testing.assert_allclose(rnn.apply(x[0]), o1, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rnn.apply(x[1]), o2, rtol=1e-5, atol=1e-5)
    rnn.destroy()


def test_two_host_threads_two_handles_two_streams(gpu):
    """SURVEY 8(b) Threading: distinct handles are independent.  Two host threads, each with its own stream (the
    current stream is per thread), drive their own LSTM handle through the persistent kernel at the same time;
    each result is checked against the oracle.  (ctypes releases the GIL inside the C calls.)"""
    import torch
    L = capi.load()
    B, I, H, T = 96, 24, 512, 40
    results, errors = {}, []

    def work(tid):
        try:
            torch.cuda.set_device(0)
            r = rng(100 + tid)
            st = torch.cuda.Stream()
            L.nntk_hip_set_stream(C.c_void_p(st.cuda_stream))
            assert L.nntk_hip_get_stream() == st.cuda_stream
            W, U, bi, bh = u(r, I, 4 * H, sc=I ** -0.5), u(r, H, 4 * H, sc=H ** -0.5), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1)
            x = u(r, B, T, I)
            lstm = NL.LSTM(I, H, True, T, v2=True)
            lstm.set_weights(W, U, bi, bh)
            with torch.cuda.stream(st):
                xd = torch.from_numpy(x).cuda()
                outs = [lstm.apply_device(xd) for _ in range(6)]
            assert L.nntk_hip_synchronize() == 0, capi.last_error()
            got = outs[-1].cpu().numpy()
            for o in outs[:-1]:
                assert torch.equal(o, outs[-1])
            ref = O.lstm(x[:4], W, U, bi, bh, v2=True)
            results[tid] = float(np.abs(got[:4] - ref).max())
            lstm.destroy()
        except Exception as e:                                     # pragma: no cover
            errors.append((tid, repr(e)))

    main_stream = L.nntk_hip_get_stream()
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errors, errors
    assert L.nntk_hip_get_stream() == main_stream                  # the workers' set_stream never touched this thread
    print("two threads: max abs err vs oracle", results)
    assert len(results) == 2 and max(results.values()) < 1e-5
    assert capi.get_option("rec_persistent") != 0 and L.nntk_hip_device_status() == 0


def test_persistent_fault_is_reported_and_heals(gpu):
    """A persistent launch whose spins run out of budget (forced here with rec_spin_us = 0, the fault-injection value) must never hand back
    garbage silently: the host-pointer call repeats itself on the per-timestep kernels and returns 0 with correct
    results; a device-pointer caller gets -1 from nntk_hip_synchronize(); later calls use the per-step kernels."""
    import torch
    L = capi.load()
    r = rng(77)
    B, I, H, T = 70, 16, 256, 30
    W, U, bi, bh = u(r, I, 3 * H, sc=0.25), u(r, H, 3 * H, sc=H ** -0.5), u(r, 3 * H, sc=0.1), u(r, 3 * H, sc=0.1)
    x = u(r, B, T, I)
    ref = O.gru(x, W, U, bi, bh)
    gru = NL.GRU(I, H, True, T)
    gru.set_weights(W, U, bi, bh)
    good = gru.apply(x)                                            # healthy persistent run
    np.testing.assert_allclose(good, ref, rtol=1e-5, atol=1e-5)

    capi.set_option("rec_spin_us", 0)                              # every hand-off poll gives up at once
    # device-pointer caller: the fault surfaces at the sync point
    xd = torch.from_numpy(x).cuda()
    gru.apply_device(xd)
    torch.cuda.synchronize()
    assert L.nntk_hip_device_status() == 1
    assert L.nntk_hip_synchronize() == -1 and "timed out" in capi.last_error()
    assert L.nntk_hip_device_status() == 0                         # reported once, then clear
    # the process now keeps to the per-step kernels (their split-K order differs from the persistent kernel's, so
    # "equal" here means equal to a run with the persistent kernel switched off, and both within tolerance of the oracle)
    after = gru.apply_device(xd).cpu().numpy()
    assert L.nntk_hip_synchronize() == 0
    capi.set_option("rec_persistent", 0)
    assert np.array_equal(after, gru.apply_device(xd).cpu().numpy())
    np.testing.assert_allclose(after, ref, rtol=1e-5, atol=1e-5)

    # host-pointer caller: re-arm the persistent kernel, keep the tiny budget -> the call heals itself
    capi.set_option("rec_persistent", 1)
    healed = gru.apply(x)
    assert capi.last_error() == ""
    assert np.array_equal(healed, after)                           # the repeated call ran on the per-step kernels
    # stateful single-sequence call: the repeated call must start from the SAME carried state
    capi.set_option("rec_persistent", 1)
    g1 = NL.GRU(I, H, True, T)
    g1.set_weights(W, U, bi, bh)
    capi.set_option("rec_spin_us", 1000000)
    a1 = g1.apply(x[0])
    capi.set_option("rec_persistent", 1); capi.set_option("rec_spin_us", 0)
    a2 = g1.apply(x[1])                                            # faults, heals, continues from a1's state
    o1, h1 = O.gru(x[0], W, U, bi, bh)
    o2, _ = O.gru(x[1], W, U, bi, bh, h0=h1)
    np.testing.assert_allclose(a1, o1, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(a2, o2, rtol=1e-5, atol=1e-5)
    g1.destroy(); gru.destroy()
    capi.set_option("rec_spin_us", "auto"); capi.set_option("rec_persistent", "auto")
    assert L.nntk_hip_synchronize() == 0


# ---- achieved error of the T ~ 1000 recurrences at the BASELINE shapes (VERDICT r01 "What's weak" #2) ----
# The gate activations are hardware exp2 / rcp forms (nntk_common.hpp), not libm expf / tanhf with a true divide
# (SURVEY a22): the deviation is a NUMBER here, against the oracle and against torch in float64, and the bound is
# what was measured x 3 (the fp32 noise floor of the reference itself at T = 1000 is ~6e-7).

def _torch64(kind, x, W, U, bi, bh):
    import torch
    H = U.shape[0]
    if kind == "lstm":
        m = torch.nn.LSTM(W.shape[0], H, batch_first=True).double()
        p = lambda a: torch.tensor(a).double()
    else:
        m = torch.nn.GRU(W.shape[0], H, batch_first=True).double()
        perm = lambda a: np.concatenate([a[..., H:2 * H], a[..., :H], a[..., 2 * H:]], axis=-1)     # [z|r|h] -> torch's [r|z|n]
        p = lambda a: torch.tensor(perm(a)).double()
    with torch.no_grad():
        m.weight_ih_l0.copy_(p(W).T); m.weight_hh_l0.copy_(p(U).T); m.bias_ih_l0.copy_(p(bi)); m.bias_hh_l0.copy_(p(bh))
        return m(torch.tensor(x).double())[0].numpy()


def _uw(r, fan, *shape):
    return r.uniform(-fan ** -0.5, fan ** -0.5, shape).astype(np.float32)


def test_achieved_error_lstm512_T996(gpu):
    r = rng(501)
    B, I, H, T = 3, 128, 512, 996
    x = r.standard_normal((B, T, I)).astype(np.float32)
    W, U, bi, bh = _uw(r, I, I, 4 * H), _uw(r, H, H, 4 * H), _uw(r, H, 4 * H), _uw(r, H, 4 * H)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    got = lstm.apply(x)
    e_or = float(np.abs(got - O.lstm(x, W, U, bi, bh, v2=True)).max())
    e_64 = float(np.abs(got - _torch64("lstm", x, W, U, bi, bh)).max())
    print("HIP LSTM(128->512, v2) T=996: max abs err vs oracle %.2e, vs torch float64 %.2e" % (e_or, e_64))
    assert e_or < 3e-6 and e_64 < 3e-6
    lstm.destroy()


def test_achieved_error_two_layer_gru256_T1000(gpu):
    r = rng(502)
    B, T = 3, 1000
    x = r.standard_normal((B, T, 128)).astype(np.float32)
    W1, U1, bi1, bh1 = _uw(r, 128, 128, 768), _uw(r, 256, 256, 768), _uw(r, 256, 768), _uw(r, 256, 768)
    W2, U2, bi2, bh2 = _uw(r, 256, 256, 768), _uw(r, 256, 256, 768), _uw(r, 256, 768), _uw(r, 256, 768)
    g1, g2 = NL.GRU(128, 256, True, T), NL.GRU(256, 256, True, T)
    g1.set_weights(W1, U1, bi1, bh1); g2.set_weights(W2, U2, bi2, bh2)
    got = g2.apply(g1.apply(x))
    ref = O.gru(O.gru(x, W1, U1, bi1, bh1), W2, U2, bi2, bh2)
    r64 = _torch64("gru", _torch64("gru", x, W1, U1, bi1, bh1).astype(np.float32), W2, U2, bi2, bh2)
    e_or, e_64 = float(np.abs(got - ref).max()), float(np.abs(got - r64).max())
    print("HIP GRU 128->256->256 T=1000: max abs err vs oracle %.2e, vs torch float64 %.2e" % (e_or, e_64))
    assert e_or < 3e-6 and e_64 < 3e-6
    # the fused single-launch form (the config-4 bench kernel) on the same inputs
    fused = NL.gru_stack2_apply(g1, g2, x)
    f_or, f_64 = float(np.abs(fused - ref).max()), float(np.abs(fused - r64).max())
    print("HIP fused GRU stack (gru2_persistent_kernel) T=1000: max abs err vs oracle %.2e, vs torch float64 %.2e, "
          "vs the two calls %.2e" % (f_or, f_64, float(np.abs(fused - got).max())))
    assert f_or < 3e-6 and f_64 < 3e-6
    g1.destroy(); g2.destroy()


# ---- streaming path: the reference's own call shape, one sequence with carried state ----

@pytest.mark.parametrize("cell,I,H", [("gru", 128, 256), ("lstm", 128, 512), ("rnn", 40, 64), ("gru", 5, 7), ("lstm", 33, 40)])
def test_streaming_calls_are_bit_identical_to_the_batch_kernels(gpu, cell, I, H):
    """GRU/LSTM/RNNApplyInference with T <= 32 take the streaming step kernel (no projection GEMM, no persistent
    launch); each output element is the same k-ordered fmaf chain the MFMA kernels compute, so the results equal the
    batch path's bit for bit -- checked against (a) the same calls with rec_stream = 0 and (b) the oracle.  (Shapes the
    MFMA / persistent kernels do not take -- H % 4 != 0, fewer than 16 inputs -- run other batch kernels with other
    summation orders; there the comparison is the oracle's tolerance.)"""
    r = rng(len(cell) * 1000 + H)
    G = {"gru": 3, "lstm": 4, "rnn": 1}[cell]
    T = 8
    x = u(r, 3 * T, I)
    W, U, bi, bh = u(r, I, G * H, sc=I ** -0.5), u(r, H, G * H, sc=H ** -0.5), u(r, G * H, sc=0.1), u(r, G * H, sc=0.1)
    mk = {"gru": lambda: NL.GRU(I, H, True, T), "lstm": lambda: NL.LSTM(I, H, True, T, v2=True), "rnn": lambda: NL.RNN(I, H, True, T)}[cell]
    outs = {}
    for mode in ("auto", "0"):
        capi.set_option("rec_stream", mode)
        l = mk()
        l.set_weights(W, U, bi, bh)
        outs[mode] = np.concatenate([l.apply(x[i * T:(i + 1) * T]) for i in range(3)])      # three calls, carried state
        st = l.state()
        outs[mode + "_state"] = st if cell != "lstm" else np.concatenate(st)
        l.destroy()
    if H % 4 == 0 and I >= 16 and G * H >= 32:
        assert np.array_equal(outs["auto"], outs["0"]) and np.array_equal(outs["auto_state"], outs["0_state"])
    else:
        np.testing.assert_allclose(outs["auto"], outs["0"], rtol=1e-5, atol=1e-5)
    ofn = {"gru": O.gru, "lstm": O.lstm, "rnn": O.rnn}[cell]
    ref = ofn(x, W, U, bi, bh)[0]
    np.testing.assert_allclose(outs["auto"], ref, rtol=1e-5, atol=1e-5)


def test_streaming_last_state_only_and_reset(gpu):
    r = rng(71)
    I, H, T = 24, 48, 5
    W, U, bi, bh = u(r, I, 4 * H, sc=0.2), u(r, H, 4 * H, sc=0.15), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1)
    x = u(r, 2 * T, I)
    l = NL.LSTM(I, H, False, T, v2=False)           # return_sequences = false, Keras one-bias form
    l.set_weights(W, U, bi, bh)
    a = l.apply(x[:T]); b = l.apply(x[T:])
    full, _, _ = O.lstm(x, W, U, bi, bh, v2=False)
    np.testing.assert_allclose(a, full[T - 1], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(b, full[2 * T - 1], rtol=1e-5, atol=1e-5)
    l.reset_state()
    np.testing.assert_allclose(l.apply(x[:T]), full[T - 1], rtol=1e-5, atol=1e-5)
    l.destroy()


# ---- fused two-layer GRU (BASELINE configs[3]) ----

@pytest.mark.parametrize("mode", ["auto", 1])        # auto: two register-resident launches when both layers' shapes qualify (any B); 1: the fused kernel
@pytest.mark.parametrize("B,I,H,T,seq", [(70, 128, 256, 40, True), (3, 24, 64, 17, True), (130, 16, 128, 9, False), (65, 40, 256, 5, True)])
def test_fused_two_layer_gru_matches_two_calls_and_oracle(gpu, B, I, H, T, seq, mode):
    """GRUStack2ApplyDevice: both layers in ONE persistent launch, layer 2 one step behind.  Against the oracle, and
    against the two single-layer calls: equal within the layer tolerance (layer 2's input projection is summed in the
    MFMA's k order instead of the GEMM kernel's), and the fused form itself is reproducible and shard-independent."""
    import torch
    r = rng(B * 7 + H)
    x = u(r, B, T, I)
    W1, U1, bi1, bh1 = u(r, I, 3 * H, sc=I ** -0.5), u(r, H, 3 * H, sc=H ** -0.5), u(r, 3 * H, sc=0.1), u(r, 3 * H, sc=0.1)
    W2, U2, bi2, bh2 = u(r, H, 3 * H, sc=H ** -0.5), u(r, H, 3 * H, sc=H ** -0.5), u(r, 3 * H, sc=0.1), u(r, 3 * H, sc=0.1)
    g1, g2 = NL.GRU(I, H, True, T), NL.GRU(H, H, seq, T)
    g1.set_weights(W1, U1, bi1, bh1); g2.set_weights(W2, U2, bi2, bh2)
    ref = O.gru(O.gru(x, W1, U1, bi1, bh1), W2, U2, bi2, bh2, return_sequences=seq)
    capi.set_option("rec_fused2", mode)
    fused = NL.gru_stack2_apply(g1, g2, x)
    if mode == 1:
        assert capi.load().nntk_hip_last_recurrent_kernel().decode().startswith("gru2_persistent_kernel")
    capi.set_option("rec_fused2", 0)
    two = NL.gru_stack2_apply(g1, g2, x)                       # falls back to the two calls
    capi.set_option("rec_fused2", mode)
    e_f, e_t, e_ft = float(np.abs(fused - ref).max()), float(np.abs(two - ref).max()), float(np.abs(fused - two).max())
    print("fused GRU stack B=%d H=%d T=%d: vs oracle %.2e (two calls %.2e), fused vs two calls %.2e" % (B, H, T, e_f, e_t, e_ft))
    assert e_f < 1e-5 and e_t < 1e-5 and e_ft < 5e-6
    xd = torch.from_numpy(x).cuda()
    a = NL.gru_stack2_apply_device(g1, g2, xd).cpu().numpy()
    assert np.array_equal(a, fused)                            # reproducible
    # a shard gives the same bits as the whole batch, whatever its size (the kernel choice depends on shape and activations only)
    for n in (B // 2 + 1, min(B, 5)):
        lo = NL.gru_stack2_apply_device(g1, g2, xd[:n].contiguous()).cpu().numpy()
        assert np.array_equal(lo, fused[:n])
    capi.set_option("rec_fused2", "auto")
    g1.destroy(); g2.destroy()


@pytest.mark.parametrize("mode", ["auto", 1])
def test_fused_two_layer_gru_several_launches_and_ragged_batch(gpu, mode):
    """B = 1100 at H = 256: 18 batch tiles of 64 rows (the last one holds 12) x 16 column tiles = 288 workgroups > 256
    CUs, so the call is TWO persistent launches (16 + 2 batch tiles) sharing one counter / hand-off area -- against the
    oracle on rows of both launches and of the ragged tile, and bit-identical to the same rows run as smaller batches."""
    import torch
    r = rng(1100)
    B, I, H, T = 1100, 40, 256, 48
    x = u(r, B, T, I)
    W1, U1, bi1, bh1 = u(r, I, 3 * H, sc=I ** -0.5), u(r, H, 3 * H, sc=H ** -0.5), u(r, 3 * H, sc=0.1), u(r, 3 * H, sc=0.1)
    W2, U2, bi2, bh2 = u(r, H, 3 * H, sc=H ** -0.5), u(r, H, 3 * H, sc=H ** -0.5), u(r, 3 * H, sc=0.1), u(r, 3 * H, sc=0.1)
    g1, g2 = NL.GRU(I, H, True, T), NL.GRU(H, H, True, T)
    g1.set_weights(W1, U1, bi1, bh1); g2.set_weights(W2, U2, bi2, bh2)
    xd = torch.from_numpy(x).cuda()
    capi.set_option("rec_fused2", mode)
    y = NL.gru_stack2_apply_device(g1, g2, xd)
    assert capi.load().nntk_hip_device_status() == 0
    rows = [0, 63, 64, 1023, 1024, 1087, 1088, 1099]
    ref = O.gru(O.gru(x[rows], W1, U1, bi1, bh1), W2, U2, bi2, bh2)
    got = y[rows].cpu().numpy()
    print("fused GRU stack B=1100 (2 launches): max abs err vs oracle %.2e" % float(np.abs(got - ref).max()))
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    tail = NL.gru_stack2_apply_device(g1, g2, xd[1000:].contiguous())           # rows 1000..1099 as their own batch
    assert torch.equal(tail, y[1000:])
    capi.set_option("rec_fused2", "auto")
    g1.destroy(); g2.destroy()


def test_fused_gru_stack_falls_back_for_other_activations(gpu):
    L = capi.load()
    r = rng(9)
    B, I, H, T = 4, 8, 16, 6
    x = u(r, B, T, I)
    W1, U1, bi1, bh1 = u(r, I, 3 * H, sc=0.3), u(r, H, 3 * H, sc=0.2), u(r, 3 * H, sc=0.1), u(r, 3 * H, sc=0.1)
    W2, U2, bi2, bh2 = u(r, H, 3 * H, sc=0.3), u(r, H, 3 * H, sc=0.2), u(r, 3 * H, sc=0.1), u(r, 3 * H, sc=0.1)
    acts = L.GRUActivationsCreate(L.ActivationFunctionCreateSigmoid(H), L.ActivationFunctionCreateReLU(H, C.c_float(0.5)), L.ActivationFunctionCreateSigmoid(H))
    g1, g2 = NL.GRU(I, H, True, T), NL.GRU(H, H, True, T, acts=acts)
    g1.set_weights(W1, U1, bi1, bh1); g2.set_weights(W2, U2, bi2, bh2)
    ref = O.gru(O.gru(x, W1, U1, bi1, bh1), W2, U2, bi2, bh2, acts=(O.ACT_SIGMOID, O.ACT_RELU, O.ACT_SIGMOID), relu_a=(1, 0.5, 1))
    np.testing.assert_allclose(NL.gru_stack2_apply(g1, g2, x), ref, rtol=1e-5, atol=1e-5)
    g1.destroy(); g2.destroy()


# ---- training, first slice: Conv1dCalculateGradient ----

@pytest.mark.parametrize("B,T,Cin,Cout,k,s", [(2, 23, 3, 4, 5, 2), (3, 40, 8, 16, 5, 1), (2, 17, 5, 3, 3, 3), (4, 300, 40, 128, 5, 1),
                                             (12, 450, 40, 128, 5, 1), (24, 400, 33, 96, 3, 1),       # d_X on the MFMA form
                                             (40, 300, 40, 128, 5, 2), (16, 500, 36, 72, 7, 1)])      # d_W on outer_mfma_kernel (stride 2; ragged tiles)
def test_conv1d_training_forward_and_gradient(gpu, B, T, Cin, Cout, k, s):
    """Conv1dCreateForTraining / ApplyTrainingBatch / CreateGradient / CalculateGradient through the C boundary against
    the oracle (reference loop order) and torch autograd (float64).  d_W, d_b accumulate into the block, d_X is overwritten."""
    import torch
    import torch.nn.functional as F
    L = capi.load()
    r = rng(B * 31 + T)
    x = u(r, B, T, Cin)
    W, b = u(r, Cout, Cin, k, sc=(Cin * k) ** -0.5), u(r, Cout, sc=0.1)
    cfg = L.Conv1dConfigCreate(Cin, Cout, k, s, T)
    tc = capi.ConvTrainingConfig(B)
    h = L.Conv1dCreateForTraining(cfg, tc)
    w = L.Conv1dGetWeights(h).contents
    C.memmove(w.W, W.ctypes.data, W.nbytes); C.memmove(w.b, b.ctypes.data, b.nbytes)
    Tout = cfg.output_size
    y = np.empty((B, Tout, Cout), np.float32)
    # wrong-mode calls behave like the reference (golden "wrong_mode_apply_inference": -1)
    assert L.Conv1dApplyInference(h, x.ctypes.data_as(capi.fp), y.ctypes.data_as(capi.fp)) == -1
    assert L.Conv1dApplyTrainingBatch(h, x.ctypes.data_as(capi.fp), y.ctypes.data_as(capi.fp)) == 0, capi.last_error()
    np.testing.assert_allclose(y, O.conv1d(x, W, b, s), rtol=1e-5, atol=1e-5)
    dout = u(r, B, Tout, Cout)
    g = L.Conv1dCreateGradient(cfg, tc)
    L.Conv1dCalculateGradient(h, g, dout.ctypes.data_as(capi.fp))
    assert capi.last_error() == ""
    gw = np.ctypeslib.as_array(g.contents.d_W, shape=(Cout, Cin, k)).copy()
    gb = np.ctypeslib.as_array(g.contents.d_b, shape=(Cout,)).copy()
    gx = np.ctypeslib.as_array(g.contents.d_X, shape=(B, T, Cin)).copy()
    dW, db, dX = O.conv1d_gradient(x, W, dout, s)
    xt, Wt = torch.tensor(x).double().requires_grad_(True), torch.tensor(W).double().requires_grad_(True)
    bt = torch.zeros(Cout).double().requires_grad_(True)
    F.conv1d(xt.transpose(1, 2), Wt, bt, stride=s).transpose(1, 2)[:, :Tout].backward(torch.tensor(dout).double())
    tol = 1.5e-7 * np.sqrt(B * Tout)                     # measured on MI355X: <= 3.6e-8 x sqrt(B Tout) x scale; bound = that x 4
    for name, got, ora, t64 in (("dW", gw, dW, Wt.grad.numpy()), ("db", gb, db, bt.grad.numpy()), ("dX", gx, dX, xt.grad.numpy())):
        sc = max(1.0, float(np.abs(t64).max()))
        e_o, e_t = float(np.abs(got - ora).max()), float(np.abs(got - t64).max())
        print("conv grad %s (%d,%d,%d,%d,%d,%d): vs oracle %.2e, vs torch float64 %.2e (scale %.1f)" % (name, B, T, Cin, Cout, k, s, e_o, e_t, sc))
        assert e_o <= tol * sc and e_t <= tol * sc
    # a second call accumulates d_W / d_b and rewrites d_X
    L.Conv1dCalculateGradient(h, g, dout.ctypes.data_as(capi.fp))
    np.testing.assert_allclose(np.ctypeslib.as_array(g.contents.d_W, shape=(Cout, Cin, k)), 2 * gw, rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(np.ctypeslib.as_array(g.contents.d_X, shape=(B, T, Cin)), gx)
    # an inference handle refuses the training call
    hi = L.Conv1dCreateForInference(cfg)
    assert L.Conv1dApplyTrainingBatch(hi, x.ctypes.data_as(capi.fp), y.ctypes.data_as(capi.fp)) == -1
    L.Conv1dDestroy(hi); L.ConvGradientDestroy(g); L.Conv1dDestroy(h)
