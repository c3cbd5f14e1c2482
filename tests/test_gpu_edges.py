"""GPU tests added in round 3: the edges VERDICT r02 / ADVICE r02 named.

* what the default (split-bf16 x 3) contraction does with non-finite, huge and denormal values, pinned: the promise is
  written in INTEGRATION.md section 6 and next to gemm_split_bf16 in the header;
* in-place edits of a single interior weight between two host-pointer calls (inference and training);
* a persistent recurrent launch next to RCCL traffic on another stream of the same device.
"""
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


# ------------------------------------------------------------ split contraction: edge values ---
# Reference arithmetic (core/default_ops.cc:224-231, layers/dense.c:122-125): a k-ordered f32 chain, so an infinite input
# gives +-inf (or NaN when it meets a zero weight or an opposite infinity), NaN stays NaN, denormals are ordinary numbers.
# The exact-f32 MFMA kernel (gemm_split_bf16 = 0) IS that chain.  The default contraction splits every operand into three
# bf16 terms; what it promises instead:
#   * a non-finite input, or one beyond bf16's largest finite value (3.39e38), makes every output whose window contains
#     it NON-FINITE (NaN where the reference says +-inf): never a finite wrong number, and nothing outside that window
#     is touched;
#   * inputs below 1.18e-38 (denormal) count as zero: absolute error <= 1.2e-38 * sum|w| per output;
#   * a WEIGHT block holding such a value is detected when it is uploaded and runs on the exact kernel ("auto" only).

def _conv_case(r, cin=24, cout=64, k=3, T=300):
    x, W, b = u(r, T, cin), u(r, cout, cin, k, sc=(cin * k) ** -0.5), u(r, cout, sc=0.2)
    return x, W, b


@pytest.mark.parametrize("bad", [np.inf, -np.inf, np.nan, 3.4e38])
def test_split_contraction_nonfinite_input_stays_nonfinite_and_local(gpu, bad):
    r = rng(17)
    cin, cout, k, T = 24, 64, 3, 300
    x, W, b = _conv_case(r, cin, cout, k, T)
    t_bad, c_bad = 150, 7
    x[t_bad, c_bad] = bad
    conv = NL.Conv1d(cin, cout, k, 1, T)
    conv.set_weights(W, b)
    got = conv.apply(x)                                     # auto = split
    capi.set_option("gemm_split_bf16", 0)
    exact = conv.apply(x)
    capi.set_option("gemm_split_bf16", "auto")
    ref = O.conv1d(x, W, b, 1)
    hit = np.zeros(got.shape[0], bool)
    hit[max(0, t_bad - k + 1):t_bad + 1] = True             # outputs whose window holds the bad sample
    # the exact kernel is the reference's chain: same class of value everywhere
    assert np.array_equal(np.isnan(exact), np.isnan(ref)) and np.array_equal(np.isinf(exact), np.isinf(ref))
    if np.isfinite(bad):
        # 3.4e38 is finite for the reference: huge finite (or overflowed) products there, non-finite here
        assert not np.isfinite(got[hit]).any()
    else:
        assert not np.isfinite(ref[hit]).any() and not np.isfinite(exact[hit]).any()
        assert not np.isfinite(got[hit]).any()
    np.testing.assert_allclose(got[~hit], ref[~hit], rtol=1e-5, atol=1e-5)
    assert np.isfinite(got[~hit]).all()
    conv.destroy()


def test_split_contraction_denormal_inputs_flush_with_a_bound(gpu):
    r = rng(18)
    cin, cout, k, T = 24, 64, 3, 300
    x, W, b = _conv_case(r, cin, cout, k, T)
    x = (x * 1e-39).astype(np.float32)                      # every input a denormal
    b[:] = 0
    conv = NL.Conv1d(cin, cout, k, 1, T)
    conv.set_weights(W, b)
    got = conv.apply(x)
    capi.set_option("gemm_split_bf16", 0)
    exact = conv.apply(x)
    capi.set_option("gemm_split_bf16", "auto")
    ref = O.conv1d(x, W, b, 1)
    bound = 1.2e-38 * np.abs(W).sum(axis=(1, 2)).max()
    assert np.abs(got.astype(np.float64) - ref).max() <= bound
    np.testing.assert_allclose(exact, ref, rtol=1e-4, atol=1e-43)      # the exact chain keeps denormals (fma vs mul+add: a few denormal quanta)
    # ordinary inputs mixed with denormals lose nothing
    x2 = u(r, T, cin)
    x2[::7] *= 1e-39
    np.testing.assert_allclose(conv.apply(x2), O.conv1d(x2, W, b, 1), rtol=1e-5, atol=1e-5)
    conv.destroy()


@pytest.mark.parametrize("bad", [np.inf, np.nan, 1e-40, 3.4e38])
def test_weights_the_split_cannot_hold_run_on_the_exact_kernel(gpu, bad):
    """A weight block with a non-finite / huge / denormal value is found at upload; "auto" then equals gemm_split_bf16=0
    bit for bit (dense and conv), and goes back to the split kernel once clean weights are uploaded again."""
    r = rng(19)
    I, N, rows = 64, 96, 200
    x, W, b = u(r, rows, I), u(r, I, N, sc=I ** -0.5), u(r, N, sc=0.1)
    Wb = W.copy()
    Wb[5, 11] = bad
    tdd = NL.TimeDistributedDense(rows, I, N)
    tdd.set_weights(Wb, b)
    auto = tdd.apply(x)
    capi.set_option("gemm_split_bf16", 0)
    exact = tdd.apply(x)
    capi.set_option("gemm_split_bf16", "auto")
    assert np.array_equal(auto, exact, equal_nan=True)
    ref = O.time_distributed_dense(x, Wb, b)
    assert np.array_equal(np.isnan(auto), np.isnan(ref)) and np.array_equal(np.isinf(auto), np.isinf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(auto[fin], ref[fin], rtol=1e-5, atol=1e-5)
    # clean weights again: back on the split kernel (differs from the exact chain in the last bits on some element)
    tdd.set_weights(W, b)
    auto2 = tdd.apply(x)
    capi.set_option("gemm_split_bf16", 0)
    exact2 = tdd.apply(x)
    capi.set_option("gemm_split_bf16", "auto")
    np.testing.assert_allclose(auto2, O.time_distributed_dense(x, W, b), rtol=1e-5, atol=1e-5)
    assert not np.array_equal(auto2, exact2)
    tdd.destroy()


# ------------------------------------------------------------ single interior weight edits ---

def test_single_interior_weight_edit_is_seen_by_the_next_host_call(gpu):
    """the reference reads the caller's block on every Apply.  One float changed in the middle of a
    large block -- far from every probe of the sampled check -- must show in the very next batch / inference call."""
    r = rng(23)
    # LSTM-512: 1.3 M floats; an element that no 16-float probe of the 257 covers
    I, H, T, B = 128, 512, 6, 3
    W, U, bi, bh = u(r, I, 4 * H, sc=I ** -0.5), u(r, H, 4 * H, sc=H ** -0.5), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1)
    x = u(r, B, T, I)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    np.testing.assert_allclose(lstm.apply(x), O.lstm(x, W, U, bi, bh, v2=True), rtol=1e-5, atol=1e-5)
    w = capi.load().LSTMGetWeights(lstm.h).contents
    U2 = U.copy()
    U2[300, 1234] += 0.75
    w.U[300 * 4 * H + 1234] = float(U2[300, 1234])           # in place, no SyncWeights
    got = lstm.apply(x)
    ref2 = O.lstm(x, W, U2, bi, bh, v2=True)
    assert np.abs(ref2 - O.lstm(x, W, U, bi, bh, v2=True)).max() > 1e-3      # the edit matters
    np.testing.assert_allclose(got, ref2, rtol=1e-5, atol=1e-5)
    lstm.destroy()
    # conv and dense: same contract
    cin, cout, k, Tc = 40, 128, 5, 64
    xc, Wc, bc = u(r, 2, Tc, cin), u(r, cout, cin, k, sc=0.1), u(r, cout, sc=0.1)
    conv = NL.Conv1d(cin, cout, k, 1, Tc)
    conv.set_weights(Wc, bc)
    conv.apply(xc)
    wc = capi.load().Conv1dGetWeights(conv.h).contents
    Wc2 = Wc.copy()
    Wc2[77, 21, 3] = 2.5
    wc.W[(77 * cin + 21) * k + 3] = 2.5
    np.testing.assert_allclose(conv.apply(xc), np.stack([O.conv1d(xi, Wc2, bc, 1) for xi in xc]), rtol=1e-5, atol=1e-5)
    conv.destroy()


def test_single_weight_edit_between_training_forward_and_gradient(gpu):
    """Forward and backward of a training step must use the same weights after an in-place edit of one float."""
    L = capi.load()
    r = rng(29)
    I, N, B = 96, 160, 32                                   # 15,360 + 160 floats: larger than the probe threshold
    x, W, b = u(r, B, I), u(r, I, N, sc=I ** -0.5), u(r, N, sc=0.1)
    cfg = L.DenseConfigCreate(I, N, None)
    d = L.DenseCreateForTraining(cfg, capi.ConvTrainingConfig(B))
    assert d
    w = L.DenseGetWeights(d).contents
    C.memmove(w.W, W.ctypes.data, W.nbytes); C.memmove(w.b, b.ctypes.data, b.nbytes)
    out = np.empty((B, N), np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    assert L.DenseApplyTrainingBatch(d, fp(x), fp(out)) == 0
    np.testing.assert_allclose(out, x @ W + b, rtol=1e-5, atol=1e-5)
    W2 = W.copy()
    W2[50, 77] = 3.0
    w.W[50 * N + 77] = 3.0
    assert L.DenseApplyTrainingBatch(d, fp(x), fp(out)) == 0
    np.testing.assert_allclose(out, x @ W2 + b, rtol=1e-5, atol=1e-5)
    # ... and the gradient's d_X = d_out W^T is taken with the edited weight too
    g = L.DenseGradientCreateFromFilter(d)
    dout = u(r, B, N)
    L.DenseCalculateGradient(d, g, fp(dout))
    dX = np.ctypeslib.as_array(g.contents.d_X, shape=(B, I))
    np.testing.assert_allclose(dX, dout @ W2.T, rtol=1e-4, atol=1e-5)
    L.DenseGradientDestroy(g)
    L.DenseDestroy(d)


# ------------------------------------------------------------ persistent kernel next to RCCL ---

def test_persistent_lstm_while_rccl_broadcasts_on_another_stream(gpu):
    """SURVEY 8(e) / VERDICT r02 #10: on the day of the 8-GPU run the persistent recurrent kernel shares its GPU with
    RCCL kernels.  Here a second host thread keeps a world-size-1 RCCL communicator busy (ncclBroadcast of a 32 MB
    block + the all-reduce barrier, on its own stream) while this thread runs persistent LSTM-512 launches that need
    all 256 CUs resident.  The results must match the oracle bit-identically to an undisturbed run, and no launch may
    have given up (nntk_hip_device_status() == 0)."""
    import torch
    L = capi.load()
    ident = (C.c_ubyte * 128)()
    if L.nntk_dist_get_unique_id(C.cast(ident, C.c_char_p)) != 0:
        pytest.skip("RCCL not loadable here: " + capi.last_error())
    assert L.nntk_dist_init(C.cast(ident, C.c_char_p), 0, 1) == 0, capi.last_error()
    try:
        r = rng(41)
        B, I, H, T = 512, 32, 512, 60                          # 8 batch tiles x 32 column tiles = 256 workgroups
        W, U, bi, bh = u(r, I, 4 * H, sc=I ** -0.5), u(r, H, 4 * H, sc=H ** -0.5), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1)
        x = u(r, B, T, I)
        lstm = NL.LSTM(I, H, True, T, v2=True)
        lstm.set_weights(W, U, bi, bh)
        xd = torch.from_numpy(x).cuda()
        quiet = lstm.apply_device(xd).clone()
        torch.cuda.synchronize()
        stop, errors, rounds = threading.Event(), [], [0]

        def traffic():
            try:
                torch.cuda.set_device(0)
                st = torch.cuda.Stream()
                L.nntk_hip_set_stream(C.c_void_p(st.cuda_stream))
                block = np.ones(8 << 20, np.float32)
                while not stop.is_set():
                    if L.nntk_dist_broadcast(block.ctypes.data_as(C.POINTER(C.c_float)), block.size, 0) != 0 or L.nntk_dist_barrier() != 0:
                        errors.append(capi.last_error())
                        return
                    rounds[0] += 1
            except Exception as e:                             # pragma: no cover
                errors.append(repr(e))

        th = threading.Thread(target=traffic)
        th.start()
        outs = []
        try:
            for _ in range(12):
                outs.append(lstm.apply_device(xd).clone())
            torch.cuda.synchronize()
        finally:
            stop.set()
            th.join(120)
        assert not errors, errors
        assert rounds[0] >= 1
        assert L.nntk_hip_device_status() == 0 and capi.get_option("rec_persistent") != 0
        for o in outs:
            assert torch.equal(o, quiet)
        ref = O.lstm(x[[0, 255, 511]], W, U, bi, bh, v2=True)
        np.testing.assert_allclose(quiet[[0, 255, 511]].cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
        print("persistent LSTM-512 x 12 launches next to %d RCCL broadcast rounds: bit-identical, no fault" % rounds[0])
        lstm.destroy()
    finally:
        assert L.nntk_dist_finalize() == 0


def test_gradient_allreduce_over_rccl_world_of_one(gpu):
    """nntk_dist_allreduce[_device]: the data-parallel training collective (in-place SUM of a gradient block over the ranks).
    One rank here (the box has one GPU): the RCCL call really runs -- in place, on the calling thread's stream -- and the sum
    over one rank leaves the block as it was; without a communicator the calls are no-ops.  A Dense gradient accumulated by
    DenseCalculateGradientDevice goes through it and then through the SGD step, all in HBM."""
    import torch
    L = capi.load()
    dp = lambda t: C.c_void_p(t.data_ptr())
    g = torch.arange(1 << 20, dtype=torch.float32, device="cuda") * 0.25
    keep = g.clone()
    assert L.nntk_dist_allreduce_device(dp(g), g.numel()) == 0           # no communicator: no-op
    ident = (C.c_ubyte * 128)()
    if L.nntk_dist_get_unique_id(C.cast(ident, C.c_char_p)) != 0:
        pytest.skip("RCCL not loadable here: " + capi.last_error())
    assert L.nntk_dist_init(C.cast(ident, C.c_char_p), 0, 1) == 0, capi.last_error()
    try:
        assert L.nntk_dist_allreduce_device(dp(g), g.numel()) == 0, capi.last_error()
        assert L.nntk_hip_synchronize() == 0
        assert torch.equal(g, keep)
        h = np.linspace(-1, 1, 4097, dtype=np.float32)
        h0 = h.copy()
        assert L.nntk_dist_allreduce(h.ctypes.data_as(capi.fp), h.size) == 0, capi.last_error()
        assert np.array_equal(h, h0)
        # a training step with its tensors in HBM: forward, gradient (accumulated on the device), all-reduce, SGD
        r = rng(8)
        B, n_in, n_out = 64, 48, 32
        x, W, b, dout = u(r, B, n_in), u(r, n_in, n_out, sc=0.3), u(r, n_out, sc=0.1), u(r, B, n_out)
        cfg = L.DenseConfigCreate(n_in, n_out, None)
        d = L.DenseCreateForTraining(cfg, capi.ConvTrainingConfig(B))
        w = L.DenseGetWeights(d).contents
        C.memmove(w.W, W.ctypes.data, W.nbytes); C.memmove(w.b, b.ctypes.data, b.nbytes)
        xd, dd = torch.from_numpy(x).cuda(), torch.from_numpy(dout).cuda()
        yd, gd, gx = torch.empty(B, n_out, device="cuda"), torch.zeros(n_in * n_out + n_out, device="cuda"), torch.empty(B, n_in, device="cuda")
        wd = torch.from_numpy(np.concatenate([W.ravel(), b])).cuda()
        assert L.DenseApplyTrainingBatchDevice(d, dp(xd), dp(yd)) == 0
        assert L.DenseCalculateGradientDevice(d, dp(gd), dp(gx), dp(dd)) == 0
        assert L.nntk_dist_allreduce_device(dp(gd), gd.numel()) == 0
        assert L.nntk_sgd_optimize_device(capi.SGD(0.1), dp(gd), dp(wd), gd.numel()) == 0
        assert L.nntk_hip_synchronize() == 0
        oW, ob, _ = O.dense_gradient(x, W, x @ W + b, x @ W + b, dout, act=None, softmax_vector_size=0)
        ref = np.concatenate([(W - np.float32(0.1) * oW).ravel(), b - np.float32(0.1) * ob])
        np.testing.assert_allclose(wd.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
        L.DenseDestroy(d)
    finally:
        assert L.nntk_dist_finalize() == 0


# ------------------------------------------------------------ stride-2 Conv1d on the MFMA kernels ---
# Reference: layers/conv_1d.c:128-140 (`input_row_offset = x * stride`).  Until round 3 every stride > 1 ran the
# one-thread-per-output VALU kernel; stride 2 (window of (128 - 1) * 2 + k rows per tile) now has its own MFMA instantiations.

@pytest.mark.parametrize("cin,cout,k,T,B", [
    (16, 48, 3, 99, 1),           # BN = 64 tile, ragged
    (40, 128, 5, 1000, 3),        # config 3 with stride 2: BN = 128 tile, 4 row tiles
    (257, 128, 5, 300, 2),        # odd channel count (ragged last chunk, 16-byte loads on 1028-byte rows)
    (24, 32, 9, 700, 2),          # BN = 32 tile, long kernel
    (33, 100, 4, 257, 1),         # Cout not a multiple of 32, Cin not a multiple of 4 (4-byte window loads)
])
def test_conv1d_stride2_mfma_matches_oracle_split_and_exact(gpu, cin, cout, k, T, B):
    r = rng(cin * 7 + cout)
    x, W, b = u(r, B, T, cin), u(r, cout, cin, k, sc=(cin * k) ** -0.5), u(r, cout, sc=0.2)
    conv = NL.Conv1d(cin, cout, k, 2, T)
    conv.set_weights(W, b)
    ref = np.stack([O.conv1d(xi, W, b, 2) for xi in x])
    got = conv.apply(x)
    capi.set_option("gemm_split_bf16", 0)
    exact = conv.apply(x)
    capi.set_option("gemm_split_bf16", "auto")
    assert got.shape == ref.shape == (B, (T - (k - 2)) // 2, cout)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(exact, ref, rtol=1e-5, atol=1e-5)
    conv.destroy()


def test_conv1d_stride2_fused_bn_relu_full_size(gpu):
    """Conv1d(40 -> 128, k = 5, stride 2) + BatchNorm + ReLU on 1024 x 1000 x 40 (config 3's sub-sampling variant):
    sampled utterances against the oracle."""
    import torch
    r = rng(3032)
    B, T, cin, cout, k = 1024, 1000, 40, 128, 5
    x = torch.randn(B, T, cin, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
    W, b = u(r, cout, cin, k, sc=(cin * k) ** -0.5), u(r, cout, sc=0.1)
    g, be, mu, var = 1 + u(r, cout, sc=0.5), u(r, cout, sc=0.5), u(r, cout, sc=0.1), 1 + u(r, cout, sc=0.5)
    conv = NL.Conv1d(cin, cout, k, 2, T)
    conv.set_weights(W, b)
    Tc = conv.out_shape[0]
    assert Tc == 498
    bn, relu = NL.BatchNorm(cout, 1e-3, Tc), NL.Activation("relu", Tc * cout, 1.0)
    bn.set_weights(g, be, mu, var)
    y = conv.apply_device(x, bn=bn, act=relu)
    for i in (0, 511, 1023):
        ref = O.activation(O.ACT_RELU, O.batch_norm(O.conv1d(x[i].cpu().numpy(), W, b, 2), g, be, mu, var, 1e-3))
        np.testing.assert_allclose(y[i].cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    # timing against the VALU kernel it replaces is in profiles/r03_conv_stride2.log (bench.py --workload conv --conv-stride 2)
    for o in (conv, bn, relu):
        o.destroy()
