"""Training path, second slice through the C boundary: activation gradients, Dense training forward and
DenseCalculateGradient, the losses and SGD -- against the oracle (reference operation order) and torch float64 autograd."""
import ctypes as C

import numpy as np
import pytest

from nntoolkitcore_amd import capi
import oracle as O

pytestmark = pytest.mark.gpu
P = lambda a: a.ctypes.data_as(capi.fp)


def rng(seed):
    return np.random.default_rng(seed)


def u(r, *shape, sc=1.0):
    return r.uniform(-sc, sc, shape).astype(np.float32)


ACTS = {"sigmoid": (O.ACT_SIGMOID, "ActivationFunctionCreateSigmoid"), "tanh": (O.ACT_TANH, "ActivationFunctionCreateTanh"),
        "identity": (O.ACT_IDENTITY, "ActivationFunctionCreateIdentity"), "relu": (O.ACT_RELU, None), "softmax": (O.ACT_SOFTMAX, None)}


def make_act(L, name, n, v=0):
    if name == "relu":
        return L.ActivationFunctionCreateReLU(n, 1.0)
    if name == "softmax":
        return L.ActivationFunctionCreateSoftmax(n // v, v)
    return getattr(L, ACTS[name][1])(n)


@pytest.mark.parametrize("name,n,v", [("sigmoid", 1000, 0), ("tanh", 777, 0), ("identity", 33, 0), ("relu", 4096, 0),
                                      ("softmax", 60, 10), ("softmax", 7, 7)])
@pytest.mark.parametrize("cached", [True, False])
def test_activation_gradient_is_bit_exact(gpu, name, n, v, cached):
    """ActivationFunctionCalculateGradient (activation.c:47-54): the derivative kernels mirror the reference's separately
    rounded operations, so given the same z / a they agree with the oracle BIT FOR BIT (sigmoid / tanh / softmax forward
    values recomputed on the device in the non-cached form are within the forward tolerance instead)."""
    L = capi.load()
    r = rng(n + 7 * cached)
    z = u(r, n, sc=2.0)
    kind = ACTS[name][0]
    a = O.activation(kind, z, softmax_vector_size=v)
    dout = u(r, n)
    h = make_act(L, name, n, v)
    out = np.empty(n, np.float32)
    L.ActivationFunctionCalculateGradient(h, P(z), P(a) if cached else None, P(dout), P(out))
    assert capi.last_error() == ""
    ref = O.activation_gradient(kind, z, a if cached else None, dout, softmax_vector_size=v)
    if cached or name in ("identity", "relu"):
        np.testing.assert_array_equal(out, ref)
    else:
        np.testing.assert_allclose(out, ref, rtol=2e-6, atol=2e-7)
    L.ActivationFunctionDestroy(h)


def test_relu_gradient_is_the_reference_clamp_not_a_step(gpu):
    L = capi.load()
    z = np.array([-1.0, 0.0, 0.25, 0.5, 1.0, 3.0], np.float32)
    dout = np.full(6, 2.0, np.float32)
    h = L.ActivationFunctionCreateReLU(6, 1.0)
    out = np.empty(6, np.float32)
    L.ActivationFunctionCalculateGradient(h, P(z), None, P(dout), P(out))
    np.testing.assert_array_equal(out, np.array([0, 0, 0.5, 1.0, 2.0, 2.0], np.float32))    # activation_default.c:118-121
    L.ActivationFunctionDestroy(h)


@pytest.mark.parametrize("B,n_in,n_out,act,v", [(5, 12, 7, None, 0), (16, 64, 32, "sigmoid", 0), (8, 33, 10, "softmax", 10),
                                                (32, 512, 1000, "tanh", 0), (3, 20, 24, "relu", 0), (64, 256, 40, "softmax", 40),
                                                (2048, 256, 320, "sigmoid", 0), (2603, 72, 1000, "tanh", 0)])
# the last two are large enough for the MFMA forms; (2603, 72, 1000): ragged tiles and an odd row count in outer_mfma_kernel
def test_dense_training_forward_and_gradient(gpu, B, n_in, n_out, act, v):
    import torch
    L = capi.load()
    r = rng(B * 13 + n_out)
    x = u(r, B, n_in)
    W, b = u(r, n_in, n_out, sc=n_in ** -0.5), u(r, n_out, sc=0.1)
    ah = make_act(L, act, n_out, v) if act else None
    kind = ACTS[act][0] if act else None
    cfg = L.DenseConfigCreate(n_in, n_out, ah)
    tc = capi.ConvTrainingConfig(B)
    h = L.DenseCreateForTraining(cfg, tc)
    w = L.DenseGetWeights(h).contents
    C.memmove(w.W, W.ctypes.data, W.nbytes); C.memmove(w.b, b.ctypes.data, b.nbytes)
    y = np.empty((B, n_out), np.float32)
    assert L.DenseApplyInference(h, P(x), P(y)) == -1                     # dense.c:136-138
    assert L.DenseApplyTrainingBatch(h, P(x), P(y)) == 0, capi.last_error()
    z, a = O.dense_forward_training(x, W, b, act=kind, softmax_vector_size=v)
    np.testing.assert_allclose(y, a, rtol=1e-5, atol=1e-6)
    dout = u(r, B, n_out)
    g = L.DenseGradientCreateFromFilter(h)
    L.DenseCalculateGradient(h, g, P(dout))
    assert capi.last_error() == ""
    gW = np.ctypeslib.as_array(g.contents.d_W, shape=(n_in, n_out)).copy()
    gb = np.ctypeslib.as_array(g.contents.d_b, shape=(n_out,)).copy()
    gX = np.ctypeslib.as_array(g.contents.d_X, shape=(B, n_in)).copy()
    oW, ob, oX = O.dense_gradient(x, W, z, a, dout, act=kind, softmax_vector_size=v)
    # torch float64 autograd of the same forward (ReLU: the reference's clamp derivative, so skip autograd there)
    refs = [("oracle", oW, ob, oX)]
    if act != "relu":
        xt, Wt, bt = (torch.tensor(t).double().requires_grad_(True) for t in (x, W, b))
        zt = xt @ Wt + bt
        at = {None: zt, "sigmoid": torch.sigmoid(zt), "tanh": torch.tanh(zt), "softmax": torch.softmax(zt, 1)}[act]
        at.backward(torch.tensor(dout).double())
        refs.append(("torch float64", Wt.grad.numpy(), bt.grad.numpy(), xt.grad.numpy()))
    tol = 2e-7 * np.sqrt(max(B, n_out))                  # measured on MI355X: <= 6.1e-8 x sqrt(max(B, n_out)) x scale; bound = that x 3.3
    for nm, rW, rb, rX in refs:
        for part, got, ref in (("dW", gW, rW), ("db", gb, rb), ("dX", gX, rX)):
            sc = max(1.0, float(np.abs(ref).max()))
            err = float(np.abs(got - ref).max())
            print("dense grad %s vs %s (%d,%d,%d,%s): %.2e (scale %.1f)" % (part, nm, B, n_in, n_out, act, err, sc))
            assert err <= tol * sc, (part, nm, err)
    # a second call accumulates d_W / d_b onto the block and rewrites d_X
    L.DenseCalculateGradient(h, g, P(dout))
    np.testing.assert_allclose(np.ctypeslib.as_array(g.contents.d_W, shape=(n_in, n_out)), 2 * gW, rtol=2e-6,
                               atol=1e-6 * max(1.0, float(np.abs(gW).max())))
    np.testing.assert_array_equal(np.ctypeslib.as_array(g.contents.d_X, shape=(B, n_in)), gX)
    # mode checks like the reference
    hi = L.DenseCreateForInference(cfg)
    assert L.DenseApplyTrainingBatch(hi, P(x), P(y)) == -1                # dense.c:145-147
    assert not L.DenseGradientCreateFromFilter(hi)                        # dense.c:107-109
    L.DenseDestroy(hi); L.DenseGradientDestroy(g); L.DenseDestroy(h)
    if ah: L.ActivationFunctionDestroy(ah)


def test_losses_and_sgd_follow_the_reference(gpu):
    L = capi.load()
    r = rng(3)
    B, c = 37, 19
    y = np.eye(c, dtype=np.float32)[r.integers(0, c, B)]
    p = O.activation(O.ACT_SOFTMAX, u(r, B, c, sc=2.0), softmax_vector_size=c).reshape(B, c)
    # MSE: per-sample sums in order on the device, batch sum in order on the host -> the oracle's value bit for bit
    assert L.mean_squared_error(P(y), P(p), c, B) == np.float32(O.mean_squared_error(y, p))
    d = np.empty_like(y)
    L.mean_squared_error_derivative(P(y), P(p), P(d), c, B)
    np.testing.assert_array_equal(d, O.mean_squared_error_derivative(y, p))
    # categorical cross-entropy: device logf is within 1 ulp of libm's
    got, ref = L.categorical_crossentropy(P(y), P(p), c, B), O.categorical_crossentropy(y, p)
    assert abs(got - ref) <= 2e-6 * abs(ref)
    L.categorical_crossentropy_derivative(P(y), P(p), P(d), c, B)
    o = O.categorical_crossentropy_derivative(y, p)                       # the reference only ever writes row 0
    np.testing.assert_array_equal(d[0], o[0])
    assert np.isnan(o[1:]).all()
    np.testing.assert_array_equal(d, -(y / p))                             # every row here
    # SGD: two roundings, no FMA
    n = 100003
    g, w = u(r, n), u(r, n)
    w2 = w.copy()
    assert L.sgd_optimize(capi.SGD(0.0371), P(g), P(w2), n) == 0, capi.last_error()
    np.testing.assert_array_equal(w2, O.sgd_optimize(0.0371, g, w))


def test_a_dense_softmax_head_trains_on_the_device_path(gpu):
    """End to end: DenseApplyTrainingBatch -> categorical_crossentropy(+derivative) -> DenseCalculateGradient ->
    sgd_optimize on the handle's own weight block (picked up by the next forward's edit check).  The loss must fall and
    the weights must follow the same trajectory as the oracle's loop within float tolerance."""
    L = capi.load()
    r = rng(11)
    B, n_in, c = 64, 20, 5
    true_W = u(r, n_in, c)
    x = u(r, B, n_in)
    labels = (x @ true_W).argmax(1)
    y = np.eye(c, dtype=np.float32)[labels]
    W, b = u(r, n_in, c, sc=0.1), np.zeros(c, np.float32)
    ah = L.ActivationFunctionCreateSoftmax(1, c)
    cfg = L.DenseConfigCreate(n_in, c, ah)
    h = L.DenseCreateForTraining(cfg, capi.ConvTrainingConfig(B))
    w = L.DenseGetWeights(h).contents
    C.memmove(w.W, W.ctypes.data, W.nbytes); C.memmove(w.b, b.ctypes.data, b.nbytes)
    oW, ob = W.copy(), b.copy()
    p, d = np.empty((B, c), np.float32), np.empty((B, c), np.float32)
    losses = []
    for it in range(30):
        assert L.DenseApplyTrainingBatch(h, P(x), P(p)) == 0, capi.last_error()
        losses.append(L.categorical_crossentropy(P(y), P(p), c, B))
        L.categorical_crossentropy_derivative(P(y), P(p), P(d), c, B)
        d /= B                                                             # mean over the batch (the caller's choice)
        g = L.DenseGradientCreateFromFilter(h)
        L.DenseCalculateGradient(h, g, P(d))
        assert L.sgd_optimize(capi.SGD(0.5), g.contents.d_W, w.W, n_in * c + c) == 0        # d_W | d_b and W | b are contiguous
        L.DenseGradientDestroy(g)
        # the same step with the oracle
        z_o, a_o = O.dense_forward_training(x, oW, ob, act=O.ACT_SOFTMAX, softmax_vector_size=c)
        d_o = (-(y / a_o) / B).astype(np.float32)
        gW, gb, _ = O.dense_gradient(x, oW, z_o, a_o, d_o, act=O.ACT_SOFTMAX, softmax_vector_size=c)
        oW, ob = O.sgd_optimize(0.5, gW, oW), O.sgd_optimize(0.5, gb, ob)
    assert losses[-1] < 0.6 * losses[0], losses
    np.testing.assert_allclose(np.ctypeslib.as_array(w.W, shape=(n_in, c)), oW, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(np.ctypeslib.as_array(w.b, shape=(c,)), ob, rtol=1e-4, atol=1e-5)
    L.DenseDestroy(h); L.ActivationFunctionDestroy(ah)


@pytest.mark.parametrize("count,mb,F", [(1, 8, 5), (13, 4, 7), (50, 16, 128), (996, 8, 128), (3, 2, 300)])
def test_batch_norm_training_forward_and_gradient(gpu, count, mb, F):
    """BatchNormCreateForTraining / ApplyTrainingBatch / CalculateGradient (batch_norm.c:191-386) against the oracle and
    torch float64 autograd.  Column sums are cut into slices here and are one chain in the scalar reference, so the
    comparison is at summation tolerance; the moving statistics land in the caller-visible weight block."""
    import torch
    L = capi.load()
    r = rng(count * 7 + F)
    N = count * mb
    x = (u(r, N, F, sc=2.0) + u(r, F, sc=1.0)).astype(np.float32)
    g, be = 1 + u(r, F, sc=0.5), u(r, F, sc=0.5)
    mm0, mv0 = u(r, F, sc=0.2), 1 + u(r, F, sc=0.3)
    eps, mom = 1e-3, 0.9
    cfg = L.BatchNormConfigCreate(F, eps, count)
    tc = L.BatchNormTrainingConfigCreate(mom, mb)
    h = L.BatchNormCreateForTraining(cfg, tc)
    w = L.BatchNormGetWeights(h).contents
    for dst, src in ((w.gamma, g), (w.beta, be), (w.moving_mean, mm0), (w.moving_variance, mv0)):
        C.memmove(dst, src.ctypes.data, src.nbytes)
    y = np.empty((N, F), np.float32)
    assert L.BatchNormApplyInference(h, P(x), P(y)) == -1                  # batch_norm.c:167-169
    assert L.BatchNormApplyTrainingBatch(h, P(x), P(y)) == 0, capi.last_error()
    o_y, o_mean, o_var, o_mm, o_mv = O.batch_norm_training_forward(x, g, be, eps, mom, mm0, mv0)
    tol = 2e-6 * max(1.0, np.sqrt(N) / 8)
    np.testing.assert_allclose(y, o_y, rtol=tol, atol=tol * 4)
    np.testing.assert_allclose(np.ctypeslib.as_array(w.moving_mean, shape=(F,)), o_mm, rtol=tol, atol=tol)
    np.testing.assert_allclose(np.ctypeslib.as_array(w.moving_variance, shape=(F,)), o_mv, rtol=tol, atol=tol)
    dout = u(r, N, F)
    gr = L.BatchNormGradientCreate(cfg, tc)
    L.BatchNormCalculateGradient(h, gr, P(dout))
    assert capi.last_error() == ""
    db = np.ctypeslib.as_array(gr.contents.d_beta, shape=(F,)).copy()
    dg = np.ctypeslib.as_array(gr.contents.d_gamma, shape=(F,)).copy()
    dx = np.ctypeslib.as_array(gr.contents.d_x, shape=(N, F)).copy()
    o_db, o_dg, o_dx = O.batch_norm_gradient(x, dout, g, o_mean, o_var, eps)
    xt, gt, bt = (torch.tensor(t).double().requires_grad_(True) for t in (x, g, be))
    mu, var = xt.mean(0), xt.var(0, unbiased=False)
    (((xt - mu) / torch.sqrt(var + eps)) * gt + bt).backward(torch.tensor(dout).double())
    gtol = 6e-6 * np.sqrt(N)
    for nm, got, ora, t64 in (("d_beta", db, o_db, bt.grad.numpy()), ("d_gamma", dg, o_dg, gt.grad.numpy()), ("d_x", dx, o_dx, xt.grad.numpy())):
        sc = max(1.0, float(np.abs(t64).max()))
        e_o, e_t = float(np.abs(got - ora).max()), float(np.abs(got - t64).max())
        print("bn grad %s (%d,%d,%d): vs oracle %.2e, vs torch float64 %.2e" % (nm, count, mb, F, e_o, e_t))
        assert e_o <= gtol * sc and e_t <= gtol * sc
    # the block is laid out d_beta | d_gamma | d_x (batch_norm.c:96-104) and a second call overwrites
    assert C.addressof(gr.contents.d_gamma.contents) - C.addressof(gr.contents.d_beta.contents) == 4 * F
    L.BatchNormCalculateGradient(h, gr, P(dout))
    np.testing.assert_array_equal(np.ctypeslib.as_array(gr.contents.d_beta, shape=(F,)), db)
    # inference handles refuse the training call; the inference path of a handle fed the trained statistics matches
    hi = L.BatchNormCreateForInference(cfg)
    assert L.BatchNormApplyTrainingBatch(hi, P(x), P(y)) == -1
    wi = L.BatchNormGetWeights(hi).contents
    for dst, src in ((wi.gamma, g), (wi.beta, be), (wi.moving_mean, o_mean), (wi.moving_variance, o_var)):
        C.memmove(dst, src.ctypes.data, src.nbytes)
    y2 = np.empty((count, F), np.float32)
    assert L.BatchNormApplyInference(hi, P(x[:count]), P(y2)) == 0
    np.testing.assert_allclose(y2, o_y[:count], rtol=tol, atol=tol * 4)
    L.BatchNormDestroy(hi); L.BatchNormGradientDestroy(gr); L.BatchNormDestroy(h)


@pytest.mark.parametrize("B,T,n_in,H,seq,acts", [
    (3, 7, 5, 4, True, ("sigmoid", "tanh", "sigmoid")),
    (4, 12, 16, 32, False, ("sigmoid", "tanh", "sigmoid")),
    (8, 50, 40, 64, True, ("sigmoid", "tanh", "sigmoid")),
    (2, 9, 6, 8, True, ("sigmoid", "relu", "tanh")),
    (16, 100, 128, 256, True, ("sigmoid", "tanh", "sigmoid")),
    # mini-batches of >= 32 sequences: the forward pass runs on the register-resident kernel (gru_rr_kernel<.., TRAIN>), which writes
    # the BPTT caches from its gate phase
    (32, 20, 64, 128, True, ("sigmoid", "tanh", "sigmoid")),
    (40, 15, 40, 64, False, ("sigmoid", "tanh", "sigmoid")),
    (64, 40, 128, 256, True, ("sigmoid", "tanh", "sigmoid")),
    (70, 9, 72, 192, True, ("sigmoid", "tanh", "sigmoid")),
    (48, 12, 200, 128, True, ("sigmoid", "tanh", "sigmoid")),          # in > 128: the <4, 4> instantiation
])
def test_gru_training_forward_and_bptt(gpu, B, T, n_in, H, seq, acts):
    """GRUCreateForTraining / ApplyTrainingBatch / GradientCreate / CalculateGradient (gru.c:232-512) through the C boundary
    against the oracle (the reference's loops) and, for the default activations, torch float64 autograd."""
    import torch
    L = capi.load()
    r = rng(B * 100 + T)
    x = u(r, B, T, n_in)
    W, U = u(r, n_in, 3 * H, sc=n_in ** -0.5), u(r, H, 3 * H, sc=H ** -0.5)
    bi, bh = u(r, 3 * H, sc=0.1), u(r, 3 * H, sc=0.1)
    ah = [make_act(L, a, H) for a in acts]                                   # z, h, r
    cfg = L.GRUConfigCreate(n_in, H, seq, T, L.GRUActivationsCreate(ah[0], ah[1], ah[2]))
    tc = capi.ConvTrainingConfig(B)
    h = L.GRUCreateForTraining(cfg, tc)
    w = L.GRUGetWeights(h).contents
    for dst, src in ((w.W, W), (w.U, U), (w.b_i, bi), (w.b_h, bh)):
        C.memmove(dst, src.ctypes.data, src.nbytes)
    n_out = (B, T, H) if seq else (B, H)
    y = np.empty(n_out, np.float32)
    assert L.GRUApplyInference(h, P(x), P(y)) == -1                          # gru.c:190-192
    assert L.GRUApplyTrainingBatch(h, P(x), P(y)) == 0, capi.last_error()
    dout = u(r, *n_out)
    kinds = tuple(ACTS[a][0] for a in acts)
    o_h, (oW, oU, obi, obh, oX) = O.gru_training(x, W, U, bi, bh, dout, return_sequences=seq, acts=kinds)
    np.testing.assert_allclose(y, o_h if seq else o_h[:, -1], rtol=2e-5, atol=2e-6)
    g = L.GRUGradientCreate(cfg, tc)
    L.GRUCalculateGradient(h, g, P(dout))
    assert capi.last_error() == ""
    gc = g.contents
    got = [np.ctypeslib.as_array(p_, shape=s).copy() for p_, s in ((gc.d_W, W.shape), (gc.d_U, U.shape), (gc.d_b_i, bi.shape),
                                                                   (gc.d_b_h, bh.shape), (gc.d_X, x.shape))]
    refs = [("oracle", (oW, oU, obi, obh, oX))]
    if acts == ("sigmoid", "tanh", "sigmoid"):
        xt, Wt, Ut, bit, bht = (torch.tensor(a).double().requires_grad_(True) for a in (x, W, U, bi, bh))
        hp = torch.zeros(B, H, dtype=torch.float64)
        outs = []
        for t in range(T):
            xw, hu = xt[:, t] @ Wt + bit, hp @ Ut + bht
            z, rr = torch.sigmoid(xw[:, :H] + hu[:, :H]), torch.sigmoid(xw[:, H:2 * H] + hu[:, H:2 * H])
            hp = (1 - z) * torch.tanh(rr * hu[:, 2 * H:] + xw[:, 2 * H:]) + z * hp
            outs.append(hp)
        hh = torch.stack(outs, 1)
        (hh if seq else hh[:, -1]).backward(torch.tensor(dout).double())
        refs.append(("torch float64", tuple(t_.grad.numpy() for t_ in (Wt, Ut, bit, bht, xt))))
    tol = 2e-7 * np.sqrt(B * T)                          # measured on MI355X: <= 5.0e-8 x sqrt(B T) x scale (oracle), 3.8e-8 (torch float64); x 4
    for nm, ref in refs:
        for part, a, b_ in zip(("dW", "dU", "dbi", "dbh", "dX"), got, ref):
            sc = max(1.0, float(np.abs(b_).max()))
            err = float(np.abs(a - b_).max())
            print("gru grad %s vs %s (%d,%d,%d,%d): %.2e (scale %.1f)" % (part, nm, B, T, n_in, H, err, sc))
            assert err <= tol * sc, (part, nm, err)
    # the block is d_W | d_U | d_b_i | d_b_h | d_X, and a second call accumulates the weight gradients
    assert C.addressof(gc.d_U.contents) - C.addressof(gc.d_W.contents) == 4 * W.size
    L.GRUCalculateGradient(h, g, P(dout))
    np.testing.assert_allclose(np.ctypeslib.as_array(gc.d_U, shape=U.shape), 2 * got[1], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(got[1]).max())))
    np.testing.assert_array_equal(np.ctypeslib.as_array(gc.d_X, shape=x.shape), got[4])
    hi = L.GRUCreateForInference(cfg)
    assert L.GRUApplyTrainingBatch(hi, P(x), P(y)) == -1                     # gru.c:247-249
    L.GRUDestroy(hi); L.RecurrentGradientDestroy(g); L.GRUDestroy(h)
    for a in ah: L.ActivationFunctionDestroy(a)


@pytest.mark.parametrize("B,T,n_in,H,seq,v2,acts", [
    (3, 7, 5, 4, True, True, None),
    (4, 12, 16, 32, False, True, None),
    (8, 50, 40, 64, True, False, None),
    (2, 9, 6, 8, True, True, ("sigmoid", "tanh", "relu", "sigmoid", "sigmoid")),
    (16, 100, 128, 256, True, True, None),
    # mini-batches of >= 32 sequences: the forward pass runs on the register-resident kernel (lstm_rr_kernel<.., TRAIN>),
    # which writes the BPTT caches from its gate phase
    (32, 20, 64, 128, True, True, None),
    (40, 15, 40, 64, False, False, None),
    (64, 60, 128, 512, True, True, None),
    (70, 9, 72, 192, True, True, None),
    (48, 12, 200, 128, True, True, None),          # in > 128: the <4, 4, TRAIN> instantiation
])
def test_lstm_training_forward_and_bptt(gpu, B, T, n_in, H, seq, v2, acts):
    """LSTMCreateForTraining / ApplyTrainingBatch / GradientCreate / CalculateGradient (lstm.c:294-556) against the oracle
    and, for the default activations, torch float64 autograd."""
    import torch
    L = capi.load()
    r = rng(B * 100 + T + 1)
    default = acts is None
    acts = acts or ("sigmoid", "sigmoid", "tanh", "sigmoid", "tanh")           # input, forget, candidate, output gate, output
    x = u(r, B, T, n_in)
    W, U = u(r, n_in, 4 * H, sc=n_in ** -0.5), u(r, H, 4 * H, sc=H ** -0.5)
    bi, bh = u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1)
    ah = [make_act(L, a, H) for a in acts]
    cfg = L.LSTMConfigCreate(n_in, H, seq, T, v2, L.LSTMActivationsCreate(*ah))
    tc = capi.ConvTrainingConfig(B)
    h = L.LSTMCreateForTraining(cfg, tc)
    w = L.LSTMGetWeights(h).contents
    for dst, src in ((w.W, W), (w.U, U), (w.b_i, bi), (w.b_h, bh)):
        C.memmove(dst, src.ctypes.data, src.nbytes)
    n_out = (B, T, H) if seq else (B, H)
    y = np.empty(n_out, np.float32)
    assert L.LSTMApplyInference(h, P(x), P(y)) == -1                         # lstm.c:242-244
    assert L.LSTMApplyTrainingBatch(h, P(x), P(y)) == 0, capi.last_error()
    dout = u(r, *n_out)
    kinds = tuple(ACTS[a][0] for a in acts)
    o_h, oref = O.lstm_training(x, W, U, bi, bh, dout, return_sequences=seq, v2=v2, acts=kinds)
    np.testing.assert_allclose(y, o_h if seq else o_h[:, -1], rtol=2e-5, atol=2e-6)
    g = L.LSTMGradientCreate(cfg, tc)
    L.LSTMCalculateGradient(h, g, P(dout))
    assert capi.last_error() == ""
    gc = g.contents
    got = [np.ctypeslib.as_array(p_, shape=s).copy() for p_, s in ((gc.d_W, W.shape), (gc.d_U, U.shape), (gc.d_b_i, bi.shape),
                                                                   (gc.d_b_h, bh.shape), (gc.d_X, x.shape))]
    refs = [("oracle", oref)]
    if default:
        xt, Wt, Ut, bit, bht = (torch.tensor(a).double().requires_grad_(True) for a in (x, W, U, bi, bh))
        hp, cp, outs = torch.zeros(B, H, dtype=torch.float64), torch.zeros(B, H, dtype=torch.float64), []
        for t in range(T):
            Z = xt[:, t] @ Wt + bit + hp @ Ut + (bht if v2 else 0)
            i, f, g_, o = torch.sigmoid(Z[:, :H]), torch.sigmoid(Z[:, H:2 * H]), torch.tanh(Z[:, 2 * H:3 * H]), torch.sigmoid(Z[:, 3 * H:])
            cp = f * cp + i * g_
            hp = o * torch.tanh(cp)
            outs.append(hp)
        hh = torch.stack(outs, 1)
        (hh if seq else hh[:, -1]).backward(torch.tensor(dout).double())
        # without v2 the forward never reads b_h, but the reference still reports d_b_h = dgates (lstm.c:415)
        refs.append(("torch float64", (Wt.grad.numpy(), Ut.grad.numpy(), bit.grad.numpy(), bht.grad.numpy() if v2 else bit.grad.numpy(), xt.grad.numpy())))
    tol = 2e-7 * np.sqrt(B * T)                          # measured on MI355X: <= 4.6e-8 x sqrt(B T) x scale (oracle), 3.8e-8 (torch float64); x 4
    for nm, ref in refs:
        for part, a, b_ in zip(("dW", "dU", "dbi", "dbh", "dX"), got, ref):
            sc = max(1.0, float(np.abs(b_).max()))
            err = float(np.abs(a - b_).max())
            print("lstm grad %s vs %s (%d,%d,%d,%d): %.2e (scale %.1f)" % (part, nm, B, T, n_in, H, err, sc))
            assert err <= tol * sc, (part, nm, err)
    L.LSTMCalculateGradient(h, g, P(dout))
    np.testing.assert_allclose(np.ctypeslib.as_array(gc.d_U, shape=U.shape), 2 * got[1], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(got[1]).max())))
    np.testing.assert_array_equal(np.ctypeslib.as_array(gc.d_X, shape=x.shape), got[4])
    hi = L.LSTMCreateForInference(cfg)
    assert L.LSTMApplyTrainingBatch(hi, P(x), P(y)) == -1                    # lstm.c:419-421
    L.LSTMDestroy(hi); L.RecurrentGradientDestroy(g); L.LSTMDestroy(h)
    for a in ah: L.ActivationFunctionDestroy(a)


def _recurrent_gradient(kind, B, T, n_in, H, seq, seed, inject_fault=False):
    """one training forward + gradient call of a GRU / LSTM with default activations: (d_W, d_U, d_b_i, d_b_h, d_X), error text"""
    L = capi.load()
    r = rng(seed)
    ng = 4 if kind == "lstm" else 3
    x = u(r, B, T, n_in)
    W, U = u(r, n_in, ng * H, sc=n_in ** -0.5), u(r, H, ng * H, sc=H ** -0.5)
    bi, bh = u(r, ng * H, sc=0.1), u(r, ng * H, sc=0.1)
    tc = capi.ConvTrainingConfig(B)
    if kind == "lstm":
        ah = [make_act(L, a, H) for a in ("sigmoid", "sigmoid", "tanh", "sigmoid", "tanh")]
        cfg = L.LSTMConfigCreate(n_in, H, seq, T, True, L.LSTMActivationsCreate(*ah))
        h = L.LSTMCreateForTraining(cfg, tc); w = L.LSTMGetWeights(h).contents
        apply_, grad_new, grad, destroy = L.LSTMApplyTrainingBatch, L.LSTMGradientCreate, L.LSTMCalculateGradient, L.LSTMDestroy
    else:
        ah = [make_act(L, a, H) for a in ("sigmoid", "tanh", "sigmoid")]
        cfg = L.GRUConfigCreate(n_in, H, seq, T, L.GRUActivationsCreate(*ah))
        h = L.GRUCreateForTraining(cfg, tc); w = L.GRUGetWeights(h).contents
        apply_, grad_new, grad, destroy = L.GRUApplyTrainingBatch, L.GRUGradientCreate, L.GRUCalculateGradient, L.GRUDestroy
    for dst, src in ((w.W, W), (w.U, U), (w.b_i, bi), (w.b_h, bh)):
        C.memmove(dst, src.ctypes.data, src.nbytes)
    n_out = (B, T, H) if seq else (B, H)
    y = np.empty(n_out, np.float32)
    assert apply_(h, P(x), P(y)) == 0, capi.last_error()
    dout = u(r, *n_out)
    g = grad_new(cfg, tc)
    if inject_fault: capi.set_option("rec_spin_us", 0)                        # the gradient call only: every poll gives up at once
    grad(h, g, P(dout))
    err = capi.last_error()
    if inject_fault: capi.set_option("rec_spin_us", "auto")
    gc = g.contents
    got = [np.ctypeslib.as_array(p_, shape=s_).copy() for p_, s_ in ((gc.d_W, W.shape), (gc.d_U, U.shape), (gc.d_b_i, bi.shape),
                                                                     (gc.d_b_h, bh.shape), (gc.d_X, x.shape))]
    L.RecurrentGradientDestroy(g); destroy(h)
    for a in ah: L.ActivationFunctionDestroy(a)
    return got, err


@pytest.mark.parametrize("kind,B,T,n_in,H,seq", [("lstm", 64, 40, 128, 512, True), ("lstm", 37, 12, 40, 64, False),
                                                 ("gru", 64, 40, 128, 256, True), ("gru", 20, 9, 16, 64, False),
                                                 ("lstm", 256, 6, 64, 512, True),      # 512 workgroups: two per CU
                                                 ("lstm", 300, 5, 64, 512, True),      # 608 workgroups: not co-resident -> per-step loop both ways
                                                 ("gru", 33, 2, 24, 64, True)])        # T = 2: one hand-off
def test_persistent_bptt_equals_the_per_step_loop(gpu, kind, B, T, n_in, H, seq):
    """train.hip bptt_persistent_kernel (the whole backward-through-time loop in one launch, carries in registers, d_gates
    exchanged through HBM) against the two-launches-per-step loop it replaces (option train_bptt = 0): the same elementwise
    code and a product that differs only in how K is chunked, so the two agree to a few roundings."""
    L = capi.load()
    a, e1 = _recurrent_gradient(kind, B, T, n_in, H, seq, 4242)
    capi.set_option("train_bptt", 0)
    b_, e2 = _recurrent_gradient(kind, B, T, n_in, H, seq, 4242)
    capi.set_option("train_bptt", "auto")
    assert e1 == "" and e2 == ""
    for nm, p_, q_ in zip(("dW", "dU", "dbi", "dbh", "dX"), a, b_):
        sc = max(1.0, float(np.abs(q_).max()))
        err = float(np.abs(p_ - q_).max())
        print("persistent vs per-step BPTT %s %s: %.2e (scale %.1f)" % (kind, nm, err, sc))
        assert err <= 2e-5 * sc, (nm, err)


def test_persistent_bptt_fault_is_reported(gpu):
    """A spin that runs out of budget (rec_spin_us = 0 injects it) must not hand back a gradient silently: the call leaves an
    error, the process moves to the per-step loop, and re-arming (rec_persistent = 1) brings the kernel back."""
    L = capi.load()
    good, e0 = _recurrent_gradient("lstm", 32, 10, 64, 128, True, 99)
    assert e0 == ""
    _, e1 = _recurrent_gradient("lstm", 32, 10, 64, 128, True, 99, inject_fault=True)
    assert "timed out" in e1
    after, e2 = _recurrent_gradient("lstm", 32, 10, 64, 128, True, 99)      # per-step loop now
    assert e2 == ""
    capi.set_option("rec_persistent", 1)                                     # re-arm
    again, e3 = _recurrent_gradient("lstm", 32, 10, 64, 128, True, 99)
    capi.set_option("rec_persistent", "auto")
    assert e3 == "" and L.nntk_hip_synchronize() == 0
    for p_, q_, s_ in zip(good, after, again):
        sc = max(1.0, float(np.abs(p_).max()))
        assert float(np.abs(p_ - q_).max()) <= 2e-5 * sc
        np.testing.assert_array_equal(p_, s_)


@pytest.mark.parametrize("B,T,n_in,H,seq,v2,act", [(3, 7, 5, 4, True, True, "tanh"), (4, 20, 16, 32, False, False, "tanh"),
                                                   (8, 60, 40, 64, True, True, "sigmoid"), (2, 9, 6, 8, True, True, "relu")])
def test_rnn_training_forward_and_bptt(gpu, B, T, n_in, H, seq, v2, act):
    """RNNCreateForTraining / ApplyTrainingBatch / GradientCreate / CalculateGradient (rnn.c:184-351)."""
    import torch
    L = capi.load()
    r = rng(B * 10 + T)
    x = u(r, B, T, n_in)
    W, U, bi, bh = u(r, n_in, H, sc=n_in ** -0.5), u(r, H, H, sc=H ** -0.5), u(r, H, sc=0.1), u(r, H, sc=0.1)
    ah = make_act(L, act, H)
    cfg = L.RNNConfigCreate(n_in, H, seq, T, v2, ah)
    tc = capi.ConvTrainingConfig(B)
    h = L.RNNCreateForTraining(cfg, tc)
    w = L.RNNGetWeights(h).contents
    for dst, src in ((w.W, W), (w.U, U), (w.b_i, bi), (w.b_h, bh)):
        C.memmove(dst, src.ctypes.data, src.nbytes)
    n_out = (B, T, H) if seq else (B, H)
    y = np.empty(n_out, np.float32)
    assert L.RNNApplyInference(h, P(x), P(y)) == -1
    assert L.RNNApplyTrainingBatch(h, P(x), P(y)) == 0, capi.last_error()
    dout = u(r, *n_out)
    o_h, oref = O.rnn_training(x, W, U, bi, bh, dout, return_sequences=seq, v2=v2, act=ACTS[act][0])
    np.testing.assert_allclose(y, o_h if seq else o_h[:, -1], rtol=2e-5, atol=2e-6)
    g = L.RNNGradientCreate(cfg, tc)
    L.RNNCalculateGradient(h, g, P(dout))
    assert capi.last_error() == ""
    gc = g.contents
    got = [np.ctypeslib.as_array(p_, shape=s).copy() for p_, s in ((gc.d_W, W.shape), (gc.d_U, U.shape), (gc.d_b_i, bi.shape),
                                                                   (gc.d_b_h, bh.shape), (gc.d_X, x.shape))]
    refs = [("oracle", oref)]
    if act != "relu":
        xt, Wt, Ut, bit, bht = (torch.tensor(a).double().requires_grad_(True) for a in (x, W, U, bi, bh))
        f = torch.tanh if act == "tanh" else torch.sigmoid
        hp, outs = torch.zeros(B, H, dtype=torch.float64), []
        for t in range(T):
            hp = f(xt[:, t] @ Wt + bit + hp @ Ut + (bht if v2 else 0))
            outs.append(hp)
        hh = torch.stack(outs, 1)
        (hh if seq else hh[:, -1]).backward(torch.tensor(dout).double())
        refs.append(("torch float64", (Wt.grad.numpy(), Ut.grad.numpy(), bit.grad.numpy(), bht.grad.numpy() if v2 else bit.grad.numpy(), xt.grad.numpy())))
    tol = 2e-7 * np.sqrt(B * T)                          # measured on MI355X: <= 4.1e-8 x sqrt(B T) x scale; x 4
    for nm, ref in refs:
        for part, a, b_ in zip(("dW", "dU", "dbi", "dbh", "dX"), got, ref):
            sc = max(1.0, float(np.abs(b_).max()))
            err = float(np.abs(a - b_).max())
            print("rnn grad %s vs %s (%d,%d,%d,%d): %.2e (scale %.1f)" % (part, nm, B, T, n_in, H, err, sc))
            assert err <= tol * sc, (part, nm, err)
    hi = L.RNNCreateForInference(cfg)
    assert L.RNNApplyTrainingBatch(hi, P(x), P(y)) == -1
    L.RNNDestroy(hi); L.RecurrentGradientDestroy(g); L.RNNDestroy(h); L.ActivationFunctionDestroy(ah)


def test_time_distributed_dense_training_is_dense_over_all_rows(gpu):
    """time_distributed_dense.c:38-67: a Dense trained on mini_batch * ts rows."""
    L = capi.load()
    r = rng(77)
    B, ts, n_in, n_out = 4, 9, 12, 10
    x, W, b = u(r, B * ts, n_in), u(r, n_in, n_out, sc=0.3), u(r, n_out, sc=0.1)
    ah = L.ActivationFunctionCreateSoftmax(1, n_out)
    cfg = L.TimeDistributedDenseConfigCreate(ts, L.DenseConfigCreate(n_in, n_out, ah))
    h = L.TimeDistributedDenseCreateForTraining(cfg, capi.ConvTrainingConfig(B))
    w = L.TimeDistributedDenseGetWeights(h).contents
    C.memmove(w.W, W.ctypes.data, W.nbytes); C.memmove(w.b, b.ctypes.data, b.nbytes)
    y = np.empty((B * ts, n_out), np.float32)
    assert L.TimeDistributedDenseApplyInference(h, P(x), P(y)) == -1
    assert L.TimeDistributedDenseApplyTrainingBatch(h, P(x), P(y)) == 0, capi.last_error()
    z, a = O.dense_forward_training(x, W, b, act=O.ACT_SOFTMAX, softmax_vector_size=n_out)
    np.testing.assert_allclose(y, a, rtol=1e-5, atol=1e-6)
    dout = u(r, B * ts, n_out)
    g = L.TimeDistributedDenseGradientCreate(h)
    L.TimeDistributedDenseCalculateGradient(h, g, P(dout))
    assert capi.last_error() == ""
    oW, ob, oX = O.dense_gradient(x, W, z, a, dout, act=O.ACT_SOFTMAX, softmax_vector_size=n_out)
    np.testing.assert_allclose(np.ctypeslib.as_array(g.contents.d_W, shape=W.shape), oW, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(np.ctypeslib.as_array(g.contents.d_b, shape=b.shape), ob, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(np.ctypeslib.as_array(g.contents.d_X, shape=x.shape), oX, rtol=1e-5, atol=2e-6)
    L.DenseGradientDestroy(g); L.TimeDistributedDenseDestroy(h); L.ActivationFunctionDestroy(ah)


def test_bidirectional_gradient_helpers(gpu):
    """bd_merge_concat_gradient / bd_merge_sum_gradient / bd_accumulate_d_x (bidirectional.c:58-108): the inverses of the
    forward helpers -- checked against them and against plain numpy."""
    L = capi.load()
    r = rng(5)
    B, T, n_in, H = 3, 7, 4, 6
    for seq in (True, False):
        cfg = L.RecurrentConfigCreate(n_in, H, seq, T)
        rows = T if seq else 1
        d_out = u(r, B, rows, 2 * H)
        f, b = np.empty((B, rows, H), np.float32), np.empty((B, rows, H), np.float32)
        L.bd_merge_concat_gradient(P(d_out), P(f), P(b), cfg, B, None)
        assert capi.last_error() == ""
        np.testing.assert_array_equal(f, d_out[..., :H]); np.testing.assert_array_equal(b, d_out[..., H:])
        back = np.empty_like(d_out)
        L.bd_merge_concat(P(f), P(b), P(back), cfg, B, None)
        np.testing.assert_array_equal(back, d_out)                         # gradient of concat = its inverse
        d_sum = u(r, B, rows, H)
        L.bd_merge_sum_gradient(P(d_sum), P(f), P(b), cfg, B)
        np.testing.assert_array_equal(f, d_sum); np.testing.assert_array_equal(b, d_sum)
    fx, bx = u(r, B, T, n_in), u(r, B, T, n_in)
    out = np.empty_like(fx)
    L.bd_accumulate_d_x(P(fx), P(bx), P(out), cfg, B)
    np.testing.assert_array_equal(out, fx + bx[:, ::-1])


# ---- device-pointer forms of the Dense / TimeDistributedDense / BatchNorm training calls (additive) ----

@pytest.mark.parametrize("kind,B,T,n_in,H,seq", [("lstm", 32, 12, 64, 128, True), ("lstm", 5, 9, 12, 16, False),
                                                 ("gru", 16, 20, 32, 64, True), ("gru", 3, 7, 5, 4, False)])
def test_recurrent_training_device_forms_equal_the_host_forms(gpu, kind, B, T, n_in, H, seq):
    """GRU / LSTM ApplyTrainingBatchDevice + CalculateGradientDevice run the same device cores on the caller's HBM tensors:
    bit-identical to the host-pointer calls, and the gradient block is accumulated onto."""
    import torch
    L = capi.load()
    r = rng(B * 7 + H)
    ng = 4 if kind == "lstm" else 3
    x = u(r, B, T, n_in)
    Wall = u(r, n_in * ng * H + H * ng * H + 2 * ng * H, sc=0.2)
    tc = capi.ConvTrainingConfig(B)
    if kind == "lstm":
        cfg = L.LSTMConfigCreate(n_in, H, seq, T, True, L.LSTMActivationsCreateDefault(H))
        h = L.LSTMCreateForTraining(cfg, tc); w = L.LSTMGetWeights(h).contents; g = L.LSTMGradientCreate(cfg, tc)
        fw, bw, fwd, bwd, de = L.LSTMApplyTrainingBatch, L.LSTMCalculateGradient, L.LSTMApplyTrainingBatchDevice, L.LSTMCalculateGradientDevice, L.LSTMDestroy
    else:
        cfg = L.GRUConfigCreate(n_in, H, seq, T, L.GRUActivationsCreateDefault(H))
        h = L.GRUCreateForTraining(cfg, tc); w = L.GRUGetWeights(h).contents; g = L.GRUGradientCreate(cfg, tc)
        fw, bw, fwd, bwd, de = L.GRUApplyTrainingBatch, L.GRUCalculateGradient, L.GRUApplyTrainingBatchDevice, L.GRUCalculateGradientDevice, L.GRUDestroy
    C.memmove(w.W, Wall.ctypes.data, Wall.nbytes)                            # W | U | b_i | b_h are one block
    n_out = (B, T, H) if seq else (B, H)
    y, dout = np.empty(n_out, np.float32), u(r, *n_out)
    assert fw(h, P(x), P(y)) == 0, capi.last_error()
    bw(h, g, P(dout))
    assert capi.last_error() == ""
    gW = np.ctypeslib.as_array(g.contents.d_W, shape=(Wall.size,)).copy()
    gX = np.ctypeslib.as_array(g.contents.d_X, shape=x.shape).copy()
    dp = lambda t: C.c_void_p(t.data_ptr())
    xd, dd = torch.from_numpy(x).cuda(), torch.from_numpy(dout).cuda()
    yd, gd, gxd = torch.empty(*n_out, device="cuda"), torch.zeros(Wall.size, device="cuda"), torch.empty(*x.shape, device="cuda")
    assert fwd(h, dp(xd), dp(yd)) == 0, capi.last_error()
    assert bwd(h, dp(gd), dp(gxd), dp(dd)) == 0, capi.last_error()
    assert L.nntk_hip_synchronize() == 0
    assert np.array_equal(yd.cpu().numpy(), y) and np.array_equal(gd.cpu().numpy(), gW) and np.array_equal(gxd.cpu().numpy(), gX)
    assert bwd(h, dp(gd), dp(gxd), dp(dd)) == 0 and L.nntk_hip_synchronize() == 0
    np.testing.assert_allclose(gd.cpu().numpy(), 2 * gW, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(gW).max())))
    assert bwd(h, None, dp(gxd), dp(dd)) == -1                                # NULL argument
    L.RecurrentGradientDestroy(g); de(h)


@pytest.mark.parametrize("B,T,Cin,Cout,k,s", [(4, 60, 8, 16, 5, 1), (12, 450, 40, 128, 5, 1), (6, 90, 12, 24, 3, 2)])
def test_conv1d_training_device_forms_equal_the_host_forms(gpu, B, T, Cin, Cout, k, s):
    """Conv1dApplyTrainingBatchDevice / Conv1dCalculateGradientDevice: the host forms' device core on the caller's HBM tensors
    (small VALU shapes, the MFMA shapes, stride 2): bit-identical, and the gradient block accumulates."""
    import torch
    L = capi.load()
    r = rng(B + T + Cout)
    x = u(r, B, T, Cin)
    W, b = u(r, Cout, Cin, k, sc=(Cin * k) ** -0.5), u(r, Cout, sc=0.1)
    cfg = L.Conv1dConfigCreate(Cin, Cout, k, s, T)
    tc = capi.ConvTrainingConfig(B)
    h = L.Conv1dCreateForTraining(cfg, tc)
    w = L.Conv1dGetWeights(h).contents
    C.memmove(w.W, W.ctypes.data, W.nbytes); C.memmove(w.b, b.ctypes.data, b.nbytes)
    Tout = cfg.output_size
    y, dout = np.empty((B, Tout, Cout), np.float32), u(r, B, Tout, Cout)
    assert L.Conv1dApplyTrainingBatch(h, P(x), P(y)) == 0, capi.last_error()
    g = L.Conv1dCreateGradient(cfg, tc)
    L.Conv1dCalculateGradient(h, g, P(dout))
    assert capi.last_error() == ""
    nW = Cout * Cin * k
    gWb = np.ctypeslib.as_array(g.contents.d_W, shape=(nW + Cout,)).copy()
    gX = np.ctypeslib.as_array(g.contents.d_X, shape=(B, T, Cin)).copy()
    dp = lambda t: C.c_void_p(t.data_ptr())
    xd, dd = torch.from_numpy(x).cuda(), torch.from_numpy(dout).cuda()
    yd, gd, gxd = torch.empty(B, Tout, Cout, device="cuda"), torch.zeros(nW + Cout, device="cuda"), torch.empty(B, T, Cin, device="cuda")
    assert L.Conv1dApplyTrainingBatchDevice(h, dp(xd), dp(yd)) == 0, capi.last_error()
    assert L.Conv1dCalculateGradientDevice(h, dp(gd), dp(gxd), dp(dd)) == 0, capi.last_error()
    assert L.nntk_hip_synchronize() == 0
    assert np.array_equal(yd.cpu().numpy(), y) and np.array_equal(gd.cpu().numpy(), gWb) and np.array_equal(gxd.cpu().numpy(), gX)
    assert L.Conv1dCalculateGradientDevice(h, dp(gd), dp(gxd), dp(dd)) == 0 and L.nntk_hip_synchronize() == 0
    np.testing.assert_allclose(gd.cpu().numpy(), 2 * gWb, rtol=1e-6, atol=1e-6 * max(1.0, float(np.abs(gWb).max())))
    assert L.Conv1dCalculateGradientDevice(h, None, dp(gxd), dp(dd)) == -1
    L.ConvGradientDestroy(g); L.Conv1dDestroy(h)


def test_dense_and_batchnorm_training_device_forms_equal_the_host_forms(gpu):
    """The device-pointer calls run the same kernels on the caller's HBM buffers: results are bit-identical to the
    host-pointer forms (which upload, call the same core and download)."""
    import torch
    L = capi.load()
    r = rng(77)
    dp = lambda t: C.c_void_p(t.data_ptr())
    # Dense with a sigmoid
    B, n_in, n_out = 96, 40, 24
    x, W, b, dout = u(r, B, n_in), u(r, n_in, n_out, sc=0.3), u(r, n_out, sc=0.1), u(r, B, n_out)
    ah = L.ActivationFunctionCreateSigmoid(n_out)
    cfg = L.DenseConfigCreate(n_in, n_out, ah)
    h = L.DenseCreateForTraining(cfg, capi.ConvTrainingConfig(B))
    w = L.DenseGetWeights(h).contents
    C.memmove(w.W, W.ctypes.data, W.nbytes); C.memmove(w.b, b.ctypes.data, b.nbytes)
    y = np.empty((B, n_out), np.float32)
    assert L.DenseApplyTrainingBatch(h, P(x), P(y)) == 0
    g = L.DenseGradientCreateFromFilter(h)
    L.DenseCalculateGradient(h, g, P(dout))
    gW = np.ctypeslib.as_array(g.contents.d_W, shape=(n_in * n_out + n_out,)).copy()
    gX = np.ctypeslib.as_array(g.contents.d_X, shape=(B, n_in)).copy()
    xd, dd = torch.from_numpy(x).cuda(), torch.from_numpy(dout).cuda()
    yd, gWd, gXd = torch.empty(B, n_out, device="cuda"), torch.zeros(n_in * n_out + n_out, device="cuda"), torch.empty(B, n_in, device="cuda")
    assert L.DenseApplyTrainingBatchDevice(h, dp(xd), dp(yd)) == 0, capi.last_error()
    assert L.DenseCalculateGradientDevice(h, dp(gWd), dp(gXd), dp(dd)) == 0, capi.last_error()
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy(), y) and np.array_equal(gWd.cpu().numpy(), gW) and np.array_equal(gXd.cpu().numpy(), gX)
    z_o, a_o = O.dense_forward_training(x, W, b, act=O.ACT_SIGMOID)
    np.testing.assert_allclose(y, a_o, rtol=1e-5, atol=1e-5)
    # accumulation onto the device gradient block: a second call adds the same amount
    assert L.DenseCalculateGradientDevice(h, dp(gWd), dp(gXd), dp(dd)) == 0
    np.testing.assert_allclose(gWd.cpu().numpy(), 2 * gW, rtol=1e-6, atol=1e-6)
    L.DenseGradientDestroy(g); L.DenseDestroy(h); L.ActivationFunctionDestroy(ah)
    # BatchNorm
    count, mb, F = 50, 8, 64
    N = count * mb
    x, dout = u(r, N, F), u(r, N, F)
    bcfg = L.BatchNormConfigCreate(F, 1e-3, count)
    btc = L.BatchNormTrainingConfigCreate(0.9, mb)
    hb = L.BatchNormCreateForTraining(bcfg, btc)
    wb = L.BatchNormGetWeights(hb).contents
    gam, bet = (1 + u(r, F, sc=0.3)), u(r, F, sc=0.3)
    C.memmove(wb.gamma, gam.ctypes.data, gam.nbytes); C.memmove(wb.beta, bet.ctypes.data, bet.nbytes)
    y = np.empty((N, F), np.float32)
    assert L.BatchNormApplyTrainingBatch(hb, P(x), P(y)) == 0
    gb = L.BatchNormGradientCreate(bcfg, btc)
    L.BatchNormCalculateGradient(hb, gb, P(dout))
    ref = [np.ctypeslib.as_array(getattr(gb.contents, k), shape=s).copy() for k, s in (("d_beta", (F,)), ("d_gamma", (F,)), ("d_x", (N, F)))]
    mm = np.ctypeslib.as_array(wb.moving_mean, shape=(F,)).copy()
    # same handle, device forms (the moving statistics advance once more: compared against a second host call below)
    xd, dd = torch.from_numpy(x).cuda(), torch.from_numpy(dout).cuda()
    yd = torch.empty(N, F, device="cuda")
    dbe, dga, dxx = torch.empty(F, device="cuda"), torch.empty(F, device="cuda"), torch.empty(N, F, device="cuda")
    assert L.BatchNormApplyTrainingBatchDevice(hb, dp(xd), dp(yd)) == 0, capi.last_error()
    assert L.BatchNormCalculateGradientDevice(hb, dp(dbe), dp(dga), dp(dxx), dp(dd)) == 0, capi.last_error()
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy(), y)
    for got, want in zip((dbe, dga, dxx), ref):
        assert np.array_equal(got.cpu().numpy(), want)
    mm2 = np.ctypeslib.as_array(wb.moving_mean, shape=(F,))
    assert not np.array_equal(mm2, mm)                       # the device call updated the caller-visible moving statistics
    L.BatchNormGradientDestroy(gb); L.BatchNormDestroy(hb)
