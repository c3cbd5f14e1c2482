"""GPU parity of the GRU on the register-resident split-bf16 kernels (csrc/hip/recurrent_rr.hip: gru_rr_kernel), the
kernels behind `bench.py --workload gru` (BASELINE configs[3], two stacked GRU-256 layers) since round 3.

Reference semantics: layers/gru.c:129-187 (cell, reset-after form), :246-293 (batch forward).  The three gates ride in
the LSTM kernel's four gate slots (z | r | h.U_h | x.W_h, zero weight blocks where a slot has no x or no h part); only
the gate arithmetic differs.  Checked against the oracle, against the exact-f32 kernels (rec_rr = 0), and for the
properties the path promises: shards give the same bits, carried state works, the two-layer stack call equals two calls.
Also here: the 256-wide-input instantiation (<4, 4>: U's low image in registers) for the LSTM.
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


def gru_weights(r, I, H):
    return u(r, I, 3 * H, sc=I ** -0.5), u(r, H, 3 * H, sc=H ** -0.5), u(r, 3 * H, sc=0.1), u(r, 3 * H, sc=0.1)


def _both(layer, xd):
    """(default path, exact-f32 path) of one device call"""
    a = layer.apply_device(xd).clone()
    capi.set_option("rec_rr", 0)
    b = layer.apply_device(xd).clone()
    capi.set_option("rec_rr", "auto")
    return a, b


@pytest.mark.parametrize("B,I,H,T,seq", [
    (64, 128, 256, 20, True),        # configs[3] layer 1: KH = 4 / KX = 2
    (64, 256, 256, 20, True),        # configs[3] layer 2: KH = 4 / KX = 4 (U's low image in registers)
    (70, 64, 128, 33, True),         # ragged second tile, KX = 1
    (130, 40, 64, 9, False),         # last state only; padded in
    (48, 128, 512, 12, True),        # KH = 8 / KX = 2
    (33, 200, 192, 11, True),        # in = 200 (padded to 256), H = 192 (12 column tiles)
    (96, 72, 320, 8, True),          # KH = 8 with padded k steps
    (32, 128, 256, 1, True),         # T = 1: the peeled pipeline's shortest form
    (32, 256, 128, 2, False),        # T = 2, KX = 4, last state only
    (1100, 64, 64, 5, True),         # more batch tiles than one launch holds at this H (4 column tiles x 18 batch tiles fit; checks the tiling loop)
])
def test_gru_rr_matches_oracle(gpu, B, I, H, T, seq):
    import torch
    L = capi.load()
    capi.set_option("rec_fk", 0)      # this file covers the split-K family on all its shapes (tests/test_gpu_fk.py: the full-K one)
    r = rng(B * 17 + H + T + I)
    x = u(r, B, T, I)
    W, U, bi, bh = gru_weights(r, I, H)
    gru = NL.GRU(I, H, seq, T)
    gru.set_weights(W, U, bi, bh)
    xd = torch.from_numpy(x).cuda()
    got, exact = _both(gru, xd)
    gru.apply_device(xd)
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith("gru_rr_kernel"), L.nntk_hip_last_recurrent_kernel()
    assert L.nntk_hip_device_status() == 0
    ref = O.gru(x, W, U, bi, bh, return_sequences=seq)
    e_rr, e_ex = float(np.abs(got.cpu().numpy() - ref).max()), float(np.abs(exact.cpu().numpy() - ref).max())
    print("gru_rr B=%d I=%d H=%d T=%d: max abs err vs oracle %.2e (exact-f32 path %.2e)" % (B, I, H, T, e_rr, e_ex))
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    assert np.array_equal(gru.apply(x), got.cpu().numpy())              # host-pointer batch call: same kernel, same bits
    gru.destroy()


def test_gru_rr_shards_are_bit_identical_and_rows_are_isolated(gpu):
    import torch
    r = rng(5)
    B, I, H, T = 150, 128, 256, 11
    x = u(r, B, T, I)
    W, U, bi, bh = gru_weights(r, I, H)
    gru = NL.GRU(I, H, True, T)
    gru.set_weights(W, U, bi, bh)
    xd = torch.from_numpy(x).cuda()
    whole = gru.apply_device(xd).clone()
    parts = torch.cat([gru.apply_device(xd[:83].contiguous()).clone(), gru.apply_device(xd[83:].contiguous()).clone()])
    assert torch.equal(whole, parts)
    xp = xd.clone()
    xp[40, 3, 17] = float("nan")
    bad = gru.apply_device(xp)
    keep = [i for i in range(B) if i != 40]
    assert torch.equal(bad[keep], whole[keep])
    assert not torch.isfinite(bad[40, 3:]).all() and torch.equal(bad[40, :3], whole[40, :3])
    gru.destroy()


def test_gru_rr_carried_state(gpu):
    """h_0 from / h_T into the handle (gru.c:189-204) through the register-resident kernel (rec_rr = 1 forces it for B = 1,
    rec_stream = 0 keeps the call off the streaming kernel): three calls equal one long oracle run."""
    r = rng(12)
    I, H, T = 64, 128, 40
    W, U, bi, bh = gru_weights(r, I, H)
    x = u(r, 3 * T, I)
    capi.set_option("rec_rr", 1)
    capi.set_option("rec_stream", 0)
    gru = NL.GRU(I, H, True, T)
    gru.set_weights(W, U, bi, bh)
    got = np.concatenate([gru.apply(x[i * T:(i + 1) * T]) for i in range(3)])
    assert capi.load().nntk_hip_last_recurrent_kernel().decode().startswith("gru_rr_kernel")
    ref, hT = O.gru(x, W, U, bi, bh)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gru.state(), hT, rtol=1e-5, atol=1e-5)
    gru.reset_state()
    np.testing.assert_allclose(gru.apply(x[:T]), ref[:T], rtol=1e-5, atol=1e-5)
    gru.destroy()
    capi.set_option("rec_rr", "auto"); capi.set_option("rec_stream", "auto")


def test_gru_rr_weight_edits_repack_the_image(gpu):
    """The four-slot image is rebuilt after the weights change (GRUSyncWeights / the host-pointer calls' edit check)."""
    import torch
    r = rng(21)
    B, I, H, T = 40, 64, 128, 6
    x = u(r, B, T, I)
    W, U, bi, bh = gru_weights(r, I, H)
    gru = NL.GRU(I, H, True, T)
    gru.set_weights(W, U, bi, bh)
    np.testing.assert_allclose(gru.apply(x), O.gru(x, W, U, bi, bh), rtol=1e-5, atol=1e-5)
    W2, U2, bi2, bh2 = gru_weights(r, I, H)
    gru.set_weights(W2, U2, bi2, bh2)
    np.testing.assert_allclose(gru.apply(x), O.gru(x, W2, U2, bi2, bh2), rtol=1e-5, atol=1e-5)
    gru.destroy()


def test_gru_stack2_takes_the_rr_pair_and_equals_two_calls(gpu):
    """GRUStack2ApplyDevice on shapes both layers' register-resident kernels take: two launches (layer 1 on the split-K kernel, the
    256-wide layer 2 on the full-K one since round 5; rec_fk = 0 keeps both on split-K), bit-identical to the two layer calls;
    rec_fused2 = 1 brings the fused exact-f32 kernel back (same tolerance, other bits)."""
    import torch
    L = capi.load()
    r = rng(31)
    B, I, H, T = 96, 128, 256, 25
    x = u(r, B, T, I)
    w1, w2 = gru_weights(r, I, H), gru_weights(r, H, H)
    g1, g2 = NL.GRU(I, H, True, T), NL.GRU(H, H, True, T)
    g1.set_weights(*w1); g2.set_weights(*w2)
    xd = torch.from_numpy(x).cuda()
    got = NL.gru_stack2_apply_device(g1, g2, xd).clone()
    assert L.nntk_hip_last_recurrent_kernel().decode() == "gru_fk_kernel<16,16,4>"
    two = g2.apply_device(g1.apply_device(xd).clone()).clone()
    assert torch.equal(got, two)
    ref = O.gru(O.gru(x, *w1), *w2)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    capi.set_option("rec_fk", 0)
    old = NL.gru_stack2_apply_device(g1, g2, xd).clone()
    assert L.nntk_hip_last_recurrent_kernel().decode() == "gru_rr_kernel<4,4>"
    assert torch.equal(old, g2.apply_device(g1.apply_device(xd).clone()))
    capi.set_option("rec_fk", "auto")
    np.testing.assert_allclose(old.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    capi.set_option("rec_fused2", 1)
    fused = NL.gru_stack2_apply_device(g1, g2, xd).clone()
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith("gru2_persistent_kernel")
    capi.set_option("rec_fused2", "auto")
    np.testing.assert_allclose(fused.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    assert np.array_equal(NL.gru_stack2_apply(g1, g2, x), got.cpu().numpy())     # host-pointer stack call
    g1.destroy(); g2.destroy()


def test_gru_rr_long_recurrence_error_at_config4_shape(gpu):
    """Achieved error of the T = 1000 recurrence at BASELINE configs[3]'s layer shapes (one 64-row tile), against the oracle and
    against the exact-f32 path: the split-bf16 contraction is an f32-accuracy product, so the two paths sit at the same distance
    from the oracle (bound = 3 x what was measured on MI355X)."""
    import torch
    r = rng(41)
    B, I, H, T = 64, 128, 256, 1000
    x = u(r, B, T, I)
    w1, w2 = gru_weights(r, I, H), gru_weights(r, H, H)
    g1, g2 = NL.GRU(I, H, True, T), NL.GRU(H, H, True, T)
    g1.set_weights(*w1); g2.set_weights(*w2)
    xd = torch.from_numpy(x).cuda()
    got = NL.gru_stack2_apply_device(g1, g2, xd).cpu().numpy()
    capi.set_option("rec_rr", 0)
    exact = NL.gru_stack2_apply_device(g1, g2, xd).cpu().numpy()
    capi.set_option("rec_rr", "auto")
    ref = O.gru(O.gru(x, *w1), *w2)
    e_rr, e_ex = float(np.abs(got - ref).max()), float(np.abs(exact - ref).max())
    print("2 x GRU-256, T=1000: max abs err vs oracle: rr pair %.2e, fused exact-f32 kernel %.2e" % (e_rr, e_ex))
    assert e_rr < 2e-5 and e_ex < 2e-5
    g1.destroy(); g2.destroy()


@pytest.mark.parametrize("B,I,H,T", [(64, 256, 256, 15), (40, 160, 128, 9)])
def test_lstm_rr_wide_input(gpu, B, I, H, T):
    """lstm_rr_kernel<4, 4>: inputs of 129..256 channels at H <= 256 (the W images take 96 KB of LDS, U's low image moves to
    registers).  Since round 5 the default for 128 < H <= 256 is the full-K kernel; rec_fk = 0 pins this one."""
    import torch
    L = capi.load()
    r = rng(B + I + H)
    x = u(r, B, T, I)
    W, U, bi, bh = u(r, I, 4 * H, sc=I ** -0.5), u(r, H, 4 * H, sc=H ** -0.5), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    xd = torch.from_numpy(x).cuda()
    capi.set_option("rec_fk", 0)
    got = lstm.apply_device(xd).cpu().numpy()
    capi.set_option("rec_fk", "auto")
    assert L.nntk_hip_last_recurrent_kernel().decode() == "lstm_rr_kernel<4,4>"
    ref = O.lstm(x, W, U, bi, bh, v2=True)
    ref = ref[0] if isinstance(ref, tuple) else ref
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    lstm.destroy()


@pytest.mark.parametrize("cell,I,H,fam", [("gru", 128, 256, "rr"), ("gru", 256, 256, "rr"), ("gru", 256, 256, "fk"), ("lstm", 256, 256, "fk"),
                                          ("lstm", 128, 512, "rr"), ("lstm", 64, 128, "rr")])
def test_rr_stale_hand_off_of_another_input_is_never_taken_for_data(gpu, cell, I, H, fam):
    """the T-deep hand-off of a launch still holds the VALID-looking fragments of the launch before it, and the repeat / shard
    tests relaunch the same input into the same scratch, where a stale block equals the right one.  Here two DIFFERENT inputs alternate
    through one layer (same hand-off scratch, the caller-visible frag3 output included): a consumer that took a stale block for data -- a
    pending mark that arrived late, a flag that overtook its data -- would mix in the other input's bits.  Both protocols: the pending
    pattern (H <= 256) and the flag words (H = 512); both families on the 256-wide shapes (the full-K one is their default)."""
    import torch
    L = capi.load()
    capi.set_option("rec_fk", "auto" if fam == "fk" else 0)
    r = rng(H + I)
    B, T = 128, 60
    G = 4 if cell == "lstm" else 3
    W, U, bi, bh = u(r, I, G * H, sc=I ** -0.5), u(r, H, G * H, sc=H ** -0.5), u(r, G * H, sc=0.1), u(r, G * H, sc=0.1)
    lay = NL.LSTM(I, H, True, T, v2=True) if cell == "lstm" else NL.GRU(I, H, True, T)
    lay.set_weights(W, U, bi, bh)
    xa, xb = u(r, B, T, I), u(r, B, T, I)
    da, db = torch.from_numpy(xa).cuda(), torch.from_numpy(xb).cuda()
    f3 = da.new_empty(L.nntk_frag3_floats(B, T, H))
    first_a, _ = NL.recurrent_apply_device_frag3(lay, x=da, want_f32=True, out_f3=f3)
    first_a = first_a.clone()
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith(cell + "_" + fam + "_kernel")
    first_b, _ = NL.recurrent_apply_device_frag3(lay, x=db, want_f32=True, out_f3=f3)
    first_b = first_b.clone()
    for _ in range(3):
        oa, _ = NL.recurrent_apply_device_frag3(lay, x=da, want_f32=True, out_f3=f3)
        assert torch.equal(oa, first_a) and torch.equal(NL.frag3_unpack_device(f3, B, T, H), first_a)
        f3.fill_(1.25)                                        # finite, non-pending garbage in EVERY block, marks of the first two steps included
        ob, _ = NL.recurrent_apply_device_frag3(lay, x=db, want_f32=True, out_f3=f3)
        assert torch.equal(ob, first_b) and torch.equal(NL.frag3_unpack_device(f3, B, T, H), first_b)
    ofn = O.lstm if cell == "lstm" else O.gru
    for xx, got in ((xa, first_a), (xb, first_b)):
        ref = ofn(xx, W, U, bi, bh, **({"v2": True} if cell == "lstm" else {}))
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    capi.set_option("rec_fk", "auto")
    lay.destroy()


def test_rr_result_does_not_depend_on_the_alignment_of_the_input_pointer(gpu):
    """a small call with an f32 input that is not 16-byte aligned used to fall to the exact-f32 kernels (another summation
    order) while the same call with an aligned pointer ran on the register-resident kernel: bits depended on pointer alignment.  A
    misaligned input is now packed into frag3 form like any shape the f32 row form does not take."""
    import torch
    L = capi.load()
    r = rng(31)
    B, I, H, T = 40, 128, 256, 7
    x = u(r, B, T, I)
    W, U, bi, bh = gru_weights(r, I, H)
    gru = NL.GRU(I, H, True, T)
    gru.set_weights(W, U, bi, bh)
    xa = torch.from_numpy(x).cuda()
    pad = torch.empty(B * T * I + 1, device="cuda")
    xm = pad[1:].view(B, T, I)
    xm.copy_(xa)
    assert xm.data_ptr() % 16 == 4
    a = gru.apply_device(xa).clone()
    b = gru.apply_device(xm).clone()
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith("gru_rr_kernel")
    assert torch.equal(a, b)
    gru.destroy()
