"""GPU parity of the register-resident split-bf16 LSTM kernel (csrc/hip/recurrent_rr.hip: lstm_rr_kernel), the kernel
behind the stack benchmark's LSTM phase: x W fused into the step, U^T in registers, h exchanged pre-split.

Reference semantics: layers/lstm.c:185-239 (cell), :426-475 (batch forward).  Checked against the oracle (and torch
float64 for the long recurrence), against the exact-f32 path (rec_rr = 0), and for the properties the path promises:
a shard gives the same bits as the whole batch, rows do not contaminate each other, carried state works.
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


def lstm_weights(r, I, H):
    return u(r, I, 4 * H, sc=I ** -0.5), u(r, H, 4 * H, sc=H ** -0.5), u(r, 4 * H, sc=0.1), u(r, 4 * H, sc=0.1)


def _took_rr(lstm, xd):
    """True when the call ran on lstm_rr_kernel: the result then differs from the exact-f32 path in the last bits."""
    a = lstm.apply_device(xd).clone()
    capi.set_option("rec_rr", 0)
    b = lstm.apply_device(xd).clone()
    capi.set_option("rec_rr", "auto")
    return not bool((a == b).all()), a, b


@pytest.mark.parametrize("B,I,H,T,seq,v2", [
    (64, 128, 512, 20, True, True),       # the stack's LSTM shape, one batch tile, KH = 8 / KX = 2
    (70, 64, 256, 33, True, False),       # ragged second tile, Keras one-bias form, KH = 4 / KX = 1
    (130, 40, 128, 9, False, True),       # last state only; in and H padded (40 -> 64, 128 -> 256 k steps)
    (33, 128, 192, 12, True, True),       # KH = 4 / KX = 2; one row in the second half-tile
    (96, 128, 320, 14, True, True),       # H between the compiled depths: 20 column tiles, padded k steps
    (40, 8, 64, 25, True, True),          # smallest shapes the kernel takes
    (200, 72, 512, 7, True, True),        # KH = 8 / KX = 2 with in = 72 (padded)
])
def test_lstm_rr_matches_oracle(gpu, B, I, H, T, seq, v2):
    import torch
    r = rng(B * 13 + H + T)
    x = u(r, B, T, I)
    W, U, bi, bh = lstm_weights(r, I, H)
    lstm = NL.LSTM(I, H, seq, T, v2=v2)
    lstm.set_weights(W, U, bi, bh)
    xd = torch.from_numpy(x).cuda()
    took, got, exact = _took_rr(lstm, xd)
    assert took, "the call did not run on lstm_rr_kernel"
    assert capi.load().nntk_hip_device_status() == 0
    ref = O.lstm(x, W, U, bi, bh, v2=v2, return_sequences=seq)
    ref = ref[0] if isinstance(ref, tuple) else ref
    e_rr, e_ex = float(np.abs(got.cpu().numpy() - ref).max()), float(np.abs(exact.cpu().numpy() - ref).max())
    print("lstm_rr B=%d I=%d H=%d T=%d: max abs err vs oracle %.2e (exact-f32 path %.2e)" % (B, I, H, T, e_rr, e_ex))
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    # host-pointer batch call: same kernel, same bits
    assert np.array_equal(lstm.apply(x), got.cpu().numpy())
    lstm.destroy()


def test_lstm_rr_shards_are_bit_identical_and_rows_are_isolated(gpu):
    import torch
    r = rng(7)
    B, I, H, T = 150, 128, 512, 11
    x = u(r, B, T, I)
    W, U, bi, bh = lstm_weights(r, I, H)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    xd = torch.from_numpy(x).cuda()
    whole = lstm.apply_device(xd).clone()
    parts = torch.cat([lstm.apply_device(xd[:83].contiguous()).clone(), lstm.apply_device(xd[83:].contiguous()).clone()])
    assert torch.equal(whole, parts)                   # a row's result does not depend on its tile / half / lane group
    assert torch.equal(whole, lstm.apply_device(xd))   # reproducible
    # one poisoned sequence (NaN and inf inputs) stays alone: every other row keeps its bits
    xp = xd.clone()
    xp[40, 3, 17] = float("nan")
    xp[101, 0, 5] = float("inf")
    bad = lstm.apply_device(xp)
    keep = [i for i in range(B) if i not in (40, 101)]
    assert torch.equal(bad[keep], whole[keep])
    assert not torch.isfinite(bad[40, 3:]).all() and torch.equal(bad[40, :3], whole[40, :3])
    lstm.destroy()


@pytest.mark.parametrize("cell", ["lstm", "gru"])
def test_rr_shards_of_any_size_equal_the_whole_batch(gpu, cell):
    """the kernel is chosen from the layer's shape and activations, never from the number of sequences in the call
    (round 3 switched kernels at B = 32, so 130 utterances over 8 GPUs differed from the single-GPU run).  B = 40 split 20 / 20,
    130 split into 8 shards of 16-17, and a single row: all bit-identical to the whole batch."""
    import torch
    r = rng(41)
    I, H, T = 128, 256, 9
    G = 4 if cell == "lstm" else 3
    W, U, bi, bh = u(r, I, G * H, sc=I ** -0.5), u(r, H, G * H, sc=H ** -0.5), u(r, G * H, sc=0.1), u(r, G * H, sc=0.1)
    layer = NL.LSTM(I, H, True, T, v2=True) if cell == "lstm" else NL.GRU(I, H, True, T)
    layer.set_weights(W, U, bi, bh)
    L = capi.load()
    for B, cuts in ((40, [0, 20, 40]), (130, [0, 17, 34, 51, 67, 83, 99, 115, 130]), (33, [0, 1, 33])):
        xd = torch.from_numpy(u(r, B, T, I)).cuda()
        whole = layer.apply_device(xd).clone()
        assert L.nntk_hip_last_recurrent_kernel().decode().startswith(cell + "_rr_kernel")
        parts = torch.cat([layer.apply_device(xd[a:b].contiguous()).clone() for a, b in zip(cuts[:-1], cuts[1:])])
        assert L.nntk_hip_last_recurrent_kernel().decode().startswith(cell + "_rr_kernel")       # ... also for a shard of one row
        assert torch.equal(whole, parts), (cell, B)
    layer.destroy()


@pytest.mark.parametrize("cell", ["lstm", "gru"])
@pytest.mark.parametrize("bad,where", [(np.inf, "W"), (-np.inf, "W"), (3.4e38, "U"), (1e-40, "U")])
def test_recurrent_weights_the_split_cannot_hold_run_on_the_exact_kernels(gpu, cell, bad, where):
    """a W or U value the bf16 split cannot represent (non-finite, above bf16's largest finite value, denormal) would go
    through the register-resident kernels' split as hi = inf, rest = inf - inf = NaN, where the reference's f32 chain gives a
    saturated, finite gate.  Such a block is found at upload and keeps the exact-f32 kernels: "auto" equals rec_rr = 0 bit for bit
    and matches the oracle; clean weights go back to the register-resident kernel."""
    import torch
    r = rng(43)
    B, I, H, T = 48, 64, 128, 6
    G = 4 if cell == "lstm" else 3
    W, U, bi, bh = u(r, I, G * H, sc=I ** -0.5), u(r, H, G * H, sc=H ** -0.5), u(r, G * H, sc=0.1), u(r, G * H, sc=0.1)
    Wb, Ub = W.copy(), U.copy()
    (Wb if where == "W" else Ub)[3, 7] = bad                       # gate block 0 (i / z), hidden unit 7: a sigmoid saturates it
    x = u(r, B, T, I)
    layer = NL.LSTM(I, H, True, T, v2=True) if cell == "lstm" else NL.GRU(I, H, True, T)
    layer.set_weights(Wb, Ub, bi, bh)
    L = capi.load()
    xd = torch.from_numpy(x).cuda()
    auto = layer.apply_device(xd).cpu().numpy()
    assert not L.nntk_hip_last_recurrent_kernel().decode().startswith(cell + "_rr_kernel")
    capi.set_option("rec_rr", 0)
    exact = layer.apply_device(xd).cpu().numpy()
    capi.set_option("rec_rr", "auto")
    assert np.array_equal(auto, exact, equal_nan=True)
    ofn = O.lstm if cell == "lstm" else O.gru
    ref = ofn(x, Wb, Ub, bi, bh, **({"v2": True} if cell == "lstm" else {}))
    ref = ref[0] if isinstance(ref, tuple) else ref
    assert np.array_equal(np.isfinite(auto), np.isfinite(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(auto[fin], ref[fin], rtol=1e-5, atol=1e-5)
    layer.set_weights(W, U, bi, bh)
    layer.sync_weights()                                           # (device-pointer calls do not look for host edits)
    layer.apply_device(xd)
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith(cell + "_rr_kernel")
    layer.destroy()


def test_lstm_rr_carried_state_and_final_state(gpu):
    """h_0 / c_0 taken from, and h_T / c_T left in, the handle (the reference's stateful single-sequence API, lstm.c:241-268)
    through the register-resident kernel (rec_rr = 1 forces it for B = 1; rec_stream = 0 keeps the call off the streaming
    kernel): three consecutive calls equal one long oracle run."""
    r = rng(11)
    I, H, T = 64, 128, 40
    W, U, bi, bh = lstm_weights(r, I, H)
    x = u(r, 3 * T, I)
    capi.set_option("rec_rr", 1)
    capi.set_option("rec_stream", 0)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    got = np.concatenate([lstm.apply(x[i * T:(i + 1) * T]) for i in range(3)])
    ref, hT, cT = O.lstm(x, W, U, bi, bh, v2=True)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    h, c = lstm.state()
    np.testing.assert_allclose(h, hT, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c, cT, rtol=1e-5, atol=2e-5)
    lstm.reset_state()
    np.testing.assert_allclose(lstm.apply(x[:T]), ref[:T], rtol=1e-5, atol=1e-5)
    lstm.destroy()


def test_lstm_rr_nondefault_activations_fall_back(gpu):
    import ctypes as C
    L = capi.load()
    r = rng(13)
    B, I, H, T = 48, 64, 128, 6
    x = u(r, B, T, I)
    W, U, bi, bh = lstm_weights(r, I, H)
    acts = L.LSTMActivationsCreate(L.ActivationFunctionCreateSigmoid(H), L.ActivationFunctionCreateSigmoid(H),
                                   L.ActivationFunctionCreateReLU(H, C.c_float(0.5)), L.ActivationFunctionCreateSigmoid(H),
                                   L.ActivationFunctionCreateTanh(H))
    lstm = NL.LSTM(I, H, True, T, v2=True, acts=acts)
    lstm.set_weights(W, U, bi, bh)
    ref = O.lstm(x, W, U, bi, bh, v2=True, acts=(O.ACT_SIGMOID, O.ACT_SIGMOID, O.ACT_RELU, O.ACT_SIGMOID, O.ACT_TANH),
                 relu_a=(1, 1, 0.5, 1, 1))
    ref = ref[0] if isinstance(ref, tuple) else ref
    np.testing.assert_allclose(lstm.apply(x), ref, rtol=1e-5, atol=1e-5)
    lstm.destroy()


def _torch64_lstm(x, W, U, bi, bh):
    import torch
    H = U.shape[0]
    m = torch.nn.LSTM(W.shape[0], H, batch_first=True).double()
    with torch.no_grad():
        m.weight_ih_l0.copy_(torch.tensor(W).double().T); m.weight_hh_l0.copy_(torch.tensor(U).double().T)
        m.bias_ih_l0.copy_(torch.tensor(bi).double()); m.bias_hh_l0.copy_(torch.tensor(bh).double())
        return m(torch.tensor(x).double())[0].numpy()


def test_achieved_error_lstm_rr_512_T996(gpu):
    """The stack's LSTM(128 -> 512, v2) over 996 steps on the register-resident kernel, one full batch tile: the
    deviation from the oracle (libm gates, scalar k order) and from torch float64 as NUMBERS, asserted at measured x 3."""
    import torch
    r = rng(501)
    B, I, H, T = 64, 128, 512, 996
    x = r.standard_normal((B, T, I)).astype(np.float32)
    uw = lambda fan, *s: r.uniform(-fan ** -0.5, fan ** -0.5, s).astype(np.float32)
    W, U, bi, bh = uw(I, I, 4 * H), uw(H, H, 4 * H), uw(H, 4 * H), uw(H, 4 * H)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    xd = torch.from_numpy(x).cuda()
    took, got, exact = _took_rr(lstm, xd)
    assert took
    rows = [0, 31, 32, 63]
    got, exact = got[rows].cpu().numpy(), exact[rows].cpu().numpy()
    ref = O.lstm(x[rows], W, U, bi, bh, v2=True)
    ref = ref[0] if isinstance(ref, tuple) else ref
    r64 = _torch64_lstm(x[rows], W, U, bi, bh)
    e_or, e_64 = float(np.abs(got - ref).max()), float(np.abs(got - r64).max())
    x_or, x_64 = float(np.abs(exact - ref).max()), float(np.abs(exact - r64).max())
    print("lstm_rr LSTM(128->512, v2) T=996: max abs err vs oracle %.2e, vs torch float64 %.2e "
          "(exact-f32 kernel: %.2e / %.2e; oracle vs float64 %.2e)" % (e_or, e_64, x_or, x_64, float(np.abs(ref - r64).max())))
    assert e_or < 3e-6 and e_64 < 3e-6
    lstm.destroy()


def test_full_size_lstm_rr_stack_shard(gpu):
    """B = 512 x T = 996: the bench's own launch (8 batch tiles x 32 column tiles = all 256 CUs).  Rows of the first and last tile
    and of both halves against the oracle; and, as a race detector for the kernel's hand-off protocol (flag words, counted drains,
    peeled half-steps), over the WHOLE batch: a second run equal bit for bit, every input / output form of the kernel (f32 rows and
    frag3 on either side) equal bit for bit, and the exact-f32 persistent kernel within the summation-order bound."""
    import torch
    r = rng(512)
    B, I, H, T = 512, 128, 512, 996
    x = torch.randn(B, T, I, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    uw = lambda fan, *s: r.uniform(-fan ** -0.5, fan ** -0.5, s).astype(np.float32)
    W, U, bi, bh = uw(I, I, 4 * H), uw(H, H, 4 * H), uw(H, 4 * H), uw(H, 4 * H)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    L = capi.load()
    y = lstm.apply_device(x).clone()                 # default route at this size: x packed into frag3, XF instantiation
    assert L.nntk_hip_device_status() == 0 and L.nntk_hip_last_recurrent_kernel().decode() == "lstm_rr_kernel<8,2>"
    rows = [0, 40, 300, 479, 511]
    ref = O.lstm(x[rows].cpu().numpy(), W, U, bi, bh, v2=True)
    ref = ref[0] if isinstance(ref, tuple) else ref
    e = float(np.abs(y[rows].cpu().numpy() - ref).max())
    print("lstm_rr B=512 T=996: max abs err vs oracle %.2e" % e)
    assert e < 3e-6
    assert torch.equal(lstm.apply_device(x), y)      # reproducible over the whole batch
    capi.set_option("rec_xf", 0)                     # the f32-row input form of the kernel
    assert torch.equal(lstm.apply_device(x), y)
    capi.set_option("rec_xf", "auto")
    _, f3 = NL.recurrent_apply_device_frag3(lstm, x=x, want_f32=False, want_f3=True)     # the bench's form: frag3 output only
    assert torch.equal(NL.frag3_unpack_device(f3, B, T, H), y)
    del f3
    capi.set_option("rec_rr", 0)                     # exact-f32 persistent kernel + projection GEMM: another summation order
    ex = lstm.apply_device(x)
    capi.set_option("rec_rr", "auto")
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith("rec_persistent_kernel")
    d = float((ex - y).abs().max())
    print("lstm_rr vs exact-f32 persistent kernel, whole batch: max abs diff %.2e" % d)
    assert d < 1e-5
    assert L.nntk_hip_device_status() == 0
    lstm.destroy()


def test_single_gpu_at_the_named_batch_of_4096(gpu):
    """north_star quotes the recurrent kernel "at batch 4096 x 1000 frames": 4096 utterances on ONE GPU are 64 batch tiles = 8
    back-to-back launches of the 256-workgroup kernel sharing the hand-off and flag buffers (T = 100 keeps the oracle rows cheap;
    the launch structure is the full-size one).  Rows of the first and the last launch against the oracle, a 512-row shard equal
    to the whole batch bit for bit, the fused LSTM -> TimeDistributedDense call equal to the two calls (lstm.c:426-475)."""
    import torch
    r = rng(4096)
    B, I, H, T, N = 4096, 128, 512, 100, 1000
    x = torch.randn(B, T, I, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))
    uw = lambda fan, *s: r.uniform(-fan ** -0.5, fan ** -0.5, s).astype(np.float32)
    W, U, bi, bh = uw(I, I, 4 * H), uw(H, H, 4 * H), uw(H, 4 * H), uw(H, 4 * H)
    Wd, bd = uw(H, H, N), uw(H, N)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    tdd = NL.TimeDistributedDense(T, H, N)
    tdd.set_weights(Wd, bd)
    L = capi.load()
    y = lstm.apply_device(x).clone()
    assert L.nntk_hip_device_status() == 0 and L.nntk_hip_last_recurrent_kernel().decode() == "lstm_rr_kernel<8,2>"
    rows = [0, 63, 511, 512, 3584, 4000, 4095]
    ref = O.lstm(x[rows].cpu().numpy(), W, U, bi, bh, v2=True)
    ref = ref[0] if isinstance(ref, tuple) else ref
    e = float(np.abs(y[rows].cpu().numpy() - ref).max())
    print("lstm_rr B=4096 T=100 (8 launches): max abs err vs oracle %.2e" % e)
    assert e < 3e-6
    assert torch.equal(lstm.apply_device(x[3584:].contiguous()), y[3584:])       # the last launch's rows as a batch of their own
    z2 = tdd.apply_device(y)
    z1 = NL.lstm_tdd_apply_device(lstm, tdd, x)       # default: FRAG2H between the layers (every launch's output wave writes its batch tiles' blocks)
    d = float((z1 - z2).abs().max())
    assert 0.0 < d < 3e-6, d
    capi.set_option("dense_f16x2", 0)                 # the frag3 route: the two calls, bit for bit
    assert torch.equal(NL.lstm_tdd_apply_device(lstm, tdd, x), z2)
    capi.set_option("dense_f16x2", "auto")
    zr = O.time_distributed_dense(ref[-1], Wd, bd)
    np.testing.assert_allclose(z1[4095].cpu().numpy(), zr, rtol=1e-4, atol=1e-5)
    assert L.nntk_hip_device_status() == 0
    lstm.destroy(); tdd.destroy()


def test_rr_requests_stay_inside_their_tensors(gpu):
    """a diagnostics build of the library (recurrent_rr.hip -DNNTK_RR_BOUNDS) records the last byte every request
    of gru_rr_kernel / lstm_rr_kernel really touches; ragged last tiles (B = 33, 65, 130, 1) through all four input / output forms
    must stay inside [0, size) of x, the f32 output, the frag3 tensors and the h_0 slot.  Runs in a child process (its own library)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rr_bounds_check.py")], capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rr bounds: 0 violation(s)" in r.stdout


@pytest.mark.parametrize("H", [128, 512])
def test_lstm_rr_fault_is_reported_and_heals(gpu, H):
    """The register-resident kernel needs all its workgroups resident, like the other persistent kernels: a poll that runs
    out of budget (forced with rec_spin_us = 0) raises the sticky fault word.  Device-pointer callers see -1 at the next
    synchronize; the host-pointer call repeats itself on the per-timestep kernels and returns correct results.
    H = 128: the pending-pattern hand-off (a look at a fetched fragment gives up); H = 512: the flag protocol (a flag poll gives up)."""
    import torch
    L = capi.load()
    r = rng(91)
    B, I, T = 70, 64, 12
    x = u(r, B, T, I)
    W, U, bi, bh = lstm_weights(r, I, H)
    ref = O.lstm(x, W, U, bi, bh, v2=True)
    lstm = NL.LSTM(I, H, True, T, v2=True)
    lstm.set_weights(W, U, bi, bh)
    xd = torch.from_numpy(x).cuda()
    good = lstm.apply_device(xd).cpu().numpy()
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith("lstm_rr_kernel")
    np.testing.assert_allclose(good, ref, rtol=1e-5, atol=1e-5)
    capi.set_option("rec_spin_us", 0)
    lstm.apply_device(xd)
    torch.cuda.synchronize()
    assert L.nntk_hip_device_status() == 1
    assert L.nntk_hip_synchronize() == -1 and "timed out" in capi.last_error()
    # the process has switched to the per-timestep kernels
    after = lstm.apply_device(xd).cpu().numpy()
    assert L.nntk_hip_synchronize() == 0
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith("rec_step_kernel")
    np.testing.assert_allclose(after, ref, rtol=1e-5, atol=1e-5)
    # host-pointer call with the persistent kernels re-armed and the budget still zero: heals itself
    capi.set_option("rec_persistent", 1)
    healed = lstm.apply(x)
    assert capi.last_error() == ""
    np.testing.assert_allclose(healed, ref, rtol=1e-5, atol=1e-5)
    capi.set_option("rec_spin_us", "auto"); capi.set_option("rec_persistent", "auto")
    assert L.nntk_hip_synchronize() == 0
    assert np.array_equal(lstm.apply_device(xd).cpu().numpy(), good)          # back on the register-resident kernel, same bits
    lstm.destroy()
