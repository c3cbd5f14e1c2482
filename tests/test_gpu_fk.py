"""GPU parity of the full-K register-resident recurrent kernels (csrc/hip/recurrent_fk.hip: gru_fk_kernel / lstm_fk_kernel), the round-5
answer to "a recurrent kernel without split-K" (option rec_fk: the default for inputs of 129..256 channels, where it measured 7-13 % ahead
of the split-K family; rec_fk = 1 also for 65..128 channels, where it measured 1-4 % behind -- DESIGN K4d).

Reference semantics: layers/gru.c:129-187, :246-293; layers/lstm.c:185-239, :426-475.  Checked against the oracle, for repeatability (the
LDS ring / pending-pattern hand-offs are race detectors in themselves: a stale fragment changes bits), for shards, carried state and the
fault path.
"""
import numpy as np
import pytest

import oracle as O
from nntoolkitcore_amd import capi, layers as NL

pytestmark = pytest.mark.gpu


def u(r, *shape, sc=1.0):
    return r.uniform(-sc, sc, shape).astype(np.float32)


def make(cell, r, I, H, seq, T):
    G = 4 if cell == "lstm" else 3
    W, U, bi, bh = u(r, I, G * H, sc=I ** -0.5), u(r, H, G * H, sc=H ** -0.5), u(r, G * H, sc=0.1), u(r, G * H, sc=0.1)
    lay = NL.LSTM(I, H, seq, T, v2=True) if cell == "lstm" else NL.GRU(I, H, seq, T)
    lay.set_weights(W, U, bi, bh)
    ref = (lambda x: O.lstm(x, W, U, bi, bh, v2=True, return_sequences=seq)) if cell == "lstm" else (lambda x: O.gru(x, W, U, bi, bh, return_sequences=seq))
    return lay, ref


@pytest.mark.parametrize("cell,B,I,H,T,seq", [
    ("gru", 64, 128, 256, 20, True),        # configs[3] layer 1: four wavefronts share a 32-row operand
    ("gru", 64, 256, 256, 20, True),        # configs[3] layer 2: the default kernel of this shape
    ("lstm", 70, 128, 256, 7, True),        # ragged second row block
    ("gru", 33, 200, 192, 11, True),        # padded input, H = 192
    ("lstm", 70, 256, 160, 3, False),       # last state only; a column tile that is half outside H
    ("gru", 32, 128, 256, 1, True),         # T = 1
    ("gru", 1, 72, 256, 40, True),          # one row
    ("lstm", 600, 100, 144, 5, True),       # many row blocks
])
def test_fk_matches_oracle_and_repeats(gpu, cell, B, I, H, T, seq):
    import torch
    L = capi.load()
    r = np.random.default_rng(B * 13 + I + H + T)
    lay, ref = make(cell, r, I, H, seq, T)
    x = u(r, B, T, I)
    xd = torch.from_numpy(x).cuda()
    capi.set_option("rec_fk", 1)
    got = lay.apply_device(xd).clone()
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith(cell + "_fk_kernel"), L.nntk_hip_last_recurrent_kernel()
    assert L.nntk_hip_device_status() == 0
    again = lay.apply_device(xd).clone()
    assert torch.equal(got, again)
    err = float(np.abs(got.cpu().numpy() - ref(x)).max())
    print("%s_fk B=%d I=%d H=%d T=%d: max abs err vs oracle %.2e" % (cell, B, I, H, T, err))
    np.testing.assert_allclose(got.cpu().numpy(), ref(x), rtol=1e-5, atol=1e-5)
    lay.destroy()


def test_fk_shards_equal_the_whole_batch_and_frag3_output_equals_f32(gpu):
    import torch
    r = np.random.default_rng(3)
    B, I, H, T = 150, 128, 256, 9
    lay, _ = make("gru", r, I, H, True, T)
    x = torch.from_numpy(u(r, B, T, I)).cuda()
    capi.set_option("rec_fk", 1)
    whole = lay.apply_device(x).clone()
    for lo, hi in [(0, 1), (1, 34), (34, 150)]:
        part = lay.apply_device(x[lo:hi].contiguous())
        assert torch.equal(part, whole[lo:hi]), (lo, hi)
    _, out_f3 = NL.recurrent_apply_device_frag3(lay, x=x, want_f32=False, want_f3=True)
    assert torch.equal(NL.frag3_unpack_device(out_f3, B, T, H), whole)
    x_f3 = NL.frag3_pack_device(x)
    o2, _ = NL.recurrent_apply_device_frag3(lay, x_f3=x_f3, batch=B)
    assert torch.equal(o2, whole)
    lay.destroy()


def test_fk_stale_hand_off_from_another_input_is_never_taken_for_data(gpu):
    """The hand-off buffer of one launch holds VALID-looking fragments of the previous launch.  Two different inputs alternate through the same
    layer (same scratch): a consumer that took a stale block for data would reproduce the other input's bits (the repeat tests
    relaunch the same input, where stale bytes equal the right ones)."""
    import torch
    r = np.random.default_rng(5)
    B, I, H, T = 96, 128, 256, 40
    lay, ref = make("gru", r, I, H, True, T)
    xa, xb = u(r, B, T, I), u(r, B, T, I)
    da, db = torch.from_numpy(xa).cuda(), torch.from_numpy(xb).cuda()
    capi.set_option("rec_fk", 1)
    first_a = lay.apply_device(da).clone()
    first_b = lay.apply_device(db).clone()
    for _ in range(4):
        assert torch.equal(lay.apply_device(da), first_a)
        assert torch.equal(lay.apply_device(db), first_b)
    np.testing.assert_allclose(first_a.cpu().numpy(), ref(xa), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(first_b.cpu().numpy(), ref(xb), rtol=1e-5, atol=1e-5)
    lay.destroy()


def test_fk_fault_is_reported(gpu):
    """A look that runs out of budget (forced with rec_spin_us = 0) raises the sticky fault word; the process then takes the per-timestep
    kernels until the persistent ones are re-armed (the contract of every persistent recurrent kernel here)."""
    import torch
    L = capi.load()
    r = np.random.default_rng(9)
    lay, ref = make("gru", r, 128, 256, True, 6)
    x = u(r, 64, 6, 128)
    xd = torch.from_numpy(x).cuda()
    capi.set_option("rec_fk", 1)
    good = lay.apply_device(xd).clone()
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith("gru_fk_kernel")
    capi.set_option("rec_spin_us", 0)
    lay.apply_device(xd)
    torch.cuda.synchronize()
    assert L.nntk_hip_device_status() == 1
    assert L.nntk_hip_synchronize() == -1 and "timed out" in capi.last_error()
    after = lay.apply_device(xd).cpu().numpy()
    assert L.nntk_hip_synchronize() == 0
    np.testing.assert_allclose(after, ref(x), rtol=1e-5, atol=1e-5)
    capi.set_option("rec_spin_us", "auto"); capi.set_option("rec_persistent", 1)
    assert torch.equal(lay.apply_device(xd), good)          # re-armed: back on the full-K kernel, same bits
    assert L.nntk_hip_last_recurrent_kernel().decode().startswith("gru_fk_kernel")
    capi.set_option("rec_persistent", "auto")
    lay.destroy()


def test_kernel_plan_names_the_full_k_family_where_it_is_the_default(gpu):
    """rec_fk: auto = inputs of 129..256 channels (where the family measured ahead of split-K), 1 = every shape it takes, 0 = never."""
    L = capi.load()
    r = np.random.default_rng(1)
    capi.set_option("rec_fk", "auto")
    lay, _ = make("gru", r, 256, 256, True, 4)
    assert "gru_fk_kernel<16,16,4>" in L.GRUKernelPlan(lay.h).decode()
    capi.set_option("rec_fk", 0)
    assert "gru_rr_kernel<4,4>" in L.GRUKernelPlan(lay.h).decode()
    lay.destroy()
    lay, _ = make("gru", r, 128, 256, True, 4)
    assert "gru_rr_kernel<4,2>" in L.GRUKernelPlan(lay.h).decode()
    capi.set_option("rec_fk", "auto")
    assert "gru_rr_kernel<4,2>" in L.GRUKernelPlan(lay.h).decode()
    capi.set_option("rec_fk", 1)
    assert "gru_fk_kernel<16,8,4>" in L.GRUKernelPlan(lay.h).decode()
    lay.destroy()
    lay, _ = make("lstm", r, 128, 512, True, 4)           # a shape the family does not take keeps the split-K kernel
    assert "lstm_rr_kernel<8,2>" in L.LSTMKernelPlan(lay.h).decode()
    lay.destroy()
