"""CPU-only: pin the oracle.  (1) against what the REAL reference returns where it can
be run here (windows, geometry: tests/golden/ref_probe.json); (2) the oracle's kissfft
restatement against its own published factorisation properties; (3) against independent
implementations (torch / scipy / numpy) -- cross-checks, not reference pins (the
reference ships no fixtures and its Eigen/kissfft backends are absent: "parity unpinned"
for the compute paths, see oracle/nnref.h)."""
import json
import os

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_probe.json")))


def rng(seed):
    return np.random.default_rng(seed)


@pytest.mark.parametrize("name", ["ones", "hann", "hamming", "periodic_hann", "periodic_hamming", "blackman"])
def test_oracle_windows_bit_exact_vs_reference(name):
    assert np.array_equal(O.window(name, 16), np.array(GOLD["windows16"][name], np.float32))
    assert np.array_equal(O.window(name, 400), np.array(GOLD["windows400"][name], np.float32))


def test_oracle_geometry_vs_reference():
    for c in GOLD["spectrogram_config"]:
        assert O.spectrogram_geometry(c["nfft"], c["window_size"], c["noverlap"], c["input_size"]) == \
            (c["step"], c["nfreq"], c["ntime_series"])
    for c in GOLD["conv1d_config"]:
        assert O.conv1d_output_size(c["input_size"], c["k"], c["stride"]) == c["output_size"]


def test_reference_weight_block_layouts_recorded():
    g = GOLD["gru_weights"]
    assert (g["U_offset"], g["b_i_offset"], g["b_h_offset"]) == (5 * 21, 5 * 21 + 7 * 21, 5 * 21 + 7 * 21 + 21)
    l = GOLD["lstm_weights"]
    assert (l["U_offset"], l["b_i_offset"], l["b_h_offset"]) == (5 * 28, 5 * 28 + 7 * 28, 5 * 28 + 7 * 28 + 28)
    assert GOLD["conv1d_weights"] == {"b_offset": 40 * 128 * 5, "all_zero": 1}
    assert GOLD["batch_norm_weights"] == {"C": 6, "beta_offset": 6, "mean_offset": 12, "var_offset": 18, "all_zero": 1}
    assert GOLD["tdd_weights"]["b_offset"] == 35


@pytest.mark.parametrize("n", [512, 256, 64, 60, 45, 16, 7, 100, 2, 1, 11 * 13])
def test_kissfft_restatement_vs_numpy(n):
    r = rng(n)
    x = (r.standard_normal(n) + 1j * r.standard_normal(n)).astype(np.complex64)
    want = np.fft.fft(x.astype(np.complex128))
    scale = np.abs(want).max()
    assert np.abs(O.kiss_fft(x) - want).max() <= 4e-6 * scale + 1e-6
    back = O.kiss_fft(O.kiss_fft(x), inverse=True) / n
    assert np.abs(back - x).max() <= 1e-5


def test_spectrogram_vs_scipy():
    import scipy.signal as ss
    x = (0.1 * rng(1).standard_normal(16000)).astype(np.float32)
    w = O.window("hann", 400)
    assert np.abs(w - ss.get_window("hann", 400, fftbins=False)).max() < 1e-7
    _, _, sx = ss.spectrogram(x.astype(np.float64), window=w.astype(np.float64), nperseg=400, noverlap=240, nfft=512,
                              mode="magnitude", scaling="spectrum", detrend=False)
    got = O.spectrogram(x, w, 512, 240)
    assert got.shape == (98, 257)
    assert np.abs(got - sx.T).max() / np.abs(sx).max() < 1e-6
    _, _, px = ss.spectrogram(x.astype(np.float64), fs=16000, window=w.astype(np.float64), nperseg=400, noverlap=240,
                              nfft=512, mode="psd", scaling="density", detrend=False)
    assert np.abs(O.spectrogram(x, w, 512, 240, mode="psd", fs=16000) - px.T).max() / np.abs(px).max() < 1e-6


def test_conv1d_vs_torch():
    import torch
    r = rng(2)
    for (cin, cout, k, s, T) in [(3, 4, 5, 2, 23), (1, 16, 9, 1, 200), (40, 128, 5, 1, 64)]:
        x, W, b = (r.standard_normal(sh).astype(np.float32) for sh in [(2, T, cin), (cout, cin, k), (cout,)])
        want = torch.nn.functional.conv1d(torch.tensor(x).transpose(1, 2), torch.tensor(W), torch.tensor(b),
                                          stride=s).transpose(1, 2).numpy()
        got = O.conv1d(x, W, b, s)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 2e-5


def _perm_zrh_to_rzn(m):
    z, r_, h = np.split(m, 3, axis=-1)
    return np.concatenate([r_, z, h], -1)


def test_gru_vs_torch_including_T1000():
    import torch
    r = rng(3)
    for (B, T, I, H) in [(3, 50, 5, 7), (1, 1000, 32, 48)]:
        x = r.standard_normal((B, T, I)).astype(np.float32)
        W = (r.standard_normal((I, 3 * H)) * I ** -0.5).astype(np.float32)
        U = (r.standard_normal((H, 3 * H)) * H ** -0.5).astype(np.float32)
        bi, bh = (0.1 * r.standard_normal(3 * H)).astype(np.float32), (0.1 * r.standard_normal(3 * H)).astype(np.float32)
        g = torch.nn.GRU(I, H, batch_first=True)
        with torch.no_grad():
            g.weight_ih_l0.copy_(torch.tensor(_perm_zrh_to_rzn(W).T))
            g.weight_hh_l0.copy_(torch.tensor(_perm_zrh_to_rzn(U).T))
            g.bias_ih_l0.copy_(torch.tensor(_perm_zrh_to_rzn(bi)))
            g.bias_hh_l0.copy_(torch.tensor(_perm_zrh_to_rzn(bh)))
            want = g(torch.tensor(x))[0].numpy()
        assert np.abs(O.gru(x, W, U, bi, bh) - want).max() < 5e-6
        last = O.gru(x, W, U, bi, bh, return_sequences=False)
        assert np.abs(last - want[:, -1]).max() < 5e-6


@pytest.mark.parametrize("nonlin,act", [("tanh", O.ACT_TANH), ("relu", O.ACT_RELU)])
def test_rnn_vs_torch_and_reference_layout(nonlin, act):
    """the one-gate cell of layers/rnn.c equals torch.nn.RNN; the weight block layout the
    real reference returns (W | U | b_i | b_h) is the one the C API reproduces."""
    import torch
    r = rng(6)
    B, T, I, H = 3, 40, 5, 7
    x = r.standard_normal((B, T, I)).astype(np.float32)
    W = (r.standard_normal((I, H)) * I ** -0.5).astype(np.float32)
    U = (r.standard_normal((H, H)) * H ** -0.5).astype(np.float32)
    bi, bh = (0.1 * r.standard_normal(H)).astype(np.float32), (0.1 * r.standard_normal(H)).astype(np.float32)
    m = torch.nn.RNN(I, H, batch_first=True, nonlinearity=nonlin)
    with torch.no_grad():
        m.weight_ih_l0.copy_(torch.tensor(W.T)); m.weight_hh_l0.copy_(torch.tensor(U.T))
        m.bias_ih_l0.copy_(torch.tensor(bi)); m.bias_hh_l0.copy_(torch.tensor(bh))
        want = m(torch.tensor(x))[0].numpy()
    assert np.abs(O.rnn(x, W, U, bi, bh, act=act) - want).max() < 5e-6
    assert np.abs(O.rnn(x, W, U, bi, bh, act=act, return_sequences=False) - want[:, -1]).max() < 5e-6
    # v2 = False drops b_h (rnn.c:158-160)
    with torch.no_grad():
        m.bias_hh_l0.zero_()
        want1 = m(torch.tensor(x))[0].numpy()
    assert np.abs(O.rnn(x, W, U, bi, bh, act=act, v2=False) - want1).max() < 5e-6
    # stateful single-sequence form = one long call
    o1, h1 = O.rnn(x[0, :15], W, U, bi, bh, act=act)
    o2, _ = O.rnn(x[0, 15:], W, U, bi, bh, h0=h1, act=act)
    assert np.array_equal(np.concatenate([o1, o2]), O.rnn(x[0], W, U, bi, bh, act=act)[0])
    g = GOLD["rnn_weights"]
    assert (g["U_offset"], g["b_i_offset"], g["b_h_offset"]) == (5 * 7, 5 * 7 + 7 * 7, 5 * 7 + 7 * 7 + 7)


def test_bidirectional_helpers_vs_reference_outputs():
    """bd_reverse_*_batch are op-free in the reference, so the REAL functions were run (oracle/ref_probe.c);
    the restatement must reproduce their output bit for bit.  The merges are checked against numpy."""
    g = GOLD["bidirectional"]
    x_in = np.arange(12, dtype=np.float32).reshape(2, 3, 2)
    x_bw = np.arange(18, dtype=np.float32).reshape(2, 3, 3)
    assert O.bd_reverse(x_in).ravel().tolist() == g["reverse_input_B2_T3_F2"]
    assert O.bd_reverse(x_bw).ravel().tolist() == g["reverse_backward_B2_T3_F3"]
    r = rng(9)
    f, b = r.standard_normal((3, 5, 4)).astype(np.float32), r.standard_normal((3, 5, 4)).astype(np.float32)
    assert np.array_equal(O.bd_merge(f, b, "concat"), np.concatenate([f, b], axis=2))
    assert np.array_equal(O.bd_merge(f, b, "sum"), f + b)
    assert (g["concat_buffer_size_seq"], g["concat_buffer_size_last"]) == (2 * 3 * 3, 2 * 1 * 3)


@pytest.mark.parametrize("v2", [True, False])
def test_lstm_vs_torch(v2):
    import torch
    r = rng(4)
    B, T, I, H = 2, 60, 6, 9
    x = r.standard_normal((B, T, I)).astype(np.float32)
    W = (r.standard_normal((I, 4 * H)) * I ** -0.5).astype(np.float32)
    U = (r.standard_normal((H, 4 * H)) * H ** -0.5).astype(np.float32)
    bi, bh = (0.1 * r.standard_normal(4 * H)).astype(np.float32), (0.1 * r.standard_normal(4 * H)).astype(np.float32)
    l = torch.nn.LSTM(I, H, batch_first=True)
    with torch.no_grad():
        l.weight_ih_l0.copy_(torch.tensor(W.T))
        l.weight_hh_l0.copy_(torch.tensor(U.T))
        l.bias_ih_l0.copy_(torch.tensor(bi))
        l.bias_hh_l0.copy_(torch.tensor(bh if v2 else np.zeros_like(bh)))
        want = l(torch.tensor(x))[0].numpy()
    assert np.abs(O.lstm(x, W, U, bi, bh, v2=v2) - want).max() < 5e-6


def test_stateful_sequence_equals_one_long_call():
    r = rng(5)
    I, H = 4, 6
    x = r.standard_normal((30, I)).astype(np.float32)
    W, U = r.standard_normal((I, 4 * H)).astype(np.float32) * .4, r.standard_normal((H, 4 * H)).astype(np.float32) * .4
    bi, bh = r.standard_normal(4 * H).astype(np.float32) * .1, r.standard_normal(4 * H).astype(np.float32) * .1
    full, hf, cf = O.lstm(x, W, U, bi, bh)
    a, h1, c1 = O.lstm(x[:11], W, U, bi, bh)
    b, h2, c2 = O.lstm(x[11:], W, U, bi, bh, h0=h1, c0=c1)
    assert np.array_equal(np.concatenate([a, b]), full) and np.array_equal(h2, hf) and np.array_equal(c2, cf)


def test_batch_norm_and_activations_vs_numpy():
    r = rng(6)
    x = r.standard_normal((13, 7)).astype(np.float32)
    g, be, mu, var = (r.uniform(.5, 1.5, 7).astype(np.float32) for _ in range(4))
    want = ((x - mu) / np.sqrt(var + np.float32(1e-3))) * g + be
    assert np.array_equal(O.batch_norm(x, g, be, mu, var, 1e-3), want.astype(np.float32))
    assert np.array_equal(O.activation(O.ACT_RELU, x, relu_a=0.5), np.maximum(x, 0) * np.float32(0.5))
    sm = O.activation(O.ACT_SOFTMAX, x, softmax_vector_size=7)
    assert np.allclose(sm.sum(-1), 1, atol=1e-6)
    assert np.allclose(O.activation(O.ACT_SIGMOID, x), 1 / (1 + np.exp(-x.astype(np.float64))), atol=1e-7)


def test_tdd_vs_numpy_and_mel():
    r = rng(7)
    x, W, b = r.standard_normal((9, 5)).astype(np.float32), r.standard_normal((5, 7)).astype(np.float32), r.standard_normal(7).astype(np.float32)
    assert np.abs(O.time_distributed_dense(x, W, b) - (x.astype(np.float64) @ W + b)).max() < 2e-6
    w = O.mel_filterbank_weights(40, 512, 16000, 20.0, 8000.0)
    assert w.shape == (257, 40) and (w >= 0).all() and (w[0] == 0).all() and w.max() <= 1.0
