"""ctypes driver for oracle/_build/libnnref_oracle.so (CPU oracle; test infrastructure).

Every wrapper takes/returns C-contiguous float32 numpy arrays and forwards to the
C restatement in nnref_signal.c / nnref_layers.c, which cite the reference
file:line they follow.  Nothing here is used by the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libnnref_oracle.so")

ACT_NONE = -1
ACT_IDENTITY, ACT_SIGMOID, ACT_TANH, ACT_RELU, ACT_SOFTMAX = 0, 1, 2, 3, 4
WIN_ONES, WIN_HANN, WIN_HAMMING, WIN_PERIODIC_HANN, WIN_PERIODIC_HAMMING, WIN_BLACKMAN = range(6)
WINDOW_KINDS = {
    "ones": WIN_ONES, "hann": WIN_HANN, "hamming": WIN_HAMMING,
    "periodic_hann": WIN_PERIODIC_HANN, "periodic_hamming": WIN_PERIODIC_HAMMING,
    "blackman": WIN_BLACKMAN,
}


def build(force=False):
    """Compile the oracle shared library (gcc, seconds)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("nnref_signal.c", "nnref_layers.c", "nnref.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "_build/libnnref_oracle.so"])
    return _LIB_PATH


_lib = None
_fp = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ref_op_vec_dot.restype = C.c_float
        _lib.ref_spectrogram_scale_magnitude.restype = C.c_float
        _lib.ref_spectrogram_scale_psd.restype = C.c_float
        _lib.ref_conv1d_output_size.restype = C.c_int
        _lib.ref_kiss_fft.restype = C.c_int
        _lib.ref_spectrogram.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_fp)


# ------------------------------------------------------------------ signal ---

def window(kind, size):
    if isinstance(kind, str):
        kind = WINDOW_KINDS[kind]
    v = np.empty(size, np.float32)
    lib().ref_window(C.c_int(kind), _p(v), C.c_int(size))
    return v


def kiss_fft(x_complex, inverse=False):
    x = np.ascontiguousarray(x_complex, dtype=np.complex64)
    out = np.empty_like(x)
    rc = lib().ref_kiss_fft(C.c_int(x.size), C.c_int(int(inverse)),
                            x.view(np.float32).ctypes.data_as(_fp), out.view(np.float32).ctypes.data_as(_fp))
    assert rc == 0
    return out


def spectrogram_geometry(nfft, window_size, noverlap, input_size):
    step, nfreq, nts = C.c_int(), C.c_int(), C.c_int()
    lib().ref_spectrogram_geometry(nfft, window_size, noverlap, input_size,
                                   C.byref(step), C.byref(nfreq), C.byref(nts))
    return step.value, nfreq.value, nts.value


def spectrogram_scale(win, mode="magnitude", fs=16000):
    win = _f32(win)
    if mode == "magnitude":
        return float(lib().ref_spectrogram_scale_magnitude(_p(win), C.c_int(win.size)))
    return float(lib().ref_spectrogram_scale_psd(_p(win), C.c_int(win.size), C.c_int(fs)))


def spectrogram(x, win, nfft, noverlap, mode="magnitude", fs=16000, fft_norm=1.0, scale=None):
    """x: [N] or [B,N] -> [nts,nfreq] or [B,nts,nfreq]"""
    x = _f32(x)
    win = _f32(win)
    squeeze = x.ndim == 1
    xb = x.reshape(1, -1) if squeeze else x
    n = xb.shape[1]
    _, nfreq, nts = spectrogram_geometry(nfft, win.size, noverlap, n)
    if scale is None:
        scale = spectrogram_scale(win, mode, fs)
    out = np.empty((xb.shape[0], nts, nfreq), np.float32)
    for b in range(xb.shape[0]):
        rc = lib().ref_spectrogram(_p(xb[b]), _p(win), _p(out[b]), nfft, win.size, noverlap, n,
                                   C.c_float(fft_norm), C.c_int(0 if mode == "magnitude" else 1),
                                   C.c_float(scale))
        assert rc == 0
    return out[0] if squeeze else out


def mel_filterbank_weights(n_mels, n_fft, sample_rate, lower_hz, upper_hz):
    nb = n_fft // 2 + 1
    w = np.empty((nb, n_mels), np.float32)
    lib().ref_mel_filterbank_weights(n_mels, n_fft, sample_rate, C.c_float(lower_hz), C.c_float(upper_hz), _p(w))
    return w


def log_mel(spec, weights):
    spec = _f32(spec)
    weights = _f32(weights)
    ts, nb = spec.shape
    out = np.empty((ts, weights.shape[1]), np.float32)
    lib().ref_log_mel(_p(spec), _p(weights), _p(out), ts, nb, weights.shape[1])
    return out


# ------------------------------------------------------------------ layers ---

def conv1d_output_size(T, k, stride):
    return int(lib().ref_conv1d_output_size(T, k, stride))


def conv1d(x, W, b, stride=1):
    """x: [T,Cin] or [B,T,Cin]; W: [Cout,Cin,k]; b: [Cout]"""
    x, W, b = _f32(x), _f32(W), _f32(b)
    squeeze = x.ndim == 2
    xb = x[None] if squeeze else x
    B, T, Cin = xb.shape
    Cout, Cin2, k = W.shape
    assert Cin == Cin2
    Tout = conv1d_output_size(T, k, stride)
    out = np.empty((B, Tout, Cout), np.float32)
    lib().ref_conv1d_batch(_p(xb), _p(W), _p(b), _p(out), B, T, Cin, Cout, k, stride)
    return out[0] if squeeze else out


def conv1d_gradient(x, W, dout, stride=1):
    """Conv1dCalculateGradient on a fresh (zeroed) gradient block: returns (dW [Cout,Cin,k], db [Cout], dX [B,T,Cin])."""
    x, W, dout = _f32(x), _f32(W), _f32(dout)
    B, T, Cin = x.shape
    Cout, _, k = W.shape
    dW, db, dX = np.zeros_like(W), np.zeros(Cout, np.float32), np.empty_like(x)
    lib().ref_conv1d_gradient(_p(x), _p(W), _p(dout), _p(dW), _p(db), _p(dX), B, T, Cin, Cout, k, stride)
    return dW, db, dX


def activation_gradient(kind, z, a, dout, softmax_vector_size=0):
    """ActivationFunctionCalculateGradient for ONE call of a handle sized to the arrays (a=None: non-cached form)."""
    dout = _f32(dout)
    z = None if z is None else _f32(z)
    a = None if a is None else _f32(a)
    out = np.empty_like(dout)
    size = dout.size if kind != ACT_SOFTMAX else dout.size // softmax_vector_size
    lib().ref_activation_gradient(C.c_int(kind), C.c_int(softmax_vector_size), _p(z) if z is not None else None,
                                  _p(a) if a is not None else None, _p(dout), _p(out), C.c_int(size))
    return out


def dense_forward_training(x, W, b, act=None, softmax_vector_size=0):
    """(z, a) [B,out] as DenseApplyTrainingBatch caches them."""
    x = _f32(x)
    z = time_distributed_dense(x, W, b)
    a = z.copy() if act is None else np.stack([activation(act, z[i], softmax_vector_size=softmax_vector_size) for i in range(x.shape[0])])
    return z, a


def dense_gradient(x, W, z, a, dout, act=None, softmax_vector_size=0, gW=None, gb=None):
    """DenseCalculateGradient: returns (gW [in,out], gb [out], dX [B,in]); gW / gb start from the given blocks (zeros)."""
    x, W, z, a, dout = _f32(x), _f32(W), _f32(z), _f32(a), _f32(dout)
    B, n_in = x.shape
    n_out = W.shape[1]
    gW = np.zeros_like(W) if gW is None else _f32(gW).copy()
    gb = np.zeros(n_out, np.float32) if gb is None else _f32(gb).copy()
    dX = np.empty_like(x)
    kind = -1 if act is None else act
    act_size = n_out if act != ACT_SOFTMAX else n_out // softmax_vector_size
    lib().ref_dense_gradient(_p(x), _p(W), _p(z), _p(a), _p(dout), C.c_int(kind), C.c_int(softmax_vector_size),
                             C.c_int(act_size), _p(gW), _p(gb), _p(dX), B, n_in, n_out)
    return gW, gb, dX


def mean_squared_error(y, p):
    y, p = _f32(y), _f32(p)
    lib().ref_mean_squared_error.restype = C.c_float
    return float(lib().ref_mean_squared_error(_p(y), _p(p), C.c_int(y.shape[1]), C.c_int(y.shape[0])))


def mean_squared_error_derivative(y, p):
    y, p = _f32(y), _f32(p)
    d = np.empty_like(y)
    lib().ref_mean_squared_error_derivative(_p(y), _p(p), _p(d), C.c_int(y.shape[1]), C.c_int(y.shape[0]))
    return d


def categorical_crossentropy(y, p):
    y, p = _f32(y), _f32(p)
    lib().ref_categorical_crossentropy.restype = C.c_float
    return float(lib().ref_categorical_crossentropy(_p(y), _p(p), C.c_int(y.shape[1]), C.c_int(y.shape[0])))


def categorical_crossentropy_derivative(y, p, fill=np.nan):
    """AS WRITTEN in the reference: only row 0 is computed; the other rows keep `fill`."""
    y, p = _f32(y), _f32(p)
    d = np.full_like(y, fill)
    lib().ref_categorical_crossentropy_derivative(_p(y), _p(p), _p(d), C.c_int(y.shape[1]), C.c_int(y.shape[0]))
    return d


def sgd_optimize(lr, g, w):
    g, w = _f32(g), _f32(w).copy()
    lib().ref_sgd_optimize(C.c_float(lr), _p(g), _p(w), C.c_int(w.size))
    return w


def batch_norm_training_forward(x, gamma, beta, eps, momentum, moving_mean, moving_var):
    """BatchNormApplyTrainingBatch on x [N, F]: returns (out, batch_mean, batch_var, new_moving_mean, new_moving_var)."""
    x = _f32(x)
    N, F = x.shape
    out, mean, var = np.empty_like(x), np.empty(F, np.float32), np.empty(F, np.float32)
    mm, mv = _f32(moving_mean).copy(), _f32(moving_var).copy()
    lib().ref_batch_norm_training_forward(_p(x), _p(_f32(gamma)), _p(_f32(beta)), C.c_float(eps), C.c_float(momentum),
                                          _p(out), _p(mean), _p(var), _p(mm), _p(mv), N, F)
    return out, mean, var, mm, mv


def batch_norm_gradient(x, dout, gamma, mean, var, eps):
    """BatchNormCalculateGradient: returns (d_beta, d_gamma, d_x)."""
    x, dout = _f32(x), _f32(dout)
    N, F = x.shape
    db, dg, dx = np.empty(F, np.float32), np.empty(F, np.float32), np.empty_like(x)
    lib().ref_batch_norm_gradient(_p(x), _p(dout), _p(_f32(gamma)), _p(_f32(mean)), _p(_f32(var)), C.c_float(eps),
                                  _p(db), _p(dg), _p(dx), N, F)
    return db, dg, dx


def batch_norm(x, gamma, beta, mean, var, eps):
    x = _f32(x)
    C_ = x.shape[-1]
    out = np.empty_like(x)
    lib().ref_batch_norm(_p(x), _p(_f32(gamma)), _p(_f32(beta)), _p(_f32(mean)), _p(_f32(var)), _p(out),
                         C.c_float(eps), C.c_int(x.size // C_), C.c_int(C_))
    return out


def activation(kind, x, relu_a=1.0, softmax_vector_size=0):
    x = _f32(x)
    out = np.empty_like(x)
    size = x.size if kind != ACT_SOFTMAX else x.size // softmax_vector_size
    lib().ref_activation(C.c_int(kind), C.c_float(relu_a), C.c_int(softmax_vector_size), _p(x), _p(out), C.c_int(size))
    return out


def _gate_scales(relu_a):
    """ReLU output scales of the gate activations for the NEXT oracle call (None = all 1)."""
    if relu_a is None:
        lib().ref_set_gate_relu_scales(None, 0)
    else:
        a = _f32(list(relu_a))
        lib().ref_set_gate_relu_scales(_p(a), int(a.size))


def gru(x, W, U, b_i, b_h, h0=None, return_sequences=True, acts=(ACT_SIGMOID, ACT_TANH, ACT_SIGMOID), relu_a=None):
    """x: [T,in] (stateful single sequence; returns (out, h_final)) or [B,T,in]
    (zero state per sequence; returns out).  acts = (z, h, r); relu_a = their ReLU output scales."""
    _gate_scales(relu_a)
    try:
        return _gru(x, W, U, b_i, b_h, h0, return_sequences, acts)
    finally:
        _gate_scales(None)


def _gru(x, W, U, b_i, b_h, h0, return_sequences, acts):
    x, W, U, b_i, b_h = _f32(x), _f32(W), _f32(U), _f32(b_i), _f32(b_h)
    H = U.shape[0]
    if x.ndim == 2:
        T, in_ = x.shape
        h = np.zeros(H, np.float32) if h0 is None else _f32(h0).copy()
        out = np.empty((T, H) if return_sequences else (H,), np.float32)
        lib().ref_gru_sequence(_p(x), _p(W), _p(U), _p(b_i), _p(b_h), _p(h), _p(out), T, in_, H,
                               int(return_sequences), acts[0], acts[1], acts[2])
        return out, h
    B, T, in_ = x.shape
    out = np.empty((B, T, H) if return_sequences else (B, H), np.float32)
    lib().ref_gru_batch(_p(x), _p(W), _p(U), _p(b_i), _p(b_h), _p(out), B, T, in_, H,
                        int(return_sequences), acts[0], acts[1], acts[2])
    return out


def gru_training(x, W, U, b_i, b_h, dout, return_sequences=True, acts=(ACT_SIGMOID, ACT_TANH, ACT_SIGMOID), grads0=None):
    """GRUApplyTrainingBatch + GRUCalculateGradient: returns (h [B,T,H], (gW, gU, gbi, gbh, dX)); grads0 = starting block."""
    x, W, U, b_i, b_h, dout = (_f32(a) for a in (x, W, U, b_i, b_h, dout))
    B, T, n_in = x.shape
    H = U.shape[0]
    h, Zg, hU = np.empty((B, T, H), np.float32), np.empty((B, T, 6 * H), np.float32), np.empty((B, T, H), np.float32)
    lib().ref_gru_training_forward(_p(x), _p(W), _p(U), _p(b_i), _p(b_h), _p(h), _p(Zg), _p(hU), B, T, n_in, H, *acts)
    if grads0 is None:
        gW, gU, gbi, gbh = np.zeros_like(W), np.zeros_like(U), np.zeros_like(b_i), np.zeros_like(b_h)
    else:
        gW, gU, gbi, gbh = (_f32(a).copy() for a in grads0)
    dX = np.empty_like(x)
    lib().ref_gru_gradient(_p(x), _p(W), _p(U), _p(h), _p(Zg), _p(hU), _p(dout), C.c_int(1 if return_sequences else 0),
                           _p(gW), _p(gU), _p(gbi), _p(gbh), _p(dX), B, T, n_in, H, *acts)
    return h, (gW, gU, gbi, gbh, dX)


def lstm_training(x, W, U, b_i, b_h, dout, return_sequences=True, v2=True,
                  acts=(ACT_SIGMOID, ACT_SIGMOID, ACT_TANH, ACT_SIGMOID, ACT_TANH), grads0=None):
    """LSTMApplyTrainingBatch + LSTMCalculateGradient: returns (h [B,T,H], (gW, gU, gbi, gbh, dX))."""
    x, W, U, b_i, b_h, dout = (_f32(a) for a in (x, W, U, b_i, b_h, dout))
    B, T, n_in = x.shape
    H = U.shape[0]
    h, c, z = np.empty((B, T, H), np.float32), np.empty((B, T, H), np.float32), np.empty((B, T, 8 * H), np.float32)
    lib().ref_lstm_training_forward(_p(x), _p(W), _p(U), _p(b_i), _p(b_h), _p(h), _p(c), _p(z), B, T, n_in, H, C.c_int(1 if v2 else 0), *acts)
    if grads0 is None:
        gW, gU, gbi, gbh = np.zeros_like(W), np.zeros_like(U), np.zeros_like(b_i), np.zeros_like(b_h)
    else:
        gW, gU, gbi, gbh = (_f32(a).copy() for a in grads0)
    dX = np.empty_like(x)
    lib().ref_lstm_gradient(_p(x), _p(W), _p(U), _p(h), _p(c), _p(z), _p(dout), C.c_int(1 if return_sequences else 0),
                            _p(gW), _p(gU), _p(gbi), _p(gbh), _p(dX), B, T, n_in, H, *acts)
    return h, (gW, gU, gbi, gbh, dX)


def rnn_training(x, W, U, b_i, b_h, dout, return_sequences=True, v2=True, act=ACT_TANH):
    """RNNApplyTrainingBatch + RNNCalculateGradient: returns (h [B,T,H], (gW, gU, gbi, gbh, dX))."""
    x, W, U, b_i, b_h, dout = (_f32(a) for a in (x, W, U, b_i, b_h, dout))
    B, T, n_in = x.shape
    H = U.shape[0]
    h, gate = np.empty((B, T, H), np.float32), np.empty((B, T, H), np.float32)
    lib().ref_rnn_training_forward(_p(x), _p(W), _p(U), _p(b_i), _p(b_h), _p(h), _p(gate), B, T, n_in, H, C.c_int(1 if v2 else 0), C.c_int(act))
    gW, gU, gbi, gbh, dX = np.zeros_like(W), np.zeros_like(U), np.zeros_like(b_i), np.zeros_like(b_h), np.empty_like(x)
    lib().ref_rnn_gradient(_p(x), _p(W), _p(U), _p(h), _p(gate), _p(dout), C.c_int(1 if return_sequences else 0),
                           _p(gW), _p(gU), _p(gbi), _p(gbh), _p(dX), B, T, n_in, H, C.c_int(act))
    return h, (gW, gU, gbi, gbh, dX)


def rnn(x, W, U, b_i, b_h, h0=None, return_sequences=True, v2=True, act=ACT_TANH, relu_a=None):
    """One-gate RNN.  x: [T,in] (stateful single sequence; returns (out, h_final)) or [B,T,in] (zero state)."""
    _gate_scales(None if relu_a is None else [relu_a])
    try:
        return _rnn(x, W, U, b_i, b_h, h0, return_sequences, v2, act)
    finally:
        _gate_scales(None)


def _rnn(x, W, U, b_i, b_h, h0, return_sequences, v2, act):
    x, W, U, b_i, b_h = _f32(x), _f32(W), _f32(U), _f32(b_i), _f32(b_h)
    H = U.shape[0]
    if x.ndim == 2:
        T, in_ = x.shape
        h = np.zeros(H, np.float32) if h0 is None else _f32(h0).copy()
        out = np.empty((T, H) if return_sequences else (H,), np.float32)
        lib().ref_rnn_sequence(_p(x), _p(W), _p(U), _p(b_i), _p(b_h), _p(h), _p(out), T, in_, H,
                               int(return_sequences), int(v2), int(act))
        return out, h
    B, T, in_ = x.shape
    out = np.empty((B, T, H) if return_sequences else (B, H), np.float32)
    lib().ref_rnn_batch(_p(x), _p(W), _p(U), _p(b_i), _p(b_h), _p(out), B, T, in_, H,
                        int(return_sequences), int(v2), int(act))
    return out


def bd_reverse(x):
    """[B,T,F] with the rows of every sequence in reverse time order (bd_reverse_input_batch / _backward_batch)."""
    x = _f32(x)
    B, T, F = x.shape
    out = np.empty_like(x)
    lib().ref_bd_reverse_batch(_p(x), _p(out), B, T, F)
    return out


def bd_merge(fwd, bwd, mode="concat"):
    """fwd, bwd: [B,rows,C] -> [B,rows,2C] (concat) or [B,rows,C] (sum)."""
    fwd, bwd = _f32(fwd), _f32(bwd)
    B, rows, Cc = fwd.shape
    if mode == "concat":
        out = np.empty((B, rows, 2 * Cc), np.float32)
        lib().ref_bd_merge_concat(_p(fwd), _p(bwd), _p(out), B, rows, Cc)
    else:
        out = np.empty((B, rows, Cc), np.float32)
        lib().ref_bd_merge_sum(_p(fwd), _p(bwd), _p(out), B, rows, Cc)
    return out


def lstm(x, W, U, b_i, b_h, h0=None, c0=None, return_sequences=True, v2=True,
         acts=(ACT_SIGMOID, ACT_SIGMOID, ACT_TANH, ACT_SIGMOID, ACT_TANH), relu_a=None):
    """acts = (input, forget, candidate, output_gate, output); relu_a = their ReLU output scales."""
    _gate_scales(relu_a)
    try:
        return _lstm(x, W, U, b_i, b_h, h0, c0, return_sequences, v2, acts)
    finally:
        _gate_scales(None)


def _lstm(x, W, U, b_i, b_h, h0, c0, return_sequences, v2, acts):
    x, W, U, b_i, b_h = _f32(x), _f32(W), _f32(U), _f32(b_i), _f32(b_h)
    H = U.shape[0]
    if x.ndim == 2:
        T, in_ = x.shape
        h = np.zeros(H, np.float32) if h0 is None else _f32(h0).copy()
        c = np.zeros(H, np.float32) if c0 is None else _f32(c0).copy()
        out = np.empty((T, H) if return_sequences else (H,), np.float32)
        lib().ref_lstm_sequence(_p(x), _p(W), _p(U), _p(b_i), _p(b_h), _p(h), _p(c), _p(out), T, in_, H,
                                int(return_sequences), int(v2), *[int(a) for a in acts])
        return out, h, c
    B, T, in_ = x.shape
    out = np.empty((B, T, H) if return_sequences else (B, H), np.float32)
    lib().ref_lstm_batch(_p(x), _p(W), _p(U), _p(b_i), _p(b_h), _p(out), B, T, in_, H,
                         int(return_sequences), int(v2), *[int(a) for a in acts])
    return out


def time_distributed_dense(x, W, b, act=ACT_NONE, relu_a=1.0, softmax_vector_size=0, act_size=None):
    """x: [..., in] -> [..., out]; rows are independent Dense applications."""
    x, W, b = _f32(x), _f32(W), _f32(b)
    in_, out_ = W.shape
    rows = x.size // in_
    out = np.empty(x.shape[:-1] + (out_,), np.float32)
    if act_size is None:
        act_size = 1 if act == ACT_SOFTMAX else out_
    lib().ref_time_distributed_dense(_p(x), _p(W), _p(b), _p(out), rows, in_, out_, C.c_int(act),
                                     C.c_float(relu_a), C.c_int(softmax_vector_size), C.c_int(act_size))
    return out
