"""Thin Python mirror of the reference's layer protocol over the C boundary
(capi.py): ``XConfigCreate -> XCreateForInference -> XGetWeights (copy weights in)
-> XApply* -> XDestroy``.  Names and argument meaning follow the reference's C API
(conv_1d.h, batch_norm.h, gru.h, lstm.h, dense.h, time_distributed_dense.h,
spectrogram.h) so parity tests read like a C caller.  No arithmetic happens here.

``apply(np_array)`` uses the host-pointer C entry points; ``apply_device(tensor)``
passes torch-ROCm device pointers to the ``*ApplyDevice`` entry points (torch is
only the owner of HBM buffers and streams).
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import check, fp

ACT_IDENTITY, ACT_SIGMOID, ACT_TANH, ACT_RELU, ACT_SOFTMAX = "identity", "sigmoid", "tanh", "relu", "softmax"


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(fp)


def _fill(ptr, arr):
    arr = _f32(arr).ravel()
    C.memmove(ptr, arr.ctypes.data, arr.nbytes)


def _dp(t):
    """device pointer of a contiguous float32 torch tensor"""
    assert t.is_cuda and t.is_contiguous() and t.dtype.is_floating_point and t.element_size() == 4
    return C.c_void_p(t.data_ptr())


def use_torch_stream():
    """Route every launch to torch's current HIP stream so torch events/timing see the kernels."""
    import torch
    capi.load().nntk_hip_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))


class Activation:
    def __init__(self, kind, size, a=1.0, vector_size=0):
        L = capi.load()
        self.kind, self.size, self.vector_size = kind, size, vector_size
        if kind == ACT_IDENTITY:
            self.h = L.ActivationFunctionCreateIdentity(size)
        elif kind == ACT_SIGMOID:
            self.h = L.ActivationFunctionCreateSigmoid(size)
        elif kind == ACT_TANH:
            self.h = L.ActivationFunctionCreateTanh(size)
        elif kind == ACT_RELU:
            self.h = L.ActivationFunctionCreateReLU(size, C.c_float(a))
        elif kind == ACT_SOFTMAX:
            self.h = L.ActivationFunctionCreateSoftmax(size, vector_size)
        else:
            raise ValueError(kind)

    def apply(self, x):
        x = _f32(x)
        out = np.empty_like(x)
        capi.load().ActivationFunctionApply(self.h, _p(x), _p(out))
        err = capi.last_error()
        if err:
            raise capi.NNTKError(err)
        return out

    def apply_device(self, x, out=None, size=0):
        out = x.new_empty(x.shape) if out is None else out
        check(capi.load().ActivationFunctionApplyDevice(self.h, _dp(x), _dp(out), size), "ActivationFunctionApplyDevice")
        return out

    def destroy(self):
        if self.h:
            capi.load().ActivationFunctionDestroy(self.h)
            self.h = None


class Conv1d:
    def __init__(self, cin, cout, k, stride, input_size):
        L = capi.load()
        self.cfg = L.Conv1dConfigCreate(cin, cout, k, stride, input_size)
        self.h = L.Conv1dCreateForInference(self.cfg)
        if not self.h:
            raise capi.NNTKError("Conv1dCreateForInference: " + capi.last_error())

    def set_weights(self, W, b):
        w = capi.load().Conv1dGetWeights(self.h).contents
        _fill(w.W, W)
        _fill(w.b, b)

    @property
    def out_shape(self):
        return (self.cfg.output_size, self.cfg.output_feature_channels)

    def apply(self, x):
        x = _f32(x)
        L = capi.load()
        if x.ndim == 2:
            out = np.empty(self.out_shape, np.float32)
            check(L.Conv1dApplyInference(self.h, _p(x), _p(out)), "Conv1dApplyInference")
        else:
            out = np.empty((x.shape[0],) + self.out_shape, np.float32)
            check(L.Conv1dApplyInferenceBatch(self.h, _p(x), _p(out), x.shape[0]), "Conv1dApplyInferenceBatch")
        return out

    def apply_device(self, x, out=None, bn=None, act=None):
        B = x.shape[0]
        if out is None:
            out = x.new_empty((B,) + self.out_shape)
        L = capi.load()
        if bn is None and act is None:
            check(L.Conv1dApplyDevice(self.h, _dp(x), _dp(out), B), "Conv1dApplyDevice")
        else:
            check(L.Conv1dBatchNormActivationApplyDevice(self.h, bn.h if bn else None, act.h if act else None,
                                                         _dp(x), _dp(out), B), "Conv1dBatchNormActivationApplyDevice")
        return out

    def apply_device_frag3(self, x, out_f3=None, bn=None, act=None):
        """Conv1dBatchNormActivationApplyDeviceFrag3: the (fused) layer output [B, Tout, Cout] as a frag3 buffer."""
        B = x.shape[0]
        L = capi.load()
        if out_f3 is None:
            out_f3 = x.new_empty(L.nntk_frag3_floats(B, *self.out_shape))
        check(L.Conv1dBatchNormActivationApplyDeviceFrag3(self.h, bn.h if bn else None, act.h if act else None,
                                                          _dp(x), _dp(out_f3), B), "Conv1dBatchNormActivationApplyDeviceFrag3")
        return out_f3

    def sync_weights(self):
        check(capi.load().Conv1dSyncWeights(self.h), "Conv1dSyncWeights")

    def destroy(self):
        if self.h:
            capi.load().Conv1dDestroy(self.h)
            self.h = None


class BatchNorm:
    def __init__(self, channels, epsilon, count):
        L = capi.load()
        self.cfg = L.BatchNormConfigCreate(channels, C.c_float(epsilon), count)
        self.h = L.BatchNormCreateForInference(self.cfg)

    def set_weights(self, gamma, beta, mean, var):
        w = capi.load().BatchNormGetWeights(self.h).contents
        _fill(w.gamma, gamma)
        _fill(w.beta, beta)
        _fill(w.moving_mean, mean)
        _fill(w.moving_variance, var)

    def apply(self, x):
        x = _f32(x)
        out = np.empty_like(x)
        check(capi.load().BatchNormApplyInference(self.h, _p(x), _p(out)), "BatchNormApplyInference")
        return out

    def apply_device(self, x, out=None):
        out = x.new_empty(x.shape) if out is None else out
        rows = x.numel() // self.cfg.feature_channels
        check(capi.load().BatchNormApplyDevice(self.h, _dp(x), _dp(out), rows), "BatchNormApplyDevice")
        return out

    def destroy(self):
        if self.h:
            capi.load().BatchNormDestroy(self.h)
            self.h = None


class _Recurrent:
    def set_weights(self, W, U, b_i, b_h):
        w = self._get_weights(self.h).contents
        _fill(w.W, W)
        _fill(w.U, U)
        _fill(w.b_i, b_i)
        _fill(w.b_h, b_h)

    def _out(self, x_shape_prefix, alloc):
        T, H = self.cfg.base.timesteps, self.cfg.base.output_feature_channels
        tail = (T, H) if self.cfg.base.return_sequences else (H,)
        return alloc(tuple(x_shape_prefix) + tail)

    def apply(self, x):
        x = _f32(x)
        if x.ndim == 2:     # one sequence, stateful
            out = self._out((), lambda s: np.empty(s, np.float32))
            check(self._apply(self.h, _p(x), _p(out)), type(self).__name__ + "ApplyInference")
        else:
            out = self._out((x.shape[0],), lambda s: np.empty(s, np.float32))
            check(self._apply_batch(self.h, _p(x), _p(out), x.shape[0]), type(self).__name__ + "ApplyInferenceBatch")
        return out

    def apply_device(self, x, out=None):
        if out is None:
            out = self._out((x.shape[0],), lambda s: x.new_empty(s))
        check(self._apply_device(self.h, _dp(x), _dp(out), x.shape[0]), type(self).__name__ + "ApplyDevice")
        return out

    def sync_weights(self):
        check(self._sync(self.h), "SyncWeights")

    def reset_state(self):
        check(self._reset(self.h), "ResetState")

    def destroy(self):
        if self.h:
            self._destroy(self.h)
            self._acts_destroy(self.acts)
            self.h = None


class GRU(_Recurrent):
    def __init__(self, in_features, hidden, return_sequences, timesteps, acts=None):
        L = capi.load()
        self.acts = L.GRUActivationsCreateDefault(hidden) if acts is None else acts
        self.cfg = L.GRUConfigCreate(in_features, hidden, return_sequences, timesteps, self.acts)
        self.h = L.GRUCreateForInference(self.cfg)
        self._get_weights, self._apply, self._apply_batch = L.GRUGetWeights, L.GRUApplyInference, L.GRUApplyInferenceBatch
        self._apply_device, self._sync, self._reset = L.GRUApplyDevice, L.GRUSyncWeights, L.GRUResetState
        self._destroy, self._acts_destroy = L.GRUDestroy, L.GRUActivationsDestroy

    def state(self):
        h = np.empty(self.cfg.base.output_feature_channels, np.float32)
        check(capi.load().GRUGetState(self.h, _p(h)), "GRUGetState")
        return h


def gru_stack2_apply_device(g1, g2, x, out=None):
    """GRUStack2ApplyDevice: two stacked GRU layers in one persistent launch (zero initial state)."""
    H, T = g2.cfg.base.output_feature_channels, g2.cfg.base.timesteps
    if out is None:
        out = x.new_empty((x.shape[0], T, H) if g2.cfg.base.return_sequences else (x.shape[0], H))
    check(capi.load().GRUStack2ApplyDevice(g1.h, g2.h, _dp(x), _dp(out), x.shape[0]), "GRUStack2ApplyDevice")
    return out


# ---- frag3 tensors (nntoolkitcore_hip.h: activations pre-split for the split-bf16 x 3 contraction, MFMA fragment order) ----

def frag3_pack_device(x):
    """[B, T, C] f32 device tensor -> its frag3 form (a flat f32-typed device buffer of nntk_frag3_floats(B, T, C) floats)."""
    B, T, Cc = x.shape
    L = capi.load()
    out = x.new_empty(L.nntk_frag3_floats(B, T, Cc))
    check(L.nntk_frag3_pack_device(_dp(x), _dp(out), B, T, Cc), "nntk_frag3_pack_device")
    return out


def frag3_unpack_device(f3, B, T, Cc):
    """frag3 buffer -> [B, T, C] f32 device tensor (exact: hi + mid + lo)."""
    out = f3.new_empty((B, T, Cc))
    check(capi.load().nntk_frag3_unpack_device(_dp(f3), _dp(out), B, T, Cc), "nntk_frag3_unpack_device")
    return out


def recurrent_apply_device_frag3(layer, x=None, x_f3=None, batch=None, want_f32=True, want_f3=False, out=None, out_f3=None):
    """<GRU|LSTM>ApplyDeviceFrag3: input as f32 `x` [B, T, in] or frag3 `x_f3` (then `batch` is required); returns (out_f32 | None,
    out_frag3 | None).  `out` / `out_f3`: preallocated outputs (they imply want_f32 / want_f3)."""
    L = capi.load()
    fn = L.LSTMApplyDeviceFrag3 if isinstance(layer, LSTM) else L.GRUApplyDeviceFrag3
    B = x.shape[0] if x is not None else batch
    src = x if x is not None else x_f3
    H, T = layer.cfg.base.output_feature_channels, layer.cfg.base.timesteps
    if out is None and want_f32:
        out = src.new_empty((B, T, H) if layer.cfg.base.return_sequences else (B, H))
    out3 = out_f3
    if out3 is None and want_f3:
        out3 = src.new_empty(L.nntk_frag3_floats(B, T, H))
    check(fn(layer.h, _dp(x) if x is not None else None, _dp(x_f3) if x_f3 is not None else None,
             _dp(out) if out is not None else None, _dp(out3) if out3 is not None else None, B), "ApplyDeviceFrag3")
    return out, out3


def tdd_apply_device_frag3(tdd, x_f3, batch, out=None):
    """TimeDistributedDenseApplyDeviceFrag3: the input [batch, ts, in] in frag3 form."""
    if out is None:
        out = x_f3.new_empty((batch, tdd.cfg.ts, tdd.cfg.dense.output_size))
    check(capi.load().TimeDistributedDenseApplyDeviceFrag3(tdd.h, _dp(x_f3), _dp(out), batch), "TimeDistributedDenseApplyDeviceFrag3")
    return out


# ---- frag2h tensors (nntoolkitcore_hip.h: a bounded activation tensor, |x| < 2, as two f16 images for the three-product contraction) ----

def frag2h_pack_device(x):
    """[B, T, C] f32 device tensor with |x| < 2 -> its frag2h form (a flat f32-typed device buffer of nntk_frag2h_floats(B, T, C) floats)."""
    B, T, Cc = x.shape
    L = capi.load()
    out = x.new_empty(L.nntk_frag2h_floats(B, T, Cc))
    check(L.nntk_frag2h_pack_device(_dp(x), _dp(out), B, T, Cc), "nntk_frag2h_pack_device")
    return out


def frag2h_unpack_device(h2, B, T, Cc):
    """frag2h buffer -> [B, T, C] f32 device tensor ((hi + lo) 2^-15: the value to 2^-23 relative at worst)."""
    out = h2.new_empty((B, T, Cc))
    check(capi.load().nntk_frag2h_unpack_device(_dp(h2), _dp(out), B, T, Cc), "nntk_frag2h_unpack_device")
    return out


def lstm_apply_device_frag2h(lstm, x=None, x_f3=None, batch=None, out_h2=None):
    """LSTMApplyDeviceFrag2h: input as f32 `x` [B, T, in] or frag3 `x_f3` (then `batch` is required); returns the sequence output in frag2h form."""
    L = capi.load()
    B = x.shape[0] if x is not None else batch
    src = x if x is not None else x_f3
    H, T = lstm.cfg.base.output_feature_channels, lstm.cfg.base.timesteps
    if out_h2 is None:
        out_h2 = src.new_empty(L.nntk_frag2h_floats(B, T, H))
    check(L.LSTMApplyDeviceFrag2h(lstm.h, _dp(x) if x is not None else None, _dp(x_f3) if x_f3 is not None else None, _dp(out_h2), B),
          "LSTMApplyDeviceFrag2h")
    return out_h2


def tdd_apply_device_frag2h(tdd, x_h2, batch, out=None):
    """TimeDistributedDenseApplyDeviceFrag2h: the input [batch, ts, in] in frag2h form."""
    if out is None:
        out = x_h2.new_empty((batch, tdd.cfg.ts, tdd.cfg.dense.output_size))
    check(capi.load().TimeDistributedDenseApplyDeviceFrag2h(tdd.h, _dp(x_h2), _dp(out), batch), "TimeDistributedDenseApplyDeviceFrag2h")
    return out


def lstm_tdd_apply_device(lstm, tdd, x, out=None):
    """LSTMTimeDistributedDenseApplyDevice: LSTM (return_sequences) -> TimeDistributedDense with the tensor in between in frag3 form."""
    if out is None:
        out = x.new_empty((x.shape[0], tdd.cfg.ts, tdd.cfg.dense.output_size))
    check(capi.load().LSTMTimeDistributedDenseApplyDevice(lstm.h, tdd.h, _dp(x), _dp(out), x.shape[0]), "LSTMTimeDistributedDenseApplyDevice")
    return out


def gru_stack2_apply(g1, g2, x):
    """GRUStack2ApplyInferenceBatch on host arrays [B, T, in]."""
    x = _f32(x)
    H, T = g2.cfg.base.output_feature_channels, g2.cfg.base.timesteps
    out = np.empty((x.shape[0], T, H) if g2.cfg.base.return_sequences else (x.shape[0], H), np.float32)
    check(capi.load().GRUStack2ApplyInferenceBatch(g1.h, g2.h, _p(x), _p(out), x.shape[0]), "GRUStack2ApplyInferenceBatch")
    return out


class RNN(_Recurrent):
    """One-gate recurrent layer (rnn.h); `act` is an ActivationFunction handle (default tanh over H)."""

    def __init__(self, in_features, hidden, return_sequences, timesteps, v2=True, act=None):
        L = capi.load()
        self.act = L.ActivationFunctionCreateTanh(hidden) if act is None else act
        self.cfg = L.RNNConfigCreate(in_features, hidden, return_sequences, timesteps, v2, self.act)
        self.h = L.RNNCreateForInference(self.cfg)
        self._get_weights, self._apply, self._apply_batch = L.RNNGetWeights, L.RNNApplyInference, L.RNNApplyInferenceBatch
        self._apply_device, self._sync, self._reset = L.RNNApplyDevice, L.RNNSyncWeights, L.RNNResetState
        self._destroy = L.RNNDestroy
        self.acts = self.act
        self._acts_destroy = L.ActivationFunctionDestroy

    def state(self):
        H = self.cfg.base.output_feature_channels
        h = np.empty(H, np.float32)
        check(capi.load().RNNGetState(self.h, _p(h)), "RNNGetState")
        return h


def bd_reverse_device(x, kind="input"):
    """[B,T,F] device tensor with every sequence's rows in reverse time order (bidirectional.h)."""
    import torch
    L = capi.load()
    B, T, F = x.shape
    cfg = capi.RecurrentConfig(F, F, True, T)
    out = torch.empty_like(x)
    fn = L.bd_reverse_input_batch_device if kind == "input" else L.bd_reverse_backward_batch_device
    check(fn(_dp(x), _dp(out), cfg, B), "bd_reverse_*_device")
    return out


def bd_merge_device(fwd, bwd, mode="concat"):
    """fwd, bwd: [B,rows,C] (or [B,C]) device tensors -> concat along features or sum."""
    import torch
    L = capi.load()
    seq = fwd.dim() == 3
    B, Cc = fwd.shape[0], fwd.shape[-1]
    rows = fwd.shape[1] if seq else 1
    cfg = capi.RecurrentConfig(Cc, Cc, seq, rows)
    if mode == "concat":
        out = torch.empty(fwd.shape[:-1] + (2 * Cc,), device=fwd.device)
        check(L.bd_merge_concat_device(_dp(fwd), _dp(bwd), _dp(out), cfg, B), "bd_merge_concat_device")
    else:
        out = torch.empty_like(fwd)
        check(L.bd_merge_sum_device(_dp(fwd), _dp(bwd), _dp(out), cfg, B), "bd_merge_sum_device")
    return out


class LSTM(_Recurrent):
    def __init__(self, in_features, hidden, return_sequences, timesteps, v2=True, acts=None):
        L = capi.load()
        self.acts = L.LSTMActivationsCreateDefault(hidden) if acts is None else acts
        self.cfg = L.LSTMConfigCreate(in_features, hidden, return_sequences, timesteps, v2, self.acts)
        self.h = L.LSTMCreateForInference(self.cfg)
        self._get_weights, self._apply, self._apply_batch = L.LSTMGetWeights, L.LSTMApplyInference, L.LSTMApplyInferenceBatch
        self._apply_device, self._sync, self._reset = L.LSTMApplyDevice, L.LSTMSyncWeights, L.LSTMResetState
        self._destroy, self._acts_destroy = L.LSTMDestroy, L.LSTMActivationsDestroy

    def state(self):
        H = self.cfg.base.output_feature_channels
        h, c = np.empty(H, np.float32), np.empty(H, np.float32)
        check(capi.load().LSTMGetState(self.h, _p(h), _p(c)), "LSTMGetState")
        return h, c


class TimeDistributedDense:
    def __init__(self, ts, in_features, out_features, act=None):
        L = capi.load()
        self.act = act
        dense = L.DenseConfigCreate(in_features, out_features, act.h if act else None)
        self.cfg = L.TimeDistributedDenseConfigCreate(ts, dense)
        self.h = L.TimeDistributedDenseCreateForInference(self.cfg)

    def set_weights(self, W, b):
        w = capi.load().TimeDistributedDenseGetWeights(self.h).contents
        _fill(w.W, W)
        _fill(w.b, b)

    def apply(self, x):
        x = _f32(x)
        L = capi.load()
        ts, out = self.cfg.ts, self.cfg.dense.output_size
        if x.ndim == 2:
            o = np.empty((ts, out), np.float32)
            check(L.TimeDistributedDenseApplyInference(self.h, _p(x), _p(o)), "TimeDistributedDenseApplyInference")
        else:
            o = np.empty((x.shape[0], ts, out), np.float32)
            check(L.TimeDistributedDenseApplyInferenceBatch(self.h, _p(x), _p(o), x.shape[0]),
                  "TimeDistributedDenseApplyInferenceBatch")
        return o

    def apply_device(self, x, out=None):
        if out is None:
            out = x.new_empty((x.shape[0], self.cfg.ts, self.cfg.dense.output_size))
        check(capi.load().TimeDistributedDenseApplyDevice(self.h, _dp(x), _dp(out), x.shape[0]),
              "TimeDistributedDenseApplyDevice")
        return out

    def destroy(self):
        if self.h:
            capi.load().TimeDistributedDenseDestroy(self.h)
            self.h = None


class Dense:
    def __init__(self, in_features, out_features, act=None):
        L = capi.load()
        self.act = act
        self.cfg = L.DenseConfigCreate(in_features, out_features, act.h if act else None)
        self.h = L.DenseCreateForInference(self.cfg)

    def set_weights(self, W, b):
        w = capi.load().DenseGetWeights(self.h).contents
        _fill(w.W, W)
        _fill(w.b, b)

    def apply(self, x):
        x = _f32(x)
        o = np.empty(self.cfg.output_size, np.float32)
        check(capi.load().DenseApplyInference(self.h, _p(x), _p(o)), "DenseApplyInference")
        return o

    def destroy(self):
        if self.h:
            capi.load().DenseDestroy(self.h)
            self.h = None


_WINDOWS = ("ones", "hann_window", "hamming_window", "periodic_hann_window", "periodic_hamming_window",
            "blackman_window")


def window(name, size):
    assert name in _WINDOWS
    v = np.empty(size, np.float32)
    getattr(capi.load(), name)(_p(v), size)
    return v


class Spectrogram:
    def __init__(self, nfft, window_size, noverlap, input_size, mode="magnitude", fs=16000, fft_norm=1.0,
                 window_name="hann_window"):
        L = capi.load()
        self.cfg = L.SpectrogramConfigCreate(nfft, window_size, noverlap, input_size, C.c_float(fft_norm))
        self.h = L.SpectrogramCreateMagnitude(self.cfg) if mode == "magnitude" else L.SpectrogramCreatePSD(self.cfg, fs)
        if window_name:
            L.SpectrogramSetWindowFunc(self.h, C.cast(getattr(L, window_name), C.c_void_p))

    @property
    def out_shape(self):
        return (self.cfg.ntime_series, self.cfg.nfreq)

    def apply(self, x):
        x = _f32(x)
        L = capi.load()
        if x.ndim == 1:
            out = np.full(self.out_shape, np.nan, np.float32)
            L.SpectrogramApply(self.h, _p(x), _p(out))
            if capi.last_error():
                raise capi.NNTKError(capi.last_error())
        else:
            out = np.empty((x.shape[0],) + self.out_shape, np.float32)
            check(L.SpectrogramApplyBatch(self.h, _p(x), _p(out), x.shape[0]), "SpectrogramApplyBatch")
        return out

    def apply_device(self, x, out=None):
        if out is None:
            out = x.new_empty((x.shape[0],) + self.out_shape)
        check(capi.load().SpectrogramApplyDevice(self.h, _dp(x), _dp(out), x.shape[0]), "SpectrogramApplyDevice")
        return out

    def destroy(self):
        if self.h:
            capi.load().SpectrogramDestroy(self.h)
            self.h = None
