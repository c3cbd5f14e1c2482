"""ctypes binding of libnntoolkitcore_hip.so -- the C boundary declared in
include/nntoolkitcore_hip.h.  This is exactly the binding a maintainer of a Python
caller of the reference would write; nothing here computes anything.

Loading fails loudly (ImportError) when the shared library has not been built:
there is no Python or CPU fallback for the product path.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libnntoolkitcore_hip.so")
if os.environ.get("NNTK_LIB"):      # A/B of two builds on one box (tools/ab_lib.sh)
    LIB_PATH = os.path.abspath(os.environ["NNTK_LIB"])

fp = C.POINTER(C.c_float)
vp = C.c_void_p


# ---- by-value config structs (layouts checked against the reference headers in tests/test_abi.py)
class Conv1dConfig(C.Structure):
    _fields_ = [("input_feature_channels", C.c_int), ("output_feature_channels", C.c_int),
                ("kernel_size", C.c_int), ("stride", C.c_int), ("input_size", C.c_int), ("output_size", C.c_int)]


class ConvTrainingConfig(C.Structure):
    _fields_ = [("mini_batch_size", C.c_int)]


class DefaultGradient(C.Structure):
    _fields_ = [("d_W", fp), ("d_b", fp), ("d_X", fp)]


class BatchNormTrainingConfig(C.Structure):
    _fields_ = [("momentum", C.c_float), ("mini_batch_size", C.c_int)]


class BatchNormGradient(C.Structure):
    _fields_ = [("d_gamma", fp), ("d_beta", fp), ("d_x", fp)]


class RecurrentGradient(C.Structure):
    _fields_ = [("d_W", fp), ("d_U", fp), ("d_b_i", fp), ("d_b_h", fp), ("d_X", fp)]


class DFTConfig(C.Structure):
    _fields_ = [("nfft", C.c_int), ("forward", C.c_bool), ("complex", C.c_bool)]


class ComplexFloatSplit(C.Structure):
    _fields_ = [("real_p", fp), ("imag_p", fp)]


class SGD(C.Structure):
    _fields_ = [("learning_rate", C.c_float)]


class DefaultWeights(C.Structure):
    _fields_ = [("W", fp), ("b", fp)]


class BatchNormWeights(C.Structure):
    _fields_ = [("gamma", fp), ("beta", fp), ("moving_mean", fp), ("moving_variance", fp)]


class BatchNormConfig(C.Structure):
    _fields_ = [("feature_channels", C.c_int), ("epsilon", C.c_float), ("count", C.c_int)]


class RecurrentWeights(C.Structure):
    _fields_ = [("W", fp), ("U", fp), ("b_i", fp), ("b_h", fp)]


class RecurrentConfig(C.Structure):
    _fields_ = [("input_feature_channels", C.c_int), ("output_feature_channels", C.c_int),
                ("return_sequences", C.c_bool), ("timesteps", C.c_int)]


class GRUActivations(C.Structure):
    _fields_ = [("z_gate_activation", vp), ("h_gate_activation", vp), ("r_gate_activation", vp)]


class GRUConfig(C.Structure):
    _fields_ = [("base", RecurrentConfig), ("activations", GRUActivations)]


class LSTMActivations(C.Structure):
    _fields_ = [("candidate_gate_activation", vp), ("input_gate_activation", vp), ("forget_gate_activation", vp),
                ("output_gate_activation", vp), ("output_activation", vp)]


class LSTMConfig(C.Structure):
    _fields_ = [("base", RecurrentConfig), ("v2", C.c_bool), ("activations", LSTMActivations)]


class RNNConfig(C.Structure):
    _fields_ = [("base", RecurrentConfig), ("v2", C.c_bool), ("activation", C.c_void_p)]


class DenseConfig(C.Structure):
    _fields_ = [("input_size", C.c_int), ("output_size", C.c_int), ("activation", vp)]


class TimeDistributedDenseConfig(C.Structure):
    _fields_ = [("dense", DenseConfig), ("ts", C.c_int)]


class SpectrogramConfig(C.Structure):
    _fields_ = [("nfft", C.c_int), ("window_size", C.c_int), ("noverlap", C.c_int), ("step", C.c_int),
                ("input_size", C.c_int), ("nfreq", C.c_int), ("ntime_series", C.c_int),
                ("fft_normalization_factor", C.c_float)]


class MelFilterBankConfig(C.Structure):
    _fields_ = [("n_mels", C.c_int), ("n_fft", C.c_int), ("sample_rate", C.c_int), ("lower_hz", C.c_float),
                ("upper_hz", C.c_float)]


WINDOW_FN = C.CFUNCTYPE(None, fp, C.c_int)
ACT_IMPL_FN = C.CFUNCTYPE(None, vp, fp, fp, C.c_int)

# name -> (restype, argtypes); this table is also the list of symbols the library must export
SIGNATURES = {
    # activation.h / activation_default.h
    "ActivationFunctionCreate": (vp, [C.c_int, vp, vp, vp, vp, vp]),
    "ActivationFunctionDestroy": (None, [vp]),
    "ActivationFunctionApply": (None, [vp, fp, fp]),
    "ActivationFunctionCreateIdentity": (vp, [C.c_int]),
    "ActivationFunctionCreateSoftmax": (vp, [C.c_int, C.c_int]),
    "ActivationFunctionCreateSigmoid": (vp, [C.c_int]),
    "ActivationFunctionCreateReLU": (vp, [C.c_int, C.c_float]),
    "ActivationFunctionCreateTanh": (vp, [C.c_int]),
    # conv_1d.h
    "Conv1dConfigCreate": (Conv1dConfig, [C.c_int] * 5),
    "Conv1dCreateForInference": (vp, [Conv1dConfig]),
    "Conv1dGetWeights": (C.POINTER(DefaultWeights), [vp]),
    "Conv1dApplyInference": (C.c_int, [vp, fp, fp]),
    "Conv1dDestroy": (None, [vp]),
    "Conv1dCreateForTraining": (vp, [Conv1dConfig, ConvTrainingConfig]),
    "Conv1dCreateGradient": (C.POINTER(DefaultGradient), [Conv1dConfig, ConvTrainingConfig]),
    "ConvGradientDestroy": (None, [C.POINTER(DefaultGradient)]),
    "Conv1dApplyTrainingBatch": (C.c_int, [vp, fp, fp]),
    "Conv1dCalculateGradient": (None, [vp, C.POINTER(DefaultGradient), fp]),
    "Conv1dApplyTrainingBatchDevice": (C.c_int, [vp, vp, vp]),
    "Conv1dCalculateGradientDevice": (C.c_int, [vp, vp, vp, vp]),
    # batch_norm.h
    "BatchNormConfigCreate": (BatchNormConfig, [C.c_int, C.c_float, C.c_int]),
    "BatchNormCreateForInference": (vp, [BatchNormConfig]),
    "BatchNormGetWeights": (C.POINTER(BatchNormWeights), [vp]),
    "BatchNormApplyInference": (C.c_int, [vp, fp, fp]),
    "BatchNormTrainingConfigCreate": (BatchNormTrainingConfig, [C.c_float, C.c_int]),
    "BatchNormCreateForTraining": (vp, [BatchNormConfig, BatchNormTrainingConfig]),
    "BatchNormGradientCreate": (C.POINTER(BatchNormGradient), [BatchNormConfig, BatchNormTrainingConfig]),
    "BatchNormGradientDestroy": (None, [C.POINTER(BatchNormGradient)]),
    "BatchNormApplyTrainingBatch": (C.c_int, [vp, fp, fp]),
    "BatchNormCalculateGradient": (None, [vp, C.POINTER(BatchNormGradient), fp]),
    "BatchNormDestroy": (None, [vp]),
    # recurrent.h / gru.h
    "RecurrentConfigCreate": (RecurrentConfig, [C.c_int, C.c_int, C.c_bool, C.c_int]),
    "GRUActivationsCreate": (GRUActivations, [vp, vp, vp]),
    "GRUActivationsCreateDefault": (GRUActivations, [C.c_int]),
    "GRUActivationsDestroy": (None, [GRUActivations]),
    "GRUConfigCreate": (GRUConfig, [C.c_int, C.c_int, C.c_bool, C.c_int, GRUActivations]),
    "GRUGetWeights": (C.POINTER(RecurrentWeights), [vp]),
    "GRUCreateForInference": (vp, [GRUConfig]),
    "GRUApplyInference": (C.c_int, [vp, fp, fp]),
    "DFTConfigCreate": (DFTConfig, [C.c_int, C.c_bool, C.c_bool]),
    "DFTSetupCreate": (vp, [DFTConfig]),
    "DFTPerform": (None, [vp, C.POINTER(ComplexFloatSplit), C.POINTER(ComplexFloatSplit)]),
    "DFTSetupDestroy": (None, [vp]),
    "split_complex": (None, [vp, C.POINTER(ComplexFloatSplit), C.c_int]),
    "join_complex_split": (None, [C.POINTER(ComplexFloatSplit), vp, C.c_int]),
    "bd_merge_concat_gradient": (None, [fp, fp, fp, RecurrentConfig, C.c_int, fp]),
    "bd_merge_sum_gradient": (None, [fp, fp, fp, RecurrentConfig, C.c_int]),
    "bd_accumulate_d_x": (None, [fp, fp, fp, RecurrentConfig, C.c_int]),
    "RNNCreateForTraining": (vp, [RNNConfig, ConvTrainingConfig]),
    "RNNGradientCreate": (C.POINTER(RecurrentGradient), [RNNConfig, ConvTrainingConfig]),
    "RNNApplyTrainingBatch": (C.c_int, [vp, fp, fp]),
    "RNNCalculateGradient": (None, [vp, C.POINTER(RecurrentGradient), fp]),
    "TimeDistributedDenseCreateForTraining": (vp, [TimeDistributedDenseConfig, ConvTrainingConfig]),
    "TimeDistributedDenseGradientCreate": (C.POINTER(DefaultGradient), [vp]),
    "TimeDistributedDenseApplyTrainingBatch": (C.c_int, [vp, fp, fp]),
    "TimeDistributedDenseCalculateGradient": (None, [vp, C.POINTER(DefaultGradient), fp]),
    "LSTMCreateForTraining": (vp, [LSTMConfig, ConvTrainingConfig]),
    "LSTMGradientCreate": (C.POINTER(RecurrentGradient), [LSTMConfig, ConvTrainingConfig]),
    "LSTMApplyTrainingBatch": (C.c_int, [vp, fp, fp]),
    "LSTMCalculateGradient": (None, [vp, C.POINTER(RecurrentGradient), fp]),
    "LSTMApplyTrainingBatchDevice": (C.c_int, [vp, vp, vp]),
    "LSTMCalculateGradientDevice": (C.c_int, [vp, vp, vp, vp]),
    "GRUApplyTrainingBatchDevice": (C.c_int, [vp, vp, vp]),
    "GRUCalculateGradientDevice": (C.c_int, [vp, vp, vp, vp]),
    "GRUCreateForTraining": (vp, [GRUConfig, ConvTrainingConfig]),
    "GRUGradientCreate": (C.POINTER(RecurrentGradient), [GRUConfig, ConvTrainingConfig]),
    "RecurrentGradientDestroy": (None, [C.POINTER(RecurrentGradient)]),
    "GRUApplyTrainingBatch": (C.c_int, [vp, fp, fp]),
    "GRUCalculateGradient": (None, [vp, C.POINTER(RecurrentGradient), fp]),
    "GRUDestroy": (None, [vp]),
    # rnn.h
    "RNNConfigCreate": (RNNConfig, [C.c_int, C.c_int, C.c_bool, C.c_int, C.c_bool, vp]),
    "RNNGetWeights": (C.POINTER(RecurrentWeights), [vp]),
    "RNNCreateForInference": (vp, [RNNConfig]),
    "RNNApplyInference": (C.c_int, [vp, fp, fp]),
    "RNNDestroy": (None, [vp]),
    # bidirectional.h (forward helpers)
    "bd_reverse_input_batch": (None, [fp, fp, RecurrentConfig, C.c_int]),
    "bd_reverse_backward_batch": (None, [fp, fp, RecurrentConfig, C.c_int]),
    "bd_merge_concat_buffer_size": (C.c_int, [RecurrentConfig]),
    "bd_merge_concat": (None, [fp, fp, fp, RecurrentConfig, C.c_int, fp]),
    "bd_merge_sum": (None, [fp, fp, fp, RecurrentConfig, C.c_int]),
    # lstm.h
    "LSTMActivationsCreate": (LSTMActivations, [vp] * 5),
    "LSTMActivationsCreateDefault": (LSTMActivations, [C.c_int]),
    "LSTMActivationsDestroy": (None, [LSTMActivations]),
    "LSTMConfigCreate": (LSTMConfig, [C.c_int, C.c_int, C.c_bool, C.c_int, C.c_bool, LSTMActivations]),
    "LSTMGetWeights": (C.POINTER(RecurrentWeights), [vp]),
    "LSTMCreateForInference": (vp, [LSTMConfig]),
    "LSTMApplyInference": (C.c_int, [vp, fp, fp]),
    "LSTMDestroy": (None, [vp]),
    # dense.h / time_distributed_dense.h
    "DenseConfigCreate": (DenseConfig, [C.c_int, C.c_int, vp]),
    "DenseCreateForInference": (vp, [DenseConfig]),
    "DenseGetWeights": (C.POINTER(DefaultWeights), [vp]),
    "DenseApplyInference": (C.c_int, [vp, fp, fp]),
    # training, second slice
    "DenseCreateForTraining": (vp, [DenseConfig, ConvTrainingConfig]),
    "DenseApplyTrainingBatch": (C.c_int, [vp, fp, fp]),
    "DenseApplyTrainingBatchDevice": (C.c_int, [vp, vp, vp]),
    "DenseCalculateGradientDevice": (C.c_int, [vp, vp, vp, vp]),
    "TimeDistributedDenseApplyTrainingBatchDevice": (C.c_int, [vp, vp, vp]),
    "TimeDistributedDenseCalculateGradientDevice": (C.c_int, [vp, vp, vp, vp]),
    "BatchNormApplyTrainingBatchDevice": (C.c_int, [vp, vp, vp]),
    "BatchNormCalculateGradientDevice": (C.c_int, [vp, vp, vp, vp, vp]),
    "DenseGradientCreate": (C.POINTER(DefaultGradient), [DenseConfig, ConvTrainingConfig]),
    "DenseGradientCreateFromFilter": (C.POINTER(DefaultGradient), [vp]),
    "DenseGradientDestroy": (None, [C.POINTER(DefaultGradient)]),
    "DenseCalculateGradient": (None, [vp, C.POINTER(DefaultGradient), fp]),
    "ActivationFunctionCalculateGradient": (None, [vp, fp, fp, fp, fp]),
    "ActivationFunctionCalculateGradientDevice": (C.c_int, [vp, vp, vp, vp, vp, C.c_int]),
    "mean_squared_error": (C.c_float, [fp, fp, C.c_int, C.c_int]),
    "mean_squared_error_derivative": (None, [fp, fp, fp, C.c_int, C.c_int]),
    "categorical_crossentropy": (C.c_float, [fp, fp, C.c_int, C.c_int]),
    "categorical_crossentropy_derivative": (None, [fp, fp, fp, C.c_int, C.c_int]),
    "sgd_optimize": (C.c_int, [SGD, fp, fp, C.c_int]),
    "nntk_mean_squared_error_device": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "nntk_categorical_crossentropy_device": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "nntk_mean_squared_error_derivative_device": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
    "nntk_categorical_crossentropy_derivative_device": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
    "nntk_sgd_optimize_device": (C.c_int, [SGD, vp, vp, C.c_long]),
    "DenseDestroy": (None, [vp]),
    "TimeDistributedDenseConfigCreate": (TimeDistributedDenseConfig, [C.c_int, DenseConfig]),
    "TimeDistributedDenseCreateForInference": (vp, [TimeDistributedDenseConfig]),
    "TimeDistributedDenseGetWeights": (C.POINTER(DefaultWeights), [vp]),
    "TimeDistributedDenseApplyInference": (C.c_int, [vp, fp, fp]),
    "TimeDistributedDenseDestroy": (None, [vp]),
    # window.h / spectrogram.h
    "hamming_window": (None, [fp, C.c_int]),
    "hann_window": (None, [fp, C.c_int]),
    "ones": (None, [fp, C.c_int]),
    "periodic_hamming_window": (None, [fp, C.c_int]),
    "periodic_hann_window": (None, [fp, C.c_int]),
    "blackman_window": (None, [fp, C.c_int]),
    "SpectrogramConfigCreate": (SpectrogramConfig, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "SpectrogramCreatePSD": (vp, [SpectrogramConfig, C.c_int]),
    "SpectrogramCreateMagnitude": (vp, [SpectrogramConfig]),
    "SpectrogramGetConfig": (SpectrogramConfig, [vp]),
    "SpectrogramSetWindowFunc": (None, [vp, vp]),
    "SpectrogramSetScaleFactor": (None, [vp, C.c_float]),
    "SpectrogramApply": (None, [vp, fp, fp]),
    "SpectrogramDestroy": (None, [vp]),
    # mel_filterbank.h / log_mel_spectrogram.h
    "MelFilterBankConfigCreate": (MelFilterBankConfig, [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]),
    "MelFilterBankCreate": (vp, [MelFilterBankConfig]),
    "MelFilterBankApply": (None, [vp, fp, fp, C.c_int]),
    "MelFilterBankDestroy": (None, [vp]),
    "LogMelSpectrogramCreate": (vp, [vp, MelFilterBankConfig]),
    "LogMelSpectrogramApply": (None, [vp, fp, fp]),
    "LogMelSpectrogramDestroy": (None, [vp]),
    # ---- additive API
    "nntk_hip_device_count": (C.c_int, []),
    "nntk_hip_set_device": (C.c_int, [C.c_int]),
    "nntk_hip_set_stream": (None, [vp]),
    "nntk_hip_get_stream": (vp, []),
    "nntk_hip_synchronize": (C.c_int, []),
    "nntk_last_error": (C.c_char_p, []),
    "nntk_version": (C.c_char_p, []),
    "nntk_build_source_hash": (C.c_char_p, []),
    "nntk_hip_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "nntk_hip_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "nntk_hip_device_status": (C.c_int, []),
    "nntk_hip_last_recurrent_kernel": (C.c_char_p, []),
    "nntk_hip_last_conv_kernel": (C.c_char_p, []),
    "nntk_hip_profile_enable": (None, [C.c_int]),
    "nntk_hip_profile_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "nntk_device_alloc": (vp, [C.c_size_t]),
    "nntk_device_free": (None, [vp]),
    "nntk_device_upload": (C.c_int, [vp, fp, C.c_size_t]),
    "nntk_device_download": (C.c_int, [fp, vp, C.c_size_t]),
    "nntk_dist_get_unique_id": (C.c_int, [C.c_char_p]),
    "nntk_dist_init": (C.c_int, [C.c_char_p, C.c_int, C.c_int]),
    "nntk_dist_rank": (C.c_int, []),
    "nntk_dist_world_size": (C.c_int, []),
    "nntk_dist_broadcast": (C.c_int, [fp, C.c_size_t, C.c_int]),
    "nntk_dist_barrier": (C.c_int, []),
    "nntk_dist_allreduce": (C.c_int, [fp, C.c_size_t]),
    "nntk_dist_allreduce_device": (C.c_int, [vp, C.c_size_t]),
    "nntk_dist_finalize": (C.c_int, []),
    "nntk_dist_shard_range": (None, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "Conv1dBroadcastWeights": (C.c_int, [vp, C.c_int]),
    "BatchNormBroadcastWeights": (C.c_int, [vp, C.c_int]),
    "GRUBroadcastWeights": (C.c_int, [vp, C.c_int]),
    "LSTMBroadcastWeights": (C.c_int, [vp, C.c_int]),
    "RNNBroadcastWeights": (C.c_int, [vp, C.c_int]),
    "DenseBroadcastWeights": (C.c_int, [vp, C.c_int]),
    "TimeDistributedDenseBroadcastWeights": (C.c_int, [vp, C.c_int]),
    "Conv1dSyncWeights": (C.c_int, [vp]),
    "BatchNormSyncWeights": (C.c_int, [vp]),
    "GRUSyncWeights": (C.c_int, [vp]),
    "LSTMSyncWeights": (C.c_int, [vp]),
    "RNNSyncWeights": (C.c_int, [vp]),
    "DenseSyncWeights": (C.c_int, [vp]),
    "TimeDistributedDenseSyncWeights": (C.c_int, [vp]),
    "Conv1dApplyInferenceBatch": (C.c_int, [vp, fp, fp, C.c_int]),
    "GRUApplyInferenceBatch": (C.c_int, [vp, fp, fp, C.c_int]),
    "LSTMApplyInferenceBatch": (C.c_int, [vp, fp, fp, C.c_int]),
    "RNNApplyInferenceBatch": (C.c_int, [vp, fp, fp, C.c_int]),
    "TimeDistributedDenseApplyInferenceBatch": (C.c_int, [vp, fp, fp, C.c_int]),
    "SpectrogramApplyBatch": (C.c_int, [vp, fp, fp, C.c_int]),
    "LogMelSpectrogramApplyBatch": (C.c_int, [vp, fp, fp, C.c_int]),
    "MelFilterBankApplyDevice": (C.c_int, [vp, vp, vp, C.c_int]),
    "LogMelSpectrogramApplyDevice": (C.c_int, [vp, vp, vp, C.c_int]),
    "SpectrogramApplyDevice": (C.c_int, [vp, vp, vp, C.c_int]),
    "Conv1dApplyDevice": (C.c_int, [vp, vp, vp, C.c_int]),
    "Conv1dBatchNormActivationApplyDevice": (C.c_int, [vp, vp, vp, vp, vp, C.c_int]),
    "Conv1dBatchNormActivationApplyDeviceFrag3": (C.c_int, [vp, vp, vp, vp, vp, C.c_int]),
    "BatchNormApplyDevice": (C.c_int, [vp, vp, vp, C.c_int]),
    "ActivationFunctionApplyDevice": (C.c_int, [vp, vp, vp, C.c_int]),
    "GRUApplyDevice": (C.c_int, [vp, vp, vp, C.c_int]),
    "GRUStack2ApplyDevice": (C.c_int, [vp, vp, vp, vp, C.c_int]),
    "GRUStack2ApplyInferenceBatch": (C.c_int, [vp, vp, fp, fp, C.c_int]),
    "LSTMApplyDevice": (C.c_int, [vp, vp, vp, C.c_int]),
    # frag3 tensors (additive)
    "nntk_frag3_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "nntk_frag3_pack_device": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int]),
    "nntk_frag3_unpack_device": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int]),
    "nntk_frag2h_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "nntk_frag2h_pack_device": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int]),
    "nntk_frag2h_unpack_device": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int]),
    "LSTMApplyDeviceFrag2h": (C.c_int, [vp, vp, vp, vp, C.c_int]),
    "TimeDistributedDenseApplyDeviceFrag2h": (C.c_int, [vp, vp, vp, C.c_int]),
    "GRUKernelPlan": (C.c_char_p, [vp]),
    "LSTMKernelPlan": (C.c_char_p, [vp]),
    "GRUApplyDeviceFrag3": (C.c_int, [vp, vp, vp, vp, vp, C.c_int]),
    "LSTMApplyDeviceFrag3": (C.c_int, [vp, vp, vp, vp, vp, C.c_int]),
    "TimeDistributedDenseApplyDeviceFrag3": (C.c_int, [vp, vp, vp, C.c_int]),
    "LSTMTimeDistributedDenseApplyDevice": (C.c_int, [vp, vp, vp, vp, C.c_int]),
    "RNNApplyDevice": (C.c_int, [vp, vp, vp, C.c_int]),
    "bd_reverse_input_batch_device": (C.c_int, [vp, vp, RecurrentConfig, C.c_int]),
    "bd_reverse_backward_batch_device": (C.c_int, [vp, vp, RecurrentConfig, C.c_int]),
    "bd_merge_concat_device": (C.c_int, [vp, vp, vp, RecurrentConfig, C.c_int]),
    "bd_merge_sum_device": (C.c_int, [vp, vp, vp, RecurrentConfig, C.c_int]),
    "DenseApplyDevice": (C.c_int, [vp, vp, vp, C.c_int]),
    "TimeDistributedDenseApplyDevice": (C.c_int, [vp, vp, vp, C.c_int]),
    "GRUResetState": (C.c_int, [vp]),
    "LSTMResetState": (C.c_int, [vp]),
    "RNNResetState": (C.c_int, [vp]),
    "RNNGetState": (C.c_int, [vp, fp]),
    "GRUGetState": (C.c_int, [vp, fp]),
    "LSTMGetState": (C.c_int, [vp, fp, fp]),
}

_lib = None


def load():
    """dlopen the product library and declare every entry point.  Needs libamdhip64
    (ROCm) at load time but no GPU: only Apply/alloc calls touch the device."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libnntoolkitcore_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python nntoolkitcore_amd/_build.py`. There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: torch-ROCm bundles its own libamdhip64.so.7.  Import it
    # first so our NEEDED libamdhip64.so.7 binds to the copy torch already loaded --
    # otherwise torch streams / device pointers would belong to a different runtime
    # instance than the one our launches go through.  (A plain C caller without torch
    # simply gets the system ROCm runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().nntk_last_error().decode()


def set_option(name, value):
    """nntk_hip_set_option: tuning / diagnostics knob by name; value "auto" restores the default."""
    rc = load().nntk_hip_set_option(name.encode(), str(value).encode())
    if rc != 0:
        raise NNTKError("nntk_hip_set_option(%s): %s" % (name, last_error()))


def get_option(name):
    v = C.c_int()
    if load().nntk_hip_get_option(name.encode(), C.byref(v)) != 0:
        raise NNTKError("nntk_hip_get_option(%s): %s" % (name, last_error()))
    return v.value


class NNTKError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise NNTKError("%s failed (rc=%d): %s" % (what, rc, last_error()))
