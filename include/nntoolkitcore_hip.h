/*
 * nntoolkitcore_hip.h -- the drop-in C boundary of the MI355X-native
 * implementation of NNToolkitCore's time-series inference path
 *   Spectrogram -> Conv1d -> BatchNorm/Activation -> GRU/LSTM -> TimeDistributedDense.
 *
 * PART 1 re-declares the reference's own layer create/apply API for that path
 * (same symbol names, argument meaning, by-value struct layouts and 0 / -1 error
 * convention), so an application that includes the reference headers links
 * against libnntoolkitcore_hip.so unchanged.  Each block cites the reference
 * header it replaces.  The struct layouts are checked against the reference's
 * real headers by tests/test_abi_and_symbols.py (golden: tests/golden/ref_probe.json).
 *
 * PART 2 is ADDITIVE: batched [batch, time, feature] entry points, device-pointer
 * variants (so a stack chains on the GPU without host round trips), fused
 * Conv1d+BatchNorm+ReLU, weight re-sync, stream / device selection and an error
 * string for the void functions.  No HIP or torch type appears in any signature:
 * device buffers are plain `float *` holding device addresses, streams `void *`.
 *
 * Host code above this header stays C; the kernels are hand-written HIP for
 * gfx950 behind the thin shim in nntoolkitcore_amd/csrc/hip/nntk_shim.h.
 * There is NO CPU fallback: if the HIP runtime or a GPU is missing, Apply calls
 * fail (-1 / nntk_last_error()).
 */
#ifndef NNTOOLKITCORE_HIP_H
#define NNTOOLKITCORE_HIP_H

#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ======================================================================= */
/* PART 1 -- reference API                                                   */
/* ======================================================================= */

/* ---- nntoolkitcore/layers/shared.h:12-34 ------------------------------- */
typedef struct { int mini_batch_size; } DefaultTrainingConfig;
typedef struct { int w; int b; int sum; } DefaultWeightsSize;
typedef struct { float *W; float *b; } DefaultWeights;

/* ---- nntoolkitcore/layers/activation.h:19-30 --------------------------- */
typedef void (*ActivationFunctionImpl)(void *, const float *, float *, int);
typedef void (*ActivationFunctionDerivative)(void *, const float *, const float *, float *, int);
typedef void (*ActivationImplementerDestroy)(void *);
typedef struct ActivationFunctionStruct *ActivationFunction;

/* Custom (host-callback) activations are accepted here for ABI compatibility
 * but cannot run on the device: a layer configured with one fails with -1. */
ActivationFunction ActivationFunctionCreate(int size, ActivationImplementerDestroy destroy_fn, void *implementer,
                                            ActivationFunctionImpl function, ActivationFunctionDerivative derivative,
                                            ActivationFunctionDerivative cached_derivative);
void ActivationFunctionDestroy(ActivationFunction filter);
/* host pointers; size fixed at create time (activation.c:23) */
void ActivationFunctionApply(ActivationFunction filter, const float *input, float *output);
/* activation.c:47-54: output = d_out * activation'(.) -- on the cached forward value `a` when the kind has a cached
 * derivative (sigmoid, tanh, softmax) and a != NULL, else on z.  Reference quirks kept: ReLU's derivative is
 * clamp(z, 0, 1) (activation_default.c:118-121), and inside ONE call every softmax vector reads d_out at the call's
 * base (activation_default.c:183).  The Device form takes device pointers; size <= 0 means the handle's size. */
void ActivationFunctionCalculateGradient(ActivationFunction filter, const float *z, const float *a, const float *d_out,
                                         float *output);
int  ActivationFunctionCalculateGradientDevice(ActivationFunction filter, const float *d_z, const float *d_a,
                                               const float *d_dout, float *d_output, int size);

/* ---- nntoolkitcore/layers/activation_default.h:19-27 ------------------- */
ActivationFunction ActivationFunctionCreateIdentity(int input_size);
ActivationFunction ActivationFunctionCreateSoftmax(int input_size, int vector_size);
ActivationFunction ActivationFunctionCreateSigmoid(int input_size);
ActivationFunction ActivationFunctionCreateReLU(int input_size, float a);
ActivationFunction ActivationFunctionCreateTanh(int input_size);

/* ---- nntoolkitcore/layers/conv_1d.h:22-55 ------------------------------ */
typedef struct {
    int input_feature_channels;
    int output_feature_channels;
    int kernel_size;
    int stride;
    int input_size;
    int output_size;
} Conv1dConfig;
typedef DefaultWeights ConvWeights;
typedef struct Conv1dStruct *Conv1d;

Conv1dConfig Conv1dConfigCreate(int input_feature_channels, int output_feature_channels, int kernel_size,
                                int stride, int inputSize);
Conv1d Conv1dCreateForInference(Conv1dConfig config);
ConvWeights *Conv1dGetWeights(Conv1d filter);            /* W [Cout][Cin][k] then b [Cout], one block */
int  Conv1dApplyInference(Conv1d filter, const float *input, float *output);   /* host, one sequence; -1 on a training handle */
void Conv1dDestroy(Conv1d filter);
/* training, first slice (layers/conv_1d.h:20, :43-51, shared.h:12-36): mini-batch forward that keeps its
 * input, and the gradient d_W / d_b (added to the block) / d_X (overwritten).  Deterministic, untuned. */
typedef DefaultTrainingConfig ConvTrainingConfig;
typedef struct { float *d_W; float *d_b; float *d_X; } DefaultGradient;      /* one zeroed block d_W | d_b | d_X */
typedef DefaultGradient ConvGradient;
Conv1d Conv1dCreateForTraining(Conv1dConfig config, ConvTrainingConfig training_config);
ConvGradient *Conv1dCreateGradient(Conv1dConfig config, ConvTrainingConfig training_config);
void ConvGradientDestroy(ConvGradient *gradient);
int  Conv1dApplyTrainingBatch(Conv1d filter, const float *input /*[mini_batch,T,Cin]*/, float *output);   /* -1 on an inference handle */
void Conv1dCalculateGradient(Conv1d filter, ConvGradient *gradient, const float *d_out /*[mini_batch,Tout,Cout]*/);
/* Additive device-pointer forms (tensors stay in HBM; enqueued on the calling thread's stream, but NOT asynchronous: each call
 * uploads the current weight block and waits for that copy, so it returns with the stream drained up to the upload; kernel-side
 * faults surface through nntk_hip_synchronize() / nntk_hip_device_status(); same kernels, same bits):
 * d_input must stay valid until the gradient call; d_grad_Wb = W [Cout][Cin][k] | b [Cout] is ADDED to, d_dX overwritten. */
int  Conv1dApplyTrainingBatchDevice(Conv1d filter, const float *d_input /*[mini_batch,T,Cin]*/, float *d_output);
int  Conv1dCalculateGradientDevice(Conv1d filter, float *d_grad_Wb, float *d_dX, const float *d_dout);

/* ---- nntoolkitcore/layers/batch_norm.h:19-66 --------------------------- */
typedef struct { float *gamma; float *beta; float *moving_mean; float *moving_variance; } BatchNormWeights;
typedef struct { int feature_channels; float epsilon; int count; } BatchNormConfig;
typedef struct BatchNormFilterStruct *BatchNorm;

BatchNormConfig BatchNormConfigCreate(int feature_channels, float epsilon, int count);
BatchNorm BatchNormCreateForInference(BatchNormConfig config);
BatchNormWeights *BatchNormGetWeights(BatchNorm filter);
int  BatchNormApplyInference(BatchNorm filter, const float *input, float *output);    /* -1 on a training-mode handle (batch_norm.c:167) */
void BatchNormDestroy(BatchNorm filter);
/* training (batch_norm.h:27-62, batch_norm.c:96-126, :191-386): forward over N = count * mini_batch rows with the batch's
 * own mean / biased variance, moving statistics updated in the weight block with `momentum`; the gradient OVERWRITES
 * d_beta, d_gamma [feature_channels] and d_x [N, feature_channels]. */
typedef struct { float momentum; int mini_batch_size; } BatchNormTrainingConfig;
typedef struct { float *d_gamma; float *d_beta; float *d_x; } BatchNormGradient;        /* block order: d_beta | d_gamma | d_x */
BatchNormTrainingConfig BatchNormTrainingConfigCreate(float momentum, int mini_batch_size);
BatchNorm BatchNormCreateForTraining(BatchNormConfig config, BatchNormTrainingConfig training_config);
BatchNormGradient *BatchNormGradientCreate(BatchNormConfig config, BatchNormTrainingConfig training_config);
void BatchNormGradientDestroy(BatchNormGradient *grad);
int  BatchNormApplyTrainingBatch(BatchNorm filter, const float *input, float *output);   /* -1 on an inference-mode handle */
void BatchNormCalculateGradient(BatchNorm filter, BatchNormGradient *gradient, float *d_out);
/* additive device-pointer forms of the two calls above (the host forms move the whole mini-batch over PCIe twice per call):
 * d_input must stay valid and unchanged until the matching gradient call; d_dbeta / d_dgamma [feature_channels] and d_dx are
 * overwritten like the host form's gradient block.  0 ok, -1 error. */
int  BatchNormApplyTrainingBatchDevice(BatchNorm filter, const float *d_input, float *d_output);
int  BatchNormCalculateGradientDevice(BatchNorm filter, float *d_dbeta, float *d_dgamma, float *d_dx, const float *d_dout);

/* ---- nntoolkitcore/layers/recurrent.h:17-56 ---------------------------- */
typedef struct { int w; int u; int b_i; int b_h; int sum; } RecurrentWeightsSize;
typedef struct { float *W; float *U; float *b_i; float *b_h; } RecurrentWeights;
typedef struct {
    int input_feature_channels;
    int output_feature_channels;
    bool return_sequences;
    int timesteps;
} RecurrentConfig;
RecurrentConfig RecurrentConfigCreate(int input_feature_channels, int output_feature_channels,
                                      bool return_sequences, int timesteps);

/* ---- nntoolkitcore/layers/gru.h:22-71 ---------------------------------- */
typedef RecurrentWeights GRUWeights;
typedef struct {
    ActivationFunction z_gate_activation;
    ActivationFunction h_gate_activation;
    ActivationFunction r_gate_activation;
} GRUActivations;
typedef struct { RecurrentConfig base; GRUActivations activations; } GRUConfig;
typedef struct GRUStruct *GRU;

GRUActivations GRUActivationsCreate(ActivationFunction z_gate_activation, ActivationFunction h_gate_activation,
                                    ActivationFunction r_gate_activation);
GRUActivations GRUActivationsCreateDefault(int size);
void GRUActivationsDestroy(GRUActivations activations);
GRUConfig GRUConfigCreate(int input_feature_channels, int output_feature_channels, bool return_sequences,
                          int timesteps, GRUActivations activations);
GRUWeights *GRUGetWeights(GRU filter);                   /* W [in,3H] | U [H,3H] | b_i [3H] | b_h [3H]; gates z,r,h */
/* training (gru.h:23-69, gru.c:232-512): forward over the mini-batch from a ZERO state per sequence keeping the gate
 * caches on the device, then back-propagation through time.  GRUCalculateGradient ADDS d_W, d_U, d_b_i, d_b_h onto the
 * block and overwrites d_X [mini_batch, timesteps, in]; d_out is [mini_batch, timesteps, out] if return_sequences else
 * [mini_batch, out].  Gate activations: built-in identity / sigmoid / tanh / ReLU. */
typedef struct { float *d_W; float *d_U; float *d_b_i; float *d_b_h; float *d_X; } RecurrentGradient;   /* one block, this order */
typedef DefaultTrainingConfig RecurrentTrainingConfig;
typedef RecurrentGradient GRUGradient;
typedef RecurrentTrainingConfig GRUTrainingConfig;
void RecurrentGradientDestroy(RecurrentGradient *gradient);
GRU  GRUCreateForTraining(GRUConfig config, GRUTrainingConfig training_config);
GRUGradient *GRUGradientCreate(GRUConfig config, GRUTrainingConfig training_config);
int  GRUApplyTrainingBatch(GRU filter, const float *input, float *output);      /* -1 on an inference-mode handle (gru.c:247) */
void GRUCalculateGradient(GRU filter, GRUGradient *gradients, float *d_out);
/* Additive device-pointer forms (tensors stay in HBM; enqueued on the calling thread's stream -- each call synchronises the stream
 * once for its weight upload, see Conv1dApplyTrainingBatchDevice; poll nntk_hip_device_status() for faults): d_input [B][T][in] must
 * stay valid until the gradient call; d_grad = W [in][3H] | U [H][3H] | b_i [3H] | b_h [3H] (the layout of the gradient block)
 * is ADDED to, d_dX [B][T][in] is overwritten.  Same kernels as the host-pointer forms: bit-identical results. */
int  GRUApplyTrainingBatchDevice(GRU filter, const float *d_input, float *d_output);
int  GRUCalculateGradientDevice(GRU filter, float *d_grad, float *d_dX, const float *d_dout);
GRU  GRUCreateForInference(GRUConfig config);
int  GRUApplyInference(GRU filter, const float *input, float *output);  /* host, one sequence, STATEFUL */
void GRUDestroy(GRU filter);

/* ---- nntoolkitcore/layers/lstm.h:20-75 --------------------------------- */
typedef RecurrentWeights LSTMWeights;
typedef struct {
    ActivationFunction candidate_gate_activation;
    ActivationFunction input_gate_activation;
    ActivationFunction forget_gate_activation;
    ActivationFunction output_gate_activation;
    ActivationFunction output_activation;
} LSTMActivations;
typedef struct { RecurrentConfig base; bool v2; LSTMActivations activations; } LSTMConfig;
typedef struct LSTMStruct *LSTM;

LSTMActivations LSTMActivationsCreate(ActivationFunction input_gate_activation,
                                      ActivationFunction forget_gate_activation,
                                      ActivationFunction candidate_gate_activation,
                                      ActivationFunction output_gate_activation,
                                      ActivationFunction output_activation);
LSTMActivations LSTMActivationsCreateDefault(int size);
void LSTMActivationsDestroy(LSTMActivations activations);
LSTMConfig LSTMConfigCreate(int input_feature_channels, int output_feature_channels, bool return_sequences,
                            int timesteps, bool v2, LSTMActivations activations);
LSTMWeights *LSTMGetWeights(LSTM filter);                /* W [in,4H] | U [H,4H] | b_i | b_h; gates i,f,g,o */
/* training (lstm.h:21-73, lstm.c:294-556): as for the GRU -- zero state per sequence, caches on the device, BPTT;
 * LSTMCalculateGradient ADDS d_W, d_U, d_b_i, d_b_h onto the block and overwrites d_X */
typedef RecurrentGradient LSTMGradient;
typedef RecurrentTrainingConfig LSTMTrainingConfig;
LSTM LSTMCreateForTraining(LSTMConfig config, LSTMTrainingConfig training_config);
LSTMGradient *LSTMGradientCreate(LSTMConfig config, LSTMTrainingConfig training_config);
int  LSTMApplyTrainingBatch(LSTM filter, const float *input, float *output);     /* -1 on an inference-mode handle (lstm.c:419) */
void LSTMCalculateGradient(LSTM filter, LSTMGradient *gradient, float *d_out);
/* device-pointer forms, as for the GRU (gate width 4H) */
int  LSTMApplyTrainingBatchDevice(LSTM filter, const float *d_input, float *d_output);
int  LSTMCalculateGradientDevice(LSTM filter, float *d_grad, float *d_dX, const float *d_dout);
LSTM LSTMCreateForInference(LSTMConfig config);
int  LSTMApplyInference(LSTM filter, const float *input, float *output); /* host, one sequence, STATEFUL */
void LSTMDestroy(LSTM filter);

/* ---- nntoolkitcore/layers/rnn.h:15-52 ------------ */
/* One-gate recurrent cell (rnn.c:144-166): h' = act((x W + b_i) + (h U [+ b_h if v2])).
 * W [in,out], U [out,out], b_i [out], b_h [out] in one block (rnn.c:36-46). */
typedef RecurrentWeights RNNWeights;
typedef struct { RecurrentConfig base; bool v2; ActivationFunction activation; } RNNConfig;
typedef struct RNNStruct *RNN;

RNNConfig RNNConfigCreate(int input_feature_channels, int output_feature_channels, bool return_sequences,
                          int timesteps, bool v2, ActivationFunction activation);
RNNWeights *RNNGetWeights(RNN filter);
/* training (rnn.h:16-48, rnn.c:184-221, :249-351): as for GRU / LSTM */
typedef RecurrentGradient RNNGradient;
typedef RecurrentTrainingConfig RNNTrainingConfig;
RNN  RNNCreateForTraining(RNNConfig config, RNNTrainingConfig training_config);
RNNGradient *RNNGradientCreate(RNNConfig config, RNNTrainingConfig training_config);
int  RNNApplyTrainingBatch(RNN filter, const float *input, float *output);
void RNNCalculateGradient(RNN filter, RNNGradient *gradients, float *d_out);
RNN  RNNCreateForInference(RNNConfig config);
/* host, one sequence, STATEFUL: T cells from the handle's h, output [T,out] or the last [out].  The
 * reference's loop (rnn.c:228-247) addresses its output at i * (i * out) and reloads h from the wrong row;
 * this is the layer its own batch forward pass (rnn.c:249-291) computes, not that indexing slip. */
int  RNNApplyInference(RNN filter, const float *input, float *output);
void RNNDestroy(RNN filter);

/* ---- nntoolkitcore/layers/bidirectional.h:15-46 (forward helpers) ------- */
/* host pointers, like the reference; the *_device forms below take device pointers */
void bd_reverse_input_batch(const float *input, float *output, RecurrentConfig config, int batch);    /* [B,T,in]  rows in reverse time order */
void bd_reverse_backward_batch(const float *input, float *output, RecurrentConfig config, int batch); /* [B,T,out] likewise */
int  bd_merge_concat_buffer_size(RecurrentConfig config);
/* output [B, rows, 2*out] = forward row | backward row (rows = timesteps if return_sequences else 1); buffer unused */
void bd_merge_concat(const float *forward_result, const float *backward_result, float *output,
                     RecurrentConfig config, int batch, float *buffer);
void bd_merge_sum(const float *forward_result, const float *backward_result, float *output,
                  RecurrentConfig config, int batch);
/* gradient helpers (bidirectional.h, bidirectional.c:58-74, :87-108); buffer unused */
void bd_merge_concat_gradient(const float *d_out, float *d_forward_out, float *d_backward_out, RecurrentConfig config,
                              int batch, float *buffer);
void bd_merge_sum_gradient(const float *d_out, float *d_forward_out, float *d_backward_out, RecurrentConfig config, int batch);
void bd_accumulate_d_x(const float *forward_dx, const float *backward_dx, float *output, RecurrentConfig config, int batch);

/* ---- nntoolkitcore/layers/dense.h:21-55 -------------------------------- */
typedef DefaultWeights DenseWeights;
typedef struct { int input_size; int output_size; ActivationFunction activation; } DenseConfig;
typedef struct DenseStruct *Dense;

DenseConfig DenseConfigCreate(int input_size, int output_size, ActivationFunction activation);
Dense DenseCreateForInference(DenseConfig config);
DenseWeights *DenseGetWeights(Dense filter);             /* W [in,out] row-major then b [out] */
int  DenseApplyInference(Dense filter, const float *input, float *output);     /* -1 on a training-mode handle (dense.c:136) */
void DenseDestroy(Dense filter);

/* training, second slice (dense.h:23-51, dense.c:85-119, :144-185): forward over the mini-batch keeping x, z, a on the
 * device; DenseCalculateGradient ADDS d_W, d_b onto the block in mini-batch order and overwrites d_X [mini_batch, in].
 * The activation must be built-in and sized to output_size (softmax: input_size * vector_size == output_size). */
typedef DefaultGradient DenseGradient;
typedef DefaultTrainingConfig DenseTrainingConfig;
Dense DenseCreateForTraining(DenseConfig config, DenseTrainingConfig training_config);
int  DenseApplyTrainingBatch(Dense filter, const float *input, float *output);  /* -1 on an inference-mode handle (dense.c:145) */
DenseGradient *DenseGradientCreate(DenseConfig config, DenseTrainingConfig training_config);
DenseGradient *DenseGradientCreateFromFilter(Dense dense);                      /* NULL on an inference-mode handle */
void DenseGradientDestroy(DenseGradient *gradient);
void DenseCalculateGradient(Dense filter, DenseGradient *gradient, float *d_out /*[mini_batch, out]*/);
/* additive device-pointer forms: d_input [mini_batch, in] must stay valid until the gradient call (it is the cached x);
 * d_grad_Wb = device [in * out + out] floats (d_W | d_b), ACCUMULATED onto like the host form; d_dX overwritten. */
int  DenseApplyTrainingBatchDevice(Dense filter, const float *d_input, float *d_output);
int  DenseCalculateGradientDevice(Dense filter, float *d_grad_Wb, float *d_dX, const float *d_dout);

/* ---- nntoolkitcore/layers/time_distributed_dense.h:19-45 --------------- */
typedef struct { DenseConfig dense; int ts; } TimeDistributedDenseConfig;
typedef struct TimeDistributedDenseStruct *TimeDistributedDense;

TimeDistributedDenseConfig TimeDistributedDenseConfigCreate(int ts, DenseConfig dense);
TimeDistributedDense TimeDistributedDenseCreateForInference(TimeDistributedDenseConfig config);
DenseWeights *TimeDistributedDenseGetWeights(TimeDistributedDense filter);
int  TimeDistributedDenseApplyInference(TimeDistributedDense filter, const float *input, float *output);
/* training (time_distributed_dense.h:24-43): a Dense trained on mini_batch * ts rows */
typedef DefaultTrainingConfig TimeDistributedDenseTrainingConfig;
TimeDistributedDense TimeDistributedDenseCreateForTraining(TimeDistributedDenseConfig config,
                                                           TimeDistributedDenseTrainingConfig training_config);
DenseGradient *TimeDistributedDenseGradientCreate(TimeDistributedDense filter);
int  TimeDistributedDenseApplyTrainingBatch(TimeDistributedDense filter, const float *input, float *output);
void TimeDistributedDenseCalculateGradient(TimeDistributedDense filter, DenseGradient *gradient, float *d_out);
int  TimeDistributedDenseApplyTrainingBatchDevice(TimeDistributedDense filter, const float *d_input, float *d_output);
int  TimeDistributedDenseCalculateGradientDevice(TimeDistributedDense filter, float *d_grad_Wb, float *d_dX, const float *d_dout);
void TimeDistributedDenseDestroy(TimeDistributedDense filter);

/* ---- nntoolkitcore/signal/dft.h:15-47 ---------------------------------- */
/* One complex DFT of any size on host split-complex buffers (un-normalised; forward = exp(-2 pi i k j / n)); the
 * reference runs kissfft (dft.c:34-47), here a direct transform with double accumulation on the device.  The
 * spectrogram kernels do not go through it. */
typedef struct DFTSetupStruct *DFTSetup;
typedef struct { int nfft; bool forward; bool complex; } DFTConfig;
typedef struct { float real; float imag; } ComplexFloat;
typedef struct { float *real_p; float *imag_p; } ComplexFloatSplit;
DFTConfig DFTConfigCreate(int nfft, bool forward, bool complex);
DFTSetup DFTSetupCreate(DFTConfig config);
void DFTPerform(DFTSetup setup, ComplexFloatSplit *input, ComplexFloatSplit *output);
void DFTSetupDestroy(DFTSetup setup);
void split_complex(const ComplexFloat *complex, ComplexFloatSplit *split, int size);
void join_complex_split(const ComplexFloatSplit *split, ComplexFloat *complex, int size);

/* ---- nntoolkitcore/signal/window.h:17-29 ------------------------------- */
typedef void (*window_fn)(float *, int);
void hamming_window(float *vector, int size);
void hann_window(float *vector, int size);
void ones(float *vector, int size);
void periodic_hamming_window(float *vector, int size);
void periodic_hann_window(float *vector, int size);
void blackman_window(float *vector, int size);

/* ---- nntoolkitcore/signal/spectrogram.h:20-44 -------------------------- */
typedef struct {
    int nfft;
    int window_size;
    int noverlap;
    int step;
    int input_size;
    int nfreq;
    int ntime_series;
    float fft_normalization_factor;
} SpectrogramConfig;
typedef struct SpectrogramStruct *Spectrogram;

SpectrogramConfig SpectrogramConfigCreate(int nfft, int window_size, int noverlap, int input_size,
                                          float fft_normalization_factor);
Spectrogram SpectrogramCreatePSD(SpectrogramConfig config, int fs);
Spectrogram SpectrogramCreateMagnitude(SpectrogramConfig config);
SpectrogramConfig SpectrogramGetConfig(Spectrogram filter);
void SpectrogramSetWindowFunc(Spectrogram filter, window_fn fn);
void SpectrogramSetScaleFactor(Spectrogram filter, float factor);
void SpectrogramApply(Spectrogram filter, const float *input, float *output);  /* host; errors via nntk_last_error() */
void SpectrogramDestroy(Spectrogram filter);

/* ---- nntoolkitcore/signal/mel_filterbank.h:14-41, log_mel_spectrogram.h:12-18 ----
 * (the first "next" row: 257 spectrogram bins -> n_mels features) */
typedef struct {
    int n_mels;
    int n_fft;
    int sample_rate;
    float lower_hz;
    float upper_hz;
} MelFilterBankConfig;
typedef struct MelFilterBankStruct *MelFilterBank;
typedef struct LogMelSpectrogramStruct *LogMelSpectrogram;

MelFilterBankConfig MelFilterBankConfigCreate(int n_mels, int n_fft, int sample_rate, float lower_hz, float upper_hz);
MelFilterBank MelFilterBankCreate(MelFilterBankConfig config);
void MelFilterBankApply(MelFilterBank filter_bank, const float *spectrogram, float *mel_spectrogram, int timesteps);
void MelFilterBankDestroy(MelFilterBank filter_bank);
LogMelSpectrogram LogMelSpectrogramCreate(Spectrogram spectrogram, MelFilterBankConfig mel_filter_bank_config);
void LogMelSpectrogramApply(LogMelSpectrogram filter, const float *input, float *output);   /* log(mel + 1.5849e-13) */
void LogMelSpectrogramDestroy(LogMelSpectrogram filter);

/* ======================================================================= */
/* PART 2 -- additive MI355X entry points                                    */
/* ======================================================================= */

/* ---- runtime ----------------------------------------------------------- */
int         nntk_hip_device_count(void);
int         nntk_hip_set_device(int device);          /* 0 ok, -1 error */
void        nntk_hip_set_stream(void *hip_stream);    /* all later launches OF THE CALLING THREAD use it; NULL = default stream */
void       *nntk_hip_get_stream(void);
int         nntk_hip_synchronize(void);               /* waits for the calling thread's current stream; -1 if a recurrent
                                                         launch faulted since the last check (see below) */
const char *nntk_last_error(void);                    /* "" when the last call succeeded */
const char *nntk_version(void);
const char *nntk_build_source_hash(void);        /* sha256/16 of the sources THIS binary was built from (nntoolkitcore_amd/_build.py) */
/* Threading (as the reference: distinct handles are independent, a handle is not re-entrant -- conv_1d.c:41,
 * gru.c:86, lstm.c:97): the current stream and the error string are per host thread; two threads may drive two
 * handles on two streams at the same time.  One handle is used by one thread and on one stream at a time
 * (nntk_hip_synchronize() before moving it to another stream).
 *
 * Tuning / diagnostics knobs by name; environment variables NNTK_<NAME> give the initial values (read once).
 *   "gemm_split_bf16" contraction of Conv1d / Dense / TimeDistributedDense / mel: auto = split-bf16 x 3 (every f32 operand as
 *                    three bf16 terms, six bf16 MFMA products accumulated in f32: f32 accuracy, not the k-ordered f32 chain),
 *                    0 = the exact-f32 chain everywhere, 1 = split also for the recurrent input projection.  EDGE VALUES under
 *                    auto: a non-finite input, or one above 3.39e38, gives NaN (where the chain gives +-inf) in exactly the outputs
 *                    whose window holds it; inputs below 1.18e-38 count as zero; a WEIGHT block holding such a value is detected
 *                    at upload and runs on the exact kernel.  INTEGRATION.md section 6 has the table, tests/test_gpu_edges.py pins it.
 *   "rec_rr"         GRU / LSTM batches: 0 = never the register-resident split-bf16 kernels (recurrent_rr.hip lstm_rr_kernel /
 *                    gru_rr_kernel; x W fused into the step, f32-accuracy contraction, not the exact-f32 chain), 1 = also below 32
 *                    sequences, auto = from 32.  Shapes: H 64..512 (multiple of 16), in <= 128, or in <= 256 when H <= 256; default
 *                    gate activations.  GRUStack2Apply* runs two such launches when both layers qualify ("rec_fused2" = 1: the fused
 *                    exact-f32 two-layer kernel instead)
 *   "train_bptt"     0 = GRU / LSTM gradients walk time with two launches per step instead of the persistent BPTT kernel
 *   "rec_persistent" 0 = per-timestep recurrent kernels only     "rec_pingpong" 0/1 = ping-pong halves off/on
 *   "rec_spin_us"    budget of the persistent kernel's spins     "gemm_tm_batch" 0/1
 *   "weights_check"  host-pointer calls look for in-place edits of the weight block before they launch:
 *                    1 (default) = the whole block is compared with the uploaded copy on every call, except the
 *                    single-sequence streaming GRU/LSTM/RNNApplyInference (T <= 32), which checks 257 probes of 64 B;
 *                    2 = the whole block everywhere, 0 = never (use <Layer>SyncWeights)
 * value "auto" restores the default.  0 ok, -1 unknown option. */
int         nntk_hip_set_option(const char *name, const char *value);
int         nntk_hip_get_option(const char *name, int *value);
/* The persistent GRU/LSTM kernel needs all its workgroups resident; if another process's kernel holds the CUs its
 * bounded spins give up and raise a sticky fault word.  Host-pointer recurrent calls notice it themselves and repeat
 * the call on the per-timestep kernels (same bits); device-pointer callers see it as -1 from nntk_hip_synchronize(),
 * or poll it without blocking here (0 healthy, 1 a completed recurrent launch of this thread has faulted).  After a
 * fault the process keeps to the per-timestep kernels. */
int         nntk_hip_device_status(void);
/* Name of the recurrent kernel the calling thread launched last ("" before the first): "lstm_rr_kernel<8,2>" (register-
 * resident split-bf16 LSTM with the fused input projection), "rec_persistent_kernel<4,LSTM>", "gru2_persistent_kernel<8>",
 * "rec_step_kernel<3,GRU>", ...  Diagnostics: which of the paths described under "rec_rr" / "rec_persistent" a call took. */
const char *nntk_hip_last_recurrent_kernel(void);
/* The same for the Conv1d / Dense / TimeDistributedDense GEMM kernels: "conv1d_mfma_bf16x3_kernel", "conv1d_mfma_bf16x3_kernel<frag3>"
 * (the frag3 epilogue of Conv1dBatchNormActivationApplyDeviceFrag3), "conv1d_flatk_bf16x3_kernel", "conv1d_mfma_kernel" (exact f32),
 * "conv1d_valu_kernel". */
const char *nntk_hip_last_conv_kernel(void);
/* Optional HIP-event spans around the recurrent kernel launches (name "rec_step"): enable,
 * run, then read the summed milliseconds, the number of kernel launches and the timesteps they
 * covered (a persistent launch covers all T of a sequence).  Reading clears the spans. */
void        nntk_hip_profile_enable(int on);
int         nntk_hip_profile_get(const char *name, double *total_ms, long *launches, long *timesteps);

/* raw device memory helpers for C callers that chain layers on the GPU */
float *nntk_device_alloc(size_t n_floats);
void   nntk_device_free(float *ptr);
int    nntk_device_upload(float *dst_device, const float *src_host, size_t n_floats);
int    nntk_device_download(float *dst_host, const float *src_device, size_t n_floats);

/* Re-upload the host weight block after the caller edited it.  The host-pointer
 * Apply* functions of Part 1 detect edits themselves (they compare the block with
 * the last uploaded copy); the *Device functions below do not. */
int Conv1dSyncWeights(Conv1d filter);
int BatchNormSyncWeights(BatchNorm filter);
int GRUSyncWeights(GRU filter);
int LSTMSyncWeights(LSTM filter);
int RNNSyncWeights(RNN filter);
int DenseSyncWeights(Dense filter);
int TimeDistributedDenseSyncWeights(TimeDistributedDense filter);

/* ---- multi-GPU: one process per GPU, utterances sharded, NO collective on the data path --------------------
 * (every utterance is independent: BatchNorm uses stored statistics, batch_norm.c:178-181; every sequence owns
 * its recurrent state).  The only communication is one RCCL broadcast of each layer's weight block from the root
 * at start-up, over xGMI.  RCCL is dlopen()ed at the first nntk_dist_* call; a single-GPU caller never loads it.
 *   rank 0:      nntk_dist_get_unique_id(id)  -> carry the 128 bytes to the other ranks (file, env, socket ...)
 *   every rank:  nntk_hip_set_device(local); nntk_dist_init(id, rank, world);
 *                <Layer>BroadcastWeights(handle, 0) for each layer;  nntk_dist_shard_range(B, world, rank, &lo, &hi);
 *                run utterances [lo, hi) with the *ApplyDevice / *ApplyInferenceBatch calls;  nntk_dist_finalize().
 * All return 0 / -1 (nntk_last_error()).  Without a communicator the broadcasts are no-ops. */
#define NNTK_DIST_ID_BYTES 128
int  nntk_dist_get_unique_id(unsigned char id[NNTK_DIST_ID_BYTES]);
int  nntk_dist_init(const unsigned char id[NNTK_DIST_ID_BYTES], int rank, int world_size);   /* collective */
int  nntk_dist_rank(void);
int  nntk_dist_world_size(void);
int  nntk_dist_broadcast(float *host_block, size_t n_floats, int root);      /* any host block, in place, blocking */
int  nntk_dist_barrier(void);
/* Data-parallel TRAINING: in-place SUM of a gradient block over the ranks (RCCL all-reduce over xGMI).  nntk_dist_allreduce: a host
 * block, staged, blocking.  nntk_dist_allreduce_device: a block in HBM (what <Layer>CalculateGradientDevice accumulates into),
 * asynchronous on the calling thread's stream.  Both are no-ops without a communicator (one rank). */
int  nntk_dist_allreduce(float *host_block, size_t n_floats);
int  nntk_dist_allreduce_device(float *d_block, size_t n_floats);
int  nntk_dist_finalize(void);
void nntk_dist_shard_range(int n_utterances, int world_size, int rank, int *lo, int *hi);
int Conv1dBroadcastWeights(Conv1d filter, int root);
int BatchNormBroadcastWeights(BatchNorm filter, int root);
int GRUBroadcastWeights(GRU filter, int root);
int LSTMBroadcastWeights(LSTM filter, int root);
int RNNBroadcastWeights(RNN filter, int root);
int DenseBroadcastWeights(Dense filter, int root);
int TimeDistributedDenseBroadcastWeights(TimeDistributedDense filter, int root);

/* ---- batched host-pointer forms: semantics of the reference's
 *      *ApplyTrainingBatch forward pass (conv_1d.c:167, gru.c:246, lstm.c:426,
 *      dense.c:144) without the training caches: input [batch, T, in],
 *      zero initial recurrent state per sequence. --------------------------- */
int Conv1dApplyInferenceBatch(Conv1d filter, const float *input, float *output, int batch);
int GRUApplyInferenceBatch(GRU filter, const float *input, float *output, int batch);
int LSTMApplyInferenceBatch(LSTM filter, const float *input, float *output, int batch);
int RNNApplyInferenceBatch(RNN filter, const float *input, float *output, int batch);   /* rnn.c:249-291 forward */
int TimeDistributedDenseApplyInferenceBatch(TimeDistributedDense filter, const float *input, float *output, int batch);
int SpectrogramApplyBatch(Spectrogram filter, const float *input, float *output, int batch);
int LogMelSpectrogramApplyBatch(LogMelSpectrogram filter, const float *input, float *output, int batch);

/* ---- device-pointer forms (all pointers are device addresses; asynchronous
 *      on the current stream; 0 ok, -1 error) ------------------------------ */
int SpectrogramApplyDevice(Spectrogram filter, const float *d_input /*[batch,input_size]*/,
                           float *d_output /*[batch,ntime_series,nfreq]*/, int batch);
int MelFilterBankApplyDevice(MelFilterBank filter_bank, const float *d_spectrogram /*[rows,nbins]*/,
                             float *d_mel /*[rows,n_mels]*/, int rows);
int LogMelSpectrogramApplyDevice(LogMelSpectrogram filter, const float *d_input /*[batch,input_size]*/,
                                 float *d_output /*[batch,ntime_series,n_mels]*/, int batch);
int Conv1dApplyDevice(Conv1d filter, const float *d_input /*[batch,T,Cin]*/,
                      float *d_output /*[batch,Tout,Cout]*/, int batch);
/* fused Conv1d -> BatchNorm(inference) -> activation in one kernel; bn and/or act may be NULL */
int Conv1dBatchNormActivationApplyDevice(Conv1d filter, BatchNorm bn, ActivationFunction act,
                                         const float *d_input, float *d_output, int batch);
int BatchNormApplyDevice(BatchNorm filter, const float *d_input, float *d_output, int rows);
int ActivationFunctionApplyDevice(ActivationFunction filter, const float *d_input, float *d_output, int size);
int GRUApplyDevice(GRU filter, const float *d_input /*[batch,T,in]*/, float *d_output, int batch);
int LSTMApplyDevice(LSTM filter, const float *d_input /*[batch,T,in]*/, float *d_output, int batch);
/* Two stacked GRU layers (layer 1 returns sequences and feeds layer 2) in ONE persistent launch: layer 2 runs one step
 * behind layer 1 inside the same kernel, so its input projection and the inter-layer [batch,T,H] tensor never reach HBM.
 * Results = GRUApplyDevice(l1) then GRUApplyDevice(l2) from zero state (gru.c:246-293 forward semantics), within the
 * layer tolerance; shapes or activations the fused kernel does not take run exactly those two calls. */
int GRUStack2ApplyDevice(GRU layer1, GRU layer2, const float *d_input /*[batch,T,in]*/, float *d_output, int batch);
int GRUStack2ApplyInferenceBatch(GRU layer1, GRU layer2, const float *input, float *output, int batch);
int RNNApplyDevice(RNN filter, const float *d_input /*[batch,T,in]*/, float *d_output, int batch);
int bd_reverse_input_batch_device(const float *d_input, float *d_output, RecurrentConfig config, int batch);
int bd_reverse_backward_batch_device(const float *d_input, float *d_output, RecurrentConfig config, int batch);
int bd_merge_concat_device(const float *d_forward, const float *d_backward, float *d_output, RecurrentConfig config, int batch);
int bd_merge_sum_device(const float *d_forward, const float *d_backward, float *d_output, RecurrentConfig config, int batch);
/* ---- frag3 tensors: activations already split for the split-bf16 x 3 contraction ---------------------------------------------
 * The default contraction of conv / dense / recurrent layers multiplies every f32 operand as three bf16 terms (x = hi + mid + lo,
 * exactly).  A FRAG3 tensor is a [batch][T][C] f32 tensor stored as those three images in MFMA fragment order:
 * [T][2 ceil(batch / 64)][ceil(C / 16)][3] blocks of 1 KB, block = 32 batch rows x 16 channels, lane 32 kh + n of a wavefront holding
 * channels 16 ks + 8 kh .. + 7 of batch row 32 ht + n as 8 consecutive bf16.  Padding CHANNELS (past C) are zeros; padding ROWS (past the
 * batch, up to the next multiple of 64) are unspecified -- nntk_frag3_pack_device writes zeros there, the recurrent kernels write the
 * state of rows that computed on zero inputs, the conv epilogue (Conv1dBatchNormActivationApplyDeviceFrag3) does not write them at all:
 * a consumer must not let them reach a real row (every consumer here masks).  6 bytes per value
 * instead of 4; in exchange a consumer's operand fetch is a run of coalesced 1 KB loads straight into MFMA registers (no LDS
 * staging, no split, no per-row requests).  The register-resident GRU / LSTM kernels produce their output in this form for free
 * (it is their inter-workgroup hand-off) and read their input from it; the dense GEMM reads it as its A operand.
 * The format is exact for every value the three bf16 terms can represent -- unpack(pack(x)) == x for finite |x| <= 3.39e38 that are not
 * denormal (a larger |x| or +-inf gives NaN, a denormal 0: INTEGRATION.md section 6) -- so every *Frag3 call equals its f32 counterpart BIT FOR BIT.
 * Scratch: a layer on the register-resident kernels keeps its T-deep hand-off (= its output in frag3 form, 6 bytes per batch * T * H value,
 * 1.5 x the f32 output) in the handle when the caller passes no d_output_frag3 -- also for plain <GRU|LSTM>ApplyDevice calls.
 *   <GRU|LSTM>ApplyDeviceFrag3: input as f32 (d_input) or frag3 (d_input_frag3), one of them NULL; output as f32 (d_output), frag3
 *   (d_output_frag3, needs return_sequences) or both, unused ones NULL.  Zero initial state per sequence.  Shapes the register-resident
 *   kernels do not take run the other kernels through f32 scratch -- the call is valid for every layer.
 *   LSTMTimeDistributedDenseApplyDevice = LSTMApplyDevice then TimeDistributedDenseApplyDevice without the f32 tensor in between
 *   (on the FRAG2H form below by default; option dense_f16x2 = 0: on frag3, bit for bit the two f32 calls). */
size_t nntk_frag3_floats(int batch, int T, int C);                       /* size of a frag3 tensor, in floats */
int nntk_frag3_pack_device(const float *d_x /*[batch,T,C]*/, float *d_frag3, int batch, int T, int C);
int nntk_frag3_unpack_device(const float *d_frag3, float *d_x /*[batch,T,C]*/, int batch, int T, int C);
/* Which kernel family the batch forms of this layer take, and -- when it is not the register-resident one -- why (shape, activations,
 * weights, options): a human-readable line, valid until the calling thread's next call of the same function.  Depends on the layer
 * only, never on the batch size. */
const char *GRUKernelPlan(GRU filter);
const char *LSTMKernelPlan(LSTM filter);
int GRUApplyDeviceFrag3(GRU filter, const float *d_input, const float *d_input_frag3, float *d_output, float *d_output_frag3, int batch);
int LSTMApplyDeviceFrag3(LSTM filter, const float *d_input, const float *d_input_frag3, float *d_output, float *d_output_frag3, int batch);
int TimeDistributedDenseApplyDeviceFrag3(TimeDistributedDense filter, const float *d_input_frag3 /*[batch,ts,in]*/, float *d_output, int batch);
/* Conv1d -> BatchNorm -> activation (bn / act may be NULL) with the output as a frag3 tensor [batch][Tout][Cout]
 * (nntk_frag3_floats(batch, Tout, Cout) floats), written by the conv kernel's epilogue: the layer in front of a recurrent layer
 * hands over the operand form directly (reference seam: layers/conv_1d.c:122-147 -> layers/lstm.c:201).  Equals
 * Conv1dBatchNormActivationApplyDevice -> nntk_frag3_pack_device bit for bit; valid for every layer (shapes the epilogue does not take
 * -- stride != 1, kernel_size > 9 -- run those two calls through scratch in the handle). */
int Conv1dBatchNormActivationApplyDeviceFrag3(Conv1d filter, BatchNorm bn, ActivationFunction act,
                                              const float *d_input /*[batch,T,Cin]*/, float *d_output_frag3, int batch);
/* ---- FRAG2H tensors: a bounded activation tensor as two f16 images, for a three-product contraction ----------------------------
 * The same block structure as frag3 -- [T][2 ceil(batch / 64)][ceil(C / 16)][2] blocks of 1 KB, same lane order -- holding hi = f16(x 2^15)
 * and lo = f16(x 2^15 - hi): |x - (hi + lo) 2^-15| <= 2^-23 |x| (at worst one f32 ulp, 0.3 ulp rms; absolute 2^-40 below |x| ~ 2^-17), 4 bytes
 * per value.  f16 ends at 65 504, so the form is DEFINED FOR |x| < 2 ONLY (a larger value becomes inf): it is what a GRU / LSTM layer with
 * the standard activations produces (|h| < 1).  The dense GEMM on it (weights as two f16 images of W 2^q, q chosen at upload so that
 * max |W| 2^q <= 32 768; weights must be finite) sums three products hi.hi + hi.lo + lo.hi per k step instead of frag3's six at the same MFMA
 * rate, f32 accumulation, one exact multiplication by 2^-(15 + q) in the epilogue.  Measured against an f64 dot product of the same f32
 * operands the error is below the frag3 / f32-input GEMM's and below that of the reference's own f32 accumulation order
 * (profiles/r05_gemm_f16x2_micro.log; tests/test_gpu_frag2h.py) -- but the results are NOT bit-identical to those routes.
 *   LSTMApplyDeviceFrag2h: the layer's sequence output in this form; -1 for non-standard activations / return_sequences == false.
 *     For 256 < H <= 512 the call runs the HF instantiation of the register-resident kernel: the RECURRENCE ITSELF multiplies h -- the same
 *     bounded operand -- as two f16 images against two f16 images of U 2^q (three products per k step; the input projection x.W keeps its
 *     three bf16 images and six products), and the kernel's hand-off buffer IS the frag2h tensor.  Same tolerance against the reference as the
 *     six-product kernel (tests/test_gpu_frag2h.py: T = 996 against the oracle and float64), not the same bits; option rec_hf = 0 keeps the
 *     six-product recurrence (its output wave then writes the form).  Which kernel runs depends on the layer only, never on the call.
 *   TimeDistributedDenseApplyDeviceFrag2h: valid for every layer (shapes / weights the f16 kernel does not take unpack to f32).
 *   LSTMTimeDistributedDenseApplyDevice takes this route by default when it applies (option dense_f16x2 = 0: the frag3 route). */
size_t nntk_frag2h_floats(int batch, int T, int C);
int nntk_frag2h_pack_device(const float *d_x /*[batch,T,C], |x| < 2*/, float *d_frag2h, int batch, int T, int C);
int nntk_frag2h_unpack_device(const float *d_frag2h, float *d_x /*[batch,T,C]*/, int batch, int T, int C);
int LSTMApplyDeviceFrag2h(LSTM filter, const float *d_input, const float *d_input_frag3, float *d_output_frag2h, int batch);
int TimeDistributedDenseApplyDeviceFrag2h(TimeDistributedDense filter, const float *d_input_frag2h /*[batch,ts,in]*/, float *d_output, int batch);
int LSTMTimeDistributedDenseApplyDevice(LSTM lstm, TimeDistributedDense tdd, const float *d_input /*[batch,T,in]*/,
                                        float *d_output /*[batch,T,out]*/, int batch);
int DenseApplyDevice(Dense filter, const float *d_input /*[rows,in]*/, float *d_output /*[rows,out]*/, int rows);
int TimeDistributedDenseApplyDevice(TimeDistributedDense filter, const float *d_input, float *d_output, int batch);

/* recurrent state of the stateful single-sequence API (gru.c:201, lstm.c:264-265) */
int GRUResetState(GRU filter);
int LSTMResetState(LSTM filter);
int RNNResetState(RNN filter);
int GRUGetState(GRU filter, float *h_host);
int RNNGetState(RNN filter, float *h_host);
int LSTMGetState(LSTM filter, float *h_host, float *c_host);

/* ---- nntoolkitcore/train/loss.h:17-23, train/optimizers.h:12-16 ------- */
/* Host pointers, the reference's names and operation order (per-sample sums in order, batch sum in order).  One
 * documented difference: categorical_crossentropy_derivative writes EVERY row; the reference's loop (loss.c:47-52)
 * forgets the row offset and only ever writes row 0 (identical here). */
float mean_squared_error(float *y, float *y_pred, int size, int batch);
void  mean_squared_error_derivative(float *y, float *y_pred, float *d_y_pred, int size, int batch);
float categorical_crossentropy(float *y, float *y_pred, int c, int batch);
void  categorical_crossentropy_derivative(float *y, float *y_pred, float *d_y_pred, int c, int batch);
typedef struct { float learning_rate; } SGD;
int   sgd_optimize(SGD optimizer, float *gradient, float *weights, int size);    /* w -= g * lr, two roundings */
/* device-pointer forms */
int nntk_mean_squared_error_device(const float *d_y, const float *d_pred, int size, int batch, float *loss);
int nntk_categorical_crossentropy_device(const float *d_y, const float *d_pred, int c, int batch, float *loss);
int nntk_mean_squared_error_derivative_device(const float *d_y, const float *d_pred, float *d_out, int size, int batch);
int nntk_categorical_crossentropy_derivative_device(const float *d_y, const float *d_pred, float *d_out, int c, int batch);
int nntk_sgd_optimize_device(SGD optimizer, const float *d_gradient, float *d_weights, long size);

#ifdef __cplusplus
}
#endif
#endif /* NNTOOLKITCORE_HIP_H */
