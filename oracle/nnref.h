/*
 * oracle/nnref.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic of the reference's time-series
 * inference path (Spectrogram -> Conv1d -> BatchNorm/Activation -> GRU/LSTM ->
 * TimeDistributedDense).  Every function cites the reference file:line whose
 * operation ORDER it follows (all fp32, libm expf/tanhf/sqrtf, true division,
 * left-to-right accumulation), so results agree with a scalar x86 build of the
 * reference.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library.  The product (nntoolkitcore_amd/) never links or calls it.
 *
 * PARITY PINNING STATUS
 *   - window functions: pinned against the reference's own signal/window.c
 *     compiled in place (oracle/_ref, see oracle/Makefile target `ref`).
 *   - config geometry + struct ABI: pinned against the reference headers
 *     (oracle/ref_probe.c compiled against /root/reference headers).
 *   - compute paths (conv/bn/act/gru/lstm/dense/spectrogram): PARITY UNPINNED
 *     by the reference.  The reference ships no tests, fixtures or golden
 *     vectors, and its Linux backend cannot be built here: core/default_ops.cc
 *     needs Eigen and signal/dft.c needs kissfft, both empty, un-vendored
 *     submodules (.gitmodules:1-6) that are absent from this image.  The
 *     restatement is instead cross-checked against independent implementations
 *     (torch.nn.GRU/LSTM/conv1d, scipy.signal.spectrogram, numpy) in
 *     tests/test_oracle.py.
 *
 * Third-party arithmetic restated from published algorithms:
 *   - Eigen (gitlab.com/libeigen/eigen, commit unpinned in the reference):
 *     only op_mat_mul / op_mat_transp use it (core/default_ops.cc:729-747).
 *     Restated as the reference's OWN scalar forms op_mat_mul_c /
 *     op_mat_transp_c (core/default_ops.cc:707-727).
 *   - kissfft (github.com/mborgerding/kissfft, commit unpinned): restated from
 *     its published kiss_fft.c (v1.3.x/131 mixed-radix decimation-in-time:
 *     kf_factor, kf_work, kf_bfly2/3/4/5/generic, twiddles = cos/sin evaluated
 *     in double and rounded to float).
 */
#ifndef NNREF_ORACLE_H
#define NNREF_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* activation kinds (shared numbering with the product's nntk_hip.h) */
enum {
    REF_ACT_IDENTITY = 0,
    REF_ACT_SIGMOID  = 1,
    REF_ACT_TANH     = 2,
    REF_ACT_RELU     = 3,
    REF_ACT_SOFTMAX  = 4
};

/* window kinds */
enum {
    REF_WIN_ONES = 0,
    REF_WIN_HANN = 1,
    REF_WIN_HAMMING = 2,
    REF_WIN_PERIODIC_HANN = 3,
    REF_WIN_PERIODIC_HAMMING = 4,
    REF_WIN_BLACKMAN = 5
};

/* ---- ops (core/default_ops.cc scalar paths) ---- */
float ref_op_vec_dot(const float *a, const float *b, int size);
void  ref_op_mat_mul(const float *a, const float *b, float *c, int M, int N, int K);
void  ref_op_mat_transp(const float *a, float *b, int M, int N);

/* ---- signal ---- */
void ref_window(int kind, float *v, int size);
/* complex forward/inverse DFT, interleaved (re,im) in and out, kissfft order */
int  ref_kiss_fft(int nfft, int inverse, const float *in_interleaved, float *out_interleaved);
/* derived geometry (signal/spectrogram.c:59-70) */
void ref_spectrogram_geometry(int nfft, int window_size, int noverlap, int input_size,
                              int *step, int *nfreq, int *ntime_series);
float ref_spectrogram_scale_magnitude(const float *window, int window_size);
float ref_spectrogram_scale_psd(const float *window, int window_size, int fs);
/* mode 0 = magnitude, 1 = psd.  out is [ntime_series, nfreq] */
int  ref_spectrogram(const float *input, const float *window, float *out,
                     int nfft, int window_size, int noverlap, int input_size,
                     float fft_norm, int mode, float scale_factor);

/* ---- layers ---- */
int  ref_conv1d_output_size(int input_size, int kernel_size, int stride);
/* one sequence: in [T,Cin] -> out [Tout,Cout]; W [Cout][Cin][k], b [Cout] */
void ref_conv1d(const float *in, const float *W, const float *b, float *out,
                int T, int Cin, int Cout, int k, int stride);
void ref_conv1d_batch(const float *in, const float *W, const float *b, float *out,
                      int B, int T, int Cin, int Cout, int k, int stride);

/* training, first slice: Conv1dCalculateGradient (conv_1d.c:185-245): dW, db are added to, dX overwritten */
void ref_conv1d_gradient(const float *in, const float *W, const float *dout, float *dW, float *db, float *dX,
                         int B, int T, int Cin, int Cout, int k, int stride);

/* training, second slice: activation gradients, Dense gradient, losses, SGD (see nnref_layers.c for file:line) */
void ref_activation_gradient(int kind, int vector_size, const float *z, const float *a, const float *dout, float *out, int size);
void ref_dense_gradient(const float *x, const float *W, const float *z, const float *a, const float *dout,
                        int act_kind, int vector_size, int act_size, float *gW, float *gb, float *dX, int B, int in, int out);
float ref_mean_squared_error(const float *y, const float *p, int size, int batch);
void  ref_mean_squared_error_derivative(const float *y, const float *p, float *d, int size, int batch);
float ref_categorical_crossentropy(const float *y, const float *p, int c, int batch);
void  ref_categorical_crossentropy_derivative(const float *y, const float *p, float *d, int c, int batch);
void  ref_sgd_optimize(float lr, const float *g, float *w, int size);
void ref_gru_training_forward(const float *x, const float *W, const float *U, const float *b_i, const float *b_h,
                              float *h, float *Zg, float *hU, int B, int T, int in, int H, int act_z, int act_h, int act_r);
void ref_gru_gradient(const float *x, const float *W, const float *U, const float *h, const float *Zg, const float *hU,
                      const float *dout, int return_sequences, float *gW, float *gU, float *gbi, float *gbh, float *dX,
                      int B, int T, int in, int H, int act_z, int act_h, int act_r);
void ref_lstm_training_forward(const float *x, const float *W, const float *U, const float *b_i, const float *b_h,
                               float *h, float *c, float *zifgo, int B, int T, int in, int H, int v2,
                               int act_i, int act_f, int act_g, int act_o, int act_out);
void ref_lstm_gradient(const float *x, const float *W, const float *U, const float *h, const float *c, const float *zifgo,
                       const float *dout, int return_sequences, float *gW, float *gU, float *gbi, float *gbh, float *dX,
                       int B, int T, int in, int H, int act_i, int act_f, int act_g, int act_o, int act_out);
void ref_rnn_training_forward(const float *x, const float *W, const float *U, const float *b_i, const float *b_h,
                              float *h, float *gate, int B, int T, int in, int H, int v2, int act);
void ref_rnn_gradient(const float *x, const float *W, const float *U, const float *h, const float *gate, const float *dout,
                      int return_sequences, float *gW, float *gU, float *gbi, float *gbh, float *dX,
                      int B, int T, int in, int H, int act);
void ref_batch_norm_training_forward(const float *x, const float *gamma, const float *beta, float eps, float momentum,
                                     float *out, float *mean, float *var, float *moving_mean, float *moving_var, int N, int F);
void ref_batch_norm_gradient(const float *x, const float *dout, const float *gamma, const float *mean, const float *var,
                             float eps, float *d_beta, float *d_gamma, float *d_x, int N, int F);

void ref_batch_norm(const float *in, const float *gamma, const float *beta,
                    const float *mean, const float *variance, float *out,
                    float epsilon, int count, int C);

/* size = number of elements (number of vectors for softmax), like the reference's input_size */
void ref_activation(int kind, float relu_a, int softmax_vector_size,
                    const float *in, float *out, int size);

/* GRU: W [in,3H], U [H,3H], b_i[3H], b_h[3H]; gate order z,r,h.
 * h_state [H] is read as the initial state and holds the final state on return.
 * out is [T,H] if return_sequences else [H]. */
void ref_gru_sequence(const float *x, const float *W, const float *U,
                      const float *b_i, const float *b_h, float *h_state, float *out,
                      int T, int in, int H, int return_sequences,
                      int act_z, int act_h, int act_r);
/* ReLU output scale per gate activation (index = position in the act_* lists; default 1) for the next calls of this thread */
void ref_set_gate_relu_scales(const float *a, int n);
void ref_gru_batch(const float *x, const float *W, const float *U,
                   const float *b_i, const float *b_h, float *out,
                   int B, int T, int in, int H, int return_sequences,
                   int act_z, int act_h, int act_r);

/* RNN (one gate): W [in,H], U [H,H], b_i[H], b_h[H] (b_h used only if v2) */
void ref_rnn_sequence(const float *x, const float *W, const float *U,
                      const float *b_i, const float *b_h, float *h_state, float *out,
                      int T, int in, int H, int return_sequences, int v2, int act);
void ref_rnn_batch(const float *x, const float *W, const float *U,
                   const float *b_i, const float *b_h, float *out,
                   int B, int T, int in, int H, int return_sequences, int v2, int act);
/* bidirectional forward helpers */
void ref_bd_reverse_batch(const float *in, float *out, int B, int T, int F);
void ref_bd_merge_concat(const float *fwd, const float *bwd, float *out, int B, int rows, int C);
void ref_bd_merge_sum(const float *fwd, const float *bwd, float *out, int B, int rows, int C);

/* LSTM: W [in,4H], U [H,4H], b_i[4H], b_h[4H] (b_h used only if v2); gate order i,f,g,o */
void ref_lstm_sequence(const float *x, const float *W, const float *U,
                       const float *b_i, const float *b_h, float *h_state, float *c_state,
                       float *out, int T, int in, int H, int return_sequences, int v2,
                       int act_i, int act_f, int act_g, int act_o, int act_out);
void ref_lstm_batch(const float *x, const float *W, const float *U,
                    const float *b_i, const float *b_h, float *out,
                    int B, int T, int in, int H, int return_sequences, int v2,
                    int act_i, int act_f, int act_g, int act_o, int act_out);

/* Dense: W [in,out] row-major, b [out]; act_kind < 0 means "no activation handle" */
void ref_dense(const float *x, const float *W, const float *b, float *out,
               int in, int out_size, int act_kind, float relu_a, int softmax_vector_size, int act_size);
/* TimeDistributedDense over ts rows sharing weights */
void ref_time_distributed_dense(const float *x, const float *W, const float *b, float *out,
                                int ts, int in, int out_size,
                                int act_kind, float relu_a, int softmax_vector_size, int act_size);

/* ---- next-row components ---- */
/* weights [nbins, n_mels] as built by signal/mel_filterbank.c:43-102 */
void ref_mel_filterbank_weights(int n_mels, int n_fft, int sample_rate, float lower_hz, float upper_hz,
                                float *weights);
void ref_log_mel(const float *spec, const float *weights, float *out, int ts, int nbins, int n_mels);

#ifdef __cplusplus
}
#endif
#endif
