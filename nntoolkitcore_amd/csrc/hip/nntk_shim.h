/*
 * nntk_shim.h -- the thin C-ABI between the C host layer (csrc/host) and the
 * hand-written HIP kernels (csrc/hip).  Plain pointers and ints only; no
 * HIP types.  All `d_` pointers are device addresses.  Every launcher enqueues on
 * the current stream (nntk_shim_stream) and returns 0, or -1 after recording a
 * message retrievable with nntk_shim_error().
 */
#ifndef NNTK_SHIM_H
#define NNTK_SHIM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* activation kinds understood by the device code */
enum {
    NNTK_ACT_NONE     = -1,   /* no activation handle (dense.c:127-133 copy branch) */
    NNTK_ACT_IDENTITY = 0,
    NNTK_ACT_SIGMOID  = 1,
    NNTK_ACT_TANH     = 2,
    NNTK_ACT_RELU     = 3,
    NNTK_ACT_SOFTMAX  = 4,
    NNTK_ACT_LOG_EPS  = 5,    /* log(x + a): log-mel epilogue (log_mel_spectrogram.c:34-35); a rides in relu_a */
    NNTK_ACT_CUSTOM   = 100   /* host callback: not runnable on the device */
};

/* ---- runtime --------------------------------------------------------------- */
int         nntk_shim_device_count(void);
int         nntk_shim_set_device(int device);
int         nntk_shim_get_device(void);            /* the calling thread's current device, -1 without one */
void        nntk_shim_set_stream(void *stream);
void       *nntk_shim_get_stream(void);
int         nntk_shim_synchronize(void);
const char *nntk_shim_error(void);
void        nntk_shim_set_error(const char *msg);
void        nntk_shim_clear_error(void);

/* tuning / diagnostics knobs by name ("rec_persistent", "rec_pingpong", "gemm_tm_batch", ...; value "auto" restores
 * the default).  Environment variables NNTK_<NAME> give the initial values, read once. */
int nntk_shim_set_option(const char *name, const char *value);
int nntk_shim_get_option(const char *name, int *value);
/* sticky fault word of the persistent recurrent kernel (see runtime.hip) */
int nntk_shim_take_fault(void);            /* after a stream sync: 1 = a launch faulted (cleared, per-step kernels from now on) */
int nntk_shim_persistent_disabled(void);
const char *nntk_shim_last_conv_kernel(void);  /* ... of the Conv1d / dense GEMM kernel ("conv1d_mfma_bf16x3_kernel<frag3>", ...) */
const char *nntk_shim_last_rec_kernel(void);   /* name of the recurrent kernel this thread launched last ("" before the first) */
int nntk_shim_device_status(void);         /* non-blocking: 1 = a completed recurrent launch has faulted */

/* HIP-event spans around the recurrent step launches (off by default) */
void nntk_shim_profile_enable(int on);
int  nntk_shim_profile_get(const char *name, double *total_ms, long *launches, long *timesteps);

void *nntk_shim_malloc(size_t bytes);
void  nntk_shim_free(void *d_ptr);
void *nntk_shim_host_alloc(size_t bytes);              /* pinned, zero-filled */
void  nntk_shim_host_free(void *ptr);
int   nntk_shim_upload(void *d_dst, const void *h_src, size_t bytes);     /* blocking */
int   nntk_shim_upload_async(void *d_dst, const void *h_src, size_t bytes);        /* async on stream (pinned source) */
int   nntk_shim_download(void *h_dst, const void *d_src, size_t bytes);   /* blocking; -1 if a recurrent launch faulted */
int   nntk_shim_download_nocheck(void *h_dst, const void *d_src, size_t bytes);    /* blocking; leaves a fault for nntk_shim_take_fault */
int   nntk_shim_download_rows(void *h_dst, const void *d_src, size_t spitch, size_t width, size_t height);   /* strided rows, packed at the destination */
int   nntk_shim_copy_d2d(void *d_dst, const void *d_src, size_t bytes);   /* async on stream */
int   nntk_shim_copy_rows_d2d(void *d_dst, const void *d_src, size_t spitch, size_t width, size_t height);   /* strided rows -> packed, async */
int   nntk_shim_memset(void *d_ptr, int value, size_t bytes);             /* async on stream */

/* ---- K2/K3: implicit-GEMM conv1d / dense on f32 MFMA with fused epilogue ----
 * out[b, x, o] = act( bn( bias[o] + sum_{kk,i} in[b, x*stride+kk, i] * Wp[(kk*Cin_p + i), o] ) )
 *   d_in   [B, T, Cin]           channels-last
 *   d_wp   packed weights [Cout_p][k*Cin_p], K-contiguous, zero padded (see nntk_shim_conv_pack_sizes)
 *   d_bias [Cout]
 *   d_bn   NULL or 6*Cout floats: gamma | beta | mean | variance (batch_norm.c:79-84 order) | sd | 1/sd, the last
 *          two written by nntk_shim_bn_derive(d_bn, eps, Cout) after an upload of the first four
 *   out_mode 0: out[b, x, :] at row b*Tout + x;  1: time-major row x*B + b (for recurrent input projections)
 * Dense / TimeDistributedDense / input projection = the k=1, stride=1 case with T = rows.
 */
void nntk_shim_conv_pack_sizes(int Cin, int Cout, int k, int *Cin_p, int *Cout_p);
/* d_wp of nntk_shim_conv1d is the packed f32 matrix [Cout_p][k * Cin_p] FOLLOWED BY its three bf16 split images
 * (hi | mid | lo, each Cout_p * k * Cin_p bf16 in MFMA fragment order) written by
 * nntk_shim_split_bf16x3(d_wp, d_wp + n, Cout_p, k * Cin_p); the images are read only by the split-bf16 kernel */
int  nntk_shim_bn_derive(float *d_block, float eps, int C);
int  nntk_shim_split_bf16x3(const float *d_src, void *d_dst, int rows, int ktot);
/* on = 1: this packed weight block holds a non-finite, > 3.39e38 or denormal value, which the bf16 split cannot represent
 * exactly -- the "auto" contraction takes the exact-f32 kernel for it.  on = 0 (re-upload of clean weights, free) clears. */
void nntk_shim_weights_exact_only(const void *d_wp, int on);
int  nntk_shim_conv1d(const float *d_in, const float *d_wp, const float *d_bias, const float *d_bn,
                      float bn_eps, int act_kind, float relu_a, float *d_out,
                      int B, int T, int Cin, int Cout, int k, int stride, int Tout, int out_mode);

/* ---- training, first slice: Conv1dCalculateGradient (conv_1d.c:185-245); untuned VALU kernels, deterministic ---- */
/* Conv1d (+ BatchNorm + activation) with a frag3 tensor as output (conv1d.hip conv_epilogue_frag3); 1 = this form does not take the call */
int  nntk_shim_conv1d_frag3(const float *d_in, const float *d_wp, const float *d_bias, const float *d_bn,
                            float bn_eps, int act_kind, float relu_a, float *d_out_f3,
                            int B, int T, int Cin, int Cout, int k, int stride, int Tout);
/* flat-K split convolution (conv1d_flatk.hip; stride 1, Cin % 8 == 0, Cin % 16 != 0): d_wpf = [Cout_p][Kf_p] with K = tap * Cin + channel,
 * Kf_p = k * Cin rounded up to 16, followed by its split images (nntk_upload_packed_weights).  Returns 1 when not taken. */
int nntk_shim_conv1d_flatk(const float *d_in, const float *d_wpf, const float *d_bias, const float *d_bn,
                           float bn_eps, int act_kind, float relu_a, float *d_out,
                           int B, int T, int Cin, int Cout, int k, int Tout);
size_t nntk_shim_conv1d_grad_scratch_floats(int Cin, int Cout, int k);
int nntk_shim_conv1d_grad(const float *d_in, const float *d_W, const float *d_dout, float *d_dW, float *d_db,
                          float *d_dX, float *d_scratch, int B, int T, int Cin, int Cout, int k, int stride, int Tout);

/* ---- standalone BatchNorm (batch_norm.c:140-163 op order) and activations -- */
int nntk_shim_batch_norm(const float *d_in, const float *d_bn /*gamma|beta|mean|var*/, float eps,
                         float *d_out, long rows, int C);
int nntk_shim_activation(int kind, float relu_a, int softmax_vector_size,
                         const float *d_in, float *d_out, long n_elems);

/* ---- bidirectional helpers (layers/bidirectional.c): rows of [B, T, F] in reverse time order; row-wise
 *      concatenation [rows, C] | [rows, C] -> [rows, 2C]; elementwise sum */
int nntk_shim_reverse_time(const float *d_in, float *d_out, long B, int T, int F);
/* ---- training, second slice (train.hip): reference operation order throughout ---- */
int nntk_shim_activation_grad(int kind, int vector_size, int vectors_per_call, const float *d_z, const float *d_a,
                              const float *d_dout, float *d_out, long n);
int nntk_shim_dense_grad(const float *d_x, const float *d_W /*[in,out] caller layout*/, const float *d_dz, float *d_gW,
                         float *d_gb, float *d_dX, int B, int in, int out);
int nntk_shim_loss_rows(int kind /*0 mse, 1 categorical ce*/, const float *d_y, const float *d_pred, float *d_per_row, int size, int batch);
int nntk_shim_loss_grad(int kind, const float *d_y, const float *d_pred, float *d_out, int size, int batch);
int nntk_shim_sgd(float lr, const float *d_grad, float *d_w, long n);
/* BatchNorm training (batch_norm.c:191-386): x, d_out [N, F]; d_block = gamma | beta | ...; d_stats [8][F] = mean | variance |
 * var_eps | sqrt_var | d_beta | d_gamma | d_var | d_mu; d_partial [slices][3][F] with slices from nntk_shim_bn_train_slices */
/* GRU training (gru.c:246-512): caller-layout weights W [in][3H], U [H][3H]; caches h [B][T][H], Zg [B][T][6H], hU [B][T][H];
 * acts / scales in the order z, h, r */
int nntk_shim_gru_train_forward(const float *d_x, const float *d_W, const float *d_U, const float *d_bi, const float *d_bh,
                                float *d_h, float *d_Zg, float *d_hU, int B, int T, int in, int H, const int *acts, const float *scales);
int nntk_shim_gru_train_backward(const float *d_dout, const float *d_UT /*U transposed [3H][H]*/, const float *d_h, const float *d_Zg, const float *d_hU,
                                 float *d_dxW, float *d_dhU, float *d_work /*5*B*H*/, int B, int T, int H, int return_sequences, const int *acts);
/* LSTM training (lstm.c:185-239, :294-556): W [in][4H], U [H][4H]; caches h, c [B][T][H], zifgo [B][T][8H]; acts i,f,g,o,out */
int nntk_shim_lstm_train_forward(const float *d_x, const float *d_W, const float *d_U, const float *d_bi, const float *d_bh,
                                 float *d_h, float *d_c, float *d_zifgo, int B, int T, int in, int H, int v2, const int *acts, const float *scales);
int nntk_shim_lstm_train_backward(const float *d_dout, const float *d_UT /*U transposed [4H][H]*/, const float *d_c, const float *d_zifgo, float *d_dG,
                                  float *d_work /*6*B*H*/, int B, int T, int H, int return_sequences, const int *acts, const float *scales);
/* RNN training (rnn.c:144-221, :249-351): W [in][H], U [H][H]; caches h, gate [B][T][H] */
int nntk_shim_rnn_train_forward(const float *d_x, const float *d_W, const float *d_U, const float *d_bi, const float *d_bh,
                                float *d_h, float *d_gate, int B, int T, int in, int H, int v2, int act, float scale);
int nntk_shim_rnn_train_backward(const float *d_dout, const float *d_UT /*U transposed [H][H]*/, const float *d_h, const float *d_gate, float *d_dG,
                                 float *d_work /*2*B*H*/, int B, int T, int H, int return_sequences, int act);
/* MFMA forms of the large training products: C [M][N] (+)= A [M][K] x Bw [N][K]^T through the inference GEMM kernel (Bw is
 * packed per call into d_pack >= nntk_shim_gemm_nt_scratch_floats(N, K) floats; accumulate uses d_tmp [M][N]) */
size_t nntk_shim_gemm_nt_scratch_floats(int N, int K);
int nntk_shim_gemm_nt(const float *d_A, const float *d_Bw, float *d_C, float *d_pack, float *d_tmp, long M, int N, int K, int accumulate);
int nntk_shim_transpose(const float *d_src, float *d_dst, long R, int C, int shift_T);
/* MFMA form of the Conv1d input gradient (stride 1): the forward kernel on zero-padded d_out with flipped weights (train.hip) */
size_t nntk_shim_conv_dx_pack_floats(int Cin, int Cout, int k);
int nntk_shim_conv_dx_mfma(const float *d_dout, const float *d_W, float *d_dX, float *d_pad, float *d_wpack,
                           int B, int T, int Cin, int Cout, int k, int Tout);
/* C [I][K] += A^T B over `rows` rows, c [K] += column sums of B (a_shift_T > 0: A is h [B][T][I] and row (b,t) uses h_{t-1}) */
size_t nntk_shim_outer_scratch_floats(int I, int K);
int nntk_shim_outer_accumulate(const float *d_A, const float *d_B, float *d_C, float *d_c, float *d_scratch, long rows, int I, int K, int a_shift_T);
/* out [rows][I] = d [rows][K] times M [I][K]^T, k-ordered */
int nntk_shim_rows_times_rowmat(const float *d_d, const float *d_M, float *d_out, long rows, int I, int K);
int nntk_shim_bn_train_slices(long N, int *rows_per_slice);
int nntk_shim_bn_train_forward(const float *d_x, const float *d_block, float eps, float *d_stats, float *d_partial, float *d_out, long N, int F);
int nntk_shim_bn_train_backward(const float *d_x, const float *d_dout, const float *d_block, float *d_stats, float *d_partial, float *d_dx, long N, int F);
int nntk_shim_concat2(const float *d_a, const float *d_b, float *d_out, long rows, int C);
int nntk_shim_add2(const float *d_a, const float *d_b, float *d_out, long n);
/* signal/dft.h: complex DFT of any size n (direct, double accumulation); d_tw = 2n floats from nntk_shim_dft_twiddles */
int nntk_shim_dft_twiddles(float *d_tw, int n);
int nntk_shim_dft(const float *d_re, const float *d_im, const float *d_tw, float *d_ore, float *d_oim, int n, int inverse);
int nntk_shim_split2(const float *d_in, float *d_a, float *d_b, long rows, int C);

/* ---- K4: recurrent layers -------------------------------------------------
 * d_xw   [T, B, G*H] time-major input projections INCLUDING b_i (from nntk_shim_conv1d out_mode 1)
 * d_ut   packed recurrent weights U^T: [G*H, H] (row n = column n of U)
 * d_bh   [G*H] recurrent bias (NULL = none; LSTM v2=false)
 * d_h0 / d_c0  [B, H] initial state or NULL for zeros
 * d_out  [B, T, H] if return_sequences else [B, H]
 * d_hT / d_cT  [B, H] final state (may be NULL)
 * d_work scratch >= nntk_shim_recurrent_work_floats(B, H) floats
 * acts   GRU: {z, h, r};  LSTM: {i, f, g, o, out}
 * act_scales  parallel to acts: ReLU's output scale `a` (activation_default.c:123-129), ignored for other kinds; NULL = 1
 */
/* simple RNN cell (rnn.c:144-166): one gate, h' = act(xW + hU + b); same buffers as nntk_shim_gru */
int nntk_shim_rnn(const float *d_xw, const float *d_ut, const float *d_bh, const float *d_h0,
                  float *d_out, float *d_hT, float *d_work, int B, int T, int H,
                  int return_sequences, int act, float act_scale);
size_t nntk_shim_recurrent_work_floats(int B, int H);
int nntk_shim_gru(const float *d_xw, const float *d_ut, const float *d_bh,
                  const float *d_h0, float *d_out, float *d_hT, float *d_work,
                  int B, int T, int H, int return_sequences, const int acts[3], const float act_scales[3]);
int nntk_shim_lstm(const float *d_xw, const float *d_ut, const float *d_bh,
                   const float *d_h0, const float *d_c0, float *d_out, float *d_hT, float *d_cT,
                   float *d_work, int B, int T, int H, int return_sequences, const int acts[5], const float act_scales[5]);

/* Register-resident split-bf16 recurrent kernels with the input projection fused into the step (recurrent_rr.hip): standard
 * activations, H % 16 == 0, 64 <= H <= 512, in <= 128 (in <= 256 when H <= 256); f32 input: in % 8 == 0; frag3 input: any in.
 * d_x [B][T][in] f32 -- or NULL with d_xf3 = the same tensor in frag3 form (frag3.hip); d_img = weight images made by
 * nntk_shim_lstm_rr_pack from the per-gate U^T (d_ut) and the packed W^T (d_wp); d_bh NULL for the one-bias form.
 * d_hseq (nntk_shim_rr_hseq_floats) receives the layer output of every step in frag3 form -- it is the kernels' hand-off buffer;
 * d_out (f32) may be NULL when the caller consumes d_hseq.  d_work: nntk_shim_lstm_rr_work_floats.
 * nntk_shim_lstm_rr / nntk_shim_gru_rr return 1 when the shape / configuration is not taken (nothing launched). */
size_t nntk_shim_lstm_rr_image_floats(int H, int in);           /* 0: shape not taken (f32 input) */
size_t nntk_shim_rr_image_floats_xf(int H, int in);             /* 0: shape not taken (frag3 input) */
size_t nntk_shim_lstm_rr_work_floats(int B, int H);
size_t nntk_shim_rr_hseq_floats(int B, int T, int H);
int nntk_shim_lstm_rr_pack(const float *d_ut, const float *d_wp, float *d_img, int H, int in);
int nntk_shim_lstm_rr_pack_raw(const float *d_U /*[H][4H]*/, const float *d_W /*[in][4H]*/, float *d_img, int H, int in);
/* GRU on the same kernels (gru_rr_kernel): image from the four-slot matrices, d_b4 [4H]; see recurrent_rr.hip */
/* full-K family (recurrent_fk.hip: no split-K; H <= 256, frag3 input): own weight images d_img4 (NULL: family not used).  A shape this
 * family takes ALWAYS runs on it (the host packs an f32 input into frag3 form first), so a row's bits never depend on the call's size */
size_t nntk_shim_fk_image_floats(int H, int in);                /* 0: shape not taken (or option rec_fk = 0) */
int nntk_shim_fk_pack(const float *d_ut, const float *d_wp, float *d_img4, int H, int in);
int nntk_shim_fk_pack_raw(const float *d_U /*[H][4H]*/, const float *d_W /*[in][4H]*/, float *d_img4, int H, int in);
int nntk_shim_gru_rr(const float *d_x, const void *d_xf3, const float *d_img, const float *d_img4, const float *d_b4, const float *d_h0, float *d_out,
                     float *d_hseq, float *d_hT, float *d_work, int B, int T, int in, int H, int return_sequences, int x_tm, int out_tm);
int nntk_shim_gru_rr_train_forward(const float *d_x, const float *d_img, const float *d_b4, float *d_h, float *d_hU, float *d_Zg,
                                   float *d_hseq, float *d_work, int B, int T, int in, int H);
int nntk_shim_lstm_rr_train_forward(const float *d_x, const float *d_img, const float *d_bi, const float *d_bh,
                                    float *d_h, float *d_c, float *d_zifgo, float *d_hseq, float *d_work, int B, int T, int in, int H);
int nntk_shim_lstm_rr(const float *d_x, const void *d_xf3, const float *d_img, const float *d_img4, const float *d_bi, const float *d_bh,
                      const float *d_h0, const float *d_c0, float *d_out, float *d_out_h2 /* frag2h form of the sequence output, or NULL; needs d_out == NULL */,
                      float *d_hseq, float *d_hT, float *d_cT, float *d_work, int B, int T, int in, int H, int return_sequences);

/* The HF instantiations of lstm_rr_kernel (H > 256): the h part of Z on two f16 images (three products per k step), the hand-off = the layer's
 * sequence output as a FRAG2H tensor (d_h2: nntk_shim_frag2h_floats(B, T, H) floats).  Zero initial state, x as a frag3 tensor.  Images:
 * nntk_shim_lstm_rr_pack_hf with uscale = a power of two with max |U| uscale <= 32768 and wscale = 32768 uscale; z_scale = 1 / wscale. */
int nntk_shim_lstm_rr_hf_ok(int H, int in);                     /* 256 < H <= 512, H % 16 == 0, in <= 256 (U has no LDS image: a 256-wide W fits), option rec_hf */
size_t nntk_shim_lstm_rr_hf_image_floats(int H, int in);
int nntk_shim_lstm_rr_pack_hf(const float *d_ut, const float *d_wp, float *d_img, int H, int in, float uscale, float wscale);
int nntk_shim_lstm_rr_hf(const void *d_xf3, const float *d_img, const float *d_bi, const float *d_bh, float *d_h2, float *d_work,
                         int B, int T, int in, int H, float z_scale);

/* ---- frag3 tensors (frag3.hip): a [B][T][C] f32 tensor as three bf16 images (x = hi + mid + lo exactly) in MFMA fragment order,
 * [T][2 ceil(B / 64)][ceil(C / 16)][3] blocks of 1 KB.  nntk_shim_dense_frag3: out [B][T][N] = act(h . W + b) with h in frag3 form and
 * d_wp the packed weights of a Dense layer; returns 1 when the shape is not taken. */
size_t nntk_shim_frag3_floats(int B, int T, int C);
int nntk_shim_frag3_pack(const float *d_x, void *d_frag, int B, int T, int C);
int nntk_shim_frag3_unpack(const void *d_frag, float *d_x, int B, int T, int C);
/* FRAG2H (frag3.hip): the same tensor as two f16 images of x * 2^15 (|x| < 2), [T][2 ceil(B / 64)][ceil(C / 16)][2] blocks of 1 KB; the
 * dense GEMM on that form sums three products per k step.  d_wh2: the two f16 images of W * w_scale made by nntk_shim_split_f16x2 from the
 * packed f32 matrix ([N_p][K_p], nntk_shim_conv_pack_sizes; 2 * N_p * K_p halves), w_scale a power of two with max |W| w_scale <= 32768.
 * nntk_shim_dense_frag2h returns 1 when the shape / configuration is not taken (probe = 1: asks only, launches nothing). */
size_t nntk_shim_frag2h_floats(int B, int T, int C);
int nntk_shim_frag2h_pack(const float *d_x, void *d_frag, int B, int T, int C);
int nntk_shim_frag2h_unpack(const void *d_frag, float *d_x, int B, int T, int C);
int nntk_shim_split_f16x2(const float *d_src, void *d_dst, int rows, int ktot, float scale);
int nntk_shim_dense_frag2h(const void *d_frag, const void *d_wh2, float w_scale, const float *d_bias, int act_kind, float relu_a,
                           float *d_out, int B, int T, int K, int N, int probe);
int nntk_shim_dense_frag3(const void *d_frag, const float *d_wp, const float *d_bias, int act_kind, float relu_a,
                          float *d_out, int B, int T, int K, int N);

/* fused two-layer GRU (standard activations, zero initial state, both layers H units): layer 2's input projection and
 * recurrence run inside layer 1's persistent launch, one step behind.  d_wt2 = W2^T packed like U^T.  Returns 1 when
 * the shape is not taken (nothing launched). */
size_t nntk_shim_gru2_work_floats(int B, int H);
int nntk_shim_gru2(const float *d_xw1, const float *d_ut1, const float *d_bh1, const float *d_wt2,
                   const float *d_bi2, const float *d_ut2, const float *d_bh2, float *d_out, float *d_out1,
                   float *d_work, int B, int T, int H, int return_sequences);

/* streaming form: ONE sequence, T small; x [T, in] and out ([T, H] or [H]) may be pinned host memory (read / written by
 * the kernel directly, no copies); state h[cur] (c[cur]) -> h[(cur + T) & 1]; the last launch stores `seq` to *flag (a
 * pinned host word) once all outputs are visible to the host.  Bit-compatible with nntk_shim_gru / _lstm / _rnn. */
int nntk_shim_rec_stream(int G, int is_lstm, const float *x, const float *d_wp, const float *d_bi,
                         const float *d_ut, const float *d_bh, float *d_h0, float *d_h1, float *d_c0, float *d_c1,
                         int cur, float *out, int T, int in, int H, int return_sequences,
                         const int *acts, const float *scales, unsigned *d_done, unsigned *flag, unsigned seq);

/* ---- multi-GPU: one RCCL communicator per process, weight-block broadcast at start-up (dist.hip) ---- */
int nntk_shim_dist_unique_id(unsigned char *id128);
int nntk_shim_dist_init(const unsigned char *id128, int rank, int world);
int nntk_shim_dist_rank(void);
int nntk_shim_dist_world(void);
int nntk_shim_dist_broadcast_host(float *block, size_t n, int root);
int nntk_shim_add_into(float *d_dst, const float *d_src, long n);      /* dst += src, async */
int nntk_shim_dist_barrier(void);
int nntk_shim_dist_allreduce_device(float *d_block, size_t n);    /* in-place sum over the ranks, async on the stream */
int nntk_shim_dist_allreduce_host(float *block, size_t n);          /* staged, blocking */
int nntk_shim_dist_finalize(void);

/* ---- K1: framed STFT magnitude / PSD ---------------------------------------
 * d_in [B, input_size], d_window [window_size], d_out [B, nts, nfreq]
 * d_twiddle [nfft] complex interleaved: exp(-2*pi*i*m/nfft), evaluated in double on the host
 * mode 0: sqrt(re^2+im^2)/scale   mode 1: psd (spectrogram.c:41-47)
 */
int nntk_shim_spectrogram(const float *d_in, const float *d_window, const float *d_twiddle, float *d_out,
                          int B, int input_size, int nfft, int window_size, int step,
                          int nfreq, int nts, float fft_norm, int mode, float scale);

/* K1 with the mel projection (+ optional log(x + eps)) fused into its output stage; returns 1 when the configuration
 * is not taken by the fused kernel (nothing launched) */
int nntk_shim_spectrogram_mel(const float *d_in, const float *d_window, const float *d_twiddle, float *d_out,
                              int B, int input_size, int nfft, int window_size, int step,
                              int nfreq, int nts, float fft_norm, int mode, float scale,
                              const int *d_mel_tab, const float *d_mel_w, int n_mels, float eps, int do_log);

#ifdef __cplusplus
}
#endif
#endif
