/*
 * nntk_internal.h -- private helpers of the C host layer.  The host layer owns
 * handles, the caller-visible (pinned) weight blocks, lazy upload / repacking and
 * the recurrent state; all GPU work goes through csrc/hip/nntk_shim.h.
 */
#ifndef NNTK_INTERNAL_H
#define NNTK_INTERNAL_H

#include <stddef.h>
#include "nntoolkitcore_hip.h"
#include "../hip/nntk_shim.h"

/* One contiguous, zero-initialised, caller-visible weight block (the reference's
 * f_malloc'd block: weights_private.c:16-21, recurrent_private.c:29-36,
 * batch_norm.c:79-84), pinned for fast upload, plus a shadow of what the device
 * currently holds so host-pointer Apply calls can detect in-place edits. */
typedef struct {
    float *host;
    float *shadow;
    size_t n;
    int uploaded;
} nntk_wblock;

int  nntk_wblock_init(nntk_wblock *wb, size_t n_floats);
void nntk_wblock_free(nntk_wblock *wb);
/* 1 if the device copy is missing or (check_edits && host != shadow) */
int  nntk_wblock_dirty(const nntk_wblock *wb, int check_edits);
void nntk_wblock_mark_uploaded(nntk_wblock *wb);

/* growable device scratch */
typedef struct { float *p; size_t cap; } nntk_devbuf;
float *nntk_devbuf_reserve(nntk_devbuf *b, size_t n_floats);
void   nntk_devbuf_free(nntk_devbuf *b);

/* Scratch that belongs to no handle (losses, the optimiser, the gradient products, the bidirectional helpers): one set of
 * growable buffers per host thread AND device -- a thread that switches devices gets that device's set, never a pointer
 * into another GPU's memory -- released by a thread-exit hook (the former bare _Thread_local buffers leaked
 * device memory with every thread that ever computed a gradient). */
enum { NNTK_TS_A, NNTK_TS_B, NNTK_TS_C, NNTK_TS_AT, NNTK_TS_BT, NNTK_TS_PACK, NNTK_TS_TMP, NNTK_TS_SCR,
       NNTK_TS_BD_A, NNTK_TS_BD_B, NNTK_TS_BD_OUT, NNTK_TS_SLOTS };
nntk_devbuf *nntk_thread_scratch(int slot);

/* upload a host array into a fresh/reused device buffer */
int nntk_upload_floats(float **d_dst, const float *h_src, size_t n);

/* pack a row-major [K, N] matrix into the conv/GEMM kernel's [N_p][K_p] (K-contiguous) layout and upload */
int nntk_upload_gemm_weights(float **d_wp, const float *W, int K, int N);
/* the scale of a weight block's two f16 images (frag3.hip FRAG2H): the largest power of two with max |W| * scale <= 32 768; 0 = not available
 * (a non-finite value, or a block whose largest magnitude no power of two brings there) */
float nntk_f16_scale(const float *W, size_t n);
/* training products (train.c): VALU in the reference's order when small, the MFMA GEMM when large */
int nntk_train_outer_accumulate(const float *d_A, const float *d_B, float *d_C, float *d_c, long rows, int I, int K, int a_shift_T);
int nntk_train_rows_times_rowmat(const float *d_d, const float *d_M, float *d_out, long rows, int I, int K);
int nntk_upload_packed_weights(float **d_wp, const float *h_packed, int rows, int ktot);

struct ActivationFunctionStruct {
    int kind;                 /* NNTK_ACT_* */
    int input_size;
    float relu_a;
    int vector_size;          /* softmax */
    /* custom host-callback activations (activation.h:19-26) */
    void *implementer;
    ActivationImplementerDestroy destroy_fn;
    ActivationFunctionImpl function;
    ActivationFunctionDerivative derivative;
    ActivationFunctionDerivative cached_derivative;
};

/* kind usable inside a fused kernel epilogue? (identity / sigmoid / tanh / relu, or NULL handle) */
int nntk_act_fusable(ActivationFunction a);
int nntk_act_kind(ActivationFunction a);   /* NNTK_ACT_NONE for NULL */

/* accessors used across host files */
const float *nntk_batch_norm_device_block(BatchNorm bn, int check_edits);   /* gamma|beta|mean|var or NULL on error */
int   nntk_batch_norm_channels(BatchNorm bn);
float nntk_batch_norm_epsilon(BatchNorm bn);

const float *nntk_mel_weights(MelFilterBank bank);   /* host [nbins, n_mels] */
int nntk_spectrogram_apply_mel_device(Spectrogram filter, const float *d_input, float *d_output, int batch,
                                      const int *d_mel_tab, const float *d_mel_w, int n_mels, float eps, int do_log);

/* recurrent.c, for the fused LSTM -> TimeDistributedDense call in dense.c */
void nntk_lstm_dims(LSTM f, int *T, int *in, int *H, int *return_sequences);
float *nntk_lstm_frag3_scratch(LSTM f, int batch);       /* the handle's own frag3 output buffer [batch][T][H] */
float *nntk_lstm_frag2h_scratch(LSTM f, int batch);      /* ... and its FRAG2H one */
/* the LSTM's sequence output as a FRAG2H tensor: 0 done, 1 not available for this layer (non-standard activations), -1 error */
int nntk_lstm_apply_device_h2(LSTM f, const float *d_in, const float *d_in_f3, float *d_out_h2, int B);

void nntk_set_error(const char *msg);
#define NNTK_FAIL(msg) do { nntk_set_error(msg); return -1; } while (0)

#endif
