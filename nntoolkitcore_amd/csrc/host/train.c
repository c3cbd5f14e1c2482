/*
 * train.c -- host layer of the training path's second slice: the reference's train/loss.{h,c},
 * train/optimizers.{h,c} and ActivationFunctionCalculateGradient (layers/activation.c:47-54), under their own names
 * and host-pointer signatures, plus device-pointer forms for callers that keep tensors in HBM.  The kernels
 * (csrc/hip/train.hip) follow the reference's operation order.
 */
#include <stdlib.h>
#include <string.h>
#include "nntk_internal.h"

/* scratch for the host-pointer forms: per thread and device, freed at thread exit (nntk_thread_scratch, runtime.c) */
#define t_a (*nntk_thread_scratch(NNTK_TS_A))
#define t_b (*nntk_thread_scratch(NNTK_TS_B))
#define t_c (*nntk_thread_scratch(NNTK_TS_C))

/* ---- the three large products of every gradient: VALU kernels in the reference's order for small shapes, the MFMA GEMM
 *      (csrc/hip/train.hip "MFMA forms") once rows * I * K passes 2^27 multiply-adds ---- */
#define t_at (*nntk_thread_scratch(NNTK_TS_AT))
#define t_bt (*nntk_thread_scratch(NNTK_TS_BT))
#define t_pack (*nntk_thread_scratch(NNTK_TS_PACK))
#define t_tmp (*nntk_thread_scratch(NNTK_TS_TMP))
#define t_scr (*nntk_thread_scratch(NNTK_TS_SCR))
#define NNTK_TRAIN_MFMA_MACS ((double)(1 << 27))

/* C [I][K] += A [rows][I]^T B [rows][K];  c [K] += column sums of B.  a_shift_T > 0: A is h [B][T][I], row (b,t) uses h_{t-1} */
int nntk_train_outer_accumulate(const float *d_A, const float *d_B, float *d_C, float *d_c, long rows, int I, int K, int a_shift_T) {
    if (rows <= 0 || K <= 0) return 0;
    /* large products take the row-sliced MFMA form inside the call (train.hip outer_mfma_kernel), small ones the VALU dots */
    float *scr = nntk_devbuf_reserve(&t_scr, nntk_shim_outer_scratch_floats(I, K));
    if (!scr) return -1;
    return nntk_shim_outer_accumulate(d_A, d_B, d_C, d_c, scr, rows, I, K, a_shift_T);
}
/* out [rows][I] = d [rows][K] M [I][K]^T */
int nntk_train_rows_times_rowmat(const float *d_d, const float *d_M, float *d_out, long rows, int I, int K) {
    if (rows <= 0 || I <= 0) return 0;
    const int mfma = (double)rows * I * K >= NNTK_TRAIN_MFMA_MACS && K >= 16 && I >= 32;
    if (!mfma) return nntk_shim_rows_times_rowmat(d_d, d_M, d_out, rows, I, K);
    float *pack = nntk_devbuf_reserve(&t_pack, nntk_shim_gemm_nt_scratch_floats(I, K));
    if (!pack) return -1;
    return nntk_shim_gemm_nt(d_d, d_M, d_out, pack, NULL, rows, I, K, 0);
}

/* ---- activation gradient (activation.c:47-54): cached derivative on `a` when there is one, else derivative on z ---- */
int ActivationFunctionCalculateGradientDevice(ActivationFunction filter, const float *d_z, const float *d_a,
                                              const float *d_dout, float *d_output, int size) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("ActivationFunctionCalculateGradientDevice: NULL handle");
    long n = size > 0 ? size : filter->input_size;
    int vpc = 1;
    if (filter->kind == NNTK_ACT_SOFTMAX) { vpc = filter->input_size; n *= filter->vector_size; }
    /* ReLU and identity have no cached derivative in the reference (activation_default.c:106, :138): z is used */
    return nntk_shim_activation_grad(filter->kind, filter->vector_size, vpc, d_z, d_a, d_dout, d_output, n);
}

void ActivationFunctionCalculateGradient(ActivationFunction filter, const float *z, const float *a, const float *d_out,
                                         float *output) {
    nntk_shim_clear_error();
    if (!filter) { nntk_set_error("ActivationFunctionCalculateGradient: NULL handle"); return; }
    if (filter->kind == NNTK_ACT_CUSTOM) {          /* the caller's own host functions */
        if (filter->cached_derivative == NULL || a == NULL) {
            if (filter->derivative) filter->derivative(filter->implementer, z, d_out, output, filter->input_size);
        } else {
            filter->cached_derivative(filter->implementer, a, d_out, output, filter->input_size);
        }
        return;
    }
    long n = filter->kind == NNTK_ACT_SOFTMAX ? (long)filter->input_size * filter->vector_size : filter->input_size;
    if (n <= 0) return;
    if ((filter->kind == NNTK_ACT_RELU || filter->kind == NNTK_ACT_IDENTITY) && !z && filter->kind == NNTK_ACT_RELU) {
        nntk_set_error("ActivationFunctionCalculateGradient: ReLU needs z"); return;
    }
    float *d = nntk_devbuf_reserve(&t_a, (size_t)4 * n);
    if (!d) return;
    float *dz = d, *da = d + n, *dd = d + 2 * n, *dout = d + 3 * n;
    /* softmax without a cached output: recompute it from z first (activation_default.c:187-190) */
    const int need_fwd = filter->kind == NNTK_ACT_SOFTMAX && !a;
    if (z && nntk_shim_upload(dz, z, (size_t)n * sizeof(float))) return;
    if (a && nntk_shim_upload(da, a, (size_t)n * sizeof(float))) return;
    if (need_fwd) {
        if (!z) { nntk_set_error("ActivationFunctionCalculateGradient: softmax needs z or a"); return; }
        if (nntk_shim_activation(NNTK_ACT_SOFTMAX, 1.f, filter->vector_size, dz, da, n)) return;
    }
    if (nntk_shim_upload(dd, d_out, (size_t)n * sizeof(float))) return;
    if (nntk_shim_activation_grad(filter->kind, filter->vector_size, filter->input_size, z ? dz : NULL,
                                  (a || need_fwd) ? da : NULL, dd, dout, n)) return;
    nntk_shim_download(output, dout, (size_t)n * sizeof(float));
}

/* ---- losses (train/loss.c) ---- */
static int loss_value(int kind, const float *d_y, const float *d_pred, int size, int batch, float *loss) {
    *loss = 0.0f;
    if (size <= 0 || batch <= 0) return 0;
    float *d_rows = nntk_devbuf_reserve(&t_c, (size_t)batch);
    if (!d_rows) return -1;
    if (nntk_shim_loss_rows(kind, d_y, d_pred, d_rows, size, batch)) return -1;
    float *rows = (float *)malloc((size_t)batch * sizeof(float));
    if (!rows) NNTK_FAIL("out of host memory");
    if (nntk_shim_download(rows, d_rows, (size_t)batch * sizeof(float))) { free(rows); return -1; }
    float acc = 0.0f;
    for (int b = 0; b < batch; ++b) acc += rows[b];          /* loss.c:15-22 / :36-44: summed over the batch in order */
    free(rows);
    *loss = acc / (float)batch;
    return 0;
}
int nntk_mean_squared_error_device(const float *d_y, const float *d_pred, int size, int batch, float *loss) {
    nntk_shim_clear_error();
    return loss_value(0, d_y, d_pred, size, batch, loss);
}
int nntk_categorical_crossentropy_device(const float *d_y, const float *d_pred, int c, int batch, float *loss) {
    nntk_shim_clear_error();
    return loss_value(1, d_y, d_pred, c, batch, loss);
}
int nntk_mean_squared_error_derivative_device(const float *d_y, const float *d_pred, float *d_out, int size, int batch) {
    nntk_shim_clear_error();
    return nntk_shim_loss_grad(0, d_y, d_pred, d_out, size, batch);
}
int nntk_categorical_crossentropy_derivative_device(const float *d_y, const float *d_pred, float *d_out, int c, int batch) {
    nntk_shim_clear_error();
    return nntk_shim_loss_grad(1, d_y, d_pred, d_out, c, batch);
}

static int stage2(const float *y, const float *p, size_t n, float **dy, float **dp) {
    float *d = nntk_devbuf_reserve(&t_a, 2 * n);
    if (!d) return -1;
    *dy = d; *dp = d + n;
    if (nntk_shim_upload(*dy, y, n * sizeof(float))) return -1;
    return nntk_shim_upload(*dp, p, n * sizeof(float));
}
float mean_squared_error(float *y, float *y_pred, int size, int batch) {
    nntk_shim_clear_error();
    float *dy, *dp, loss = 0.0f;
    if ((size_t)size * batch == 0 || stage2(y, y_pred, (size_t)size * batch, &dy, &dp)) return 0.0f;
    loss_value(0, dy, dp, size, batch, &loss);
    return loss;
}
float categorical_crossentropy(float *y, float *y_pred, int c, int batch) {
    nntk_shim_clear_error();
    float *dy, *dp, loss = 0.0f;
    if ((size_t)c * batch == 0 || stage2(y, y_pred, (size_t)c * batch, &dy, &dp)) return 0.0f;
    loss_value(1, dy, dp, c, batch, &loss);
    return loss;
}
static void loss_grad_host(int kind, float *y, float *y_pred, float *d_y_pred, int size, int batch) {
    nntk_shim_clear_error();
    size_t n = (size_t)size * batch;
    float *dy, *dp;
    if (n == 0 || stage2(y, y_pred, n, &dy, &dp)) return;
    float *dd = nntk_devbuf_reserve(&t_b, n);
    if (!dd || nntk_shim_loss_grad(kind, dy, dp, dd, size, batch)) return;
    nntk_shim_download(d_y_pred, dd, n * sizeof(float));
}
void mean_squared_error_derivative(float *y, float *y_pred, float *d_y_pred, int size, int batch) {
    loss_grad_host(0, y, y_pred, d_y_pred, size, batch);
}
void categorical_crossentropy_derivative(float *y, float *y_pred, float *d_y_pred, int c, int batch) {
    loss_grad_host(1, y, y_pred, d_y_pred, c, batch);
}

/* ---- SGD (train/optimizers.c:13-19) ---- */
int nntk_sgd_optimize_device(SGD optimizer, const float *d_gradient, float *d_weights, long size) {
    nntk_shim_clear_error();
    return nntk_shim_sgd(optimizer.learning_rate, d_gradient, d_weights, size);
}
int sgd_optimize(SGD optimizer, float *gradient, float *weights, int size) {
    nntk_shim_clear_error();
    if (size <= 0) return 0;
    float *dg, *dw;
    if (stage2(gradient, weights, (size_t)size, &dg, &dw)) return -1;
    if (nntk_shim_sgd(optimizer.learning_rate, dg, dw, size)) return -1;
    return nntk_shim_download(weights, dw, (size_t)size * sizeof(float));
}
