/*
 * dense.c -- Dense and TimeDistributedDense host layer.  Reference:
 * layers/dense.c:24-28 (weight block W [in,out] | b [out]), :122-142 (z = x*W + b,
 * a = activation(z)), layers/time_distributed_dense.c:31-58 (ts Dense
 * applications sharing weights).  On the device every row (timestep, batch
 * element) is one row of a single MFMA GEMM with the bias and the activation
 * fused; softmax (which spans a row) runs as a second wave-per-row kernel.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "nntk_internal.h"

struct DenseStruct {
    DenseConfig config;
    DenseWeights *weights;
    nntk_wblock wb;
    float *d_wp, *d_bias;
    /* FRAG2H operand form of W (frag3.hip): two f16 images of W * wh2_scale behind the bf16 ones' pattern; wh2_scale = 0: not available
     * (a non-finite weight, or a block whose largest magnitude no power of two brings to 32 768) */
    void *d_wh2;
    float wh2_scale;
    nntk_devbuf d_in, d_out;
    /* training (dense.c:18-48): x | z | a kept from DenseApplyTrainingBatch for DenseCalculateGradient */
    int training, mini_batch;
    nntk_devbuf d_x, d_z, d_a, d_dz, d_dout, d_grad, d_wraw, d_dx;
    const float *d_x_cur;       /* the mini-batch input of the last training forward: d_x.p (host form) or the caller's device buffer */
};

/* dense.c:67-73 */
DenseConfig DenseConfigCreate(int input_size, int output_size, ActivationFunction activation) {
    DenseConfig c;
    memset(&c, 0, sizeof(c));
    c.input_size = input_size;
    c.output_size = output_size;
    c.activation = activation;
    return c;
}

/* dense.c:77-83 */
Dense DenseCreateForInference(DenseConfig config) {
    nntk_shim_clear_error();
    Dense f = (Dense)calloc(1, sizeof(struct DenseStruct));
    if (!f) return NULL;
    f->config = config;
    size_t w = (size_t)config.input_size * config.output_size;
    if (nntk_wblock_init(&f->wb, w + config.output_size)) { free(f); return NULL; }
    f->weights = (DenseWeights *)malloc(sizeof(DenseWeights));
    f->weights->W = f->wb.host;
    f->weights->b = f->wb.host + w;
    return f;
}

DenseWeights *DenseGetWeights(Dense filter) { return filter->weights; }

void DenseDestroy(Dense filter) {
    if (!filter) return;
    nntk_shim_synchronize();
    nntk_shim_free(filter->d_wp);
    nntk_shim_free(filter->d_wh2);
    nntk_shim_free(filter->d_bias);
    nntk_devbuf_free(&filter->d_in);
    nntk_devbuf_free(&filter->d_out);
    nntk_devbuf_free(&filter->d_x); nntk_devbuf_free(&filter->d_z); nntk_devbuf_free(&filter->d_a);
    nntk_devbuf_free(&filter->d_dz); nntk_devbuf_free(&filter->d_dout); nntk_devbuf_free(&filter->d_grad);
    nntk_devbuf_free(&filter->d_wraw); nntk_devbuf_free(&filter->d_dx);
    nntk_wblock_free(&filter->wb);
    free(filter->weights);
    free(filter);
}

static int dense_upload(Dense f) {
    if (nntk_upload_gemm_weights(&f->d_wp, f->weights->W, f->config.input_size, f->config.output_size)) return -1;
    if (nntk_upload_floats(&f->d_bias, f->weights->b, (size_t)f->config.output_size)) return -1;
    {
        int K_p, N_p;
        nntk_shim_conv_pack_sizes(f->config.input_size, f->config.output_size, 1, &K_p, &N_p);
        const size_t n_w = (size_t)K_p * N_p;
        f->wh2_scale = nntk_f16_scale(f->weights->W, (size_t)f->config.input_size * f->config.output_size);
        if (f->wh2_scale > 0.f) {
            if (!f->d_wh2 && !(f->d_wh2 = nntk_shim_malloc(n_w * 2 * sizeof(unsigned short)))) return -1;
            if (nntk_shim_split_f16x2(f->d_wp, f->d_wh2, N_p, K_p, f->wh2_scale)) return -1;
        }
    }
    nntk_wblock_mark_uploaded(&f->wb);
    return 0;
}
static int dense_ensure(Dense f, int check_edits) {
    if (nntk_wblock_dirty(&f->wb, check_edits)) return dense_upload(f);
    return 0;
}
int DenseSyncWeights(Dense filter) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("DenseSyncWeights: NULL handle");
    nntk_shim_synchronize();
    return dense_upload(filter);
}

int DenseBroadcastWeights(Dense filter, int root) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("DenseBroadcastWeights: NULL handle");
    if (nntk_shim_dist_broadcast_host(filter->wb.host, filter->wb.n, root)) return -1;
    return dense_upload(filter);
}

static int dense_rows_device(Dense f, const float *d_in, float *d_out, long rows) {
    if (rows <= 0) return 0;
    if (rows > 0x7fffffffL) NNTK_FAIL("dense: too many rows");
    ActivationFunction act = f->config.activation;
    int kind = nntk_act_kind(act);
    if (kind == NNTK_ACT_CUSTOM)
        NNTK_FAIL("dense: custom host-callback activation cannot run on the device");
    int fused = nntk_act_fusable(act);
    if (nntk_shim_conv1d(d_in, f->d_wp, f->d_bias, NULL, 0.f, fused ? kind : NNTK_ACT_IDENTITY,
                         act ? act->relu_a : 1.f, d_out, 1, (int)rows, f->config.input_size,
                         f->config.output_size, 1, 1, (int)rows, 0))
        return -1;
    if (!fused) {   /* softmax over each row's vectors (activation_default.c:157-167) */
        if ((long)act->input_size * act->vector_size != f->config.output_size)
            NNTK_FAIL("dense: softmax input_size * vector_size must equal the dense output_size");
        return nntk_shim_activation(NNTK_ACT_SOFTMAX, 1.f, act->vector_size, d_out, d_out,
                                    rows * (long)f->config.output_size);
    }
    return 0;
}

int DenseApplyDevice(Dense filter, const float *d_input, float *d_output, int rows) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("DenseApplyDevice: NULL handle");
    if (dense_ensure(filter, 0)) return -1;
    return dense_rows_device(filter, d_input, d_output, rows);
}

/* rows (b, t) of a frag3 tensor [B][T][in]: the register-direct GEMM (frag3.hip dense_frag3_kernel) when it takes the shape, else the
 * tensor is unpacked (exactly) and the ordinary GEMM runs -- same split products in the same order, same bits */
static int dense_frag3_device(Dense f, const float *d_in_f3, float *d_out, int B, int T) {
    if (B <= 0 || T <= 0) return 0;
    ActivationFunction act = f->config.activation;
    int kind = nntk_act_kind(act);
    if (kind == NNTK_ACT_CUSTOM) NNTK_FAIL("dense: custom host-callback activation cannot run on the device");
    const int fused = nntk_act_fusable(act);
    int rc = nntk_shim_dense_frag3(d_in_f3, f->d_wp, f->d_bias, fused ? kind : NNTK_ACT_IDENTITY, act ? act->relu_a : 1.f, d_out,
                                   B, T, f->config.input_size, f->config.output_size);
    if (rc < 0) return -1;
    if (rc == 1) {
        float *xs = nntk_devbuf_reserve(&f->d_in, (size_t)B * T * f->config.input_size);
        if (!xs || nntk_shim_frag3_unpack(d_in_f3, xs, B, T, f->config.input_size)) return -1;
        return dense_rows_device(f, xs, d_out, (long)B * T);
    }
    if (!fused) {
        if ((long)act->input_size * act->vector_size != f->config.output_size)
            NNTK_FAIL("dense: softmax input_size * vector_size must equal the dense output_size");
        return nntk_shim_activation(NNTK_ACT_SOFTMAX, 1.f, act->vector_size, d_out, d_out, (long)B * T * f->config.output_size);
    }
    return 0;
}

/* rows (b, t) of a FRAG2H tensor [B][T][in] (frag3.hip: two f16 images, three products per k step): dense_frag3_kernel's f16 instantiation when
 * it takes the shape and W has the form, else the tensor is unpacked and the ordinary GEMM runs (other products: results agree to the
 * tolerance of the layer, not bit for bit -- see the header) */
static int dense_frag2h_takes(Dense f, float *d_out, int B, int T) {
    ActivationFunction act = f->config.activation;
    int kind = nntk_act_kind(act);
    if (kind == NNTK_ACT_CUSTOM || !(f->wh2_scale > 0.f)) return 0;
    const int fused = nntk_act_fusable(act);
    return nntk_shim_dense_frag2h(NULL, f->d_wh2, f->wh2_scale, f->d_bias, fused ? kind : NNTK_ACT_IDENTITY, act ? act->relu_a : 1.f, d_out,
                                  B, T, f->config.input_size, f->config.output_size, 1) == 0;
}
static int dense_frag2h_device(Dense f, const float *d_in_h2, float *d_out, int B, int T) {
    if (B <= 0 || T <= 0) return 0;
    ActivationFunction act = f->config.activation;
    int kind = nntk_act_kind(act);
    if (kind == NNTK_ACT_CUSTOM) NNTK_FAIL("dense: custom host-callback activation cannot run on the device");
    const int fused = nntk_act_fusable(act);
    int rc = f->wh2_scale > 0.f ? nntk_shim_dense_frag2h(d_in_h2, f->d_wh2, f->wh2_scale, f->d_bias, fused ? kind : NNTK_ACT_IDENTITY,
                                                         act ? act->relu_a : 1.f, d_out, B, T, f->config.input_size, f->config.output_size, 0) : 1;
    if (rc < 0) return -1;
    if (rc == 1) {
        float *xs = nntk_devbuf_reserve(&f->d_in, (size_t)B * T * f->config.input_size);
        if (!xs || nntk_shim_frag2h_unpack(d_in_h2, xs, B, T, f->config.input_size)) return -1;
        return dense_rows_device(f, xs, d_out, (long)B * T);
    }
    if (!fused) {
        if ((long)act->input_size * act->vector_size != f->config.output_size)
            NNTK_FAIL("dense: softmax input_size * vector_size must equal the dense output_size");
        return nntk_shim_activation(NNTK_ACT_SOFTMAX, 1.f, act->vector_size, d_out, d_out, (long)B * T * f->config.output_size);
    }
    return 0;
}

static int dense_rows_host(Dense f, const float *input, float *output, long rows) {
    if (rows <= 0) return 0;
    if (dense_ensure(f, 1)) return -1;
    size_t n_in = (size_t)rows * f->config.input_size, n_out = (size_t)rows * f->config.output_size;
    float *d_in = nntk_devbuf_reserve(&f->d_in, n_in);
    float *d_out = nntk_devbuf_reserve(&f->d_out, n_out);
    if (!d_in || !d_out) return -1;
    if (nntk_shim_upload(d_in, input, n_in * sizeof(float))) return -1;
    if (dense_rows_device(f, d_in, d_out, rows)) return -1;
    return nntk_shim_download(output, d_out, n_out * sizeof(float));
}

/* dense.c:135-142 (-1 for a training-mode handle, :136-138) */
int DenseApplyInference(Dense filter, const float *input, float *output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("DenseApplyInference: NULL handle");
    if (filter->training) NNTK_FAIL("DenseApplyInference: the handle was created for training");
    return dense_rows_host(filter, input, output, 1);
}

/* ---- training, second slice: dense.c:85-119 (create, gradient block), :144-162 (forward over the
 *      mini-batch keeping x, z, a), :164-185 (DenseCalculateGradient) ---- */
Dense DenseCreateForTraining(DenseConfig config, DenseTrainingConfig training_config) {
    Dense f = DenseCreateForInference(config);
    if (!f) return NULL;
    f->training = 1;
    f->mini_batch = training_config.mini_batch_size;
    return f;
}

/* one zeroed block d_W | d_b | d_X (weights_private.c:29-36) */
DenseGradient *DenseGradientCreate(DenseConfig config, DenseTrainingConfig training_config) {
    DenseGradient *g = (DenseGradient *)malloc(sizeof(DenseGradient));
    if (!g) return NULL;
    size_t w = (size_t)config.input_size * config.output_size;
    size_t x = (size_t)training_config.mini_batch_size * config.input_size;
    g->d_W = (float *)calloc(w + config.output_size + x + 1, sizeof(float));
    if (!g->d_W) { free(g); return NULL; }
    g->d_b = g->d_W + w;
    g->d_X = g->d_b + config.output_size;
    return g;
}
DenseGradient *DenseGradientCreateFromFilter(Dense dense) {
    if (!dense || !dense->training) return NULL;                      /* dense.c:107-109 */
    DenseTrainingConfig tc;
    tc.mini_batch_size = dense->mini_batch;
    return DenseGradientCreate(dense->config, tc);
}
void DenseGradientDestroy(DenseGradient *gradient) {
    if (!gradient) return;
    free(gradient->d_W);
    free(gradient);
}

static int dense_act_matches(Dense f) {
    ActivationFunction act = f->config.activation;
    if (!act) return 1;
    if (act->kind == NNTK_ACT_CUSTOM) return 0;
    long n = act->kind == NNTK_ACT_SOFTMAX ? (long)act->input_size * act->vector_size : act->input_size;
    return n == f->config.output_size;
}

/* the forward pass on device buffers: x kept (by pointer when the caller's buffer is a device buffer), z and a cached */
static int dense_train_forward_device(Dense filter, const float *d_x, float *d_out_or_null) {
    const int B = filter->mini_batch, in = filter->config.input_size, out = filter->config.output_size;
    float *d_z = nntk_devbuf_reserve(&filter->d_z, (size_t)B * out);
    float *d_a = nntk_devbuf_reserve(&filter->d_a, (size_t)B * out);
    if (!d_z || !d_a) return -1;
    /* z = x W + b for the whole mini-batch (one GEMM), then a = activation(z) as its own pass: z is needed later */
    if (nntk_shim_conv1d(d_x, filter->d_wp, filter->d_bias, NULL, 0.f, NNTK_ACT_IDENTITY, 1.f, d_z, 1, B, in, out, 1, 1, B, 0))
        return -1;
    ActivationFunction act = filter->config.activation;
    if (act) {
        if (nntk_shim_activation(act->kind, act->relu_a, act->vector_size, d_z, d_a, (long)B * out)) return -1;
    } else if (nntk_shim_copy_d2d(d_a, d_z, (size_t)B * out * sizeof(float))) {
        return -1;
    }
    filter->d_x_cur = d_x;
    if (d_out_or_null && nntk_shim_copy_d2d(d_out_or_null, d_a, (size_t)B * out * sizeof(float))) return -1;
    return 0;
}
static int dense_train_check(Dense filter, const char *who) {
    if (!filter) { nntk_set_error("Dense training call: NULL handle"); return -1; }
    if (!filter->training) { nntk_set_error(who); return -1; }      /* dense.c:145-147 */
    if (!dense_act_matches(filter)) { nntk_set_error("Dense training: the activation must be built-in and sized to the dense output_size"); return -1; }
    return 0;
}

int DenseApplyTrainingBatch(Dense filter, const float *input, float *output) {
    nntk_shim_clear_error();
    if (dense_train_check(filter, "DenseApplyTrainingBatch: the handle was created for inference")) return -1;
    const int B = filter->mini_batch, in = filter->config.input_size, out = filter->config.output_size;
    if (B <= 0) return 0;
    if (dense_ensure(filter, 1)) return -1;
    float *d_x = nntk_devbuf_reserve(&filter->d_x, (size_t)B * in);
    if (!d_x) return -1;
    if (nntk_shim_upload(d_x, input, (size_t)B * in * sizeof(float))) return -1;
    if (dense_train_forward_device(filter, d_x, NULL)) return -1;
    return nntk_shim_download(output, filter->d_a.p, (size_t)B * out * sizeof(float));
}
/* Device-pointer form (additive): d_input [mini_batch, in] and d_output [mini_batch, out] are device buffers; d_input must
 * stay valid and unchanged until the matching DenseCalculateGradientDevice (it is the cached x).  Weight edits are picked up
 * like in the host form (whole-block compare). */
int DenseApplyTrainingBatchDevice(Dense filter, const float *d_input, float *d_output) {
    nntk_shim_clear_error();
    if (dense_train_check(filter, "DenseApplyTrainingBatchDevice: the handle was created for inference")) return -1;
    if (filter->mini_batch <= 0) return 0;
    if (dense_ensure(filter, 1)) return -1;
    return dense_train_forward_device(filter, d_input, d_output);
}

/* gradient on device buffers: d_grad = d_W | d_b (accumulated onto), d_dx overwritten */
static int dense_gradient_device(Dense filter, const float *d_dout, float *d_grad, float *d_dx) {
    const int B = filter->mini_batch, in = filter->config.input_size, out = filter->config.output_size;
    const size_t w = (size_t)in * out;
    float *d_dz = nntk_devbuf_reserve(&filter->d_dz, (size_t)B * out);
    float *d_wraw = nntk_devbuf_reserve(&filter->d_wraw, w);
    if (!d_dz || !d_wraw) return -1;
    if (nntk_shim_upload(d_wraw, filter->weights->W, w * sizeof(float))) return -1;             /* caller layout [in, out] */
    ActivationFunction act = filter->config.activation;
    const float *dz = d_dout;
    if (act) {      /* dz = d_out * activation'(z)  (per sample in the reference; the kernels are elementwise / per vector) */
        int vpc = act->kind == NNTK_ACT_SOFTMAX ? act->input_size : 1;
        if (nntk_shim_activation_grad(act->kind, act->vector_size, vpc, filter->d_z.p, filter->d_a.p, d_dout, d_dz, (long)B * out)) return -1;
        dz = d_dz;
    }
    if ((double)B * in * out < (double)(1 << 27)) {       /* small: the reference's mini-batch order exactly */
        if (nntk_shim_dense_grad(filter->d_x_cur, d_wraw, dz, d_grad, d_grad + w, d_dx, B, in, out)) return -1;
    } else {                                              /* large: the MFMA GEMM (train.c) */
        if (nntk_train_outer_accumulate(filter->d_x_cur, dz, d_grad, d_grad + w, B, in, out, 0)) return -1;
        if (nntk_train_rows_times_rowmat(dz, d_wraw, d_dx, B, in, out)) return -1;
    }
    return 0;
}

/* d_W and d_b are accumulated onto the caller's block in mini-batch order (default_gradient_sum, weights_private.c:43-48),
 * d_X is overwritten.  void in the reference; errors through nntk_last_error(). */
void DenseCalculateGradient(Dense filter, DenseGradient *gradient, float *d_out) {
    nntk_shim_clear_error();
    if (!filter || !gradient || !d_out) { nntk_set_error("DenseCalculateGradient: NULL argument"); return; }
    if (!filter->training || !filter->d_x_cur) { nntk_set_error("DenseCalculateGradient: run DenseApplyTrainingBatch on a training handle first"); return; }
    const int B = filter->mini_batch, in = filter->config.input_size, out = filter->config.output_size;
    const size_t w = (size_t)in * out;
    float *d_dout = nntk_devbuf_reserve(&filter->d_dout, (size_t)B * out);
    float *d_grad = nntk_devbuf_reserve(&filter->d_grad, w + out);
    float *d_dx = nntk_devbuf_reserve(&filter->d_dx, (size_t)B * in);
    if (!d_dout || !d_grad || !d_dx) return;
    if (nntk_shim_upload(d_dout, d_out, (size_t)B * out * sizeof(float))) return;
    if (nntk_shim_upload(d_grad, gradient->d_W, (w + out) * sizeof(float))) return;          /* d_W | d_b are contiguous */
    if (dense_gradient_device(filter, d_dout, d_grad, d_dx)) return;
    if (nntk_shim_download(gradient->d_W, d_grad, (w + out) * sizeof(float))) return;
    nntk_shim_download(gradient->d_X, d_dx, (size_t)B * in * sizeof(float));
}
/* Device-pointer form (additive): d_grad_Wb = device [in * out + out] floats (d_W | d_b), ACCUMULATED onto like the host form;
 * d_dX = device [mini_batch, in], overwritten; d_dout = device [mini_batch, out].  0 ok, -1 error. */
int DenseCalculateGradientDevice(Dense filter, float *d_grad_Wb, float *d_dX, const float *d_dout) {
    nntk_shim_clear_error();
    if (!filter || !d_grad_Wb || !d_dX || !d_dout) NNTK_FAIL("DenseCalculateGradientDevice: NULL argument");
    if (!filter->training || !filter->d_x_cur) NNTK_FAIL("DenseCalculateGradientDevice: run DenseApplyTrainingBatch[Device] on a training handle first");
    return dense_gradient_device(filter, d_dout, d_grad_Wb, d_dX);
}

/* ========================= TimeDistributedDense =========================== */

struct TimeDistributedDenseStruct {
    TimeDistributedDenseConfig config;
    Dense dense;
};

/* time_distributed_dense.c:18-23 */
TimeDistributedDenseConfig TimeDistributedDenseConfigCreate(int ts, DenseConfig dense) {
    TimeDistributedDenseConfig c;
    memset(&c, 0, sizeof(c));
    c.dense = dense;
    c.ts = ts;
    return c;
}

/* time_distributed_dense.c:31-35 */
TimeDistributedDense TimeDistributedDenseCreateForInference(TimeDistributedDenseConfig config) {
    TimeDistributedDense f = (TimeDistributedDense)calloc(1, sizeof(struct TimeDistributedDenseStruct));
    if (!f) return NULL;
    f->config = config;
    f->dense = DenseCreateForInference(config.dense);
    if (!f->dense) { free(f); return NULL; }
    return f;
}
/* training (time_distributed_dense.c:38-67): a Dense trained on mini_batch * ts rows */
TimeDistributedDense TimeDistributedDenseCreateForTraining(TimeDistributedDenseConfig config,
                                                           TimeDistributedDenseTrainingConfig training_config) {
    TimeDistributedDense f = (TimeDistributedDense)calloc(1, sizeof(struct TimeDistributedDenseStruct));
    if (!f) return NULL;
    f->config = config;
    DenseTrainingConfig dc;
    dc.mini_batch_size = training_config.mini_batch_size * config.ts;
    f->dense = DenseCreateForTraining(config.dense, dc);
    if (!f->dense) { free(f); return NULL; }
    return f;
}
DenseGradient *TimeDistributedDenseGradientCreate(TimeDistributedDense filter) {
    return filter ? DenseGradientCreateFromFilter(filter->dense) : NULL;
}
int TimeDistributedDenseApplyTrainingBatch(TimeDistributedDense filter, const float *input, float *output) {
    if (!filter) { nntk_shim_clear_error(); NNTK_FAIL("TimeDistributedDenseApplyTrainingBatch: NULL handle"); }
    return DenseApplyTrainingBatch(filter->dense, input, output);
}
void TimeDistributedDenseCalculateGradient(TimeDistributedDense filter, DenseGradient *gradient, float *d_out) {
    if (!filter) { nntk_shim_clear_error(); nntk_set_error("TimeDistributedDenseCalculateGradient: NULL handle"); return; }
    DenseCalculateGradient(filter->dense, gradient, d_out);
}
int TimeDistributedDenseApplyTrainingBatchDevice(TimeDistributedDense filter, const float *d_input, float *d_output) {
    if (!filter) { nntk_shim_clear_error(); NNTK_FAIL("TimeDistributedDenseApplyTrainingBatchDevice: NULL handle"); }
    return DenseApplyTrainingBatchDevice(filter->dense, d_input, d_output);
}
int TimeDistributedDenseCalculateGradientDevice(TimeDistributedDense filter, float *d_grad_Wb, float *d_dX, const float *d_dout) {
    if (!filter) { nntk_shim_clear_error(); NNTK_FAIL("TimeDistributedDenseCalculateGradientDevice: NULL handle"); }
    return DenseCalculateGradientDevice(filter->dense, d_grad_Wb, d_dX, d_dout);
}
DenseWeights *TimeDistributedDenseGetWeights(TimeDistributedDense filter) { return DenseGetWeights(filter->dense); }
void TimeDistributedDenseDestroy(TimeDistributedDense filter) {
    if (!filter) return;
    DenseDestroy(filter->dense);
    free(filter);
}
int TimeDistributedDenseSyncWeights(TimeDistributedDense filter) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("TimeDistributedDenseSyncWeights: NULL handle");
    return DenseSyncWeights(filter->dense);
}

int TimeDistributedDenseBroadcastWeights(TimeDistributedDense filter, int root) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("TimeDistributedDenseBroadcastWeights: NULL handle");
    return DenseBroadcastWeights(filter->dense, root);
}

/* time_distributed_dense.c:52-58 (always returns 0 there; here -1 on device errors) */
int TimeDistributedDenseApplyInference(TimeDistributedDense filter, const float *input, float *output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("TimeDistributedDenseApplyInference: NULL handle");
    if (filter->dense->training) NNTK_FAIL("TimeDistributedDenseApplyInference: the handle was created for training");
    return dense_rows_host(filter->dense, input, output, filter->config.ts);
}
int TimeDistributedDenseApplyInferenceBatch(TimeDistributedDense filter, const float *input, float *output, int batch) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("TimeDistributedDenseApplyInferenceBatch: NULL handle");
    return dense_rows_host(filter->dense, input, output, (long)batch * filter->config.ts);
}
/* additive: the input as a frag3 tensor [batch][ts][input_size] (nntk_frag3_pack_device, or a recurrent layer's frag3 output) */
int TimeDistributedDenseApplyDeviceFrag3(TimeDistributedDense filter, const float *d_input_frag3, float *d_output, int batch) {
    nntk_shim_clear_error();
    if (!filter || !d_input_frag3 || !d_output) NNTK_FAIL("TimeDistributedDenseApplyDeviceFrag3: NULL argument");
    if (dense_ensure(filter->dense, 0)) return -1;
    return dense_frag3_device(filter->dense, d_input_frag3, d_output, batch, filter->config.ts);
}
/* additive: the input as a FRAG2H tensor [batch][ts][input_size] (LSTMApplyDeviceFrag2h, nntk_frag2h_pack_device) */
int TimeDistributedDenseApplyDeviceFrag2h(TimeDistributedDense filter, const float *d_input_frag2h, float *d_output, int batch) {
    nntk_shim_clear_error();
    if (!filter || !d_input_frag2h || !d_output) NNTK_FAIL("TimeDistributedDenseApplyDeviceFrag2h: NULL argument");
    if (dense_ensure(filter->dense, 0)) return -1;
    return dense_frag2h_device(filter->dense, d_input_frag2h, d_output, batch, filter->config.ts);
}
/* additive: LSTM (return_sequences) -> TimeDistributedDense without an f32 tensor in between.
 * Default (option dense_f16x2 = -1 / 1, an LSTM with the standard activations, finite dense weights, a shape dense_frag3_kernel takes): the
 * LSTM hands h over as a FRAG2H tensor -- two f16 images of h * 2^15, |h| < 1: the hand-off buffer of the HF kernel (H > 256, whose recurrence
 * runs on that form too), else written by the output wave of the six-product kernel -- and the dense GEMM sums three products per k step
 * on it and on W's two f16 images: half the MFMA work of the frag3 route, operands rounded to 2^-23 relative (at worst one f32 ulp), measured
 * error against f64 below the frag3 route's (frag3.hip; tests/test_gpu_frag2h.py).  Not bit-identical to the two separate f32 calls.
 * Otherwise / option dense_f16x2 = 0: the LSTM's hand-off buffer is its output in frag3 form and the dense GEMM reads it as its A operand:
 * results = LSTMApplyDevice then TimeDistributedDenseApplyDevice, bit for bit (the frag3 images are the f32 values exactly, and the GEMM
 * sums the same products in the same order). */
int LSTMTimeDistributedDenseApplyDevice(LSTM lstm, TimeDistributedDense tdd, const float *d_input, float *d_output, int batch) {
    nntk_shim_clear_error();
    if (!lstm || !tdd || !d_input || !d_output) NNTK_FAIL("LSTMTimeDistributedDenseApplyDevice: NULL argument");
    int T, in, H, seq;
    nntk_lstm_dims(lstm, &T, &in, &H, &seq);
    if (!seq || tdd->config.ts != T || tdd->dense->config.input_size != H)
        NNTK_FAIL("LSTMTimeDistributedDenseApplyDevice: the LSTM must return sequences and feed the dense layer (ts = timesteps, input_size = H)");
    if (batch <= 0) return 0;
    if (dense_ensure(tdd->dense, 0)) return -1;
    if (dense_frag2h_takes(tdd->dense, d_output, batch, T)) {
        float *d_h2 = nntk_lstm_frag2h_scratch(lstm, batch);
        if (!d_h2) return -1;
        int rc = nntk_lstm_apply_device_h2(lstm, d_input, NULL, d_h2, batch);
        if (rc < 0) return -1;
        if (rc == 0) return dense_frag2h_device(tdd->dense, d_h2, d_output, batch, T);
    }
    float *d_h3 = nntk_lstm_frag3_scratch(lstm, batch);
    if (!d_h3) return -1;
    if (LSTMApplyDeviceFrag3(lstm, d_input, NULL, NULL, d_h3, batch)) return -1;
    if (dense_ensure(tdd->dense, 0)) return -1;
    return dense_frag3_device(tdd->dense, d_h3, d_output, batch, T);
}
int TimeDistributedDenseApplyDevice(TimeDistributedDense filter, const float *d_input, float *d_output, int batch) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("TimeDistributedDenseApplyDevice: NULL handle");
    if (dense_ensure(filter->dense, 0)) return -1;
    return dense_rows_device(filter->dense, d_input, d_output, (long)batch * filter->config.ts);
}
