/*
 * dense.c -- Dense and TimeDistributedDense host layer.  Reference:
 * layers/dense.c:24-28 (weight block W [in,out] | b [out]), :122-142 (z = x*W + b,
 * a = activation(z)), layers/time_distributed_dense.c:31-58 (ts Dense
 * applications sharing weights).  On the device every row (timestep, batch
 * element) is one row of a single MFMA GEMM with the bias and the activation
 * fused; softmax (which spans a row) runs as a second wave-per-row kernel.
 */
#include <stdlib.h>
#include <string.h>
#include "nntk_internal.h"

struct DenseStruct {
    DenseConfig config;
    DenseWeights *weights;
    nntk_wblock wb;
    float *d_wp, *d_bias;
    nntk_devbuf d_in, d_out;
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
    nntk_shim_free(filter->d_bias);
    nntk_devbuf_free(&filter->d_in);
    nntk_devbuf_free(&filter->d_out);
    nntk_wblock_free(&filter->wb);
    free(filter->weights);
    free(filter);
}

static int dense_upload(Dense f) {
    if (nntk_upload_gemm_weights(&f->d_wp, f->weights->W, f->config.input_size, f->config.output_size)) return -1;
    if (nntk_upload_floats(&f->d_bias, f->weights->b, (size_t)f->config.output_size)) return -1;
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

/* dense.c:135-142 */
int DenseApplyInference(Dense filter, const float *input, float *output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("DenseApplyInference: NULL handle");
    return dense_rows_host(filter, input, output, 1);
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
    return dense_rows_host(filter->dense, input, output, filter->config.ts);
}
int TimeDistributedDenseApplyInferenceBatch(TimeDistributedDense filter, const float *input, float *output, int batch) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("TimeDistributedDenseApplyInferenceBatch: NULL handle");
    return dense_rows_host(filter->dense, input, output, (long)batch * filter->config.ts);
}
int TimeDistributedDenseApplyDevice(TimeDistributedDense filter, const float *d_input, float *d_output, int batch) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("TimeDistributedDenseApplyDevice: NULL handle");
    if (dense_ensure(filter->dense, 0)) return -1;
    return dense_rows_device(filter->dense, d_input, d_output, (long)batch * filter->config.ts);
}
