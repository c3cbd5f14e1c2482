/*
 * conv_1d.c + batch_norm host layer -- reference: layers/conv_1d.{h,c} (forward,
 * :77-165) and layers/batch_norm.{h,c} (inference, :61-189).  Handles own the
 * caller-visible weight block; the device copy is packed for the implicit-GEMM
 * kernel (csrc/hip/conv1d.hip) on first use.
 */
#include <stdlib.h>
#include <string.h>
#include "nntk_internal.h"

/* =============================== Conv1d =================================== */

struct Conv1dStruct {
    Conv1dConfig config;
    ConvWeights *weights;
    nntk_wblock wb;
    float *d_wp, *d_bias;
    float *d_wpf;               /* the same weights packed with a flat K axis (conv1d_flatk.hip), when the shape takes that kernel */
    int flatk_ok;
    nntk_devbuf d_in, d_out;
    /* training mode (conv_1d.c:104-108): mini-batch size, the last forward pass's input kept on the device for the
     * gradient, and the gradient scratch */
    int training, mini_batch;
    nntk_devbuf d_cache, d_dout, d_grad, d_wraw, d_scratch, d_pad, d_wpk;
    const float *d_x_cur;     /* the last training forward's input: d_cache (host-pointer call) or the caller's device tensor */
};

/* conv_1d.c:77-87 */
Conv1dConfig Conv1dConfigCreate(int input_feature_channels, int output_feature_channels, int kernel_size,
                                int stride, int inputSize) {
    Conv1dConfig c;
    c.input_feature_channels = input_feature_channels;
    c.output_feature_channels = output_feature_channels;
    c.kernel_size = kernel_size;
    c.stride = stride;
    c.input_size = inputSize;
    c.output_size = (inputSize - (kernel_size - stride)) / stride;
    return c;
}

/* conv_1d.c:89-102 with the weight block of weights_private.c:16-21: W then b */
Conv1d Conv1dCreateForInference(Conv1dConfig config) {
    nntk_shim_clear_error();
    Conv1d f = (Conv1d)calloc(1, sizeof(struct Conv1dStruct));
    if (!f) return NULL;
    f->config = config;
    size_t w = (size_t)config.kernel_size * config.input_feature_channels * config.output_feature_channels;
    if (nntk_wblock_init(&f->wb, w + config.output_feature_channels)) { free(f); return NULL; }
    f->weights = (ConvWeights *)malloc(sizeof(ConvWeights));
    f->weights->W = f->wb.host;
    f->weights->b = f->wb.host + w;
    return f;
}

/* conv_1d.c:104-108 */
Conv1d Conv1dCreateForTraining(Conv1dConfig config, ConvTrainingConfig training_config) {
    Conv1d f = Conv1dCreateForInference(config);
    if (!f) return NULL;
    f->training = 1;
    f->mini_batch = training_config.mini_batch_size;
    return f;
}

ConvWeights *Conv1dGetWeights(Conv1d filter) { return filter->weights; }

void Conv1dDestroy(Conv1d filter) {
    if (!filter) return;
    nntk_shim_synchronize();
    nntk_shim_free(filter->d_wp);
    nntk_shim_free(filter->d_wpf);
    nntk_shim_free(filter->d_bias);
    nntk_devbuf_free(&filter->d_in);
    nntk_devbuf_free(&filter->d_out);
    nntk_devbuf_free(&filter->d_cache); nntk_devbuf_free(&filter->d_dout); nntk_devbuf_free(&filter->d_grad);
    nntk_devbuf_free(&filter->d_wraw); nntk_devbuf_free(&filter->d_scratch);
    nntk_devbuf_free(&filter->d_pad); nntk_devbuf_free(&filter->d_wpk);
    nntk_wblock_free(&filter->wb);
    free(filter->weights);
    free(filter);
}

/* W [Cout][Cin][k] (conv_1d.c:129-139 indexing) -> Wp [Cout_p][kk*Cin_p + i] (K-contiguous, zero padded) */
static int conv_upload(Conv1d f) {
    const Conv1dConfig *c = &f->config;
    int Cin = c->input_feature_channels, Cout = c->output_feature_channels, k = c->kernel_size;
    int Cin_p, Cout_p;
    nntk_shim_conv_pack_sizes(Cin, Cout, k, &Cin_p, &Cout_p);
    size_t n = (size_t)k * Cin_p * Cout_p;
    float *tmp = (float *)calloc(n, sizeof(float));
    if (!tmp) NNTK_FAIL("out of host memory while packing conv weights");
    const float *W = f->weights->W;
    for (int o = 0; o < Cout; ++o)
        for (int i = 0; i < Cin; ++i)
            for (int kk = 0; kk < k; ++kk)
                tmp[((size_t)o * k + kk) * Cin_p + i] = W[((size_t)o * Cin + i) * k + kk];
    int rc = nntk_upload_packed_weights(&f->d_wp, tmp, Cout_p, k * Cin_p);
    free(tmp);
    if (rc) return rc;
    /* flat K axis (K = tap * Cin + channel, padded once to a multiple of 16) for channel counts that are multiples of 8 but not of
     * 16: no per-tap channel padding, 13 % fewer MFMAs at Conv1d(40 -> 128, k = 5).  A weight block the bf16 split cannot hold keeps
     * the exact-f32 kernel, which only exists in the chunked form. */
    f->flatk_ok = 0;
    if (c->stride == 1 && Cin % 8 == 0 && Cin % 16 != 0 && k >= 2 && Cout_p % 64 == 0) {
        int odd = 0;
        for (size_t i = 0; i < (size_t)Cout * Cin * k && !odd; ++i) {
            union { float f; unsigned u; } v = { W[i] };
            unsigned a = v.u & 0x7fffffffu;
            odd = a > 0x7f7f0000u || (a - 1u < 0x007fffffu);
        }
        if (!odd) {
            const int Kf_p = (k * Cin + 15) & ~15;
            float *t2 = (float *)calloc((size_t)Cout_p * Kf_p, sizeof(float));
            if (!t2) NNTK_FAIL("out of host memory while packing conv weights");
            for (int o = 0; o < Cout; ++o)
                for (int i = 0; i < Cin; ++i)
                    for (int kk = 0; kk < k; ++kk)
                        t2[(size_t)o * Kf_p + (size_t)kk * Cin + i] = W[((size_t)o * Cin + i) * k + kk];
            rc = nntk_upload_packed_weights(&f->d_wpf, t2, Cout_p, Kf_p);
            free(t2);
            if (rc) return rc;
            f->flatk_ok = 1;
        }
    }
    if (nntk_upload_floats(&f->d_bias, f->weights->b, (size_t)Cout)) return -1;
    nntk_wblock_mark_uploaded(&f->wb);
    return 0;
}

static int conv_ensure(Conv1d f, int check_edits) {
    if (nntk_wblock_dirty(&f->wb, check_edits)) return conv_upload(f);
    return 0;
}

int Conv1dSyncWeights(Conv1d filter) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("Conv1dSyncWeights: NULL handle");
    nntk_shim_synchronize();
    return conv_upload(filter);
}

/* multi-GPU: the root's weight block replaces every rank's (one RCCL broadcast over xGMI), then re-upload */
int Conv1dBroadcastWeights(Conv1d filter, int root) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("Conv1dBroadcastWeights: NULL handle");
    if (nntk_shim_dist_broadcast_host(filter->wb.host, filter->wb.n, root)) return -1;
    return conv_upload(filter);
}

static int conv_launch(Conv1d f, const float *d_bn, float eps, int act_kind, float relu_a,
                       const float *d_in, float *d_out, int batch) {
    const Conv1dConfig *c = &f->config;
    if (f->flatk_ok) {
        int rc = nntk_shim_conv1d_flatk(d_in, f->d_wpf, f->d_bias, d_bn, eps, act_kind, relu_a, d_out, batch, c->input_size,
                                        c->input_feature_channels, c->output_feature_channels, c->kernel_size, c->output_size);
        if (rc <= 0) return rc;
    }
    return nntk_shim_conv1d(d_in, f->d_wp, f->d_bias, d_bn, eps, act_kind, relu_a, d_out, batch, c->input_size,
                            c->input_feature_channels, c->output_feature_channels, c->kernel_size, c->stride,
                            c->output_size, 0);
}

int Conv1dApplyDevice(Conv1d filter, const float *d_input, float *d_output, int batch) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("Conv1dApplyDevice: NULL handle");
    if (conv_ensure(filter, 0)) return -1;
    return conv_launch(filter, NULL, 0.f, NNTK_ACT_IDENTITY, 1.f, d_input, d_output, batch);
}

int Conv1dBatchNormActivationApplyDevice(Conv1d filter, BatchNorm bn, ActivationFunction act,
                                         const float *d_input, float *d_output, int batch) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("Conv1dBatchNormActivationApplyDevice: NULL conv handle");
    if (conv_ensure(filter, 0)) return -1;
    const float *d_bn = NULL;
    float eps = 0.f;
    if (bn) {
        if (nntk_batch_norm_channels(bn) != filter->config.output_feature_channels)
            NNTK_FAIL("fused conv+bn: BatchNorm feature_channels must equal conv output channels");
        d_bn = nntk_batch_norm_device_block(bn, 0);
        if (!d_bn) return -1;
        eps = nntk_batch_norm_epsilon(bn);
    }
    if (!nntk_act_fusable(act)) NNTK_FAIL("fused conv+bn+act: activation must be identity/sigmoid/tanh/relu");
    int kind = act ? act->kind : NNTK_ACT_IDENTITY;
    float a = act ? act->relu_a : 1.0f;
    return conv_launch(filter, d_bn, eps, kind, a, d_input, d_output, batch);
}

/* additive: the fused layer's output as a frag3 tensor [batch][Tout][Cout] (include/nntoolkitcore_hip.h "frag3 tensors") -- what a
 * register-resident GRU / LSTM layer or TimeDistributedDenseApplyDeviceFrag3 reads -- written by the conv kernel's own epilogue
 * (conv_1d.c:122-147 feeding lstm.c:201 without the f32 tensor and the pack pass in between).  Bit-identical to
 * Conv1dBatchNormActivationApplyDevice -> nntk_frag3_pack_device; shapes the frag3 epilogue does not take (stride != 1, k > 9, the
 * flat-K and exact-f32 kernels) run exactly those two calls through scratch in the handle: valid for every layer. */
int Conv1dBatchNormActivationApplyDeviceFrag3(Conv1d filter, BatchNorm bn, ActivationFunction act,
                                              const float *d_input, float *d_output_frag3, int batch) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("Conv1dBatchNormActivationApplyDeviceFrag3: NULL conv handle");
    if (!d_input || !d_output_frag3) NNTK_FAIL("Conv1dBatchNormActivationApplyDeviceFrag3: NULL tensor");
    if (batch <= 0) return 0;
    if (conv_ensure(filter, 0)) return -1;
    const Conv1dConfig *c = &filter->config;
    const float *d_bn = NULL;
    float eps = 0.f;
    if (bn) {
        if (nntk_batch_norm_channels(bn) != c->output_feature_channels)
            NNTK_FAIL("fused conv+bn: BatchNorm feature_channels must equal conv output channels");
        d_bn = nntk_batch_norm_device_block(bn, 0);
        if (!d_bn) return -1;
        eps = nntk_batch_norm_epsilon(bn);
    }
    if (!nntk_act_fusable(act)) NNTK_FAIL("fused conv+bn+act: activation must be identity/sigmoid/tanh/relu");
    int kind = act ? act->kind : NNTK_ACT_IDENTITY;
    float a = act ? act->relu_a : 1.0f;
    if (!filter->flatk_ok) {        /* (the flat-K kernel sums in another order: a layer that takes it keeps it, through scratch) */
        int rc = nntk_shim_conv1d_frag3(d_input, filter->d_wp, filter->d_bias, d_bn, eps, kind, a, d_output_frag3, batch, c->input_size,
                                        c->input_feature_channels, c->output_feature_channels, c->kernel_size, c->stride, c->output_size);
        if (rc <= 0) return rc;
    }
    float *d_out = nntk_devbuf_reserve(&filter->d_out, (size_t)batch * c->output_size * c->output_feature_channels);
    if (!d_out) return -1;
    if (conv_launch(filter, d_bn, eps, kind, a, d_input, d_out, batch)) return -1;
    return nntk_shim_frag3_pack(d_out, d_output_frag3, batch, c->output_size, c->output_feature_channels);
}

int Conv1dApplyInferenceBatch(Conv1d filter, const float *input, float *output, int batch) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("Conv1dApplyInferenceBatch: NULL handle");
    if (batch <= 0) return 0;
    const Conv1dConfig *c = &filter->config;
    if (conv_ensure(filter, 1)) return -1;
    size_t n_in = (size_t)batch * c->input_size * c->input_feature_channels;
    size_t n_out = (size_t)batch * c->output_size * c->output_feature_channels;
    float *d_in = nntk_devbuf_reserve(&filter->d_in, n_in);
    float *d_out = nntk_devbuf_reserve(&filter->d_out, n_out);
    if (!d_in || !d_out) return -1;
    if (nntk_shim_upload(d_in, input, n_in * sizeof(float))) return -1;
    if (conv_launch(filter, NULL, 0.f, NNTK_ACT_IDENTITY, 1.f, d_in, d_out, batch)) return -1;
    return nntk_shim_download(output, d_out, n_out * sizeof(float));
}

/* conv_1d.c:149-155: one [T, Cin] sequence -> [Tout, Cout]; -1 for a training-mode handle (:150-152, pinned by the real
 * reference: tests/golden/ref_probe.json "wrong_mode_apply_inference") */
int Conv1dApplyInference(Conv1d filter, const float *input, float *output) {
    nntk_shim_clear_error();
    if (filter && filter->training) NNTK_FAIL("Conv1dApplyInference: the handle was created for training");
    return Conv1dApplyInferenceBatch(filter, input, output, 1);
}

/* ---- training, first slice: forward over the mini-batch with the input kept for the gradient
 *      (conv_1d.c:167-183) and Conv1dCalculateGradient (conv_1d.c:185-245) ---- */
int Conv1dApplyTrainingBatch(Conv1d filter, const float *input, float *output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("Conv1dApplyTrainingBatch: NULL handle");
    if (!filter->training) NNTK_FAIL("Conv1dApplyTrainingBatch: the handle was created for inference");     /* conv_1d.c:168-170 */
    const Conv1dConfig *c = &filter->config;
    const int B = filter->mini_batch;
    if (B <= 0) return 0;
    if (conv_ensure(filter, 1)) return -1;
    size_t n_in = (size_t)B * c->input_size * c->input_feature_channels;
    size_t n_out = (size_t)B * c->output_size * c->output_feature_channels;
    float *d_in = nntk_devbuf_reserve(&filter->d_cache, n_in);
    float *d_out = nntk_devbuf_reserve(&filter->d_out, n_out);
    if (!d_in || !d_out) return -1;
    if (nntk_shim_upload(d_in, input, n_in * sizeof(float))) return -1;
    if (conv_launch(filter, NULL, 0.f, NNTK_ACT_IDENTITY, 1.f, d_in, d_out, B)) return -1;
    filter->d_x_cur = d_in;
    return nntk_shim_download(output, d_out, n_out * sizeof(float));
}
/* device-pointer form: d_input [B][T][Cin] must stay valid until Conv1dCalculateGradientDevice; d_output [B][Tout][Cout] */
int Conv1dApplyTrainingBatchDevice(Conv1d filter, const float *d_input, float *d_output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("Conv1dApplyTrainingBatchDevice: NULL handle");
    if (!filter->training) NNTK_FAIL("Conv1dApplyTrainingBatchDevice: the handle was created for inference");
    if (filter->mini_batch <= 0) return 0;
    if (!d_input || !d_output) NNTK_FAIL("Conv1dApplyTrainingBatchDevice: NULL argument");
    if (conv_ensure(filter, 1)) return -1;
    if (conv_launch(filter, NULL, 0.f, NNTK_ACT_IDENTITY, 1.f, d_input, d_output, filter->mini_batch)) return -1;
    filter->d_x_cur = d_input;
    return 0;
}

/* conv_1d.c:157-161 with weights_private.c:29-36: ONE zeroed block d_W | d_b | d_X */
ConvGradient *Conv1dCreateGradient(Conv1dConfig config, ConvTrainingConfig training_config) {
    ConvGradient *g = (ConvGradient *)malloc(sizeof(ConvGradient));
    if (!g) return NULL;
    size_t w = (size_t)config.kernel_size * config.input_feature_channels * config.output_feature_channels;
    size_t x = (size_t)training_config.mini_batch_size * config.input_size * config.input_feature_channels;
    g->d_W = (float *)calloc(w + config.output_feature_channels + x + 1, sizeof(float));
    if (!g->d_W) { free(g); return NULL; }
    g->d_b = g->d_W + w;
    g->d_X = g->d_b + config.output_feature_channels;
    return g;
}
void ConvGradientDestroy(ConvGradient *gradient) {
    if (!gradient) return;
    free(gradient->d_W);
    free(gradient);
}

/* d_W and d_b are ADDED to the gradient block (as default_gradient_sum does, weights_private.c:50-55), d_X is overwritten
 * (conv_1d.c:242).  void in the reference; errors through nntk_last_error(). */
/* d_dW_db [Cout][Cin][k] | [Cout] is OVERWRITTEN (the callers add it onto their blocks), d_dX [B][T][Cin] too */
static int conv_gradient_dev(Conv1d filter, const float *d_dout, float *d_dW_db, float *d_dX) {
    const Conv1dConfig *c = &filter->config;
    const int B = filter->mini_batch, Cin = c->input_feature_channels, Cout = c->output_feature_channels, k = c->kernel_size;
    const size_t w = (size_t)k * Cin * Cout;
    float *d_wraw = nntk_devbuf_reserve(&filter->d_wraw, w);
    float *d_scr = nntk_devbuf_reserve(&filter->d_scratch, nntk_shim_conv1d_grad_scratch_floats(Cin, Cout, k));
    if (!d_wraw || !d_scr) return -1;
    if (nntk_shim_upload(d_wraw, filter->weights->W, w * sizeof(float))) return -1;       /* caller layout [Cout][Cin][k] */
    const long rows = (long)B * c->output_size;
    if (c->stride == 1 && c->output_size > 0 && (double)rows * Cout * k * Cin >= (double)(1 << 27) && Cin >= 32 && Cout >= 16) {
        /* large, stride 1: d_X on the MFMA forward kernel (train.hip); d_W, d_b on the row-sliced MFMA product or the sliced dots */
        const int T = c->input_size, Tout = c->output_size;
        float *d_pad = nntk_devbuf_reserve(&filter->d_pad, (size_t)B * (Tout + 2 * (k - 1)) * Cout);
        float *d_wpk = nntk_devbuf_reserve(&filter->d_wpk, nntk_shim_conv_dx_pack_floats(Cin, Cout, k));
        if (!d_pad || !d_wpk) return -1;
        if (nntk_shim_conv1d_grad(filter->d_x_cur, d_wraw, d_dout, d_dW_db, d_dW_db + w, NULL, d_scr, B, T, Cin, Cout, k, 1, Tout)) return -1;
        return nntk_shim_conv_dx_mfma(d_dout, d_wraw, d_dX, d_pad, d_wpk, B, T, Cin, Cout, k, Tout);
    }
    return nntk_shim_conv1d_grad(filter->d_x_cur, d_wraw, d_dout, d_dW_db, d_dW_db + w, d_dX, d_scr,
                                 B, c->input_size, Cin, Cout, k, c->stride, c->output_size);
}
void Conv1dCalculateGradient(Conv1d filter, ConvGradient *gradient, const float *d_out) {
    nntk_shim_clear_error();
    if (!filter || !gradient) { nntk_set_error("Conv1dCalculateGradient: NULL argument"); return; }
    if (!filter->training || !filter->d_x_cur) { nntk_set_error("Conv1dCalculateGradient: run Conv1dApplyTrainingBatch on a training handle first"); return; }
    const Conv1dConfig *c = &filter->config;
    const int B = filter->mini_batch, Cin = c->input_feature_channels, Cout = c->output_feature_channels, k = c->kernel_size;
    const size_t w = (size_t)k * Cin * Cout, n_x = (size_t)B * c->input_size * Cin, n_do = (size_t)B * c->output_size * Cout;
    float *d_dout = nntk_devbuf_reserve(&filter->d_dout, n_do);
    float *d_grad = nntk_devbuf_reserve(&filter->d_grad, w + Cout + n_x);
    if (!d_dout || !d_grad) return;
    if (nntk_shim_upload(d_dout, d_out, n_do * sizeof(float))) return;
    if (conv_gradient_dev(filter, d_dout, d_grad, d_grad + w + Cout)) return;
    float *tmp = (float *)malloc((w + Cout) * sizeof(float));
    if (!tmp) { nntk_set_error("out of host memory"); return; }
    if (nntk_shim_download(tmp, d_grad, (w + Cout) * sizeof(float)) == 0 &&
        nntk_shim_download(gradient->d_X, d_grad + w + Cout, n_x * sizeof(float)) == 0) {
        for (size_t i = 0; i < w; ++i) gradient->d_W[i] += tmp[i];
        for (int i = 0; i < Cout; ++i) gradient->d_b[i] += tmp[w + i];
    }
    free(tmp);
}
/* device-pointer form: d_grad_Wb = W [Cout][Cin][k] | b [Cout] (the gradient block's layout) is ADDED to, d_dX [B][T][Cin] overwritten */
int Conv1dCalculateGradientDevice(Conv1d filter, float *d_grad_Wb, float *d_dX, const float *d_dout) {
    nntk_shim_clear_error();
    if (!filter || !d_grad_Wb || !d_dX || !d_dout) NNTK_FAIL("Conv1dCalculateGradientDevice: NULL argument");
    if (!filter->training || !filter->d_x_cur) NNTK_FAIL("Conv1dCalculateGradientDevice: run Conv1dApplyTrainingBatch[Device] on a training handle first");
    const Conv1dConfig *c = &filter->config;
    const size_t w = (size_t)c->kernel_size * c->input_feature_channels * c->output_feature_channels;
    float *d_grad = nntk_devbuf_reserve(&filter->d_grad, w + c->output_feature_channels);
    if (!d_grad) return -1;
    if (conv_gradient_dev(filter, d_dout, d_grad, d_dX)) return -1;
    return nntk_shim_add_into(d_grad_Wb, d_grad, (long)(w + c->output_feature_channels));
}

/* ============================== BatchNorm ================================= */

struct BatchNormFilterStruct {
    BatchNormConfig config;
    BatchNormWeights *weights;
    nntk_wblock wb;           /* gamma | beta | moving_mean | moving_variance (batch_norm.c:79-84) */
    float *d_block;
    nntk_devbuf d_io;
    /* training (batch_norm.c:20-64): the mini-batch input is kept; x_mu, x_norm, ... are recomputed from it */
    int training, mini_batch, have_batch;
    float momentum;
    nntk_devbuf d_x, d_dout, d_stats, d_partial, d_res;
    const float *d_x_cur;       /* input of the last training forward: d_x.p (host form) or the caller's device buffer */
};

/* batch_norm.c:65-71 */
BatchNormConfig BatchNormConfigCreate(int feature_channels, float epsilon, int count) {
    BatchNormConfig c;
    c.feature_channels = feature_channels;
    c.epsilon = epsilon;
    c.count = count;
    return c;
}

/* batch_norm.c:73-86: all four vectors zero-initialised, gamma included */
BatchNorm BatchNormCreateForInference(BatchNormConfig config) {
    nntk_shim_clear_error();
    BatchNorm f = (BatchNorm)calloc(1, sizeof(struct BatchNormFilterStruct));
    if (!f) return NULL;
    f->config = config;
    int C = config.feature_channels;
    if (nntk_wblock_init(&f->wb, (size_t)4 * C)) { free(f); return NULL; }
    f->weights = (BatchNormWeights *)malloc(sizeof(BatchNormWeights));
    f->weights->gamma = f->wb.host;
    f->weights->beta = f->wb.host + C;
    f->weights->moving_mean = f->wb.host + 2 * C;
    f->weights->moving_variance = f->wb.host + 3 * C;
    return f;
}

BatchNormWeights *BatchNormGetWeights(BatchNorm filter) { return filter->weights; }

void BatchNormDestroy(BatchNorm filter) {
    if (!filter) return;
    nntk_shim_synchronize();
    nntk_shim_free(filter->d_block);
    nntk_devbuf_free(&filter->d_io);
    nntk_devbuf_free(&filter->d_x); nntk_devbuf_free(&filter->d_dout); nntk_devbuf_free(&filter->d_stats);
    nntk_devbuf_free(&filter->d_partial); nntk_devbuf_free(&filter->d_res);
    nntk_wblock_free(&filter->wb);
    free(filter->weights);
    free(filter);
}

static int bn_upload(BatchNorm f) {
    /* device block = the four vectors + sd | 1/sd derived on the device (the fused conv epilogue reads those) */
    int C = f->config.feature_channels;
    if (!f->d_block) {
        f->d_block = (float *)nntk_shim_malloc((size_t)6 * C * sizeof(float));
        if (!f->d_block) return -1;
    }
    if (nntk_shim_upload(f->d_block, f->wb.host, f->wb.n * sizeof(float))) return -1;
    if (nntk_shim_bn_derive(f->d_block, f->config.epsilon, C)) return -1;
    nntk_wblock_mark_uploaded(&f->wb);
    return 0;
}

const float *nntk_batch_norm_device_block(BatchNorm bn, int check_edits) {
    if (nntk_wblock_dirty(&bn->wb, check_edits) && bn_upload(bn)) return NULL;
    return bn->d_block;
}
int nntk_batch_norm_channels(BatchNorm bn) { return bn->config.feature_channels; }
float nntk_batch_norm_epsilon(BatchNorm bn) { return bn->config.epsilon; }

int BatchNormSyncWeights(BatchNorm filter) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("BatchNormSyncWeights: NULL handle");
    nntk_shim_synchronize();
    return bn_upload(filter);
}

int BatchNormBroadcastWeights(BatchNorm filter, int root) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("BatchNormBroadcastWeights: NULL handle");
    if (nntk_shim_dist_broadcast_host(filter->wb.host, filter->wb.n, root)) return -1;
    return bn_upload(filter);
}

int BatchNormApplyDevice(BatchNorm filter, const float *d_input, float *d_output, int rows) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("BatchNormApplyDevice: NULL handle");
    const float *blk = nntk_batch_norm_device_block(filter, 0);
    if (!blk) return -1;
    return nntk_shim_batch_norm(d_input, blk, filter->config.epsilon, d_output, rows, filter->config.feature_channels);
}

/* ---- training: batch_norm.c:96-126 (create), :191-262 (forward with batch statistics + moving
 *      statistics update), :264-386 (gradient) ---- */
BatchNormTrainingConfig BatchNormTrainingConfigCreate(float momentum, int mini_batch_size) {
    BatchNormTrainingConfig c;
    c.momentum = momentum;
    c.mini_batch_size = mini_batch_size;
    return c;
}
BatchNorm BatchNormCreateForTraining(BatchNormConfig config, BatchNormTrainingConfig training_config) {
    BatchNorm f = BatchNormCreateForInference(config);
    if (!f) return NULL;
    f->training = 1;
    f->momentum = training_config.momentum;
    f->mini_batch = training_config.mini_batch_size;
    return f;
}
/* ONE zeroed block d_beta | d_gamma | d_x (batch_norm.c:96-104: note the order) */
BatchNormGradient *BatchNormGradientCreate(BatchNormConfig config, BatchNormTrainingConfig training_config) {
    BatchNormGradient *g = (BatchNormGradient *)malloc(sizeof(BatchNormGradient));
    if (!g) return NULL;
    size_t F = (size_t)config.feature_channels;
    size_t n = 2 * F + F * config.count * training_config.mini_batch_size;
    g->d_beta = (float *)calloc(n + 1, sizeof(float));
    if (!g->d_beta) { free(g); return NULL; }
    g->d_gamma = g->d_beta + F;
    g->d_x = g->d_gamma + F;
    return g;
}
void BatchNormGradientDestroy(BatchNormGradient *grad) {
    if (!grad) return;
    free(grad->d_beta);
    free(grad);
}

/* forward on device buffers; the batch statistics stay on the device, the moving statistics are updated on the caller-visible
 * weight block (batch_norm.c:247-257) in the reference's operation order */
static int bn_train_forward_device(BatchNorm filter, const float *d_x, float *d_o) {
    const int F = filter->config.feature_channels;
    const long N = (long)filter->config.count * filter->mini_batch;
    const float *blk = nntk_batch_norm_device_block(filter, 1);
    if (!blk) return -1;
    int rps, slices = nntk_shim_bn_train_slices(N, &rps);
    float *d_stats = nntk_devbuf_reserve(&filter->d_stats, (size_t)8 * F);
    float *d_part = nntk_devbuf_reserve(&filter->d_partial, (size_t)slices * 3 * F);
    if (!d_stats || !d_part) return -1;
    if (nntk_shim_bn_train_forward(d_x, blk, filter->config.epsilon, d_stats, d_part, d_o, N, F)) return -1;
    float *ms = (float *)malloc((size_t)2 * F * sizeof(float));
    if (!ms) NNTK_FAIL("out of host memory");
    if (nntk_shim_download(ms, d_stats, (size_t)2 * F * sizeof(float))) { free(ms); return -1; }
    const float m = filter->momentum;
    const volatile float one_minus = 1 - m;
    for (int f = 0; f < F; ++f) {
        volatile float b = ms[f] * one_minus;
        volatile float mm = filter->weights->moving_mean[f] * m;
        filter->weights->moving_mean[f] = b + mm;
        b = ms[F + f] * one_minus;
        mm = filter->weights->moving_variance[f] * m;
        filter->weights->moving_variance[f] = b + mm;
    }
    free(ms);
    filter->d_x_cur = d_x;
    filter->have_batch = 1;
    return 0;
}

int BatchNormApplyTrainingBatch(BatchNorm filter, const float *input, float *output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("BatchNormApplyTrainingBatch: NULL handle");
    if (!filter->training) NNTK_FAIL("BatchNormApplyTrainingBatch: the handle was created for inference");   /* batch_norm.c:192-194 */
    const int F = filter->config.feature_channels;
    const long N = (long)filter->config.count * filter->mini_batch;
    if (N <= 0 || F <= 0) return 0;
    float *d_x = nntk_devbuf_reserve(&filter->d_x, (size_t)N * F);
    float *d_o = nntk_devbuf_reserve(&filter->d_res, (size_t)N * F);
    if (!d_x || !d_o) return -1;
    if (nntk_shim_upload(d_x, input, (size_t)N * F * sizeof(float))) return -1;
    if (bn_train_forward_device(filter, d_x, d_o)) return -1;
    return nntk_shim_download(output, d_o, (size_t)N * F * sizeof(float));
}
/* Device-pointer form (additive): d_input / d_output are device buffers of count * mini_batch rows; d_input must stay valid and
 * unchanged until the matching BatchNormCalculateGradientDevice (x_mu and x_norm are recomputed from it). */
int BatchNormApplyTrainingBatchDevice(BatchNorm filter, const float *d_input, float *d_output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("BatchNormApplyTrainingBatchDevice: NULL handle");
    if (!filter->training) NNTK_FAIL("BatchNormApplyTrainingBatchDevice: the handle was created for inference");
    if ((long)filter->config.count * filter->mini_batch <= 0 || filter->config.feature_channels <= 0) return 0;
    return bn_train_forward_device(filter, d_input, d_output);
}

/* d_beta, d_gamma, d_x are all OVERWRITTEN (op_vec_sum stores, batch_norm.c:286, :297, :384).  void in the reference;
 * errors through nntk_last_error(). */
void BatchNormCalculateGradient(BatchNorm filter, BatchNormGradient *gradient, float *d_out) {
    nntk_shim_clear_error();
    if (!filter || !gradient || !d_out) { nntk_set_error("BatchNormCalculateGradient: NULL argument"); return; }
    if (!filter->training || !filter->have_batch) { nntk_set_error("BatchNormCalculateGradient: run BatchNormApplyTrainingBatch on a training handle first"); return; }
    const int F = filter->config.feature_channels;
    const long N = (long)filter->config.count * filter->mini_batch;
    const float *blk = nntk_batch_norm_device_block(filter, 1);      /* gamma may have been edited (an optimizer step) */
    if (!blk) return;
    float *d_dout = nntk_devbuf_reserve(&filter->d_dout, (size_t)N * F);
    float *d_dx = nntk_devbuf_reserve(&filter->d_res, (size_t)N * F);
    if (!d_dout || !d_dx) return;
    if (nntk_shim_upload(d_dout, d_out, (size_t)N * F * sizeof(float))) return;
    if (nntk_shim_bn_train_backward(filter->d_x_cur, d_dout, blk, filter->d_stats.p, filter->d_partial.p, d_dx, N, F)) return;
    if (nntk_shim_download(gradient->d_beta, filter->d_stats.p + (size_t)4 * F, (size_t)F * sizeof(float))) return;
    if (nntk_shim_download(gradient->d_gamma, filter->d_stats.p + (size_t)5 * F, (size_t)F * sizeof(float))) return;
    nntk_shim_download(gradient->d_x, d_dx, (size_t)N * F * sizeof(float));
}
/* Device-pointer form (additive): d_dbeta, d_dgamma [feature_channels] and d_dx [rows, feature_channels] are device buffers,
 * all overwritten; d_dout device [rows, feature_channels].  0 ok, -1 error. */
int BatchNormCalculateGradientDevice(BatchNorm filter, float *d_dbeta, float *d_dgamma, float *d_dx, const float *d_dout) {
    nntk_shim_clear_error();
    if (!filter || !d_dbeta || !d_dgamma || !d_dx || !d_dout) NNTK_FAIL("BatchNormCalculateGradientDevice: NULL argument");
    if (!filter->training || !filter->have_batch) NNTK_FAIL("BatchNormCalculateGradientDevice: run BatchNormApplyTrainingBatch[Device] on a training handle first");
    const int F = filter->config.feature_channels;
    const long N = (long)filter->config.count * filter->mini_batch;
    const float *blk = nntk_batch_norm_device_block(filter, 1);
    if (!blk) return -1;
    if (nntk_shim_bn_train_backward(filter->d_x_cur, d_dout, blk, filter->d_stats.p, filter->d_partial.p, d_dx, N, F)) return -1;
    if (nntk_shim_copy_d2d(d_dbeta, filter->d_stats.p + (size_t)4 * F, (size_t)F * sizeof(float))) return -1;
    return nntk_shim_copy_d2d(d_dgamma, filter->d_stats.p + (size_t)5 * F, (size_t)F * sizeof(float));
}

/* batch_norm.c:166-189: `count` rows of `feature_channels`; -1 for a training-mode handle (:167-169) */
int BatchNormApplyInference(BatchNorm filter, const float *input, float *output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("BatchNormApplyInference: NULL handle");
    if (filter->training) NNTK_FAIL("BatchNormApplyInference: the handle was created for training");
    const float *blk = nntk_batch_norm_device_block(filter, 1);
    if (!blk) return -1;
    size_t n = (size_t)filter->config.count * filter->config.feature_channels;
    if (n == 0) return 0;
    float *d = nntk_devbuf_reserve(&filter->d_io, n);
    if (!d) return -1;
    if (nntk_shim_upload(d, input, n * sizeof(float))) return -1;
    if (nntk_shim_batch_norm(d, blk, filter->config.epsilon, d, filter->config.count, filter->config.feature_channels))
        return -1;
    return nntk_shim_download(output, d, n * sizeof(float));
}
