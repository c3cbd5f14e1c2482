/*
 * oracle/nnref_layers.c -- CPU ORACLE (test infrastructure, not product code).
 * Restates layers/conv_1d.c, batch_norm.c, activation_default.c, gru.c, lstm.c,
 * dense.c, time_distributed_dense.c (forward/inference only) in the reference's
 * operation order.  "Parity unpinned" for these paths: see nnref.h header.
 */
#include "nnref.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------- activations --- */

/* layers/activation_default.c:28-33 sigmoid (neg, exp, +1, reciprocal),
 * :65-67 tanh, :98-103 identity, :123-129 relu (max then optional scale),
 * :149-167 softmax (no max subtraction; size = number of vectors). */
void ref_activation(int kind, float relu_a, int softmax_vector_size,
                    const float *in, float *out, int size) {
    switch (kind) {
    case REF_ACT_IDENTITY:
        if (in != out) memmove(out, in, sizeof(float) * (size_t)size);
        break;
    case REF_ACT_SIGMOID:
        for (int i = 0; i < size; ++i) {
            float v = -in[i];
            v = expf(v);
            v = v + 1;
            out[i] = 1 / v;
        }
        break;
    case REF_ACT_TANH:
        for (int i = 0; i < size; ++i) out[i] = tanhf(in[i]);
        break;
    case REF_ACT_RELU:
        for (int i = 0; i < size; ++i) out[i] = fmaxf(in[i], 0);
        if (relu_a != 1.0)
            for (int i = 0; i < size; ++i) out[i] = out[i] * relu_a;
        break;
    case REF_ACT_SOFTMAX:
        for (int v = 0; v < size; ++v) {
            const float *x = in + (size_t)v * softmax_vector_size;
            float *y = out + (size_t)v * softmax_vector_size;
            for (int i = 0; i < softmax_vector_size; ++i) y[i] = expf(x[i]);
            float sum = 0.0f;
            for (int i = 0; i < softmax_vector_size; ++i) sum += y[i];
            for (int i = 0; i < softmax_vector_size; ++i) y[i] = y[i] / sum;
        }
        break;
    default: break;
    }
}

/* ---------------------------------------------------------------- conv1d --- */

/* layers/conv_1d.c:77-87 */
int ref_conv1d_output_size(int input_size, int kernel_size, int stride) {
    return (input_size - (kernel_size - stride)) / stride;
}

/* layers/conv_1d.c:122-147: transpose [T,Cin]->[Cin,T], then per (o, x):
 * result = 0; for i: result += dot_k(row_i[x*s ..], W[o,i,:]); result += b[o] */
void ref_conv1d(const float *in, const float *W, const float *b, float *out,
                int T, int Cin, int Cout, int k, int stride) {
    int Tout = ref_conv1d_output_size(T, k, stride);
    float *tr = (float *)malloc(sizeof(float) * (size_t)T * Cin);
    ref_op_mat_transp(in, tr, Cin, T);
    for (int o = 0; o < Cout; ++o) {
        const float *Wo = W + (size_t)o * Cin * k;
        for (int x = 0; x < Tout; ++x) {
            float result = 0.0f;
            int off = x * stride;
            for (int i = 0; i < Cin; ++i)
                result += ref_op_vec_dot(tr + (size_t)i * T + off, Wo + i * k, k);
            result += b[o];
            out[(size_t)x * Cout + o] = result;
        }
    }
    free(tr);
}

/* layers/conv_1d.c:167-183 forward part */
void ref_conv1d_batch(const float *in, const float *W, const float *b, float *out,
                      int B, int T, int Cin, int Cout, int k, int stride) {
    int Tout = ref_conv1d_output_size(T, k, stride);
    for (int n = 0; n < B; ++n)
        ref_conv1d(in + (size_t)n * T * Cin, W, b, out + (size_t)n * Tout * Cout, T, Cin, Cout, k, stride);
}

/* ------------------------------------------------------------ batch norm --- */

/* layers/conv_1d.c:185-245 (Conv1dCalculateGradient) + weights_private.c:50-55 (default_gradient_sum), in the
 * reference's loop order: per batch entry b: d_b += rows of d_out; for out_f, out_n, in_f: d_W[out_f][in_f][:] +=
 * x_row[:] * d_o, d_X_transposed[in_f][out_n*stride + :] += W[out_f][in_f][:] * d_o; then per-batch d_W, d_b summed over b.
 * dW [Cout][Cin][k] and db [Cout] are ADDED to (the caller's zeroed gradient block), dX [B][T][Cin] is overwritten. */
void ref_conv1d_gradient(const float *in, const float *W, const float *dout, float *dW, float *db, float *dX,
                         int B, int T, int Cin, int Cout, int k, int stride) {
    int Tout = ref_conv1d_output_size(T, k, stride);
    if (Tout < 0) Tout = 0;
    size_t w = (size_t)Cout * Cin * k;
    float *bw = (float *)calloc(w ? w : 1, sizeof(float)), *bb = (float *)calloc((size_t)Cout, sizeof(float));
    float *xt = (float *)calloc((size_t)Cin * T + 1, sizeof(float)), *dxt = (float *)calloc((size_t)Cin * T + 1, sizeof(float));
    for (int b = 0; b < B; ++b) {
        memset(bw, 0, w * sizeof(float)); memset(bb, 0, (size_t)Cout * sizeof(float));
        memset(dxt, 0, (size_t)Cin * T * sizeof(float));
        for (int t = 0; t < T; ++t) for (int i = 0; i < Cin; ++i) xt[(size_t)i * T + t] = in[((size_t)b * T + t) * Cin + i];
        const float *d = dout + (size_t)b * Tout * Cout;
        for (int o = 0; o < Tout; ++o) for (int f = 0; f < Cout; ++f) bb[f] = bb[f] + d[(size_t)o * Cout + f];
        for (int of = 0; of < Cout; ++of)
            for (int on = 0; on < Tout; ++on) {
                const float d_o = d[(size_t)on * Cout + of];
                for (int f = 0; f < Cin; ++f) {
                    const float *row = xt + (size_t)f * T + (size_t)on * stride;
                    const float *wp = W + ((size_t)of * Cin + f) * k;
                    float *dw = bw + ((size_t)of * Cin + f) * k;
                    float *dx = dxt + (size_t)f * T + (size_t)on * stride;
                    for (int kk = 0; kk < k; ++kk) { float p = row[kk] * d_o; dw[kk] = dw[kk] + p; }
                    for (int kk = 0; kk < k; ++kk) { float p = wp[kk] * d_o; dx[kk] = dx[kk] + p; }
                }
            }
        for (int t = 0; t < T; ++t) for (int i = 0; i < Cin; ++i) dX[((size_t)b * T + t) * Cin + i] = dxt[(size_t)i * T + t];
        for (size_t e = 0; e < w; ++e) dW[e] = bw[e] + dW[e];
        for (int f = 0; f < Cout; ++f) db[f] = bb[f] + db[f];
    }
    free(bw); free(bb); free(xt); free(dxt);
}

/* layers/batch_norm.c:140-163, :166-189:
 * ((x - mean) / sqrt(var + eps)) * gamma + beta, each op a separate rounding */
void ref_batch_norm(const float *in, const float *gamma, const float *beta,
                    const float *mean, const float *variance, float *out,
                    float epsilon, int count, int C) {
    for (int r = 0; r < count; ++r) {
        const float *x = in + (size_t)r * C;
        float *y = out + (size_t)r * C;
        for (int c = 0; c < C; ++c) {
            float x_mu = x[c] - mean[c];
            float var_eps = variance[c] + epsilon;
            float sqrt_var = sqrtf(var_eps);
            float x_norm = x_mu / sqrt_var;
            float gx = x_norm * gamma[c];
            y[c] = gx + beta[c];
        }
    }
}

/* ------------------------------------------------------------------- GRU --- */

/* ReLU output scale `a` of each gate's ActivationFunction handle (activation_default.c:123-129: the cell calls
 * ActivationFunctionApply on the gate's own handle, gru.c:157-173 / lstm.c:213-237 / rnn.c:163, so a ReLU gate
 * carries its a).  Index = position in the act_* argument lists below (GRU: z, h, r; LSTM: i, f, g, o, out; RNN: 0).
 * Default 1; set with ref_set_gate_relu_scales before a call. */
static _Thread_local float g_gate_a[5] = {1.f, 1.f, 1.f, 1.f, 1.f};
void ref_set_gate_relu_scales(const float *a, int n) {
    for (int i = 0; i < 5; ++i) g_gate_a[i] = (a && i < n) ? a[i] : 1.0f;
}

static void vec_add(const float *a, const float *b, float *c, int n) {
    for (int i = 0; i < n; ++i) c[i] = a[i] + b[i];
}
static void vec_mul(const float *a, const float *b, float *c, int n) {
    for (int i = 0; i < n; ++i) c[i] = a[i] * b[i];
}

/* layers/gru.c:129-187 (GRUCellForward).  buf needs 14*H floats. */
static void gru_cell(const float *x, const float *W, const float *U, const float *b_i, const float *b_h,
                     const float *h_pr, float *ht, float *buf, int in, int H,
                     int act_z, int act_h, int act_r) {
    float *Z_zr = buf;               /* 6H: Z_z, Z_r, Z_h~, z, r, h~ */
    float *x_W = buf + 6 * H;        /* 3H */
    ref_op_mat_mul(x, W, x_W, 1, 3 * H, in);
    vec_add(x_W, b_i, x_W, 3 * H);
    float *h_pr_U = x_W + 3 * H;     /* 3H */
    ref_op_mat_mul(h_pr, U, h_pr_U, 1, 3 * H, H);
    vec_add(h_pr_U, b_h, h_pr_U, 3 * H);
    vec_add(x_W, h_pr_U, Z_zr, 2 * H);
    float *z = Z_zr + 3 * H;
    float *r = z + H;
    ref_activation(act_z, g_gate_a[0], 0, Z_zr, z, H);
    ref_activation(act_r, g_gate_a[2], 0, Z_zr + H, r, H);
    float *Z_h = Z_zr + 2 * H;
    float *h_tilda = r + H;
    vec_mul(r, h_pr_U + 2 * H, Z_h, H);
    vec_add(Z_h, x_W + 2 * H, Z_h, H);
    ref_activation(act_h, g_gate_a[1], 0, Z_h, h_tilda, H);
    float *minus_z = h_pr_U + 3 * H; /* H */
    for (int i = 0; i < H; ++i) minus_z[i] = -z[i];
    for (int i = 0; i < H; ++i) minus_z[i] = minus_z[i] + 1;
    vec_mul(minus_z, h_tilda, minus_z, H);
    float *z_h = minus_z + H;        /* H */
    vec_mul(z, h_pr, z_h, H);
    vec_add(minus_z, z_h, ht, H);
}

/* layers/gru.c:189-204 (GRUApplyInference): state carried in h_state */
void ref_gru_sequence(const float *x, const float *W, const float *U,
                      const float *b_i, const float *b_h, float *h_state, float *out,
                      int T, int in, int H, int return_sequences,
                      int act_z, int act_h, int act_r) {
    float *buf = (float *)calloc((size_t)14 * H, sizeof(float));
    for (int t = 0; t < T; ++t) {
        float *o = out + (return_sequences ? (size_t)t * H : 0);
        gru_cell(x + (size_t)t * in, W, U, b_i, b_h, h_state, o, buf, in, H, act_z, act_h, act_r);
        memcpy(h_state, o, sizeof(float) * (size_t)H);
    }
    free(buf);
}

/* layers/gru.c:246-293 forward semantics: zero state per sequence (:260) */
void ref_gru_batch(const float *x, const float *W, const float *U,
                   const float *b_i, const float *b_h, float *out,
                   int B, int T, int in, int H, int return_sequences,
                   int act_z, int act_h, int act_r) {
    float *h = (float *)malloc(sizeof(float) * (size_t)H);
    for (int n = 0; n < B; ++n) {
        memset(h, 0, sizeof(float) * (size_t)H);
        float *o = out + (return_sequences ? (size_t)n * T * H : (size_t)n * H);
        ref_gru_sequence(x + (size_t)n * T * in, W, U, b_i, b_h, h, o, T, in, H, return_sequences,
                         act_z, act_h, act_r);
    }
    free(h);
}

/* ------------------------------------------------------------------ LSTM --- */

/* layers/lstm.c:185-239 (LSTMCellForward).  buf needs 15*H floats. */
static void lstm_cell(const float *x, const float *W, const float *U, const float *b_i, const float *b_h,
                      const float *c_prev, const float *h_prev, float *c, float *h, float *buf,
                      int in, int H, int v2, int act_i, int act_f, int act_g, int act_o, int act_out) {
    float *Z = buf;                  /* 8H: Z(4H), i, f, g, o */
    float *u_H = buf + 8 * H;        /* 4H, then i_g, f_c_pr, c_tanh */
    ref_op_mat_mul(x, W, Z, 1, 4 * H, in);
    vec_add(Z, b_i, Z, 4 * H);
    ref_op_mat_mul(h_prev, U, u_H, 1, 4 * H, H);
    if (v2) vec_add(u_H, b_h, u_H, 4 * H);
    vec_add(Z, u_H, Z, 4 * H);
    float *ig = Z + 4 * H;
    ref_activation(act_i, g_gate_a[0], 0, Z, ig, H);
    float *fg = ig + H;
    ref_activation(act_f, g_gate_a[1], 0, Z + H, fg, H);
    float *gg = fg + H;
    ref_activation(act_g, g_gate_a[2], 0, Z + 2 * H, gg, H);
    float *og = gg + H;
    ref_activation(act_o, g_gate_a[3], 0, Z + 3 * H, og, H);
    float *i_g = u_H + 4 * H;
    vec_mul(ig, gg, i_g, H);
    float *f_c = i_g + H;
    vec_mul(fg, c_prev, f_c, H);
    vec_add(f_c, i_g, c, H);
    float *c_t = f_c + H;
    ref_activation(act_out, g_gate_a[4], 0, c, c_t, H);
    vec_mul(og, c_t, h, H);
}

/* layers/lstm.c:241-268 (LSTMApplyInference) */
void ref_lstm_sequence(const float *x, const float *W, const float *U,
                       const float *b_i, const float *b_h, float *h_state, float *c_state,
                       float *out, int T, int in, int H, int return_sequences, int v2,
                       int act_i, int act_f, int act_g, int act_o, int act_out) {
    float *buf = (float *)calloc((size_t)15 * H, sizeof(float));
    float *state = (float *)malloc(sizeof(float) * (size_t)H);
    for (int t = 0; t < T; ++t) {
        float *o = out + (return_sequences ? (size_t)t * H : 0);
        lstm_cell(x + (size_t)t * in, W, U, b_i, b_h, c_state, h_state, state, o, buf, in, H, v2,
                  act_i, act_f, act_g, act_o, act_out);
        memcpy(h_state, o, sizeof(float) * (size_t)H);
        memcpy(c_state, state, sizeof(float) * (size_t)H);
    }
    free(buf); free(state);
}

/* layers/lstm.c:426-475 forward semantics: zero state per sequence (:439) */
void ref_lstm_batch(const float *x, const float *W, const float *U,
                    const float *b_i, const float *b_h, float *out,
                    int B, int T, int in, int H, int return_sequences, int v2,
                    int act_i, int act_f, int act_g, int act_o, int act_out) {
    float *h = (float *)malloc(sizeof(float) * (size_t)H);
    float *c = (float *)malloc(sizeof(float) * (size_t)H);
    for (int n = 0; n < B; ++n) {
        memset(h, 0, sizeof(float) * (size_t)H);
        memset(c, 0, sizeof(float) * (size_t)H);
        float *o = out + (return_sequences ? (size_t)n * T * H : (size_t)n * H);
        ref_lstm_sequence(x + (size_t)n * T * in, W, U, b_i, b_h, h, c, o, T, in, H, return_sequences, v2,
                          act_i, act_f, act_g, act_o, act_out);
    }
    free(h); free(c);
}

/* ------------------------------------------------------------------- RNN --- */

/* layers/rnn.c:144-166 (RNNCellForward): x_W = x W + b_i; h_U = h U (+ b_h if v2); gate = h_U + x_W; h' = act(gate).
 * buf needs 3*H floats. */
static void rnn_cell(const float *x, const float *W, const float *U, const float *b_i, const float *b_h,
                     const float *h_prev, float *h, float *buf, int in, int H, int v2, int act) {
    float *x_W = buf, *h_U = buf + H, *gate = buf + 2 * H;
    ref_op_mat_mul(x, W, x_W, 1, H, in);
    vec_add(x_W, b_i, x_W, H);
    ref_op_mat_mul(h_prev, U, h_U, 1, H, H);
    if (v2) vec_add(h_U, b_h, h_U, H);
    vec_add(h_U, x_W, gate, H);
    ref_activation(act, g_gate_a[0], 0, gate, h, H);
}

/* The layer that layers/rnn.c:249-291 (the batch forward pass) computes, one sequence, state carried in
 * h_state.  RNNApplyInference itself (rnn.c:228-247) writes cell i's output at output + i*(i*out) and then
 * reloads h from output + i*out -- an indexing slip ("note and do not replicate"). */
void ref_rnn_sequence(const float *x, const float *W, const float *U,
                      const float *b_i, const float *b_h, float *h_state, float *out,
                      int T, int in, int H, int return_sequences, int v2, int act) {
    float *buf = (float *)calloc((size_t)3 * H, sizeof(float));
    for (int t = 0; t < T; ++t) {
        float *o = out + (return_sequences ? (size_t)t * H : 0);
        rnn_cell(x + (size_t)t * in, W, U, b_i, b_h, h_state, o, buf, in, H, v2, act);
        memcpy(h_state, o, sizeof(float) * (size_t)H);
    }
    free(buf);
}

/* layers/rnn.c:249-291 forward semantics: zero state per sequence (:260) */
void ref_rnn_batch(const float *x, const float *W, const float *U,
                   const float *b_i, const float *b_h, float *out,
                   int B, int T, int in, int H, int return_sequences, int v2, int act) {
    float *h = (float *)malloc(sizeof(float) * (size_t)H);
    for (int n = 0; n < B; ++n) {
        memset(h, 0, sizeof(float) * (size_t)H);
        float *o = out + (return_sequences ? (size_t)n * T * H : (size_t)n * H);
        ref_rnn_sequence(x + (size_t)n * T * in, W, U, b_i, b_h, h, o, T, in, H, return_sequences, v2, act);
    }
    free(h);
}

/* --------------------------------------------------------- bidirectional --- */

/* layers/bidirectional.c:11-23 (reverse / reverse_batch): rows of each [T, F] sequence in reverse order */
void ref_bd_reverse_batch(const float *in, float *out, int B, int T, int F) {
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t)
            memcpy(out + ((size_t)b * T + t) * F, in + ((size_t)b * T + (T - 1 - t)) * F, sizeof(float) * (size_t)F);
}

/* layers/bidirectional.c:42-58 (bd_merge_concat): transpose both results into one [2C, rows] buffer and
 * transpose that back, i.e. out[b][r] = forward[b][r] | backward[b][r] */
void ref_bd_merge_concat(const float *fwd, const float *bwd, float *out, int B, int rows, int C) {
    float *buffer = (float *)malloc(sizeof(float) * (size_t)2 * rows * C);
    size_t size = (size_t)rows * C;
    for (int b = 0; b < B; ++b) {
        ref_op_mat_transp(fwd + b * size, buffer, C, rows);
        ref_op_mat_transp(bwd + b * size, buffer + size, C, rows);
        ref_op_mat_transp(buffer, out + 2 * b * size, rows, 2 * C);
    }
    free(buffer);
}

/* layers/bidirectional.c:76-85 (bd_merge_sum) */
void ref_bd_merge_sum(const float *fwd, const float *bwd, float *out, int B, int rows, int C) {
    vec_add(fwd, bwd, out, B * rows * C);
}

/* ----------------------------------------------------------------- dense --- */

/* layers/dense.c:122-142: z = x*W + b ; a = activation(z) (or copy if none).
 * The activation handle carries its own size (act_size), as in the reference. */
void ref_dense(const float *x, const float *W, const float *b, float *out,
               int in, int out_size, int act_kind, float relu_a, int softmax_vector_size, int act_size) {
    ref_op_mat_mul(x, W, out, 1, out_size, in);
    vec_add(out, b, out, out_size);
    if (act_kind >= 0) ref_activation(act_kind, relu_a, softmax_vector_size, out, out, act_size);
}

/* layers/time_distributed_dense.c:52-58 */
void ref_time_distributed_dense(const float *x, const float *W, const float *b, float *out,
                                int ts, int in, int out_size,
                                int act_kind, float relu_a, int softmax_vector_size, int act_size) {
    for (int t = 0; t < ts; ++t)
        ref_dense(x + (size_t)t * in, W, b, out + (size_t)t * out_size, in, out_size,
                  act_kind, relu_a, softmax_vector_size, act_size);
}

/* ======================= training, second slice ======================= */

/* layers/activation.c:47-54 with activation_default.c:38-51 (sigmoid), :70-82 (tanh), :94-96 (identity), :118-121
 * (ReLU), :169-190 (softmax).  a == NULL selects the non-cached derivative (forward recomputed from z); kinds without a
 * cached derivative (identity, ReLU) always use z.  size = elements, or vectors for softmax.  The softmax loop reads
 * d_out at the CALL's base for every vector (activation_default.c:183) -- restated as it is. */
void ref_activation_gradient(int kind, int vector_size, const float *z, const float *a, const float *dout, float *out, int size) {
    if (kind == REF_ACT_SOFTMAX) {
        int v = vector_size;
        float *fwd = NULL;
        if (!a) {
            fwd = (float *)malloc((size_t)size * v * sizeof(float));
            ref_activation(REF_ACT_SOFTMAX, 1.0f, v, z, fwd, size);
            a = fwd;
        }
        float *m = (float *)malloc((size_t)v * v * sizeof(float));
        float *tmp = (float *)malloc((size_t)v * sizeof(float));
        for (int index = 0; index < size; ++index) {
            const float *in = a + (size_t)index * v;
            for (int i = 0; i < v; ++i)
                for (int j = 0; j < v; ++j)
                    m[i * v + j] = i == j ? in[i] * (1 - in[i]) : -1 * in[i] * in[j];
            ref_op_mat_mul(dout, m, tmp, 1, v, v);
            memcpy(out + (size_t)index * v, tmp, (size_t)v * sizeof(float));
        }
        free(m); free(tmp); free(fwd);
        return;
    }
    for (int i = 0; i < size; ++i) {
        float r;
        if (kind == REF_ACT_SIGMOID || kind == REF_ACT_TANH) {
            float s;
            if (a) s = a[i];
            else ref_activation(kind, 1.0f, 0, z + i, &s, 1);
            if (kind == REF_ACT_SIGMOID) { float t = -s; t = t + 1; t = s * t; r = t * dout[i]; }
            else { float t = s * s; t = -t; t = t + 1; r = t * dout[i]; }
        } else if (kind == REF_ACT_RELU) {
            float c = fmaxf(fminf(z[i], 1.0f), 0.0f);
            r = c * dout[i];
        } else {
            r = dout[i];
        }
        out[i] = r;
    }
}

/* layers/dense.c:164-185 + weights_private.c:43-48.  act_kind < 0: no activation handle.  act_size = the handle's
 * input_size (elements, or vectors for softmax) -- the reference calls the activation once per sample.  gW [in,out] and
 * gb [out] are ADDED to in mini-batch order, dX [B,in] is overwritten.  z, a: [B,out] cached by the forward pass. */
void ref_dense_gradient(const float *x, const float *W, const float *z, const float *a, const float *dout,
                        int act_kind, int vector_size, int act_size, float *gW, float *gb, float *dX, int B, int in, int out) {
    float *dz = (float *)malloc((size_t)out * sizeof(float));
    float *dW = (float *)malloc((size_t)in * out * sizeof(float));
    for (int b = 0; b < B; ++b) {
        if (act_kind >= 0) {
            int cached = act_kind == REF_ACT_SIGMOID || act_kind == REF_ACT_TANH || act_kind == REF_ACT_SOFTMAX;
            ref_activation_gradient(act_kind, vector_size, z + (size_t)b * out, cached ? a + (size_t)b * out : NULL,
                                    dout + (size_t)b * out, dz, act_size);
        } else {
            memcpy(dz, dout + (size_t)b * out, (size_t)out * sizeof(float));
        }
        ref_op_mat_mul(x + (size_t)b * in, dz, dW, in, out, 1);              /* d_W = x^T dz */
        ref_op_mat_mul(W, dz, dX + (size_t)b * in, in, 1, out);              /* d_X = W dz */
        for (size_t e = 0; e < (size_t)in * out; ++e) gW[e] = dW[e] + gW[e];
        for (int o = 0; o < out; ++o) gb[o] = dz[o] + gb[o];
    }
    free(dz); free(dW);
}

/* train/loss.c:13-24 */
float ref_mean_squared_error(const float *y, const float *p, int size, int batch) {
    float loss = 0.0f;
    for (int b = 0; b < batch; ++b) {
        float one = 0.0f;
        for (int i = 0; i < size; ++i) { float d = y[(size_t)b * size + i] - p[(size_t)b * size + i]; d = d * d; one = one + d; }
        loss += one / (float)size;
    }
    return loss / (float)batch;
}
/* train/loss.c:26-32 */
void ref_mean_squared_error_derivative(const float *y, const float *p, float *d, int size, int batch) {
    for (int b = 0; b < batch; ++b) {
        float k = -2.0f / (float)(size * batch);
        for (int i = 0; i < size; ++i) { float t = y[(size_t)b * size + i] - p[(size_t)b * size + i]; d[(size_t)b * size + i] = t * k; }
    }
}
/* train/loss.c:34-45 */
float ref_categorical_crossentropy(const float *y, const float *p, int c, int batch) {
    float loss = 0.0f;
    for (int b = 0; b < batch; ++b) {
        float one = 0.0f;
        for (int i = 0; i < c; ++i) { float t = logf(p[(size_t)b * c + i]); t = t * y[(size_t)b * c + i]; one = one + t; }
        loss += -one;
    }
    return loss / (float)batch;
}
/* train/loss.c:47-52 AS WRITTEN: the loop body ignores b, so only row 0 is ever written (batch times) */
void ref_categorical_crossentropy_derivative(const float *y, const float *p, float *d, int c, int batch) {
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < c; ++i) { float t = y[i] / p[i]; d[i] = t * -1.0f; }
}
/* train/optimizers.c:13-19 */
void ref_sgd_optimize(float lr, const float *g, float *w, int size) {
    for (int i = 0; i < size; ++i) { float t = g[i] * lr; w[i] = w[i] - t; }
}

/* layers/batch_norm.c:191-262: batch mean / biased variance per feature (column sums in row order), batch_norm() with
 * them, then the moving statistics: buffer = stat * (1 - momentum); moving = moving * momentum; moving = buffer + moving.
 * x, out [N, F]; mean, var [F] are outputs; moving_mean / moving_var [F] are updated in place. */
void ref_batch_norm_training_forward(const float *x, const float *gamma, const float *beta, float eps, float momentum,
                                     float *out, float *mean, float *var, float *moving_mean, float *moving_var, int N, int F) {
    for (int f = 0; f < F; ++f) {
        float s = 0.0f;
        for (int n = 0; n < N; ++n) s = s + x[(size_t)n * F + f];
        mean[f] = s / (float)N;
    }
    for (int f = 0; f < F; ++f) {
        float s = 0.0f;
        for (int n = 0; n < N; ++n) { float d = x[(size_t)n * F + f] + -mean[f]; d = d * d; s = s + d; }
        var[f] = s / (float)N;
    }
    ref_batch_norm(x, gamma, beta, mean, var, out, eps, N, F);
    float one_minus = 1 - momentum;
    for (int f = 0; f < F; ++f) {
        float b = mean[f] * one_minus; float m = moving_mean[f] * momentum; moving_mean[f] = b + m;
        b = var[f] * one_minus; m = moving_var[f] * momentum; moving_var[f] = b + m;
    }
}

/* layers/batch_norm.c:264-386 in its operation order (mean, var = the batch statistics of the forward pass).
 * d_beta, d_gamma [F] and d_x [N, F] are overwritten. */
void ref_batch_norm_gradient(const float *x, const float *dout, const float *gamma, const float *mean, const float *var,
                             float eps, float *d_beta, float *d_gamma, float *d_x, int N, int F) {
    float *d_var = (float *)malloc((size_t)F * sizeof(float)), *d_mu = (float *)malloc((size_t)F * sizeof(float));
    float *sqrt_var = (float *)malloc((size_t)F * sizeof(float));
    for (int f = 0; f < F; ++f) {
        float var_eps = var[f] + eps;
        sqrt_var[f] = sqrtf(var_eps);
        float sb = 0.0f, sg = 0.0f, si = 0.0f;
        for (int n = 0; n < N; ++n) {
            float d = dout[(size_t)n * F + f];
            float x_mu = x[(size_t)n * F + f] - mean[f];
            float x_norm = x_mu / sqrt_var[f];
            float dxn = d * gamma[f];
            sb = sb + d;
            float t = d * x_norm; sg = sg + t;
            t = dxn * x_mu; si = si + t;
        }
        d_beta[f] = sb; d_gamma[f] = sg;
        float ds = si * -1.0f; ds = ds / var_eps;          /* d_sqrt_var */
        float dv = ds / sqrt_var[f]; dv = dv / 2.0f; dv = dv / (float)N;
        d_var[f] = dv;
    }
    for (int f = 0; f < F; ++f) {
        float s = 0.0f;
        for (int n = 0; n < N; ++n) {
            float x_mu = x[(size_t)n * F + f] - mean[f];
            float dxn = dout[(size_t)n * F + f] * gamma[f];
            float a = dxn / sqrt_var[f];
            float b = x_mu * 2; b = b * d_var[f];
            float dxm = a + b;
            d_x[(size_t)n * F + f] = dxm;
            s = s + dxm;
        }
        d_mu[f] = s * (-1.0f / (float)N);
    }
    for (int n = 0; n < N; ++n)
        for (int f = 0; f < F; ++f) d_x[(size_t)n * F + f] = d_x[(size_t)n * F + f] + d_mu[f];
    free(d_var); free(d_mu); free(sqrt_var);
}

/* layers/gru.c:246-293 (GRUApplyTrainingBatch): zero state per sequence; caches Z_gates [B][T][6H] = Z_z | Z_r | Z_h~ |
 * z | r | h~, h_pr_Uh [B][T][H], h [B][T][H] */
void ref_gru_training_forward(const float *x, const float *W, const float *U, const float *b_i, const float *b_h,
                              float *h, float *Zg, float *hU, int B, int T, int in, int H, int act_z, int act_h, int act_r) {
    float *buf = (float *)calloc((size_t)14 * H, sizeof(float)), *state = (float *)malloc((size_t)H * sizeof(float));
    for (int b = 0; b < B; ++b) {
        memset(state, 0, (size_t)H * sizeof(float));
        for (int t = 0; t < T; ++t) {
            size_t row = (size_t)b * T + t;
            gru_cell(x + row * in, W, U, b_i, b_h, state, h + row * H, buf, in, H, act_z, act_h, act_r);
            memcpy(Zg + row * 6 * H, buf, (size_t)6 * H * sizeof(float));
            memcpy(hU + row * H, buf + 9 * H + 2 * H, (size_t)H * sizeof(float));      /* h_pr_U[2H..3H) */
            memcpy(state, h + row * H, (size_t)H * sizeof(float));
            memset(buf, 0, (size_t)14 * H * sizeof(float));
        }
    }
    free(buf); free(state);
}

/* layers/gru.c:295-512 (GRUCellBackward + GRUCalculateGradient) in its operation order: for b, for t = T-1 .. 0, the
 * cell's d_W_t, d_U_t, d_bi_t, d_bh_t are added onto gW [in][3H], gU [H][3H], gbi, gbh [3H]; dX [B][T][in] overwritten. */
void ref_gru_gradient(const float *x, const float *W, const float *U, const float *h, const float *Zg, const float *hU,
                      const float *dout, int return_sequences, float *gW, float *gU, float *gbi, float *gbh, float *dX,
                      int B, int T, int in, int H, int act_z, int act_h, int act_r) {
    int G = 3 * H;
    float *dW = (float *)malloc((size_t)in * G * sizeof(float)), *dU = (float *)malloc((size_t)H * G * sizeof(float));
    float *d_x_W = (float *)malloc((size_t)G * sizeof(float)), *d_h_pr_U = (float *)malloc((size_t)G * sizeof(float));
    float *tmp = (float *)malloc((size_t)8 * H * sizeof(float)), *dh_carry = (float *)malloc((size_t)H * sizeof(float));
    float *d_h_t = tmp, *d_h_prev_1 = tmp + H, *d_h_tilda = tmp + 2 * H, *d_z_t = tmp + 3 * H, *d_z_h = tmp + 4 * H,
          *d_r_t = tmp + 5 * H, *d_h_prev_2 = tmp + 6 * H, *d_zr = tmp + 7 * H;
    (void)d_zr;
    for (int b = 0; b < B; ++b)
        for (int t = T - 1; t >= 0; --t) {
            size_t row = (size_t)b * T + t;
            const float *Z = Zg + row * 6 * H, *z = Z + 3 * H, *r = Z + 4 * H, *ht = Z + 5 * H;
            const float *h_prev = t == 0 ? NULL : h + (row - 1) * H;
            for (int j = 0; j < H; ++j) {
                float d_o = return_sequences ? dout[row * H + j] : (t == T - 1 ? dout[(size_t)b * H + j] : 0.0f);
                d_h_t[j] = (t == T - 1 ? 0.0f : dh_carry[j]) + d_o;
            }
            vec_mul(z, d_h_t, d_h_prev_1, H);
            for (int j = 0; j < H; ++j) { float m = -z[j]; m = m * d_h_t[j]; d_h_tilda[j] = m + d_h_t[j]; }
            for (int j = 0; j < H; ++j) { float d = h_prev ? h_prev[j] - ht[j] : -ht[j]; d_z_t[j] = d * d_h_t[j]; }
            ref_activation_gradient(act_h, 0, Z + 2 * H, ht, d_h_tilda, d_z_h, H);
            vec_mul(hU + row * H, d_z_h, d_r_t, H);
            memcpy(d_x_W + 2 * H, d_z_h, (size_t)H * sizeof(float));
            vec_mul(r, d_z_h, d_h_pr_U + 2 * H, H);
            ref_activation_gradient(act_z, 0, Z, z, d_z_t, d_x_W, H);
            ref_activation_gradient(act_r, 0, Z + H, r, d_r_t, d_x_W + H, H);
            memcpy(d_h_pr_U, d_x_W, (size_t)2 * H * sizeof(float));
            ref_op_mat_mul(W, d_x_W, dX + row * in, in, 1, G);
            ref_op_mat_mul(U, d_h_pr_U, d_h_prev_2, H, 1, G);
            vec_add(d_h_prev_1, d_h_prev_2, dh_carry, H);
            ref_op_mat_mul(x + row * in, d_x_W, dW, in, G, 1);
            if (h_prev) ref_op_mat_mul(h_prev, d_h_pr_U, dU, H, G, 1);
            else memset(dU, 0, (size_t)H * G * sizeof(float));
            for (size_t e = 0; e < (size_t)in * G; ++e) gW[e] = gW[e] + dW[e];
            for (size_t e = 0; e < (size_t)H * G; ++e) gU[e] = gU[e] + dU[e];
            for (int e = 0; e < G; ++e) { gbi[e] = gbi[e] + d_x_W[e]; gbh[e] = gbh[e] + d_h_pr_U[e]; }
        }
    free(dW); free(dU); free(d_x_W); free(d_h_pr_U); free(tmp); free(dh_carry);
}

/* layers/lstm.c:418-475 (LSTMApplyTrainingBatch): zero state per sequence; caches zifgo [B][T][8H], c [B][T][H], h [B][T][H] */
void ref_lstm_training_forward(const float *x, const float *W, const float *U, const float *b_i, const float *b_h,
                               float *h, float *c, float *zifgo, int B, int T, int in, int H, int v2,
                               int act_i, int act_f, int act_g, int act_o, int act_out) {
    float *buf = (float *)calloc((size_t)15 * H, sizeof(float));
    float *hs = (float *)malloc((size_t)H * sizeof(float)), *cs = (float *)malloc((size_t)H * sizeof(float));
    for (int b = 0; b < B; ++b) {
        memset(hs, 0, (size_t)H * sizeof(float)); memset(cs, 0, (size_t)H * sizeof(float));
        for (int t = 0; t < T; ++t) {
            size_t row = (size_t)b * T + t;
            lstm_cell(x + row * in, W, U, b_i, b_h, cs, hs, c + row * H, h + row * H, buf, in, H, v2, act_i, act_f, act_g, act_o, act_out);
            memcpy(zifgo + row * 8 * H, buf, (size_t)8 * H * sizeof(float));
            memcpy(hs, h + row * H, (size_t)H * sizeof(float));
            memcpy(cs, c + row * H, (size_t)H * sizeof(float));
        }
    }
    free(buf); free(hs); free(cs);
}

/* layers/lstm.c:294-416 (LSTMCellBackward) + :477-556 (LSTMCalculateGradient) in its operation order */
void ref_lstm_gradient(const float *x, const float *W, const float *U, const float *h, const float *c, const float *zifgo,
                       const float *dout, int return_sequences, float *gW, float *gU, float *gbi, float *gbh, float *dX,
                       int B, int T, int in, int H, int act_i, int act_f, int act_g, int act_o, int act_out) {
    int G = 4 * H;
    float *dW = (float *)malloc((size_t)in * G * sizeof(float)), *dU = (float *)malloc((size_t)H * G * sizeof(float));
    float *dg = (float *)malloc((size_t)G * sizeof(float)), *tmp = (float *)malloc((size_t)8 * H * sizeof(float));
    float *dh = (float *)malloc((size_t)H * sizeof(float)), *dc_carry = (float *)malloc((size_t)H * sizeof(float));
    float *d_h_t = tmp, *d_a_O = tmp + H, *d_a_C = tmp + 2 * H, *d_c_t = tmp + 3 * H, *d_a = tmp + 4 * H, *tc = tmp + 5 * H;
    for (int b = 0; b < B; ++b)
        for (int t = T - 1; t >= 0; --t) {
            size_t row = (size_t)b * T + t;
            const float *Z = zifgo + row * 8 * H, *it = Z + 4 * H, *ft = Z + 5 * H, *gt = Z + 6 * H, *ot = Z + 7 * H;
            const float *c_t = c + row * H, *c_prev = t == 0 ? NULL : c + (row - 1) * H, *h_prev = t == 0 ? NULL : h + (row - 1) * H;
            for (int j = 0; j < H; ++j) {
                float d_o = return_sequences ? dout[row * H + j] : (t == T - 1 ? dout[(size_t)b * H + j] : 0.0f);
                d_h_t[j] = (t == T - 1 ? 0.0f : dh[j]) + d_o;
            }
            ref_activation(act_out, g_gate_a[4], 0, c_t, tc, H);
            vec_mul(d_h_t, tc, d_a_O, H);
            ref_activation_gradient(act_o, 0, Z + 3 * H, ot, d_a_O, dg + 3 * H, H);
            vec_mul(d_h_t, ot, d_a_C, H);
            ref_activation_gradient(act_out, 0, c_t, NULL, d_a_C, d_c_t, H);
            if (t != T - 1) vec_add(d_c_t, dc_carry, d_c_t, H);
            vec_mul(d_c_t, gt, d_a, H);
            ref_activation_gradient(act_i, 0, Z, it, d_a, dg, H);
            if (!c_prev) memset(dg + H, 0, (size_t)H * sizeof(float));
            else { vec_mul(c_prev, d_c_t, d_a, H); ref_activation_gradient(act_f, 0, Z + H, ft, d_a, dg + H, H); }
            vec_mul(d_c_t, it, d_a, H);
            ref_activation_gradient(act_g, 0, Z + 2 * H, gt, d_a, dg + 2 * H, H);
            vec_mul(d_c_t, ft, dc_carry, H);
            ref_op_mat_mul(W, dg, dX + row * in, in, 1, G);
            ref_op_mat_mul(U, dg, dh, H, 1, G);
            ref_op_mat_mul(x + row * in, dg, dW, in, G, 1);
            if (h_prev) ref_op_mat_mul(h_prev, dg, dU, H, G, 1);
            else memset(dU, 0, (size_t)H * G * sizeof(float));
            for (size_t e = 0; e < (size_t)in * G; ++e) gW[e] = gW[e] + dW[e];
            for (size_t e = 0; e < (size_t)H * G; ++e) gU[e] = gU[e] + dU[e];
            for (int e = 0; e < G; ++e) { gbi[e] = gbi[e] + dg[e]; gbh[e] = gbh[e] + dg[e]; }
        }
    free(dW); free(dU); free(dg); free(tmp); free(dh); free(dc_carry);
}

/* layers/rnn.c:249-291 (RNNApplyTrainingBatch) + :184-221, :293-351 (RNNCellBackward, RNNCalculateGradient) */
void ref_rnn_training_forward(const float *x, const float *W, const float *U, const float *b_i, const float *b_h,
                              float *h, float *gate, int B, int T, int in, int H, int v2, int act) {
    float *buf = (float *)calloc((size_t)3 * H, sizeof(float)), *hs = (float *)malloc((size_t)H * sizeof(float));
    for (int b = 0; b < B; ++b) {
        memset(hs, 0, (size_t)H * sizeof(float));
        for (int t = 0; t < T; ++t) {
            size_t row = (size_t)b * T + t;
            rnn_cell(x + row * in, W, U, b_i, b_h, hs, h + row * H, buf, in, H, v2, act);
            memcpy(gate + row * H, buf + 2 * H, (size_t)H * sizeof(float));
            memcpy(hs, h + row * H, (size_t)H * sizeof(float));
        }
    }
    free(buf); free(hs);
}
void ref_rnn_gradient(const float *x, const float *W, const float *U, const float *h, const float *gate, const float *dout,
                      int return_sequences, float *gW, float *gU, float *gbi, float *gbh, float *dX,
                      int B, int T, int in, int H, int act) {
    float *dW = (float *)malloc((size_t)in * H * sizeof(float)), *dU = (float *)malloc((size_t)H * H * sizeof(float));
    float *dg = (float *)malloc((size_t)H * sizeof(float)), *dh = (float *)malloc((size_t)H * sizeof(float));
    float *d_h_t = (float *)malloc((size_t)H * sizeof(float));
    for (int b = 0; b < B; ++b)
        for (int t = T - 1; t >= 0; --t) {
            size_t row = (size_t)b * T + t;
            const float *h_prev = t == 0 ? NULL : h + (row - 1) * H;
            for (int j = 0; j < H; ++j) {
                float d_o = return_sequences ? dout[row * H + j] : (t == T - 1 ? dout[(size_t)b * H + j] : 0.0f);
                d_h_t[j] = (t == T - 1 ? 0.0f : dh[j]) + d_o;
            }
            int cached = act == REF_ACT_SIGMOID || act == REF_ACT_TANH;
            ref_activation_gradient(act, 0, gate + row * H, cached ? h + row * H : NULL, d_h_t, dg, H);
            ref_op_mat_mul(W, dg, dX + row * in, in, 1, H);
            ref_op_mat_mul(U, dg, dh, H, 1, H);
            ref_op_mat_mul(x + row * in, dg, dW, in, H, 1);
            if (h_prev) ref_op_mat_mul(h_prev, dg, dU, H, H, 1);
            else memset(dU, 0, (size_t)H * H * sizeof(float));
            for (size_t e = 0; e < (size_t)in * H; ++e) gW[e] = gW[e] + dW[e];
            for (size_t e = 0; e < (size_t)H * H; ++e) gU[e] = gU[e] + dU[e];
            for (int e = 0; e < H; ++e) { gbi[e] = gbi[e] + dg[e]; gbh[e] = gbh[e] + dg[e]; }
        }
    free(dW); free(dU); free(dg); free(dh); free(d_h_t);
}
