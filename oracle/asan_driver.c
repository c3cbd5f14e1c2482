/* asan_driver.c -- runs every oracle entry point over small and edge-case shapes under AddressSanitizer + UBSan
 * (sanitizers on the CPU side only).  Test infrastructure: built by `make -C oracle asan`, run by
 * tests/test_oracle.py::test_oracle_is_clean_under_asan_and_ubsan.  Exact-size heap buffers, so any out-of-bounds
 * index in the restatement (or a VLA-sized stack overrun like the reference's own spectrogram.c:120) is reported. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nnref.h"

static float *rnd(size_t n, unsigned seed) {
    float *p = (float *)malloc((n ? n : 1) * sizeof(float));
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; p[i] = ((int)(s >> 9) % 2001 - 1000) * 1e-3f; }
    return p;
}
static float *buf(size_t n) { return (float *)malloc((n ? n : 1) * sizeof(float)); }

int main(void) {
    /* windows + fft + spectrogram: geometries from the GPU test matrix, incl. non powers of two and one-frame inputs */
    const int geo[][4] = {{512, 400, 240, 16000}, {512, 512, 0, 5120}, {512, 400, 399, 900}, {256, 200, 120, 8000},
                          {64, 48, 16, 1000}, {60, 45, 15, 777}, {16, 16, 8, 40}, {512, 40, 8, 2000}, {8, 8, 0, 8}};
    for (unsigned g = 0; g < sizeof(geo) / sizeof(geo[0]); ++g) {
        int nfft = geo[g][0], win = geo[g][1], nov = geo[g][2], n = geo[g][3], step, nfreq, nts;
        ref_spectrogram_geometry(nfft, win, nov, n, &step, &nfreq, &nts);
        float *w = buf(win), *x = rnd(n, g), *out = buf((size_t)nts * nfreq);
        for (int kind = 0; kind <= REF_WIN_BLACKMAN; ++kind) ref_window(kind, w, win);
        ref_window(REF_WIN_HANN, w, win);
        for (int mode = 0; mode < 2; ++mode)
            if (ref_spectrogram(x, w, out, nfft, win, nov, n, mode ? 0.5f : 1.0f, mode,
                                mode ? ref_spectrogram_scale_psd(w, win, 16000) : ref_spectrogram_scale_magnitude(w, win))) return 2;
        float *ci = rnd(2 * (size_t)nfft, g + 50), *co = buf(2 * (size_t)nfft);
        if (ref_kiss_fft(nfft, 0, ci, co) || ref_kiss_fft(nfft, 1, co, ci)) return 3;
        free(w); free(x); free(out); free(ci); free(co);
    }
    /* mel */
    { float *wm = buf(257 * 40), *sp = rnd(9 * 257, 3), *lm = buf(9 * 40);
      ref_mel_filterbank_weights(40, 512, 16000, 20.f, 8000.f, wm);
      for (int i = 0; i < 9 * 257; ++i) sp[i] = sp[i] < 0 ? -sp[i] : sp[i];
      ref_log_mel(sp, wm, lm, 9, 257, 40); free(wm); free(sp); free(lm); }
    /* conv / bn / activations: stride > 1, k > T (empty output), ragged channels */
    const int cv[][5] = {{23, 3, 4, 5, 2}, {1000, 40, 128, 5, 1}, {4, 2, 3, 5, 1}, {16, 1, 16, 9, 1}, {7, 5, 1, 7, 3}};
    for (unsigned c = 0; c < sizeof(cv) / sizeof(cv[0]); ++c) {
        int T = cv[c][0], ci = cv[c][1], co = cv[c][2], k = cv[c][3], st = cv[c][4], To = ref_conv1d_output_size(T, k, st);
        if (To < 0) To = 0;
        float *x = rnd((size_t)2 * T * ci, c), *W = rnd((size_t)co * ci * k, c + 9), *b = rnd(co, c + 5), *y = buf((size_t)2 * To * co);
        ref_conv1d(x, W, b, y, T, ci, co, k, st);
        ref_conv1d_batch(x, W, b, y, 2, T, ci, co, k, st);
        float *g = rnd(co, 1), *be = rnd(co, 2), *mu = rnd(co, 3), *va = rnd(co, 4);
        for (int i = 0; i < co; ++i) va[i] = 1.0f + 0.5f * va[i];
        ref_batch_norm(y, g, be, mu, va, y, 1e-3f, 2 * To, co);
        for (int a = 0; a <= REF_ACT_RELU; ++a) ref_activation(a, 0.5f, 0, y, y, 2 * To * co);
        if (To > 0) ref_activation(REF_ACT_SOFTMAX, 1.f, co, y, y, 2 * To);
        free(x); free(W); free(b); free(y); free(g); free(be); free(mu); free(va);
    }
    /* recurrent layers: T = 1, H not a multiple of anything, both return modes, v2 on/off, ReLU gate scales */
    const int rc[][4] = {{1, 5, 7, 1}, {3, 5, 7, 11}, {2, 16, 33, 4}};
    for (unsigned c = 0; c < sizeof(rc) / sizeof(rc[0]); ++c) {
        int B = rc[c][0], I = rc[c][1], H = rc[c][2], T = rc[c][3];
        for (int seq = 0; seq < 2; ++seq) {
            float *x = rnd((size_t)B * T * I, c);
            size_t no = seq ? (size_t)B * T * H : (size_t)B * H;
            { float *W = rnd((size_t)I * 3 * H, 1), *U = rnd((size_t)H * 3 * H, 2), *bi = rnd(3 * H, 3), *bh = rnd(3 * H, 4), *o = buf(no), *h = rnd(H, 5);
              ref_gru_batch(x, W, U, bi, bh, o, B, T, I, H, seq, REF_ACT_SIGMOID, REF_ACT_TANH, REF_ACT_SIGMOID);
              const float sc[3] = {1.f, 0.5f, 1.f}; ref_set_gate_relu_scales(sc, 3);
              ref_gru_sequence(x, W, U, bi, bh, h, o, T, I, H, seq, REF_ACT_SIGMOID, REF_ACT_RELU, REF_ACT_SIGMOID);
              ref_set_gate_relu_scales(NULL, 0);
              free(W); free(U); free(bi); free(bh); free(o); free(h); }
            for (int v2 = 0; v2 < 2; ++v2) {
              float *W = rnd((size_t)I * 4 * H, 1), *U = rnd((size_t)H * 4 * H, 2), *bi = rnd(4 * H, 3), *bh = rnd(4 * H, 4), *o = buf(no), *h = rnd(H, 5), *cs = rnd(H, 6);
              ref_lstm_batch(x, W, U, bi, bh, o, B, T, I, H, seq, v2, REF_ACT_SIGMOID, REF_ACT_SIGMOID, REF_ACT_TANH, REF_ACT_SIGMOID, REF_ACT_TANH);
              ref_lstm_sequence(x, W, U, bi, bh, h, cs, o, T, I, H, seq, v2, REF_ACT_SIGMOID, REF_ACT_SIGMOID, REF_ACT_TANH, REF_ACT_SIGMOID, REF_ACT_TANH);
              float *Wr = rnd((size_t)I * H, 7), *Ur = rnd((size_t)H * H, 8);
              ref_rnn_batch(x, Wr, Ur, bi, bh, o, B, T, I, H, seq, v2, REF_ACT_TANH);
              ref_rnn_sequence(x, Wr, Ur, bi, bh, h, o, T, I, H, seq, v2, REF_ACT_RELU);
              free(W); free(U); free(bi); free(bh); free(o); free(h); free(cs); free(Wr); free(Ur); }
            free(x);
        }
    }
    /* dense / tdd incl. softmax(1, V), bidirectional helpers */
    { int ts = 9, I = 5, V = 7;
      float *x = rnd((size_t)ts * I, 1), *W = rnd((size_t)I * V, 2), *b = rnd(V, 3), *y = buf((size_t)ts * V);
      ref_dense(x, W, b, y, I, V, -1, 1.f, 0, V);
      ref_time_distributed_dense(x, W, b, y, ts, I, V, -1, 1.f, 0, V);
      ref_time_distributed_dense(x, W, b, y, ts, I, V, REF_ACT_SOFTMAX, 1.f, V, 1);
      ref_time_distributed_dense(x, W, b, y, ts, I, V, REF_ACT_RELU, 0.25f, 0, V);
      float *r = buf((size_t)ts * V), *cat = buf((size_t)2 * ts * V);
      ref_bd_reverse_batch(y, r, 3, 3, V); ref_bd_merge_concat(y, r, cat, 3, 3, V); ref_bd_merge_sum(y, r, cat, 3, 3, V);
      float *t = buf((size_t)I * V); ref_op_mat_transp(W, t, V, I); (void)ref_op_vec_dot(W, W, I * V);
      free(x); free(W); free(b); free(y); free(r); free(cat); free(t); }
    /* training restatements: activation gradients, dense / batch-norm gradients, losses, SGD, GRU / LSTM / RNN BPTT
     * (exact-size buffers; T = 1 and return_sequences = 0 included) */
    { int B = 3, I = 5, V = 6;
      float *x = rnd((size_t)B * I, 1), *W = rnd((size_t)I * V, 2), *z = rnd((size_t)B * V, 3), *a = buf((size_t)B * V), *d = rnd((size_t)B * V, 4);
      float *gW = rnd((size_t)I * V, 5), *gb = rnd(V, 6), *dX = buf((size_t)B * I), *o = buf((size_t)B * V);
      const int kinds[] = {REF_ACT_IDENTITY, REF_ACT_SIGMOID, REF_ACT_TANH, REF_ACT_RELU};
      for (int k = 0; k < 4; ++k) {
          ref_activation(kinds[k], 1.f, 0, z, a, B * V);
          ref_activation_gradient(kinds[k], 0, z, a, d, o, B * V);
          ref_activation_gradient(kinds[k], 0, z, NULL, d, o, B * V);
          ref_dense_gradient(x, W, z, a, d, kinds[k], 0, V, gW, gb, dX, B, I, V);
      }
      ref_activation(REF_ACT_SOFTMAX, 1.f, V, z, a, B);
      ref_activation_gradient(REF_ACT_SOFTMAX, V, z, a, d, o, B);
      ref_activation_gradient(REF_ACT_SOFTMAX, V, z, NULL, d, o, B);
      ref_dense_gradient(x, W, z, a, d, REF_ACT_SOFTMAX, V, 1, gW, gb, dX, B, I, V);
      ref_dense_gradient(x, W, z, a, d, -1, 0, V, gW, gb, dX, B, I, V);
      (void)ref_mean_squared_error(a, z, V, B); ref_mean_squared_error_derivative(a, z, o, V, B);
      (void)ref_categorical_crossentropy(a, a, V, B); ref_categorical_crossentropy_derivative(a, a, o, V, B);
      ref_sgd_optimize(0.1f, gW, W, I * V);
      float *mean = buf(V), *var = buf(V), *mm = rnd(V, 7), *mv = rnd(V, 8), *g = rnd(V, 9), *be = rnd(V, 10);
      for (int i = 0; i < V; ++i) mv[i] = mv[i] * mv[i] + 0.1f;
      ref_batch_norm_training_forward(z, g, be, 1e-3f, 0.9f, o, mean, var, mm, mv, B, V);
      ref_batch_norm_gradient(z, d, g, mean, var, 1e-3f, gb, be, a, B, V);
      free(x); free(W); free(z); free(a); free(d); free(gW); free(gb); free(dX); free(o); free(mean); free(var); free(mm); free(mv); free(g); free(be); }
    for (int T = 1; T <= 4; T += 3)
        for (int seq = 0; seq < 2; ++seq) {
            int B = 2, I = 3, H = 4;
            size_t rows = (size_t)B * T;
            float *x = rnd(rows * I, 1), *dout = rnd(seq ? rows * H : (size_t)B * H, 2), *dX = buf(rows * I);
            float *h = buf(rows * H), *c = buf(rows * H), *Z = buf(rows * 8 * H), *hU = buf(rows * H);
            for (int G = 1; G <= 4; G += (G == 1 ? 2 : 1)) {
                float *W = rnd((size_t)I * G * H, 3), *U = rnd((size_t)H * G * H, 4), *bi = rnd((size_t)G * H, 5), *bh = rnd((size_t)G * H, 6);
                float *gW = rnd((size_t)I * G * H, 7), *gU = rnd((size_t)H * G * H, 8), *gbi = rnd((size_t)G * H, 9), *gbh = rnd((size_t)G * H, 10);
                if (G == 3) {
                    ref_gru_training_forward(x, W, U, bi, bh, h, Z, hU, B, T, I, H, REF_ACT_SIGMOID, REF_ACT_TANH, REF_ACT_SIGMOID);
                    ref_gru_gradient(x, W, U, h, Z, hU, dout, seq, gW, gU, gbi, gbh, dX, B, T, I, H, REF_ACT_SIGMOID, REF_ACT_TANH, REF_ACT_SIGMOID);
                } else if (G == 4) {
                    ref_lstm_training_forward(x, W, U, bi, bh, h, c, Z, B, T, I, H, seq, REF_ACT_SIGMOID, REF_ACT_SIGMOID, REF_ACT_TANH, REF_ACT_SIGMOID, REF_ACT_TANH);
                    ref_lstm_gradient(x, W, U, h, c, Z, dout, seq, gW, gU, gbi, gbh, dX, B, T, I, H, REF_ACT_SIGMOID, REF_ACT_SIGMOID, REF_ACT_TANH, REF_ACT_SIGMOID, REF_ACT_RELU);
                } else {
                    ref_rnn_training_forward(x, W, U, bi, bh, h, hU, B, T, I, H, seq, REF_ACT_TANH);
                    ref_rnn_gradient(x, W, U, h, hU, dout, seq, gW, gU, gbi, gbh, dX, B, T, I, H, REF_ACT_TANH);
                }
                free(W); free(U); free(bi); free(bh); free(gW); free(gU); free(gbi); free(gbh);
            }
            free(x); free(dout); free(dX); free(h); free(c); free(Z); free(hU);
        }
    puts("oracle asan driver: ok");
    return 0;
}
