/*
 * oracle/ref_probe.c -- TEST INFRASTRUCTURE.  Runs the parts of the REAL
 * reference that build in this image without any stand-in, and prints what they
 * return as JSON (committed as tests/golden/ref_probe.json).
 *
 * What can be built honestly: every reference .c file compiles, but the compute
 * backend (core/default_ops.cc -> Eigen, signal/dft.c -> kissfft) does not, so
 * oracle/_ref/libnnref_partial.so is linked WITHOUT those two files and with
 * op_* / DFT* left undefined.  This probe dlopen()s it RTLD_LAZY and calls only
 * functions whose call graph never reaches an undefined symbol:
 *   - window functions                      (signal/window.c)
 *   - *ConfigCreate geometry                (spectrogram.c:59, conv_1d.c:77, ...)
 *   - *CreateForInference + *GetWeights     (weight-block layout, zero init)
 * It also prints sizeof/offsetof of the by-value config structs from the
 * reference's own headers (the drop-in ABI).
 *
 * Build + run: `make -C oracle ref` (needs /root/reference; outputs only under
 * oracle/_ref/).
 */
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#include "nntoolkitcore/signal/spectrogram.h"
#include "nntoolkitcore/signal/window.h"
#include "nntoolkitcore/layers/conv_1d.h"
#include "nntoolkitcore/layers/batch_norm.h"
#include "nntoolkitcore/layers/gru.h"
#include "nntoolkitcore/layers/lstm.h"
#include "nntoolkitcore/layers/rnn.h"
#include "nntoolkitcore/layers/bidirectional.h"
#include "nntoolkitcore/layers/dense.h"
#include "nntoolkitcore/layers/time_distributed_dense.h"
#include "nntoolkitcore/layers/activation_default.h"
#include "nntoolkitcore/signal/mel_filterbank.h"

static void *lib;
#define SYM(type, name) type name##_p = (type)dlsym(lib, #name); \
    if (!name##_p) { fprintf(stderr, "missing %s\n", #name); return 2; }

static void print_floats(const char *key, const float *v, int n, int last) {
    printf("  \"%s\": [", key);
    for (int i = 0; i < n; ++i) printf("%s%.9g", i ? ", " : "", v[i]);
    printf("]%s\n", last ? "" : ",");
}

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "oracle/_ref/libnnref_partial.so";
    lib = dlopen(path, RTLD_LAZY | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }

    typedef SpectrogramConfig (*spec_cfg_fn)(int, int, int, int, float);
    typedef Conv1dConfig (*conv_cfg_fn)(int, int, int, int, int);
    typedef void (*win_fn)(float *, int);
    SYM(spec_cfg_fn, SpectrogramConfigCreate)
    SYM(conv_cfg_fn, Conv1dConfigCreate)
    SYM(win_fn, hann_window) SYM(win_fn, hamming_window) SYM(win_fn, ones)
    SYM(win_fn, periodic_hann_window) SYM(win_fn, periodic_hamming_window) SYM(win_fn, blackman_window)

    typedef Conv1d (*conv_create_fn)(Conv1dConfig);
    typedef ConvWeights *(*conv_w_fn)(Conv1d);
    typedef void (*conv_destroy_fn)(Conv1d);
    SYM(conv_create_fn, Conv1dCreateForInference) SYM(conv_w_fn, Conv1dGetWeights) SYM(conv_destroy_fn, Conv1dDestroy)

    typedef GRUActivations (*gru_act_fn)(int);
    typedef GRUConfig (*gru_cfg_fn)(int, int, bool, int, GRUActivations);
    typedef GRU (*gru_create_fn)(GRUConfig);
    typedef GRUWeights *(*gru_w_fn)(GRU);
    SYM(gru_act_fn, GRUActivationsCreateDefault) SYM(gru_cfg_fn, GRUConfigCreate)
    SYM(gru_create_fn, GRUCreateForInference) SYM(gru_w_fn, GRUGetWeights)

    typedef LSTMActivations (*lstm_act_fn)(int);
    typedef LSTMConfig (*lstm_cfg_fn)(int, int, bool, int, bool, LSTMActivations);
    typedef LSTM (*lstm_create_fn)(LSTMConfig);
    typedef LSTMWeights *(*lstm_w_fn)(LSTM);
    SYM(lstm_act_fn, LSTMActivationsCreateDefault) SYM(lstm_cfg_fn, LSTMConfigCreate)
    SYM(lstm_create_fn, LSTMCreateForInference) SYM(lstm_w_fn, LSTMGetWeights)

    typedef RNNConfig (*rnn_cfg_fn)(int, int, bool, int, bool, ActivationFunction);
    typedef RNN (*rnn_create_fn)(RNNConfig);
    typedef RNNWeights *(*rnn_w_fn)(RNN);
    typedef void (*bd_rev_fn)(const float *, float *, RecurrentConfig, int);
    typedef int (*bd_size_fn)(RecurrentConfig);
    SYM(rnn_cfg_fn, RNNConfigCreate) SYM(rnn_create_fn, RNNCreateForInference) SYM(rnn_w_fn, RNNGetWeights)
    SYM(bd_rev_fn, bd_reverse_input_batch) SYM(bd_rev_fn, bd_reverse_backward_batch) SYM(bd_size_fn, bd_merge_concat_buffer_size)

    typedef BatchNormConfig (*bn_cfg_fn)(int, float, int);
    typedef BatchNorm (*bn_create_fn)(BatchNormConfig);
    typedef BatchNormWeights *(*bn_w_fn)(BatchNorm);
    SYM(bn_cfg_fn, BatchNormConfigCreate) SYM(bn_create_fn, BatchNormCreateForInference) SYM(bn_w_fn, BatchNormGetWeights)

    typedef DenseConfig (*dense_cfg_fn)(int, int, ActivationFunction);
    typedef TimeDistributedDenseConfig (*tdd_cfg_fn)(int, DenseConfig);
    typedef TimeDistributedDense (*tdd_create_fn)(TimeDistributedDenseConfig);
    typedef DenseWeights *(*tdd_w_fn)(TimeDistributedDense);
    SYM(dense_cfg_fn, DenseConfigCreate) SYM(tdd_cfg_fn, TimeDistributedDenseConfigCreate)
    SYM(tdd_create_fn, TimeDistributedDenseCreateForInference) SYM(tdd_w_fn, TimeDistributedDenseGetWeights)

    printf("{\n");

    /* ---- ABI: sizes and offsets of by-value structs ---- */
    printf("  \"abi\": {\n");
#define SZ(T) printf("    \"sizeof_%s\": %zu,\n", #T, sizeof(T))
#define OFF(T, f) printf("    \"offsetof_%s_%s\": %zu,\n", #T, #f, offsetof(T, f))
    SZ(SpectrogramConfig); OFF(SpectrogramConfig, nfft); OFF(SpectrogramConfig, window_size);
    OFF(SpectrogramConfig, noverlap); OFF(SpectrogramConfig, step); OFF(SpectrogramConfig, input_size);
    OFF(SpectrogramConfig, nfreq); OFF(SpectrogramConfig, ntime_series); OFF(SpectrogramConfig, fft_normalization_factor);
    SZ(Conv1dConfig); OFF(Conv1dConfig, input_feature_channels); OFF(Conv1dConfig, output_feature_channels);
    OFF(Conv1dConfig, kernel_size); OFF(Conv1dConfig, stride); OFF(Conv1dConfig, input_size); OFF(Conv1dConfig, output_size);
    SZ(BatchNormConfig); OFF(BatchNormConfig, feature_channels); OFF(BatchNormConfig, epsilon); OFF(BatchNormConfig, count);
    SZ(RecurrentConfig); OFF(RecurrentConfig, input_feature_channels); OFF(RecurrentConfig, output_feature_channels);
    OFF(RecurrentConfig, return_sequences); OFF(RecurrentConfig, timesteps);
    SZ(GRUActivations); OFF(GRUActivations, z_gate_activation); OFF(GRUActivations, h_gate_activation); OFF(GRUActivations, r_gate_activation);
    SZ(GRUConfig); OFF(GRUConfig, base); OFF(GRUConfig, activations);
    SZ(LSTMActivations); OFF(LSTMActivations, candidate_gate_activation); OFF(LSTMActivations, input_gate_activation);
    OFF(LSTMActivations, forget_gate_activation); OFF(LSTMActivations, output_gate_activation); OFF(LSTMActivations, output_activation);
    SZ(LSTMConfig); OFF(LSTMConfig, base); OFF(LSTMConfig, v2); OFF(LSTMConfig, activations);
    SZ(RNNConfig); OFF(RNNConfig, base); OFF(RNNConfig, v2); OFF(RNNConfig, activation);
    SZ(DenseConfig); OFF(DenseConfig, input_size); OFF(DenseConfig, output_size); OFF(DenseConfig, activation);
    SZ(TimeDistributedDenseConfig); OFF(TimeDistributedDenseConfig, dense); OFF(TimeDistributedDenseConfig, ts);
    SZ(MelFilterBankConfig); OFF(MelFilterBankConfig, n_mels); OFF(MelFilterBankConfig, n_fft); OFF(MelFilterBankConfig, sample_rate);
    OFF(MelFilterBankConfig, lower_hz); OFF(MelFilterBankConfig, upper_hz);
    SZ(DefaultWeights); SZ(RecurrentWeights); SZ(BatchNormWeights);
    printf("    \"end\": 0\n  },\n");

    /* ---- geometry ---- */
    static const int spec_cases[][4] = { /* nfft, win, noverlap, input_size */
        {512, 400, 240, 16000}, {512, 400, 240, 160240}, {256, 200, 120, 8000}, {64, 48, 16, 1000},
        {512, 512, 0, 5120}, {128, 100, 99, 500}, {16, 16, 8, 40}, {60, 45, 15, 777} };
    printf("  \"spectrogram_config\": [\n");
    int ns = (int)(sizeof(spec_cases) / sizeof(spec_cases[0]));
    for (int i = 0; i < ns; ++i) {
        SpectrogramConfig c = SpectrogramConfigCreate_p(spec_cases[i][0], spec_cases[i][1], spec_cases[i][2], spec_cases[i][3], 1.0f);
        printf("    {\"nfft\": %d, \"window_size\": %d, \"noverlap\": %d, \"input_size\": %d, \"step\": %d, \"nfreq\": %d, \"ntime_series\": %d}%s\n",
               c.nfft, c.window_size, c.noverlap, c.input_size, c.step, c.nfreq, c.ntime_series, i + 1 < ns ? "," : "");
    }
    printf("  ],\n");
    static const int conv_cases[][5] = { /* Cin, Cout, k, stride, T */
        {1, 16, 9, 1, 16000}, {40, 128, 5, 1, 1000}, {257, 128, 5, 1, 1000}, {3, 4, 5, 2, 23},
        {2, 3, 3, 3, 30}, {5, 7, 1, 1, 11}, {4, 4, 7, 4, 50}, {8, 2, 4, 2, 9} };
    printf("  \"conv1d_config\": [\n");
    int nc = (int)(sizeof(conv_cases) / sizeof(conv_cases[0]));
    for (int i = 0; i < nc; ++i) {
        Conv1dConfig c = Conv1dConfigCreate_p(conv_cases[i][0], conv_cases[i][1], conv_cases[i][2], conv_cases[i][3], conv_cases[i][4]);
        printf("    {\"cin\": %d, \"cout\": %d, \"k\": %d, \"stride\": %d, \"input_size\": %d, \"output_size\": %d}%s\n",
               c.input_feature_channels, c.output_feature_channels, c.kernel_size, c.stride, c.input_size, c.output_size,
               i + 1 < nc ? "," : "");
    }
    printf("  ],\n");

    /* ---- weight-block layouts (float offsets from the block base) + zero init ---- */
    {
        Conv1d f = Conv1dCreateForInference_p(Conv1dConfigCreate_p(40, 128, 5, 1, 1000));
        ConvWeights *w = Conv1dGetWeights_p(f);
        int zero = 1;
        for (int i = 0; i < 40 * 128 * 5 + 128; ++i) zero &= (w->W[i] == 0.0f);
        printf("  \"conv1d_weights\": {\"b_offset\": %td, \"all_zero\": %d},\n", w->b - w->W, zero);
        Conv1dDestroy_p(f);
    }
    {
        GRU g = GRUCreateForInference_p(GRUConfigCreate_p(5, 7, true, 11, GRUActivationsCreateDefault_p(7)));
        GRUWeights *w = GRUGetWeights_p(g);
        int zero = 1;
        for (int i = 0; i < 5 * 21 + 7 * 21 + 42; ++i) zero &= (w->W[i] == 0.0f);
        printf("  \"gru_weights\": {\"in\": 5, \"out\": 7, \"U_offset\": %td, \"b_i_offset\": %td, \"b_h_offset\": %td, \"all_zero\": %d},\n",
               w->U - w->W, w->b_i - w->W, w->b_h - w->W, zero);
    }
    {
        LSTM l = LSTMCreateForInference_p(LSTMConfigCreate_p(5, 7, true, 11, true, LSTMActivationsCreateDefault_p(7)));
        LSTMWeights *w = LSTMGetWeights_p(l);
        printf("  \"lstm_weights\": {\"in\": 5, \"out\": 7, \"U_offset\": %td, \"b_i_offset\": %td, \"b_h_offset\": %td},\n",
               w->U - w->W, w->b_i - w->W, w->b_h - w->W);
    }
    {
        BatchNorm b = BatchNormCreateForInference_p(BatchNormConfigCreate_p(6, 1e-3f, 10));
        BatchNormWeights *w = BatchNormGetWeights_p(b);
        int zero = 1;
        for (int i = 0; i < 24; ++i) zero &= (w->gamma[i] == 0.0f);
        printf("  \"batch_norm_weights\": {\"C\": 6, \"beta_offset\": %td, \"mean_offset\": %td, \"var_offset\": %td, \"all_zero\": %d},\n",
               w->beta - w->gamma, w->moving_mean - w->gamma, w->moving_variance - w->gamma, zero);
    }
    {
        TimeDistributedDense t = TimeDistributedDenseCreateForInference_p(
            TimeDistributedDenseConfigCreate_p(9, DenseConfigCreate_p(5, 7, NULL)));
        DenseWeights *w = TimeDistributedDenseGetWeights_p(t);
        printf("  \"tdd_weights\": {\"in\": 5, \"out\": 7, \"b_offset\": %td},\n", w->b - w->W);
    }

    /* ---- RNN weight block layout and the op-free bidirectional helpers ---- */
    {
        RNN r = RNNCreateForInference_p(RNNConfigCreate_p(5, 7, true, 11, true, NULL));
        RNNWeights *w = RNNGetWeights_p(r);
        printf("  \"rnn_weights\": {\"in\": 5, \"out\": 7, \"U_offset\": %td, \"b_i_offset\": %td, \"b_h_offset\": %td},\n",
               w->U - w->W, w->b_i - w->W, w->b_h - w->W);
        /* batch 2, 3 timesteps, 2 input / 3 output channels, values 0, 1, 2, ... */
        RecurrentConfig rc = {2, 3, true, 3};
        float in[12], out_in[12], bw[18], out_bw[18];
        for (int i = 0; i < 12; ++i) in[i] = (float)i;
        for (int i = 0; i < 18; ++i) bw[i] = (float)i;
        bd_reverse_input_batch_p(in, out_in, rc, 2);
        bd_reverse_backward_batch_p(bw, out_bw, rc, 2);
        printf("  \"bidirectional\": {\n  ");
        print_floats("reverse_input_B2_T3_F2", out_in, 12, 0);
        printf("  ");
        print_floats("reverse_backward_B2_T3_F3", out_bw, 18, 0);
        RecurrentConfig last = {2, 3, false, 3};
        printf("    \"concat_buffer_size_seq\": %d, \"concat_buffer_size_last\": %d\n  },\n",
               bd_merge_concat_buffer_size_p(rc), bd_merge_concat_buffer_size_p(last));
    }

    /* ---- round 2: the remaining op-free behaviour of the hot path's boundary ---- */
    {
        /* (1) wrong-mode handles: *ApplyInference on a handle made by *CreateForTraining returns -1 before any
         *     arithmetic (conv_1d.c:150-152, gru.c:190-192, lstm.c:242-244, dense.c:136-138, batch_norm.c:167-169) */
        typedef Conv1d (*conv_train_fn)(Conv1dConfig, ConvTrainingConfig);
        typedef int (*conv_apply_fn)(Conv1d, const float *, float *);
        typedef GRU (*gru_train_fn)(GRUConfig, GRUTrainingConfig);
        typedef int (*gru_apply_fn)(GRU, const float *, float *);
        typedef LSTM (*lstm_train_fn)(LSTMConfig, LSTMTrainingConfig);
        typedef int (*lstm_apply_fn)(LSTM, const float *, float *);
        typedef Dense (*dense_train_fn)(DenseConfig, DenseTrainingConfig);
        typedef int (*dense_apply_fn)(Dense, const float *, float *);
        typedef BatchNorm (*bn_train_fn)(BatchNormConfig, BatchNormTrainingConfig);
        typedef int (*bn_apply_fn)(BatchNorm, const float *, float *);
        SYM(conv_train_fn, Conv1dCreateForTraining) SYM(conv_apply_fn, Conv1dApplyInference)
        SYM(gru_train_fn, GRUCreateForTraining) SYM(gru_apply_fn, GRUApplyInference)
        SYM(lstm_train_fn, LSTMCreateForTraining) SYM(lstm_apply_fn, LSTMApplyInference)
        SYM(dense_train_fn, DenseCreateForTraining) SYM(dense_apply_fn, DenseApplyInference)
        SYM(bn_train_fn, BatchNormCreateForTraining) SYM(bn_apply_fn, BatchNormApplyInference)
        float in[64] = {0}, out[64] = {0};
        Conv1d c = Conv1dCreateForTraining_p(Conv1dConfigCreate_p(2, 3, 2, 1, 8), (ConvTrainingConfig){2});
        GRU g = GRUCreateForTraining_p(GRUConfigCreate_p(2, 3, true, 2, GRUActivationsCreateDefault_p(3)), (GRUTrainingConfig){2});
        LSTM l = LSTMCreateForTraining_p(LSTMConfigCreate_p(2, 3, true, 2, true, LSTMActivationsCreateDefault_p(3)), (LSTMTrainingConfig){2});
        Dense d = DenseCreateForTraining_p(DenseConfigCreate_p(2, 3, NULL), (DenseTrainingConfig){2});
        BatchNormTrainingConfig btc; btc.momentum = 0.9f; btc.mini_batch_size = 2;
        BatchNorm bnh = BatchNormCreateForTraining_p(BatchNormConfigCreate_p(3, 1e-3f, 2), btc);
        printf("  \"wrong_mode_apply_inference\": {\"conv1d\": %d, \"gru\": %d, \"lstm\": %d, \"dense\": %d, \"batch_norm\": %d},\n",
               Conv1dApplyInference_p(c, in, out), GRUApplyInference_p(g, in, out), LSTMApplyInference_p(l, in, out),
               DenseApplyInference_p(d, in, out), BatchNormApplyInference_p(bnh, in, out));
    }
    {
        /* (1b) gradient block of the training path's first slice (conv_1d.c:157-161, weights_private.c:29-36) */
        typedef ConvGradient *(*conv_grad_fn)(Conv1dConfig, ConvTrainingConfig);
        SYM(conv_grad_fn, Conv1dCreateGradient)
        Conv1dConfig cc = Conv1dConfigCreate_p(3, 4, 5, 2, 23);
        ConvGradient *g = Conv1dCreateGradient_p(cc, (ConvTrainingConfig){2});
        int zero = 1;
        for (int i = 0; i < 4 * 3 * 5 + 4 + 2 * 23 * 3; ++i) zero &= g->d_W[i] == 0.0f;
        printf("  \"conv1d_gradient_block\": {\"d_b_offset\": %td, \"d_X_offset\": %td, \"all_zero\": %d},\n", g->d_b - g->d_W, g->d_X - g->d_W, zero);
    }
    {
        /* (2) activation handles: identity copies exactly the size given at create (activation.c:23-25,
         *     activation_default.c:98-103); a custom handle's callback receives (implementer, input, output, size) */
        typedef ActivationFunction (*act_id_fn)(int);
        typedef ActivationFunction (*act_create_fn)(int, ActivationImplementerDestroy, void *, ActivationFunctionImpl,
                                                    ActivationFunctionDerivative, ActivationFunctionDerivative);
        typedef void (*act_apply_fn)(ActivationFunction, const float *, float *);
        typedef void (*act_destroy_fn)(ActivationFunction);
        SYM(act_id_fn, ActivationFunctionCreateIdentity) SYM(act_create_fn, ActivationFunctionCreate)
        SYM(act_apply_fn, ActivationFunctionApply) SYM(act_destroy_fn, ActivationFunctionDestroy)
        float in[8] = {1, 2, 3, 4, 5, 6, 7, 8}, out[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        ActivationFunction id = ActivationFunctionCreateIdentity_p(5);
        ActivationFunctionApply_p(id, in, out);
        printf("  \"identity_size5_on_8\": [%g, %g, %g, %g, %g, %g, %g, %g],\n", out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7]);
        static struct { void *impl; const float *in; float *out; int size; int destroyed; } seen;
        int token = 42;
        void cb(void *impl, const float *i, float *o, int n) { seen.impl = impl; seen.in = i; seen.out = o; seen.size = n; }
        void dtor(void *p) { seen.destroyed = (p == seen.impl); }
        ActivationFunction cu = ActivationFunctionCreate_p(7, dtor, &token, cb, NULL, NULL);
        ActivationFunctionApply_p(cu, in, out);
        ActivationFunctionDestroy_p(cu);
        printf("  \"custom_activation_callback\": {\"implementer_passed\": %d, \"input_passed\": %d, \"output_passed\": %d, \"size\": %d, \"destroy_called_with_implementer\": %d},\n",
               seen.impl == (void *)&token, seen.in == in, seen.out == out, seen.size, seen.destroyed);
    }
    {
        /* (3) mel filter bank config (mel_filterbank.c:32-41): by-value struct, field order is ABI */
        typedef MelFilterBankConfig (*mel_cfg_fn)(int, int, int, float, float);
        SYM(mel_cfg_fn, MelFilterBankConfigCreate)
        MelFilterBankConfig m = MelFilterBankConfigCreate_p(40, 512, 16000, 20.0f, 8000.0f);
        printf("  \"mel_config\": {\"n_mels\": %d, \"n_fft\": %d, \"sample_rate\": %d, \"lower_hz\": %g, \"upper_hz\": %g, \"sizeof\": %zu},\n",
               m.n_mels, m.n_fft, m.sample_rate, m.lower_hz, m.upper_hz, sizeof(MelFilterBankConfig));
    }

    /* ---- windows (size 16 and the first/last 8 taps of size 400) ---- */
    float w16[16], w400[400];
    struct { const char *name; win_fn fn; } wins[] = {
        {"ones", ones_p}, {"hann", hann_window_p}, {"hamming", hamming_window_p},
        {"periodic_hann", periodic_hann_window_p}, {"periodic_hamming", periodic_hamming_window_p},
        {"blackman", blackman_window_p} };
    printf("  \"windows16\": {\n");
    for (int i = 0; i < 6; ++i) {
        wins[i].fn(w16, 16);
        printf("  ");
        print_floats(wins[i].name, w16, 16, i == 5);
    }
    printf("  },\n  \"windows400\": {\n");
    for (int i = 0; i < 6; ++i) {
        wins[i].fn(w400, 400);
        printf("  ");
        print_floats(wins[i].name, w400, 400, i == 5);
    }
    printf("  }\n}\n");
    return 0;
}
