/* A plain C application written ONLY against the reference's public API and include paths
 * (conv_1d.h, batch_norm.h, activation_default.h, gru.h, time_distributed_dense.h,
 * spectrogram.h) -- the way README.md:25 says the library is used.  It is compiled and
 * linked against libnntoolkitcore_hip.so unchanged by tests/test_c_dropin.py, which checks
 * its outputs against the CPU oracle.  usage: dropin_caller <dir>  (reads *.bin, writes out_*.bin) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nntoolkitcore/signal/spectrogram.h"
#include "nntoolkitcore/layers/conv_1d.h"
#include "nntoolkitcore/layers/batch_norm.h"
#include "nntoolkitcore/layers/activation_default.h"
#include "nntoolkitcore/layers/gru.h"
#include "nntoolkitcore/layers/time_distributed_dense.h"

static float *rd(const char *dir, const char *name, size_t n) {
    char path[512];
    snprintf(path, sizeof path, "%s/%s.bin", dir, name);
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    float *p = malloc(n * sizeof(float));
    if (fread(p, sizeof(float), n, f) != n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    return p;
}
static void wr(const char *dir, const char *name, const float *p, size_t n) {
    char path[512];
    snprintf(path, sizeof path, "%s/%s.bin", dir, name);
    FILE *f = fopen(path, "wb");
    fwrite(p, sizeof(float), n, f);
    fclose(f);
}

int main(int argc, char **argv) {
    const char *dir = argc > 1 ? argv[1] : ".";
    enum { N = 4240, NFFT = 512, WIN = 400, NOV = 240, C1 = 32, K = 5, H = 48, V = 10 };

    /* audio -> magnitude spectrogram with a hann window */
    SpectrogramConfig scfg = SpectrogramConfigCreate(NFFT, WIN, NOV, N, 1.0f);
    Spectrogram spec = SpectrogramCreateMagnitude(scfg);
    SpectrogramSetWindowFunc(spec, hann_window);
    const int T = scfg.ntime_series, F = scfg.nfreq;               /* 25 x 257 */
    float *audio = rd(dir, "audio", 2 * N);
    float *s = malloc(sizeof(float) * T * F);

    Conv1dConfig ccfg = Conv1dConfigCreate(F, C1, K, 1, T);
    Conv1d conv = Conv1dCreateForInference(ccfg);
    const int Tc = ccfg.output_size;
    ConvWeights *cw = Conv1dGetWeights(conv);
    float *Wc = rd(dir, "conv_W", (size_t)C1 * F * K), *bc = rd(dir, "conv_b", C1);
    memcpy(cw->W, Wc, sizeof(float) * C1 * F * K);
    memcpy(cw->b, bc, sizeof(float) * C1);

    BatchNorm bn = BatchNormCreateForInference(BatchNormConfigCreate(C1, 1e-3f, Tc));
    BatchNormWeights *bw = BatchNormGetWeights(bn);
    float *bnw = rd(dir, "bn", 4 * C1);
    memcpy(bw->gamma, bnw, sizeof(float) * C1);
    memcpy(bw->beta, bnw + C1, sizeof(float) * C1);
    memcpy(bw->moving_mean, bnw + 2 * C1, sizeof(float) * C1);
    memcpy(bw->moving_variance, bnw + 3 * C1, sizeof(float) * C1);
    ActivationFunction relu = ActivationFunctionCreateReLU(Tc * C1, 1.0f);

    GRUConfig gcfg = GRUConfigCreate(C1, H, true, Tc, GRUActivationsCreateDefault(H));
    GRU gru = GRUCreateForInference(gcfg);
    GRUWeights *gw = GRUGetWeights(gru);
    float *gW = rd(dir, "gru_W", C1 * 3 * H), *gU = rd(dir, "gru_U", H * 3 * H);
    float *gbi = rd(dir, "gru_bi", 3 * H), *gbh = rd(dir, "gru_bh", 3 * H);
    memcpy(gw->W, gW, sizeof(float) * C1 * 3 * H);
    memcpy(gw->U, gU, sizeof(float) * H * 3 * H);
    memcpy(gw->b_i, gbi, sizeof(float) * 3 * H);
    memcpy(gw->b_h, gbh, sizeof(float) * 3 * H);

    TimeDistributedDense tdd = TimeDistributedDenseCreateForInference(
        TimeDistributedDenseConfigCreate(Tc, DenseConfigCreate(H, V, ActivationFunctionCreateSoftmax(1, V))));
    DenseWeights *dw = TimeDistributedDenseGetWeights(tdd);
    float *dW = rd(dir, "tdd_W", H * V), *db = rd(dir, "tdd_b", V);
    memcpy(dw->W, dW, sizeof(float) * H * V);
    memcpy(dw->b, db, sizeof(float) * V);

    float *c = malloc(sizeof(float) * Tc * C1), *h = malloc(sizeof(float) * Tc * H), *y = malloc(sizeof(float) * 2 * Tc * V);
    for (int chunk = 0; chunk < 2; ++chunk) {           /* two consecutive chunks: GRU state carries over */
        SpectrogramApply(spec, audio + chunk * N, s);
        if (Conv1dApplyInference(conv, s, c) != 0) return 3;
        if (BatchNormApplyInference(bn, c, c) != 0) return 4;
        ActivationFunctionApply(relu, c, c);
        if (GRUApplyInference(gru, c, h) != 0) return 5;
        if (TimeDistributedDenseApplyInference(tdd, h, y + chunk * Tc * V) != 0) return 6;
    }
    wr(dir, "out_y", y, (size_t)2 * Tc * V);
    printf("T=%d F=%d Tc=%d\n", T, F, Tc);

    TimeDistributedDenseDestroy(tdd);
    GRUDestroy(gru);
    GRUActivationsDestroy(gcfg.activations);
    ActivationFunctionDestroy(relu);
    BatchNormDestroy(bn);
    Conv1dDestroy(conv);
    SpectrogramDestroy(spec);
    return 0;
}
