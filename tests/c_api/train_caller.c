/* A training loop written against the reference's own API (gru.h, dense.h, activation.h, train/loss.h,
 * train/optimizers.h): GRU(return_sequences = false) -> Dense + softmax -> categorical cross-entropy, plain SGD on the
 * handles' weight blocks.  Compiled unchanged against libnntoolkitcore_hip.so (tests/test_c_dropin.py); prints the loss
 * per step and exits non-zero unless it falls by 40 %. */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "nntoolkitcore/layers/gru.h"
#include "nntoolkitcore/layers/dense.h"
#include "nntoolkitcore/train/loss.h"
#include "nntoolkitcore/train/optimizers.h"

static unsigned s = 12345u;
static float frand(void) { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) % 2001 - 1000) * 1e-3f; }

int main(void) {
    enum { B = 32, T = 12, IN = 6, H = 16, C = 3, STEPS = 40 };
    GRUConfig gcfg = GRUConfigCreate(IN, H, false, T, GRUActivationsCreateDefault(H));
    GRU gru = GRUCreateForTraining(gcfg, (GRUTrainingConfig){B});
    ActivationFunction sm = ActivationFunctionCreateSoftmax(1, C);
    DenseConfig dcfg = DenseConfigCreate(H, C, sm);
    Dense head = DenseCreateForTraining(dcfg, (DenseTrainingConfig){B});
    GRUWeights *gw = GRUGetWeights(gru);
    DenseWeights *dw = DenseGetWeights(head);
    const int n_gru = IN * 3 * H + H * 3 * H + 6 * H, n_head = H * C + C;
    for (int i = 0; i < n_gru; ++i) gw->W[i] = 0.3f * frand();            /* W | U | b_i | b_h are one block */
    for (int i = 0; i < n_head; ++i) dw->W[i] = 0.3f * frand();
    /* class = which third of the sequence carries the large first feature */
    static float x[B * T * IN], y[B * C], h[B * H], p[B * C], d_p[B * C];
    for (int b = 0; b < B; ++b) {
        int cls = b % C;
        for (int c = 0; c < C; ++c) y[b * C + c] = c == cls ? 1.f : 0.f;
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < IN; ++i)
                x[(b * T + t) * IN + i] = 0.1f * frand() + ((i == 0 && t / (T / C) == cls) ? 1.0f : 0.f);
    }
    float first = 0.f, last = 0.f;
    for (int it = 0; it < STEPS; ++it) {
        if (GRUApplyTrainingBatch(gru, x, h) || DenseApplyTrainingBatch(head, h, p)) { fprintf(stderr, "forward failed\n"); return 2; }
        float loss = categorical_crossentropy(y, p, C, B);
        categorical_crossentropy_derivative(y, p, d_p, C, B);
        for (int i = 0; i < B * C; ++i) d_p[i] /= (float)B;
        DenseGradient *dg = DenseGradientCreateFromFilter(head);
        GRUGradient *gg = GRUGradientCreate(gcfg, (GRUTrainingConfig){B});
        DenseCalculateGradient(head, dg, d_p);
        GRUCalculateGradient(gru, gg, dg->d_X);                            /* d loss / d h  ->  BPTT */
        sgd_optimize((SGD){0.5f}, dg->d_W, dw->W, n_head);
        sgd_optimize((SGD){0.5f}, gg->d_W, gw->W, n_gru);
        DenseGradientDestroy(dg);
        RecurrentGradientDestroy(gg);
        printf("step %d loss %.6f\n", it, loss);
        if (it == 0) first = loss;
        last = loss;
        if (!isfinite(loss)) return 3;
    }
    GRUDestroy(gru); DenseDestroy(head); ActivationFunctionDestroy(sm); GRUActivationsDestroy(gcfg.activations);
    return last < 0.6f * first ? 0 : 1;
}
