/* stream_latency.c -- latency of the reference-shaped single-sequence calls (host pointers, carried state) measured
 * from C, the way the reference's callers use them: GRU(128 -> 256) and LSTM(128 -> 512, v2).
 * build: gcc -O2 -Iinclude tests/c_api/stream_latency.c -Lnntoolkitcore_amd/lib -lnntoolkitcore_hip -Wl,-rpath,$PWD/nntoolkitcore_amd/lib -o /tmp/stream_latency */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "nntoolkitcore/layers/gru.h"
#include "nntoolkitcore/layers/lstm.h"

static double now_us(void) {
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}
static void fill(float *p, size_t n, unsigned seed, float sc) {
    unsigned s = seed * 2654435761u + 1u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; p[i] = (((int)(s >> 9) % 2001) - 1000) * 1e-3f * sc; }
}
static int cmp(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y; }

int main(int argc, char **argv) {
    const char *opt = argc > 1 ? argv[1] : "auto";       /* value of the rec_stream option: "0" = batch path */
    if (nntk_hip_set_option("rec_stream", opt)) { fprintf(stderr, "%s\n", nntk_last_error()); return 1; }
    const int Ts[] = {1, 2, 5, 10, 16, 20, 50};
    for (int cell = 0; cell < 2; ++cell) {
        const int I = 128, H = cell ? 512 : 256, G = cell ? 4 : 3;
        for (unsigned ti = 0; ti < sizeof(Ts) / sizeof(Ts[0]); ++ti) {
            const int T = Ts[ti];
            float *x = malloc(sizeof(float) * T * I), *y = malloc(sizeof(float) * T * H);
            fill(x, (size_t)T * I, 7, 1.0f);
            void *h;
            RecurrentWeights *w;
            if (cell) { LSTM l = LSTMCreateForInference(LSTMConfigCreate(I, H, true, T, true, LSTMActivationsCreateDefault(H))); h = l; w = LSTMGetWeights(l); }
            else      { GRU g = GRUCreateForInference(GRUConfigCreate(I, H, true, T, GRUActivationsCreateDefault(H))); h = g; w = GRUGetWeights(g); }
            if (!h) { fprintf(stderr, "create failed: %s\n", nntk_last_error()); return 1; }
            fill(w->W, (size_t)I * G * H, 1, 0.09f); fill(w->U, (size_t)H * G * H, 2, 0.05f);
            fill(w->b_i, (size_t)G * H, 3, 0.1f); fill(w->b_h, (size_t)G * H, 4, 0.1f);
            const int n = 300;
            double *dt = malloc(sizeof(double) * n);
            for (int i = 0; i < 20 + n; ++i) {
                const double t0 = now_us();
                const int rc = cell ? LSTMApplyInference((LSTM)h, x, y) : GRUApplyInference((GRU)h, x, y);
                if (rc) { fprintf(stderr, "apply failed: %s\n", nntk_last_error()); return 1; }
                if (i >= 20) dt[i - 20] = now_us() - t0;
            }
            qsort(dt, n, sizeof(double), cmp);
            printf("%s  T=%3d  rec_stream=%s  median %7.1f us  p10 %7.1f  p90 %7.1f  (%.1f us per frame)\n", cell ? "LSTM-512" : "GRU-256 ", T, opt,
                   dt[n / 2], dt[n / 10], dt[n * 9 / 10], dt[n / 2] / T);
            if (cell) LSTMDestroy((LSTM)h); else GRUDestroy((GRU)h);
            free(x); free(y); free(dt);
        }
    }
    return 0;
}
