/* frag3_caller.c -- a C caller of the additive frag3 entry points (include/nntoolkitcore_hip.h): an LSTM feeding a
 * TimeDistributedDense with the tensor in between kept in frag3 form on the GPU (lstm.c:426-475 then time_distributed_dense.c:52-58
 * semantics).  It runs the tail of the stack three ways -- two f32 device calls, the fused call, the piece-by-piece frag3 calls -- and
 * writes each result; tests/test_c_dropin.py checks them against each other (the f32 and frag3 routes bit for bit; the fused call, which
 * hands h over as two f16 images by default, within its stated rounding and bit for bit with NNTK_DENSE_F16X2=0) and against the CPU oracle.
 *   frag3_caller <dir>      <dir>: shape.txt "B T I H V", x.bin, W.bin, U.bin, bi.bin, bh.bin, dW.bin, db.bin */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nntoolkitcore/layers/lstm.h"
#include "nntoolkitcore/layers/time_distributed_dense.h"

static float *slurp(const char *dir, const char *name, size_t n) {
    char path[1024]; snprintf(path, sizeof(path), "%s/%s", dir, name);
    FILE *f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
    float *p = malloc(n * sizeof(float));
    if (fread(p, sizeof(float), n, f) != n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f); return p;
}
static void dump(const char *dir, const char *name, const float *d_src, size_t n) {
    float *h = malloc(n * sizeof(float));
    if (nntk_device_download(h, d_src, n) != 0) { fprintf(stderr, "download: %s\n", nntk_last_error()); exit(3); }
    char path[1024]; snprintf(path, sizeof(path), "%s/%s", dir, name);
    FILE *f = fopen(path, "wb"); fwrite(h, sizeof(float), n, f); fclose(f); free(h);
}
#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, nntk_last_error()); return 3; } } while (0)

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const char *dir = argv[1];
    char path[1024];
    int B, T, I, H, V;
    snprintf(path, sizeof(path), "%s/shape.txt", dir);
    FILE *sf = fopen(path, "r"); if (!sf || fscanf(sf, "%d %d %d %d %d", &B, &T, &I, &H, &V) != 5) return 2; fclose(sf);
    CHECK(nntk_hip_set_device(0));

    LSTM lstm = LSTMCreateForInference(LSTMConfigCreate(I, H, true, T, true, LSTMActivationsCreateDefault(H)));
    TimeDistributedDense tdd = TimeDistributedDenseCreateForInference(TimeDistributedDenseConfigCreate(T, DenseConfigCreate(H, V, NULL)));
    if (!lstm || !tdd) { fprintf(stderr, "create: %s\n", nntk_last_error()); return 3; }
    LSTMWeights *w = LSTMGetWeights(lstm);
    float *W = slurp(dir, "W.bin", (size_t)I * 4 * H), *U = slurp(dir, "U.bin", (size_t)H * 4 * H);
    float *bi = slurp(dir, "bi.bin", 4 * H), *bh = slurp(dir, "bh.bin", 4 * H);
    memcpy(w->W, W, sizeof(float) * I * 4 * H); memcpy(w->U, U, sizeof(float) * H * 4 * H);
    memcpy(w->b_i, bi, sizeof(float) * 4 * H); memcpy(w->b_h, bh, sizeof(float) * 4 * H);
    DenseWeights *dw = TimeDistributedDenseGetWeights(tdd);
    float *dW = slurp(dir, "dW.bin", (size_t)H * V), *db = slurp(dir, "db.bin", V);
    memcpy(dw->W, dW, sizeof(float) * H * V); memcpy(dw->b, db, sizeof(float) * V);
    CHECK(LSTMSyncWeights(lstm));
    CHECK(TimeDistributedDenseSyncWeights(tdd));
    printf("%s\n", LSTMKernelPlan(lstm));

    const size_t nx = (size_t)B * T * I, nh = (size_t)B * T * H, ny = (size_t)B * T * V;
    float *x = slurp(dir, "x.bin", nx);
    float *d_x = nntk_device_alloc(nx), *d_h = nntk_device_alloc(nh), *d_y = nntk_device_alloc(ny);
    float *d_h3 = nntk_device_alloc(nntk_frag3_floats(B, T, H)), *d_x3 = nntk_device_alloc(nntk_frag3_floats(B, T, I));
    if (!d_x || !d_h || !d_y || !d_h3 || !d_x3) return 4;
    CHECK(nntk_device_upload(d_x, x, nx));

    CHECK(LSTMApplyDevice(lstm, d_x, d_h, B));                                  /* 1: through an f32 tensor */
    CHECK(TimeDistributedDenseApplyDevice(tdd, d_h, d_y, B));
    dump(dir, "out_f32.bin", d_y, ny);
    CHECK(LSTMTimeDistributedDenseApplyDevice(lstm, tdd, d_x, d_y, B));         /* 2: the fused call (FRAG2H between the layers by default) */
    dump(dir, "out_fused.bin", d_y, ny);
    CHECK(nntk_frag3_pack_device(d_x, d_x3, B, T, I));                          /* 3: piece by piece, frag3 on both sides of the LSTM */
    CHECK(LSTMApplyDeviceFrag3(lstm, NULL, d_x3, NULL, d_h3, B));
    CHECK(TimeDistributedDenseApplyDeviceFrag3(tdd, d_h3, d_y, B));
    dump(dir, "out_frag3.bin", d_y, ny);
    CHECK(nntk_frag3_unpack_device(d_h3, d_h, B, T, H));                        /* the LSTM output itself, back in f32 */
    dump(dir, "out_h.bin", d_h, nh);
    CHECK(nntk_hip_synchronize());

    nntk_device_free(d_x); nntk_device_free(d_h); nntk_device_free(d_y); nntk_device_free(d_h3); nntk_device_free(d_x3);
    TimeDistributedDenseDestroy(tdd);
    LSTMDestroy(lstm);
    return 0;
}
