/* dist_caller.c -- a C caller of the multi-GPU boundary: one process per GPU, rank 0 owns the weights,
 * ONE RCCL broadcast per layer at start-up, then every rank runs its own utterance shard with no communication.
 *   dist_caller <rank> <world> <dir>
 * <dir> holds x.bin [B, T, I], W.bin, U.bin, bi.bin, bh.bin (read by rank 0 only) and shape.txt "B T I H";
 * writes <dir>/out_<rank>.bin = GRU outputs of this rank's shard.  The 128-byte RCCL id travels through <dir>/id.bin. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "nntoolkitcore/layers/gru.h"

static float *slurp(const char *dir, const char *name, size_t n) {
    char path[1024]; snprintf(path, sizeof(path), "%s/%s", dir, name);
    FILE *f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
    float *p = malloc(n * sizeof(float));
    if (fread(p, sizeof(float), n, f) != n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f); return p;
}
#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "rank %d: %s failed: %s\n", rank, #call, nntk_last_error()); return 3; } } while (0)

int main(int argc, char **argv) {
    if (argc < 4) return 1;
    const int rank = atoi(argv[1]), world = atoi(argv[2]);
    const char *dir = argv[3];
    char path[1024];
    int B, T, I, H;
    snprintf(path, sizeof(path), "%s/shape.txt", dir);
    FILE *sf = fopen(path, "r"); if (!sf || fscanf(sf, "%d %d %d %d", &B, &T, &I, &H) != 4) return 2; fclose(sf);

    const int ndev = nntk_hip_device_count();
    if (ndev < 1) { fprintf(stderr, "no GPU\n"); return 4; }
    CHECK(nntk_hip_set_device(rank % ndev));
    unsigned char id[NNTK_DIST_ID_BYTES];
    snprintf(path, sizeof(path), "%s/id.bin", dir);
    if (rank == 0) {
        CHECK(nntk_dist_get_unique_id(id));
        char tmp[1100]; snprintf(tmp, sizeof(tmp), "%s.tmp", path);
        FILE *f = fopen(tmp, "wb"); fwrite(id, 1, sizeof(id), f); fclose(f); rename(tmp, path);
    } else {
        FILE *f = NULL;
        for (int i = 0; i < 6000 && !(f = fopen(path, "rb")); ++i) usleep(10000);
        if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "rank %d: no id\n", rank); return 2; }
        fclose(f);
    }
    CHECK(nntk_dist_init(id, rank, world));
    if (nntk_dist_rank() != rank || nntk_dist_world_size() != world) return 5;

    GRU g = GRUCreateForInference(GRUConfigCreate(I, H, true, T, GRUActivationsCreateDefault(H)));
    if (!g) { fprintf(stderr, "create: %s\n", nntk_last_error()); return 3; }
    GRUWeights *w = GRUGetWeights(g);
    if (rank == 0) {                                   /* only the root has the model */
        float *W = slurp(dir, "W.bin", (size_t)I * 3 * H), *U = slurp(dir, "U.bin", (size_t)H * 3 * H);
        float *bi = slurp(dir, "bi.bin", 3 * H), *bh = slurp(dir, "bh.bin", 3 * H);
        memcpy(w->W, W, sizeof(float) * I * 3 * H); memcpy(w->U, U, sizeof(float) * H * 3 * H);
        memcpy(w->b_i, bi, sizeof(float) * 3 * H); memcpy(w->b_h, bh, sizeof(float) * 3 * H);
        free(W); free(U); free(bi); free(bh);
    }
    CHECK(GRUBroadcastWeights(g, 0));                  /* one ncclBroadcast of the packed block W | U | b_i | b_h */
    CHECK(nntk_dist_barrier());

    int lo, hi;
    nntk_dist_shard_range(B, world, rank, &lo, &hi);
    float *x = slurp(dir, "x.bin", (size_t)B * T * I);
    float *y = malloc(sizeof(float) * (size_t)(hi - lo > 0 ? hi - lo : 1) * T * H);
    if (hi > lo) CHECK(GRUApplyInferenceBatch(g, x + (size_t)lo * T * I, y, hi - lo));      /* no communication here */
    snprintf(path, sizeof(path), "%s/out_%d.bin", dir, rank);
    FILE *of = fopen(path, "wb"); fwrite(y, sizeof(float), (size_t)(hi - lo) * T * H, of); fclose(of);
    GRUDestroy(g);
    CHECK(nntk_dist_finalize());
    printf("rank %d/%d: utterances [%d, %d) done\n", rank, world, lo, hi);
    free(x); free(y);
    return 0;
}
