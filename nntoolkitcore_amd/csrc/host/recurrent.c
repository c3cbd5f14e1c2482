/*
 * recurrent.c -- GRU and LSTM host layer.  Reference: layers/recurrent.{h,c},
 * layers/private/recurrent_private.c:29-36 (one block W | U | b_i | b_h),
 * layers/gru.c (forward :13-19, :51-61, :110-204), layers/lstm.c (forward
 * :17-26, :55-65, :110-268).
 *
 * Device plan per Apply:
 *   1. xW + b_i for all timesteps: one MFMA GEMM, written time-major [T, B, G*H]
 *   2. T fused step kernels (csrc/hip/recurrent.hip) carrying h (and c)
 * The single-sequence API is stateful exactly like the reference (gru.c:201,
 * lstm.c:264-265): h/c persist in the handle between calls; the batched forms
 * start every sequence from zeros (gru.c:260, lstm.c:439).
 */
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "nntk_internal.h"

/* single-sequence calls of at most this many timesteps take the streaming kernels (one small launch per step, no
 * projection GEMM, no copies); longer ones amortise the batch path's fixed costs */
#define NNTK_STREAM_MAX_T 32

/* recurrent.c:7-19 */
RecurrentConfig RecurrentConfigCreate(int input_feature_channels, int output_feature_channels,
                                      bool return_sequences, int timesteps) {
    RecurrentConfig c;
    memset(&c, 0, sizeof(c));
    c.input_feature_channels = input_feature_channels;
    c.output_feature_channels = output_feature_channels;
    c.return_sequences = return_sequences;
    c.timesteps = timesteps;
    return c;
}

/* shared core of both layers */
typedef struct {
    int G;                      /* 1 (RNN), 3 (GRU) or 4 (LSTM) */
    int in, H, T;
    bool return_sequences;
    RecurrentWeights *weights;
    nntk_wblock wb;
    float *d_wp, *d_bi, *d_ut, *d_bh;
    float *d_wt;                /* W^T packed like U^T (only when in == H): layer-2 operand of the fused two-layer GRU */
    int wt_valid;
    int rr_exact_only;          /* W or U holds a value the bf16 split cannot represent (non-finite, > 3.39e38, denormal): exact kernels only */
    float *d_rr4;               /* the same weights as images of the full-K kernels (recurrent_fk.hip), packed with d_rr */
    float *d_rr;                /* LSTM: weight images of the register-resident split-bf16 kernel (recurrent_rr.hip), made on first use */
    int rr_valid;
    float *d_rr_hf;             /* LSTM, H > 256: the images of the HF instantiation (h.U on two f16 images), packed on first use */
    int rr_hf_valid;
    float hf_wscale;            /* ... the scale W's images (and therefore Z) carry: 2^15 * (U's scale); 0: the form does not hold these weights */
    float *d_b4, *d_b4_train;   /* GRU on those kernels: the four-slot bias vector (core_try_gru_rr; the training forward's copy) */
    float *d_rr_train;          /* ... and the training forward's own copy, re-packed every mini-batch from the raw block */
    /* persistent single-sequence state [H], double-buffered: a stateful call reads state[cur] and writes
     * state[cur ^ 1]; cur flips only once the call is known to be good, so a call that has to be repeated
     * (persistent-kernel fault, see core_apply_host) still finds its initial state intact */
    float *d_h[2], *d_c[2];
    int cur;
    /* streaming path: pinned host staging the kernels read / write directly, a device counter of finished workgroups
     * and the pinned word the last one raises */
    float *pin_in, *pin_out;
    size_t pin_in_n, pin_out_n;
    unsigned *d_done;
    volatile unsigned *flag;
    unsigned seq;
    nntk_devbuf d_in, d_out, d_xw, d_work, d_work_rr, d_rr_stage;
    nntk_devbuf d_h2;           /* the fused LSTM -> TimeDistributedDense call: the sequence output in FRAG2H form */
    nntk_devbuf d_hseq;         /* register-resident kernels: their T-deep hand-off = the layer output in frag3 form (when the caller supplies no buffer) */
    nntk_devbuf d_xf3;          /* ... and the input packed into frag3 form (when the call packs it) */
} rec_core;

/* 1 if the block holds a value the split-bf16 x 3 contraction cannot represent exactly: non-finite, above bf16's largest finite
 * value (bf16(x) = inf, and inf - inf = NaN in the split), or denormal (the bf16 MFMA flushes it).  The reference's f32 chain
 * (core/default_ops.cc:224-231) saturates or carries such values; the register-resident kernels would turn them into NaN, so a
 * flagged block keeps the exact-f32 kernels -- the same rule conv / dense / mel weights follow (runtime.c).  Branch-free so that
 * gcc vectorises it: the training forwards run it on every mini-batch. */
__attribute__((optimize("O3"))) static int rr_unsplittable(const float *v, size_t n) {
    const unsigned *u = (const unsigned *)v;
    unsigned bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const unsigned a = u[i] & 0x7fffffffu;
        bad |= (unsigned)(a > 0x7f7f0000u) | (unsigned)(a - 1u < 0x007fffffu);
    }
    return bad != 0;
}

static int core_init(rec_core *c, int G, RecurrentConfig base) {
    memset(c, 0, sizeof(*c));
    c->G = G;
    c->in = base.input_feature_channels;
    c->H = base.output_feature_channels;
    c->T = base.timesteps;
    c->return_sequences = base.return_sequences;
    size_t w = (size_t)c->in * G * c->H, u = (size_t)c->H * G * c->H, b = (size_t)G * c->H;
    if (nntk_wblock_init(&c->wb, w + u + 2 * b)) return -1;
    c->weights = (RecurrentWeights *)malloc(sizeof(RecurrentWeights));
    c->weights->W = c->wb.host;
    c->weights->U = c->weights->W + w;
    c->weights->b_i = c->weights->U + u;
    c->weights->b_h = c->weights->b_i + b;
    /* one allocation: h[0] | h[1] | c[0] | c[1], each padded to 64 floats */
    size_t hs = ((size_t)c->H + 63) & ~(size_t)63;
    float *st = (float *)nntk_shim_malloc(4 * hs * sizeof(float));
    if (!st) return -1;
    c->d_h[0] = st; c->d_h[1] = st + hs; c->d_c[0] = st + 2 * hs; c->d_c[1] = st + 3 * hs;
    c->cur = 0;
    if (nntk_shim_memset(st, 0, 4 * hs * sizeof(float))) return -1;
    return 0;
}

static void core_free(rec_core *c) {
    nntk_shim_synchronize();
    nntk_shim_free(c->d_wp); nntk_shim_free(c->d_bi); nntk_shim_free(c->d_ut); nntk_shim_free(c->d_bh); nntk_shim_free(c->d_wt);
    nntk_shim_free(c->d_rr_hf);
    nntk_shim_free(c->d_rr); nntk_shim_free(c->d_rr4); nntk_shim_free(c->d_rr_train); nntk_shim_free(c->d_b4); nntk_shim_free(c->d_b4_train);
    nntk_shim_free(c->d_h[0]);
    nntk_shim_host_free(c->pin_in); nntk_shim_host_free(c->pin_out); nntk_shim_host_free((void *)c->flag);
    nntk_shim_free(c->d_done);
    nntk_devbuf_free(&c->d_in); nntk_devbuf_free(&c->d_out); nntk_devbuf_free(&c->d_xw); nntk_devbuf_free(&c->d_work); nntk_devbuf_free(&c->d_work_rr); nntk_devbuf_free(&c->d_rr_stage);
    nntk_devbuf_free(&c->d_hseq); nntk_devbuf_free(&c->d_xf3); nntk_devbuf_free(&c->d_h2);
    nntk_wblock_free(&c->wb);
    free(c->weights);
}

/* W [in, G*H] is already the GEMM's [K, N]; U [H, G*H] is stored transposed per
 * gate as U^T [G][Hj_p][Hk_p] so each output column's K vector is contiguous. */
static int core_upload(rec_core *c) {
    int G = c->G, H = c->H;
    if (nntk_upload_gemm_weights(&c->d_wp, c->weights->W, c->in, G * H)) return -1;
    if (nntk_upload_floats(&c->d_bi, c->weights->b_i, (size_t)G * H)) return -1;
    if (nntk_upload_floats(&c->d_bh, c->weights->b_h, (size_t)G * H)) return -1;
    int Hj_p = (H + 15) & ~15, Hk_p = (H + 31) & ~31;
    size_t n = (size_t)G * Hj_p * Hk_p;
    float *tmp = (float *)calloc(n, sizeof(float));
    if (!tmp) NNTK_FAIL("out of host memory while packing recurrent weights");
    const float *U = c->weights->U;
    for (int k = 0; k < H; ++k)
        for (int g = 0; g < G; ++g)
            for (int j = 0; j < H; ++j)
                tmp[((size_t)g * Hj_p + j) * Hk_p + k] = U[(size_t)k * G * H + (size_t)g * H + j];
    int rc = nntk_upload_floats(&c->d_ut, tmp, n);
    free(tmp);
    if (rc) return rc;
    c->wt_valid = 0;
    c->rr_valid = 0;
    c->rr_hf_valid = 0;
    c->rr_exact_only = rr_unsplittable(c->weights->W, (size_t)c->in * G * H + (size_t)H * G * H);      /* W | U are contiguous */
    nntk_wblock_mark_uploaded(&c->wb);
    return 0;
}

/* W [in = H, G*H] transposed per gate into the U^T layout [G][Hj_p][Hk_p] */
static int core_ensure_wt(rec_core *c) {
    if (c->wt_valid) return 0;
    if (c->in != c->H) NNTK_FAIL("fused GRU stack: layer 2's input size must equal its hidden size");
    int G = c->G, H = c->H;
    int Hj_p = (H + 15) & ~15, Hk_p = (H + 31) & ~31;
    size_t n = (size_t)G * Hj_p * Hk_p;
    float *tmp = (float *)calloc(n, sizeof(float));
    if (!tmp) NNTK_FAIL("out of host memory while packing recurrent weights");
    const float *W = c->weights->W;
    for (int k = 0; k < H; ++k)
        for (int g = 0; g < G; ++g)
            for (int j = 0; j < H; ++j)
                tmp[((size_t)g * Hj_p + j) * Hk_p + k] = W[(size_t)k * G * H + (size_t)g * H + j];
    int rc = nntk_upload_floats(&c->d_wt, tmp, n);
    free(tmp);
    if (rc) return rc;
    c->wt_valid = 1;
    return 0;
}

static int core_ensure(rec_core *c, int check_edits) {
    if (nntk_wblock_dirty(&c->wb, check_edits)) return core_upload(c);
    return 0;
}

static int core_reset_state(rec_core *c) {
    size_t hs = ((size_t)c->H + 63) & ~(size_t)63;
    c->cur = 0;
    return nntk_shim_memset(c->d_h[0], 0, 4 * hs * sizeof(float));
}

static int core_broadcast(rec_core *c, int root) {
    if (nntk_shim_dist_broadcast_host(c->wb.host, c->wb.n, root)) return -1;
    return core_upload(c);
}

/* Which recurrent kernel a call takes depends on the layer's SHAPE, its activations and the FORM of the call -- never on the
 * number of sequences in it: a row's bits must not change with the batch (or shard) it arrives in (round 3 had a
 * B >= 32 threshold here, which made a 130-utterance batch sharded over 8 GPUs differ from the single-GPU run).
 *   zero-state batch forms ([B, T, in]: *ApplyInferenceBatch, *ApplyDevice, GRUStack2Apply*, the training forwards), qualifying
 *     shape, standard activations, splittable weights  ->  the register-resident split-bf16 kernels (recurrent_rr.hip), any B;
 *   the reference's stateful single-sequence call (*ApplyInference: carried h / c)  ->  the exact-f32 kernels, whose chain the
 *     streaming kernel (T <= 32) reproduces bit for bit, so cutting a stream into calls of any length gives the same bits.
 * Option rec_rr = 1 takes the register-resident kernels for the stateful call too; 0 never takes them. */

static int lstm_std_acts(const int *acts) {
    return acts[0] == NNTK_ACT_SIGMOID && acts[1] == NNTK_ACT_SIGMOID && acts[2] == NNTK_ACT_TANH &&
           acts[3] == NNTK_ACT_SIGMOID && acts[4] == NNTK_ACT_TANH;
}

/* How a call on the register-resident kernels takes its input and hands over its output.  The kernels' hand-off buffer is T-deep and
 * in FRAG3 form (frag3.hip: three bf16 images in MFMA fragment order), so the layer output exists in that form anyway; a consumer that
 * reads frag3 (the next layer of a stack, the dense GEMM) takes it from there and d_out may be NULL. */
typedef struct {
    const float *d_in;      /* f32 [B][T][in], or NULL when d_in_f3 is given */
    const float *d_in_f3;   /* the input in frag3 form, or NULL */
    float *d_out;           /* f32 output, or NULL */
    float *d_out_f3;        /* frag3 output [B][T][H] (nntk_frag3_floats), or NULL: handle scratch */
    float *d_out_h2;        /* LSTM only: the sequence output as a FRAG2H tensor (nntk_frag2h_floats) instead of d_out (which must be NULL), or NULL */
} rr_io;

/* decides the x form and provides it: 0 = f32 rows, 1 = frag3 (*xf3 set; packed here when the caller passed f32), 2 = shape not taken */
static int rr_input(rec_core *c, const rr_io *io, int B, const float **xf3) {
    *xf3 = io->d_in_f3;
    /* (a misaligned f32 pointer cannot take the 16-byte row requests: it is packed like a shape the f32 form does not take, so the kernel
     * family -- and the bits -- do not depend on the caller's pointer alignment) */
    const int f32_ok = io->d_in && (((size_t)io->d_in) & 15) == 0 && nntk_shim_lstm_rr_image_floats(c->H, c->in) != 0;
    const int xf_ok = nntk_shim_rr_image_floats_xf(c->H, c->in) != 0;
    if (io->d_in_f3) return xf_ok ? 1 : 2;
    int mode = -1;
    (void)nntk_shim_get_option("rec_xf", &mode);
    /* 1: always pack (one extra pass over x, then coalesced requests and no split inside the step); 0: only when the f32 path cannot
     * take the shape (in % 8 != 0); auto: pack once the call is big enough for the pass to pay (measured at the stack's LSTM, 512 x 996
     * x 128: 6.50 -> 6.16 ms including the pack).  The two x forms give the same bits (the split is exact and the kernels sum the same
     * products in the same order -- tests/test_gpu_frag3.py), so this is a speed choice that may depend on the size of the call. */
    /* a shape the full-K family takes (recurrent_fk.hip) runs on it at EVERY size -- it sums in another order than the split-K family, so
     * the choice must not depend on the call -- and that family reads frag3 only */
    const int fk = nntk_shim_fk_image_floats(c->H, c->in) != 0;
    if (f32_ok && !fk && (mode == 0 || (mode < 0 && (long)B * c->T < 8192))) return 0;
    if (!xf_ok || !io->d_in) return 2;
    float *buf = nntk_devbuf_reserve(&c->d_xf3, nntk_shim_frag3_floats(B, c->T, c->in));
    if (!buf) return -1;
    if (nntk_shim_frag3_pack(io->d_in, buf, B, c->T, c->in)) return -1;
    *xf3 = buf;
    return 1;
}

/* 0 = ran; 1 = not taken; -1 = error */
static int core_try_lstm_rr(rec_core *c, int use_bh, const int *acts, const rr_io *io, int B, int stateful) {
    int on = -1;
    (void)nntk_shim_get_option("rec_rr", &on);
    if (on == 0 || c->G != 4 || !lstm_std_acts(acts) || c->rr_exact_only) return 1;
    if (on != 1 && stateful) return 1;
    if (!nntk_shim_rr_image_floats_xf(c->H, c->in)) return 1;
    const float *xf3 = NULL;
    const int xm = rr_input(c, io, B, &xf3);
    if (xm < 0) return -1;
    if (xm == 2) return 1;
    size_t img = nntk_shim_rr_image_floats_xf(c->H, c->in);
    if (nntk_shim_fk_image_floats(c->H, c->in) && !c->d_rr4) c->rr_valid = 0;      /* (option rec_fk switched on after the images were packed) */
    if (!c->rr_valid) {
        if (!c->d_rr && !(c->d_rr = (float *)nntk_shim_malloc(img * sizeof(float)))) return -1;
        if (nntk_shim_lstm_rr_pack(c->d_ut, c->d_wp, c->d_rr, c->H, c->in)) return -1;
        const size_t img4 = nntk_shim_fk_image_floats(c->H, c->in);
        if (img4) {
            if (!c->d_rr4 && !(c->d_rr4 = (float *)nntk_shim_malloc(img4 * sizeof(float)))) return -1;
            if (nntk_shim_fk_pack(c->d_ut, c->d_wp, c->d_rr4, c->H, c->in)) return -1;
        }
        c->rr_valid = 1;
    }
    float *d_work = nntk_devbuf_reserve(&c->d_work_rr, nntk_shim_lstm_rr_work_floats(B, c->H));
    float *d_hseq = io->d_out_f3 ? io->d_out_f3 : nntk_devbuf_reserve(&c->d_hseq, nntk_shim_rr_hseq_floats(B, c->T, c->H));
    if (!d_work || !d_hseq) return -1;
    const float *h0 = stateful ? c->d_h[c->cur] : NULL, *c0 = stateful ? c->d_c[c->cur] : NULL;
    float *hT = stateful ? c->d_h[c->cur ^ 1] : NULL, *cT = stateful ? c->d_c[c->cur ^ 1] : NULL;
    return nntk_shim_lstm_rr(xm ? NULL : io->d_in, xm ? xf3 : NULL, c->d_rr, c->d_rr4, c->d_bi, use_bh ? c->d_bh : NULL, h0, c0, io->d_out, io->d_out_h2, d_hseq,
                             hT, cT, d_work, B, c->T, c->in, c->H, c->return_sequences);
}

/* GRU on the register-resident split-bf16 kernels (recurrent_rr.hip gru_rr_kernel).  The three gates ride in four slots,
 * z | r | h.U_h | x.W_h: the image is packed from [U_z | U_r | U_h | 0] and [W_z | W_r | 0 | W_h], the bias vector is
 * b_i,z + b_h,z | b_i,r + b_h,r | b_h,h | b_i,h (gru.c:144-186: the reset gate multiplies h.U_h + b_h,h only).
 * 0 = ran; 1 = not taken; -1 = error */
static int gru_std_acts(const int *acts) {
    return acts[0] == NNTK_ACT_SIGMOID && acts[1] == NNTK_ACT_TANH && acts[2] == NNTK_ACT_SIGMOID;      /* z, h, r */
}
/* packs *d_img (allocated on first use, `img` floats) and *d_b4 from the caller-layout weights W [in][3H], U [H][3H], b_i, b_h */
static int gru_rr_build_image(int in, int H, const float *W, const float *U, const float *bi, const float *bh,
                              float **d_img, size_t img, float **d_b4, nntk_devbuf *stage, float **d_img4) {
    const size_t nW = (size_t)in * 4 * H, nU = (size_t)H * 4 * H;
    float *tmp = (float *)calloc(nW + nU + 4 * (size_t)H, sizeof(float));
    if (!tmp) NNTK_FAIL("out of host memory while packing GRU weights");
    float *W4 = tmp, *U4 = tmp + nW, *b4 = U4 + nU;
    for (int k = 0; k < in; ++k) {
        memcpy(W4 + (size_t)k * 4 * H, W + (size_t)k * 3 * H, 2 * (size_t)H * sizeof(float));                 /* W_z | W_r */
        memcpy(W4 + (size_t)k * 4 * H + 3 * (size_t)H, W + (size_t)k * 3 * H + 2 * (size_t)H, (size_t)H * sizeof(float));   /* slot 3: W_h */
    }
    for (int k = 0; k < H; ++k) memcpy(U4 + (size_t)k * 4 * H, U + (size_t)k * 3 * H, 3 * (size_t)H * sizeof(float));      /* U_z | U_r | U_h | 0 */
    for (int j = 0; j < H; ++j) {
        b4[j] = bi[j] + bh[j];
        b4[H + j] = bi[H + j] + bh[H + j];
        b4[2 * H + j] = bh[2 * H + j];
        b4[3 * H + j] = bi[2 * H + j];
    }
    int rc = 0;
    float *d_tmp = nntk_devbuf_reserve(stage, nW + nU);        /* kept with the handle: stream order protects it, no synchronisation */
    if (!d_tmp) rc = -1;
    if (!rc && !*d_img && !(*d_img = (float *)nntk_shim_malloc(img * sizeof(float)))) rc = -1;
    if (!rc && !*d_b4 && !(*d_b4 = (float *)nntk_shim_malloc(4 * (size_t)H * sizeof(float)))) rc = -1;
    if (!rc) rc = nntk_shim_upload(d_tmp, tmp, (nW + nU) * sizeof(float));
    if (!rc) rc = nntk_shim_upload(*d_b4, b4, 4 * (size_t)H * sizeof(float));
    if (!rc) rc = nntk_shim_lstm_rr_pack_raw(d_tmp + nW, d_tmp, *d_img, H, in);
    const size_t img4 = d_img4 ? nntk_shim_fk_image_floats(H, in) : 0;        /* the full-K family's images of the same matrices */
    if (!rc && img4) {
        if (!*d_img4 && !(*d_img4 = (float *)nntk_shim_malloc(img4 * sizeof(float)))) rc = -1;
        if (!rc) rc = nntk_shim_fk_pack_raw(d_tmp + nW, d_tmp, *d_img4, H, in);
    }
    free(tmp);                                              /* (nntk_shim_upload has copied it) */
    return rc ? -1 : 0;
}
static int core_try_gru_rr(rec_core *c, const int *acts, const rr_io *io, int B, int stateful) {
    int on = -1;
    (void)nntk_shim_get_option("rec_rr", &on);
    if (on == 0 || c->G != 3 || !gru_std_acts(acts) || c->rr_exact_only) return 1;
    if (on != 1 && stateful) return 1;
    const int H = c->H, in = c->in;
    size_t img = nntk_shim_rr_image_floats_xf(H, in);
    if (!img) return 1;
    const float *xf3 = NULL;
    const int xm = rr_input(c, io, B, &xf3);
    if (xm < 0) return -1;
    if (xm == 2) return 1;
    if (nntk_shim_fk_image_floats(H, in) && !c->d_rr4) c->rr_valid = 0;            /* (option rec_fk switched on after the images were packed) */
    if (!c->rr_valid) {
        /* from the SHADOW, i.e. the weight version core_upload packed d_wp / d_ut from: the device-pointer calls do not look for
         * host edits (SyncWeights is their contract), and an un-synced edit must not reach this kernel alone */
        const float *sW = c->wb.shadow, *sU = sW + (size_t)in * 3 * H, *sbi = sU + (size_t)H * 3 * H, *sbh = sbi + 3 * (size_t)H;
        if (gru_rr_build_image(in, H, sW, sU, sbi, sbh, &c->d_rr, img, &c->d_b4, &c->d_rr_stage, &c->d_rr4)) return -1;
        c->rr_valid = 1;
    }
    float *d_work = nntk_devbuf_reserve(&c->d_work_rr, nntk_shim_lstm_rr_work_floats(B, H));
    float *d_hseq = io->d_out_f3 ? io->d_out_f3 : nntk_devbuf_reserve(&c->d_hseq, nntk_shim_rr_hseq_floats(B, c->T, H));
    if (!d_work || !d_hseq) return -1;
    const float *h0 = stateful ? c->d_h[c->cur] : NULL;
    float *hT = stateful ? c->d_h[c->cur ^ 1] : NULL;
    return nntk_shim_gru_rr(xm ? NULL : io->d_in, xm ? xf3 : NULL, c->d_rr, c->d_rr4, c->d_b4, h0, io->d_out, d_hseq, hT, d_work, B, c->T, in, H,
                            c->return_sequences, 0, 0);
}

/* stateful != 0: continue from / store into the handle's state (B must be 1) */
static int core_apply_device(rec_core *c, int is_lstm, int use_bh, const int *acts, const float *scales,
                             const float *d_in, float *d_out, int B, int stateful) {
    int G = c->G, H = c->H, T = c->T;
    if (B <= 0 || T <= 0) return 0;
    const rr_io io = { d_in, NULL, d_out, NULL, NULL };
    if (is_lstm) {
        int rc = core_try_lstm_rr(c, use_bh, acts, &io, B, stateful);
        if (rc <= 0) return rc;
    }
    if (!is_lstm && G == 3) {
        int rc = core_try_gru_rr(c, acts, &io, B, stateful);
        if (rc <= 0) return rc;
    }
    float *d_xw = nntk_devbuf_reserve(&c->d_xw, (size_t)T * B * G * H);
    float *d_work = nntk_devbuf_reserve(&c->d_work, nntk_shim_recurrent_work_floats(B, H));
    if (!d_xw || !d_work) return -1;
    /* input projection, rows (b, t) written at time-major row t*B + b */
    if (nntk_shim_conv1d(d_in, c->d_wp, c->d_bi, NULL, 0.f, NNTK_ACT_IDENTITY, 1.f, d_xw,
                         B, T, c->in, G * H, 1, 1, T, 1))
        return -1;
    const float *bh = use_bh ? c->d_bh : NULL;
    const float *h0 = stateful ? c->d_h[c->cur] : NULL, *c0 = stateful ? c->d_c[c->cur] : NULL;
    float *hT = stateful ? c->d_h[c->cur ^ 1] : NULL, *cT = stateful ? c->d_c[c->cur ^ 1] : NULL;
    if (G == 1)
        return nntk_shim_rnn(d_xw, c->d_ut, bh, h0, d_out, hT, d_work, B, T, H, c->return_sequences, acts[0], scales[0]);
    if (is_lstm)
        return nntk_shim_lstm(d_xw, c->d_ut, bh, h0, c0, d_out, hT, cT, d_work, B, T, H, c->return_sequences, acts, scales);
    return nntk_shim_gru(d_xw, c->d_ut, bh, h0, d_out, hT, d_work, B, T, H, c->return_sequences, acts, scales);
}

/* Device call with frag3 tensors on either side (additive API: <Layer>ApplyDeviceFrag3).  Zero state per sequence.  The
 * register-resident kernels take and produce the format natively; every other kernel goes through f32 scratch tensors (frag3 is
 * exact: unpack(pack(x)) == x), so the call works for every shape and the results do not depend on the route. */
static int core_apply_device_f3(rec_core *c, int is_lstm, int use_bh, const int *acts, const float *scales,
                                const float *d_in, const float *d_in_f3, float *d_out, float *d_out_f3, int B) {
    if (B <= 0 || c->T <= 0) return 0;
    if (!d_in && !d_in_f3) NNTK_FAIL("ApplyDeviceFrag3: no input tensor");
    if (!d_out && !d_out_f3) NNTK_FAIL("ApplyDeviceFrag3: no output tensor");
    if (d_out_f3 && !c->return_sequences) NNTK_FAIL("ApplyDeviceFrag3: a frag3 output needs return_sequences");
    const rr_io io = { d_in, d_in_f3, d_out, d_out_f3, NULL };
    int rc = 1;
    if (is_lstm) rc = core_try_lstm_rr(c, use_bh, acts, &io, B, 0);
    else if (c->G == 3) rc = core_try_gru_rr(c, acts, &io, B, 0);
    if (rc <= 0) return rc;
    const float *x = d_in;
    if (!x) {
        float *xs = nntk_devbuf_reserve(&c->d_in, (size_t)B * c->T * c->in);
        if (!xs || nntk_shim_frag3_unpack(d_in_f3, xs, B, c->T, c->in)) return -1;
        x = xs;
    }
    float *o = d_out;
    if (!o && !(o = nntk_devbuf_reserve(&c->d_out, (size_t)B * c->T * c->H))) return -1;
    if (core_apply_device(c, is_lstm, use_bh, acts, scales, x, o, B, 0)) return -1;
    return d_out_f3 ? nntk_shim_frag3_pack(o, d_out_f3, B, c->T, c->H) : 0;
}

/* Which kernel family the zero-state batch forms of a layer will take, and why not the fast one when they will not: a caller can see a
 * performance cliff (a shape outside the register-resident kernels runs 1.3-1.8x slower) before it measures one.  The answer depends on
 * the layer only -- shape, activations, weights -- never on the batch (see the comment above core_try_lstm_rr). */
static const char *core_plan(rec_core *c, int std_acts, char *buf, size_t n) {
    int on = -1;
    (void)nntk_shim_get_option("rec_rr", &on);
    const int f32_ok = nntk_shim_lstm_rr_image_floats(c->H, c->in) != 0, xf_ok = nntk_shim_rr_image_floats_xf(c->H, c->in) != 0;
    const char *cell = c->G == 4 ? "lstm" : c->G == 3 ? "gru" : "rnn";
    const char *why = NULL;
    if (c->G == 1) why = "the RNN cell has no register-resident kernel";
    else if (on == 0) why = "option rec_rr = 0";
    else if (!std_acts) why = "non-default gate activations";
    else if (c->wb.uploaded && c->rr_exact_only) why = "a weight the bf16 split cannot hold (non-finite, > 3.39e38 or denormal)";
    else if (!xf_ok) why = (c->H % 16) ? "H % 16 != 0" : (c->H < 64 || c->H > 512) ? "H outside 64..512" : "input wider than 128 (256 when H <= 256)";
    if (why)
        snprintf(buf, n, "%s: exact-f32 kernels (projection GEMM + rec_persistent_kernel, per-timestep kernels when that does not fit) -- %s", cell, why);
    else if (nntk_shim_fk_image_floats(c->H, c->in) != 0)       /* (option rec_fk = 1 and a shape that family takes) */
        snprintf(buf, n, "%s: %s_fk_kernel<16,%d,%d> (register-resident split-bf16 x 3 without split-K, input projection fused; the input is packed into frag3 form first); the stateful single-sequence call keeps the exact-f32 kernels",
                 cell, cell, c->in <= 128 ? 8 : 16, 4);
    else
        snprintf(buf, n, "%s: %s_rr_kernel<%d,%d> (register-resident split-bf16 x 3, input projection fused)%s; the stateful single-sequence call keeps the exact-f32 kernels",
                 cell, cell, c->H <= 256 ? 4 : 8, c->in <= 64 ? 1 : c->in <= 128 ? 2 : 4,
                 f32_ok ? "" : "; in % 8 != 0: the input is packed into frag3 form first");
    return buf;
}

/* The reference's own call shape -- one sequence, carried state, a handful of timesteps (gru.c:189-204,
 * lstm.c:241-268) -- is latency, not throughput: T launches of the streaming step kernel, x_t read from and h_t written
 * to pinned host memory by the kernels themselves, and the host waits on a pinned word the last workgroup raises
 * (a stream synchronisation alone costs more than the kernels).  Same bits as the batch path (recurrent.hip). */
#if defined(__x86_64__) || defined(__i386__)
#define NNTK_CPU_RELAX() __builtin_ia32_pause()
#elif defined(__aarch64__)
#define NNTK_CPU_RELAX() __asm__ __volatile__("yield")
#else
#define NNTK_CPU_RELAX() ((void)0)
#endif
static int core_apply_stream(rec_core *c, int is_lstm, int use_bh, const int *acts, const float *scales,
                             const float *input, float *output) {
    const size_t n_in = (size_t)c->T * c->in, n_out = c->return_sequences ? (size_t)c->T * c->H : (size_t)c->H;
    if (c->pin_in_n < n_in) {
        nntk_shim_synchronize();
        nntk_shim_host_free(c->pin_in);
        c->pin_in = (float *)nntk_shim_host_alloc(n_in * sizeof(float));
        c->pin_in_n = c->pin_in ? n_in : 0;
    }
    if (c->pin_out_n < n_out) {
        nntk_shim_synchronize();
        nntk_shim_host_free(c->pin_out);
        c->pin_out = (float *)nntk_shim_host_alloc(n_out * sizeof(float));
        c->pin_out_n = c->pin_out ? n_out : 0;
    }
    if (!c->flag) {
        c->flag = (volatile unsigned *)nntk_shim_host_alloc(64);
        c->d_done = (unsigned *)nntk_shim_malloc(64);
        if (c->d_done && nntk_shim_memset(c->d_done, 0, 64)) return -1;
    }
    if (!c->pin_in || !c->pin_out || !c->flag || !c->d_done) return -1;
    memcpy(c->pin_in, input, n_in * sizeof(float));
    const unsigned seq = ++c->seq ? c->seq : ++c->seq;         /* never 0: the word's initial value */
    if (nntk_shim_rec_stream(c->G, is_lstm, c->pin_in, c->d_wp, c->d_bi, c->d_ut, use_bh ? c->d_bh : NULL,
                             c->d_h[0], c->d_h[1], c->d_c[0], c->d_c[1], c->cur, c->pin_out, c->T, c->in, c->H,
                             c->return_sequences, acts, scales, c->d_done, (unsigned *)c->flag, seq))
        return -1;
    /* spin on the pinned word; after 5 ms fall back to a stream synchronisation (which also surfaces launch errors) */
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0; *c->flag != seq; ++spins) {
        if ((spins & 1023) == 1023) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 5000000L) {
                /* error paths: the last launch may not have counted every workgroup in, and only the workgroup that
                 * counts in last resets the counter -- clear it here, or every later call on this handle would wait
                 * for a count that is never reached */
                if (nntk_shim_synchronize()) { (void)nntk_shim_memset(c->d_done, 0, 64); return -1; }
                if (*c->flag != seq) {
                    (void)nntk_shim_memset(c->d_done, 0, 64);
                    (void)nntk_shim_synchronize();
                    NNTK_FAIL("streaming recurrent kernel did not complete");
                }
                break;
            }
        }
        NNTK_CPU_RELAX();
    }
    memcpy(output, c->pin_out, n_out * sizeof(float));
    c->cur = (c->cur + c->T) & 1;
    return 0;
}

static int core_apply_host(rec_core *c, int is_lstm, int use_bh, const int *acts, const float *scales,
                           const float *input, float *output, int B, int stateful) {
    if (B <= 0) return 0;
    if (stateful && B == 1 && c->T >= 1 && c->T <= NNTK_STREAM_MAX_T) {
        int on = -1;
        (void)nntk_shim_get_option("rec_stream", &on);
        if (on != 0) {
            if (core_ensure(c, 2)) return -1;          /* latency path: sampled edit check (runtime.c) */
            return core_apply_stream(c, is_lstm, use_bh, acts, scales, input, output);
        }
    }
    if (core_ensure(c, 1)) return -1;
    size_t n_in = (size_t)B * c->T * c->in;
    size_t n_out = c->return_sequences ? (size_t)B * c->T * c->H : (size_t)B * c->H;
    float *d_in = nntk_devbuf_reserve(&c->d_in, n_in);
    float *d_out = nntk_devbuf_reserve(&c->d_out, n_out);
    if (!d_in || !d_out) return -1;
    if (nntk_shim_upload(d_in, input, n_in * sizeof(float))) return -1;
    if (core_apply_device(c, is_lstm, use_bh, acts, scales, d_in, d_out, B, stateful)) return -1;
    if (nntk_shim_download_nocheck(output, d_out, n_out * sizeof(float))) return -1;
    if (nntk_shim_take_fault()) {
        /* the persistent kernel's workgroups were not all resident (another kernel held the CUs) and it gave up:
         * the library has switched to the per-timestep kernels, which need no co-residency and produce the same
         * bits; repeat this call on them.  The initial state is intact (double-buffered above). */
        if (core_apply_device(c, is_lstm, use_bh, acts, scales, d_in, d_out, B, stateful)) return -1;
        if (nntk_shim_download(output, d_out, n_out * sizeof(float))) return -1;
    }
    if (stateful) c->cur ^= 1;
    return 0;
}

/* kind + ReLU's output scale (activation_default.c:123-129) of one gate activation */
static int gate_kind(ActivationFunction a, int *kind, float *scale) {
    if (!a) NNTK_FAIL("recurrent layer: NULL gate activation");
    if (!nntk_act_fusable(a))
        NNTK_FAIL("recurrent layer: gate activations must be built-in identity/sigmoid/tanh/relu "
                  "(custom host callbacks and softmax cannot run inside the device step kernel)");
    *kind = a->kind;
    *scale = a->kind == NNTK_ACT_RELU ? a->relu_a : 1.0f;
    return 0;
}

/* ================================= GRU ==================================== */

/* training state shared by the recurrent layers (gru.c:85-108): the mini-batch input and the forward caches on the device */
typedef struct {
    int on, mini_batch, have_batch;
    const float *d_x_cur;        /* the mini-batch input of the last forward: d_x below (host-pointer calls) or the caller's device tensor */
    nntk_devbuf d_x, d_h, d_Zg, d_hU, d_dxW, d_dhU, d_work, d_raw, d_grad, d_scr, d_dout, d_dX;
} rec_train;
static void train_free(rec_train *t) {
    nntk_devbuf_free(&t->d_x); nntk_devbuf_free(&t->d_h); nntk_devbuf_free(&t->d_Zg); nntk_devbuf_free(&t->d_hU);
    nntk_devbuf_free(&t->d_dxW); nntk_devbuf_free(&t->d_dhU); nntk_devbuf_free(&t->d_work); nntk_devbuf_free(&t->d_raw);
    nntk_devbuf_free(&t->d_grad); nntk_devbuf_free(&t->d_scr); nntk_devbuf_free(&t->d_dout); nntk_devbuf_free(&t->d_dX);
}

struct GRUStruct {
    GRUConfig config;
    rec_core core;
    rec_train train;
};

/* gru.c:13-19 */
GRUConfig GRUConfigCreate(int input_feature_channels, int output_feature_channels, bool return_sequences,
                          int timesteps, GRUActivations activations) {
    GRUConfig c;
    memset(&c, 0, sizeof(c));
    c.base = RecurrentConfigCreate(input_feature_channels, output_feature_channels, return_sequences, timesteps);
    c.activations = activations;
    return c;
}

/* gru.c:206-212 */
GRUActivations GRUActivationsCreateDefault(int size) {
    GRUActivations a;
    a.z_gate_activation = ActivationFunctionCreateSigmoid(size);
    a.r_gate_activation = ActivationFunctionCreateSigmoid(size);
    a.h_gate_activation = ActivationFunctionCreateTanh(size);
    return a;
}
/* gru.c:220-230: argument order is (z, h, r) */
GRUActivations GRUActivationsCreate(ActivationFunction z_gate_activation, ActivationFunction h_gate_activation,
                                    ActivationFunction r_gate_activation) {
    GRUActivations a;
    a.z_gate_activation = z_gate_activation;
    a.h_gate_activation = h_gate_activation;
    a.r_gate_activation = r_gate_activation;
    return a;
}
void GRUActivationsDestroy(GRUActivations activations) {
    ActivationFunctionDestroy(activations.z_gate_activation);
    ActivationFunctionDestroy(activations.r_gate_activation);
    ActivationFunctionDestroy(activations.h_gate_activation);
}

GRU GRUCreateForInference(GRUConfig config) {
    nntk_shim_clear_error();
    GRU f = (GRU)calloc(1, sizeof(struct GRUStruct));
    if (!f) return NULL;
    f->config = config;
    if (core_init(&f->core, 3, config.base)) { free(f); return NULL; }
    return f;
}
GRUWeights *GRUGetWeights(GRU filter) { return filter->core.weights; }
void GRUDestroy(GRU filter) {
    if (!filter) return;
    core_free(&filter->core);   /* activations stay with the caller (gru.c:116-126) */
    train_free(&filter->train);
    free(filter);
}

static int gru_acts(GRU f, int acts[3], float scales[3]) {
    if (gate_kind(f->config.activations.z_gate_activation, &acts[0], &scales[0])) return -1;
    if (gate_kind(f->config.activations.h_gate_activation, &acts[1], &scales[1])) return -1;
    if (gate_kind(f->config.activations.r_gate_activation, &acts[2], &scales[2])) return -1;
    return 0;
}

int GRUBroadcastWeights(GRU filter, int root) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("GRUBroadcastWeights: NULL handle");
    return core_broadcast(&filter->core, root);
}
int GRUSyncWeights(GRU filter) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("GRUSyncWeights: NULL handle");
    nntk_shim_synchronize();
    return core_upload(&filter->core);
}

/* gru.c:189-204 */
/* ---- training: gru.c:232-244 (create), :246-293 (forward keeping Z_gates, h_pr_Uh, h), :295-512 (BPTT).
 *      One launch per timestep and direction, VALU dots in the reference's operation order: correct and deterministic,
 *      not tuned (csrc/hip/train.hip). ---- */
GRU GRUCreateForTraining(GRUConfig config, GRUTrainingConfig training_config) {
    GRU f = GRUCreateForInference(config);
    if (!f) return NULL;
    f->train.on = 1;
    f->train.mini_batch = training_config.mini_batch_size;
    return f;
}
/* ONE zeroed block d_W | d_U | d_b_i | d_b_h | d_X (recurrent_private.c:10-22) */
GRUGradient *GRUGradientCreate(GRUConfig config, GRUTrainingConfig training_config) {
    GRUGradient *g = (GRUGradient *)malloc(sizeof(GRUGradient));
    if (!g) return NULL;
    size_t in = (size_t)config.base.input_feature_channels, H = (size_t)config.base.output_feature_channels;
    size_t w = in * 3 * H, u = H * 3 * H, b = 3 * H;
    size_t x = (size_t)training_config.mini_batch_size * in * config.base.timesteps;
    g->d_W = (float *)calloc(w + u + 2 * b + x + 1, sizeof(float));
    if (!g->d_W) { free(g); return NULL; }
    g->d_U = g->d_W + w;
    g->d_b_i = g->d_U + u;
    g->d_b_h = g->d_b_i + b;
    g->d_X = g->d_b_h + b;
    return g;
}
void RecurrentGradientDestroy(RecurrentGradient *gradient) {
    if (!gradient) return;
    free(gradient->d_W);
    free(gradient);
}

/* forward over the mini-batch with the caches kept for the gradient; d_x is a device tensor that stays valid until the gradient call */
static int gru_train_forward_dev(GRU filter, const float *d_x) {
    int acts[3];
    float sc[3];
    if (gru_acts(filter, acts, sc)) return -1;
    rec_core *c = &filter->core;
    rec_train *t = &filter->train;
    const int B = t->mini_batch, T = c->T, in = c->in, H = c->H;
    const size_t nw = (size_t)in * 3 * H + (size_t)H * 3 * H + 6 * (size_t)H;
    float *d_h = nntk_devbuf_reserve(&t->d_h, (size_t)B * T * H);
    float *d_Zg = nntk_devbuf_reserve(&t->d_Zg, (size_t)B * T * 6 * H);
    float *d_hU = nntk_devbuf_reserve(&t->d_hU, (size_t)B * T * H);
    float *d_raw = nntk_devbuf_reserve(&t->d_raw, nw);
    if (!d_h || !d_Zg || !d_hU || !d_raw) return -1;
    if (nntk_shim_upload(d_raw, c->wb.host, nw * sizeof(float))) return -1;          /* W | U | b_i | b_h, caller layout */
    const float *dW = d_raw, *dU = dW + (size_t)in * 3 * H, *dbi = dU + (size_t)H * 3 * H, *dbh = dbi + 3 * (size_t)H;
    /* default activations, any mini-batch: ONE launch of the register-resident kernel with the caches written from
     * its gate phase (recurrent_rr.hip gru_rr_kernel<.., TRAIN>); its image is re-packed from the current weights every call */
    int ran = 0, rr_on = -1;
    (void)nntk_shim_get_option("rec_rr", &rr_on);
    size_t img = nntk_shim_lstm_rr_image_floats(H, in);
    if (rr_on != 0 && img && gru_std_acts(acts) && !rr_unsplittable(c->wb.host, (size_t)in * 3 * H + (size_t)H * 3 * H)) {
        float *d_wk = nntk_devbuf_reserve(&c->d_work_rr, nntk_shim_lstm_rr_work_floats(B, H));
        if (!d_wk) return -1;
        if (gru_rr_build_image(in, H, c->weights->W, c->weights->U, c->weights->b_i, c->weights->b_h, &c->d_rr_train, img, &c->d_b4_train, &c->d_rr_stage, NULL)) return -1;
        float *d_hs = nntk_devbuf_reserve(&c->d_hseq, nntk_shim_rr_hseq_floats(B, T, H));
        if (!d_hs) return -1;
        int rc = nntk_shim_gru_rr_train_forward(d_x, c->d_rr_train, c->d_b4_train, d_h, d_hU, d_Zg, d_hs, d_wk, B, T, in, H);
        if (rc < 0) return -1;
        ran = rc == 0;
    }
    if (!ran && nntk_shim_gru_train_forward(d_x, dW, dU, dbi, dbh, d_h, d_Zg, d_hU, B, T, in, H, acts, sc)) return -1;
    t->d_x_cur = d_x;
    t->have_batch = 1;
    return 0;
}
int GRUApplyTrainingBatch(GRU filter, const float *input, float *output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("GRUApplyTrainingBatch: NULL handle");
    if (!filter->train.on) NNTK_FAIL("GRUApplyTrainingBatch: the handle was created for inference");      /* gru.c:247-249 */
    rec_core *c = &filter->core;
    rec_train *t = &filter->train;
    const int B = t->mini_batch, T = c->T, in = c->in, H = c->H;
    if (B <= 0 || T <= 0) return 0;
    float *d_x = nntk_devbuf_reserve(&t->d_x, (size_t)B * T * in);
    if (!d_x) return -1;
    if (nntk_shim_upload(d_x, input, (size_t)B * T * in * sizeof(float))) return -1;
    if (gru_train_forward_dev(filter, d_x)) return -1;
    const float *d_h = t->d_h.p;
    if (c->return_sequences) return nntk_shim_download(output, d_h, (size_t)B * T * H * sizeof(float));
    /* the last step of every sequence (gru.c:286-291, lstm.c:466-471, rnn.c:283-288): one strided copy */
    return nntk_shim_download_rows(output, d_h + (size_t)(T - 1) * H, (size_t)T * H * sizeof(float), (size_t)H * sizeof(float), (size_t)B);
}
/* device-pointer form: d_input [B][T][in] must stay valid until GRUCalculateGradientDevice; d_output [B][T][H] or [B][H] */
int GRUApplyTrainingBatchDevice(GRU filter, const float *d_input, float *d_output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("GRUApplyTrainingBatchDevice: NULL handle");
    if (!filter->train.on) NNTK_FAIL("GRUApplyTrainingBatchDevice: the handle was created for inference");
    rec_core *c = &filter->core;
    rec_train *t = &filter->train;
    const int B = t->mini_batch, T = c->T, H = c->H;
    if (B <= 0 || T <= 0) return 0;
    if (gru_train_forward_dev(filter, d_input)) return -1;
    if (!d_output) return 0;
    if (c->return_sequences) return nntk_shim_copy_d2d(d_output, t->d_h.p, (size_t)B * T * H * sizeof(float));
    return nntk_shim_copy_rows_d2d(d_output, t->d_h.p + (size_t)(T - 1) * H, (size_t)T * H * sizeof(float), (size_t)H * sizeof(float), (size_t)B);
}

/* d_W, d_U, d_b_i, d_b_h are ADDED onto the caller's block (recurrent_gradient_sum per (b, t), gru.c:508), d_X is
 * overwritten.  void in the reference; errors through nntk_last_error(). */
static int gru_train_gradient_dev(GRU filter, const float *d_dout, float *d_grad, float *d_dX) {
    rec_core *c = &filter->core;
    rec_train *t = &filter->train;
    int acts[3];
    float sc[3];
    if (gru_acts(filter, acts, sc)) return -1;
    const int B = t->mini_batch, T = c->T, in = c->in, H = c->H;
    const size_t w = (size_t)in * 3 * H, u = (size_t)H * 3 * H, b3 = 3 * (size_t)H, rows = (size_t)B * T;
    float *d_dxW = nntk_devbuf_reserve(&t->d_dxW, rows * 3 * H);
    float *d_dhU = nntk_devbuf_reserve(&t->d_dhU, rows * 3 * H);
    float *d_work = nntk_devbuf_reserve(&t->d_work, (size_t)B * 5 * H);
    float *d_UT = nntk_devbuf_reserve(&t->d_scr, u);
    if (!d_dxW || !d_dhU || !d_work || !d_UT) return -1;
    const float *dW = t->d_raw.p, *dU = dW + w;
    if (nntk_shim_transpose(dU, d_UT, H, 3 * H, 0)) return -1;                  /* U^T [3H][H]: coalesced per-step product */
    if (nntk_shim_gru_train_backward(d_dout, d_UT, t->d_h.p, t->d_Zg.p, t->d_hU.p, d_dxW, d_dhU, d_work, B, T, H,
                                     c->return_sequences ? 1 : 0, acts)) return -1;
    /* d_W += x^T d_xW, d_b_i += colsum d_xW;  d_U += h_prev^T d_hU, d_b_h += colsum d_hU;  d_X = d_xW W^T */
    if (nntk_train_outer_accumulate(t->d_x_cur, d_dxW, d_grad, d_grad + w + u, (long)rows, in, 3 * H, 0)) return -1;
    if (nntk_train_outer_accumulate(t->d_h.p, d_dhU, d_grad + w, d_grad + w + u + b3, (long)rows, H, 3 * H, T)) return -1;
    return nntk_train_rows_times_rowmat(d_dxW, dW, d_dX, (long)rows, in, 3 * H);
}
void GRUCalculateGradient(GRU filter, GRUGradient *gradient, float *d_out) {
    nntk_shim_clear_error();
    if (!filter || !gradient || !d_out) { nntk_set_error("GRUCalculateGradient: NULL argument"); return; }
    rec_core *c = &filter->core;
    rec_train *t = &filter->train;
    if (!t->on || !t->have_batch) { nntk_set_error("GRUCalculateGradient: run GRUApplyTrainingBatch on a training handle first"); return; }
    const int B = t->mini_batch, T = c->T, in = c->in, H = c->H;
    const size_t w = (size_t)in * 3 * H, u = (size_t)H * 3 * H, b3 = 3 * (size_t)H, rows = (size_t)B * T;
    const size_t n_do = c->return_sequences ? rows * H : (size_t)B * H;
    float *d_dout = nntk_devbuf_reserve(&t->d_dout, n_do);
    float *d_grad = nntk_devbuf_reserve(&t->d_grad, w + u + 2 * b3);
    float *d_dX = nntk_devbuf_reserve(&t->d_dX, rows * in);
    if (!d_dout || !d_grad || !d_dX) return;
    if (nntk_shim_upload(d_dout, d_out, n_do * sizeof(float))) return;
    if (nntk_shim_upload(d_grad, gradient->d_W, (w + u + 2 * b3) * sizeof(float))) return;       /* the block is contiguous */
    if (gru_train_gradient_dev(filter, d_dout, d_grad, d_dX)) return;
    if (nntk_shim_download(gradient->d_W, d_grad, (w + u + 2 * b3) * sizeof(float))) return;
    nntk_shim_download(gradient->d_X, d_dX, rows * in * sizeof(float));
}
/* device-pointer form: d_grad = W [in][3H] | U [H][3H] | b_i [3H] | b_h [3H] (the gradient block's layout) is ADDED to, d_dX
 * [B][T][in] overwritten; d_dout [B][T][H] or [B][H].  Enqueued on the calling thread's stream (the forward call that precedes it
 * synchronised that stream once for its weight upload). */
int GRUCalculateGradientDevice(GRU filter, float *d_grad, float *d_dX, const float *d_dout) {
    nntk_shim_clear_error();
    if (!filter || !d_grad || !d_dX || !d_dout) NNTK_FAIL("GRUCalculateGradientDevice: NULL argument");
    if (!filter->train.on || !filter->train.have_batch) NNTK_FAIL("GRUCalculateGradientDevice: run GRUApplyTrainingBatch[Device] on a training handle first");
    return gru_train_gradient_dev(filter, d_dout, d_grad, d_dX);
}

int GRUApplyInference(GRU filter, const float *input, float *output) {
    nntk_shim_clear_error();
    int acts[3];
    float sc[3];
    if (!filter) NNTK_FAIL("GRUApplyInference: NULL handle");
    if (filter->train.on) NNTK_FAIL("GRUApplyInference: the handle was created for training");           /* gru.c:190-192 */
    if (gru_acts(filter, acts, sc)) return -1;
    return core_apply_host(&filter->core, 0, 1, acts, sc, input, output, 1, 1);
}
/* gru.c:246-293 forward semantics */
int GRUApplyInferenceBatch(GRU filter, const float *input, float *output, int batch) {
    nntk_shim_clear_error();
    int acts[3];
    float sc[3];
    if (!filter) NNTK_FAIL("GRUApplyInferenceBatch: NULL handle");
    if (gru_acts(filter, acts, sc)) return -1;
    return core_apply_host(&filter->core, 0, 1, acts, sc, input, output, batch, 0);
}
int GRUApplyDevice(GRU filter, const float *d_input, float *d_output, int batch) {
    nntk_shim_clear_error();
    int acts[3];
    float sc[3];
    if (!filter) NNTK_FAIL("GRUApplyDevice: NULL handle");
    if (gru_acts(filter, acts, sc)) return -1;
    if (core_ensure(&filter->core, 0)) return -1;
    return core_apply_device(&filter->core, 0, 1, acts, sc, d_input, d_output, batch, 0);
}
int GRUApplyDeviceFrag3(GRU filter, const float *d_input, const float *d_input_frag3, float *d_output, float *d_output_frag3, int batch) {
    nntk_shim_clear_error();
    int acts[3];
    float sc[3];
    if (!filter) NNTK_FAIL("GRUApplyDeviceFrag3: NULL handle");
    if (gru_acts(filter, acts, sc)) return -1;
    if (core_ensure(&filter->core, 0)) return -1;
    return core_apply_device_f3(&filter->core, 0, 1, acts, sc, d_input, d_input_frag3, d_output, d_output_frag3, batch);
}
const char *GRUKernelPlan(GRU filter) {
    static _Thread_local char buf[320];
    if (!filter) return "";
    int a[3]; float sc[3];
    if (gru_acts(filter, a, sc)) return "gru: invalid gate activations";
    return core_plan(&filter->core, gru_std_acts(a), buf, sizeof buf);
}
int GRUResetState(GRU filter) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("GRUResetState: NULL handle");
    return core_reset_state(&filter->core);
}
int GRUGetState(GRU filter, float *h_host) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("GRUGetState: NULL handle");
    return nntk_shim_download(h_host, filter->core.d_h[filter->core.cur], (size_t)filter->core.H * sizeof(float));
}

/* ---- two stacked GRU layers in one persistent launch (BASELINE configs[3]; recurrent.hip gru2_persistent_kernel) ----
 * Same results as GRUApplyDevice(l1) followed by GRUApplyDevice(l2) on zero initial state (layer 1 bit for bit, layer 2
 * within the layer tolerance: its input projection is summed in another order); falls back to exactly those two calls
 * for shapes / activations the fused kernel does not take. */
static int gru_default_acts(GRU f) {
    int a[3]; float sc[3];
    if (gru_acts(f, a, sc)) return -1;
    return (a[0] == NNTK_ACT_SIGMOID && a[1] == NNTK_ACT_TANH && a[2] == NNTK_ACT_SIGMOID) ? 1 : 0;
}
int GRUStack2ApplyDevice(GRU l1, GRU l2, const float *d_input, float *d_output, int batch) {
    nntk_shim_clear_error();
    if (!l1 || !l2) NNTK_FAIL("GRUStack2ApplyDevice: NULL handle");
    rec_core *c1 = &l1->core, *c2 = &l2->core;
    if (!c1->return_sequences || c2->in != c1->H || c2->T != c1->T)
        NNTK_FAIL("GRUStack2ApplyDevice: layer 1 must return sequences and feed layer 2 (in2 = H1, same timesteps)");
    if (batch <= 0) return 0;
    if (core_ensure(c1, 0) || core_ensure(c2, 0)) return -1;
    const int H = c1->H, T = c1->T, B = batch;
    int d1 = gru_default_acts(l1), d2 = gru_default_acts(l2);
    if (d1 < 0 || d2 < 0) return -1;
    /* both layers on the register-resident split-bf16 kernels (gru_rr_kernel, one launch per layer, projections fused) when their
     * shapes allow it: measured faster than the fused exact-f32 kernel below (DESIGN K4b); rec_fused2 = 1 forces the fused kernel */
    int rr_on = -1, fused_on = -1;
    (void)nntk_shim_get_option("rec_rr", &rr_on);
    (void)nntk_shim_get_option("rec_fused2", &fused_on);
    const int rr_pair = rr_on != 0 && d1 && d2 && !c1->rr_exact_only && !c2->rr_exact_only &&
                        nntk_shim_rr_image_floats_xf(c1->H, c1->in) && nntk_shim_rr_image_floats_xf(c2->H, c2->in);
    if (d1 && d2 && c2->H == H && !(rr_pair && fused_on != 1)) {
        if (core_ensure_wt(c2)) return -1;
        float *d_xw = nntk_devbuf_reserve(&c1->d_xw, (size_t)T * B * 3 * H);
        float *d_work = nntk_devbuf_reserve(&c1->d_work, nntk_shim_gru2_work_floats(B, H) > nntk_shim_recurrent_work_floats(B, H)
                                                           ? nntk_shim_gru2_work_floats(B, H) : nntk_shim_recurrent_work_floats(B, H));
        if (!d_xw || !d_work) return -1;
        if (nntk_shim_conv1d(d_input, c1->d_wp, c1->d_bi, NULL, 0.f, NNTK_ACT_IDENTITY, 1.f, d_xw, B, T, c1->in, 3 * H, 1, 1, T, 1))
            return -1;
        int rc = nntk_shim_gru2(d_xw, c1->d_ut, c1->d_bh, c2->d_wt, c2->d_bi, c2->d_ut, c2->d_bh, d_output, NULL, d_work,
                                B, T, H, c2->return_sequences);
        if (rc <= 0) return rc;
    }
    /* the register-resident pair: layer 1 publishes h1_t ALREADY SPLIT, in layer 2's operand order, into a T-deep buffer -- its own
     * hand-off, so not one store more than a single layer issues -- and layer 2 reads that buffer as its x operand: 12 coalesced 1 KB
     * requests per half-step, no f32 inter-layer tensor, no split in the consumer (DESIGN K4b).  The values are the f32 h1_t exactly
     * (hi + mid + lo), so the result equals GRUApplyDevice(l1) then GRUApplyDevice(l2) bit for bit. */
    if (rr_pair) {
        int a1[3], a2[3];
        float s1[3], s2[3];
        if (gru_acts(l1, a1, s1) || gru_acts(l2, a2, s2)) return -1;
        float *d_h1 = nntk_devbuf_reserve(&c1->d_hseq, nntk_shim_rr_hseq_floats(B, T, H));
        if (!d_h1) return -1;
        const rr_io io1 = { d_input, NULL, NULL, d_h1, NULL };
        int rc = core_try_gru_rr(c1, a1, &io1, B, 0);
        if (rc < 0) return -1;
        if (rc == 0) {
            const rr_io io2 = { NULL, d_h1, d_output, NULL, NULL };
            rc = core_try_gru_rr(c2, a2, &io2, B, 0);
            if (rc <= 0) return rc;
            /* layer 2 not taken after all (cannot happen for a pair that qualified): both layers again, through an f32 tensor */
        }
    }
    float *d_mid = nntk_devbuf_reserve(&c1->d_out, (size_t)B * T * H);
    if (!d_mid) return -1;
    if (GRUApplyDevice(l1, d_input, d_mid, batch)) return -1;
    return GRUApplyDevice(l2, d_mid, d_output, batch);
}
int GRUStack2ApplyInferenceBatch(GRU l1, GRU l2, const float *input, float *output, int batch) {
    nntk_shim_clear_error();
    if (!l1 || !l2) NNTK_FAIL("GRUStack2ApplyInferenceBatch: NULL handle");
    if (batch <= 0) return 0;
    rec_core *c1 = &l1->core, *c2 = &l2->core;
    if (core_ensure(c1, 1) || core_ensure(c2, 1)) return -1;
    size_t n_in = (size_t)batch * c1->T * c1->in;
    size_t n_out = c2->return_sequences ? (size_t)batch * c2->T * c2->H : (size_t)batch * c2->H;
    float *d_in = nntk_devbuf_reserve(&c1->d_in, n_in);
    float *d_out = nntk_devbuf_reserve(&c2->d_out, n_out);
    if (!d_in || !d_out) return -1;
    if (nntk_shim_upload(d_in, input, n_in * sizeof(float))) return -1;
    if (GRUStack2ApplyDevice(l1, l2, d_in, d_out, batch)) return -1;
    if (nntk_shim_download_nocheck(output, d_out, n_out * sizeof(float))) return -1;
    if (nntk_shim_take_fault()) {      /* persistent launch faulted: repeat on the per-timestep kernels (see core_apply_host) */
        if (GRUStack2ApplyDevice(l1, l2, d_in, d_out, batch)) return -1;
        if (nntk_shim_download(output, d_out, n_out * sizeof(float))) return -1;
    }
    return 0;
}

/* ================================= LSTM =================================== */

struct LSTMStruct {
    LSTMConfig config;
    rec_core core;
    rec_train train;            /* d_Zg = zifgo [B][T][8H], d_hU = the cell state c [B][T][H] */
};

/* lstm.c:17-26: argument order (input, forget, candidate, output_gate, output) */
LSTMActivations LSTMActivationsCreate(ActivationFunction input_gate_activation,
                                      ActivationFunction forget_gate_activation,
                                      ActivationFunction candidate_gate_activation,
                                      ActivationFunction output_gate_activation,
                                      ActivationFunction output_activation) {
    LSTMActivations a;
    a.candidate_gate_activation = candidate_gate_activation;
    a.input_gate_activation = input_gate_activation;
    a.forget_gate_activation = forget_gate_activation;
    a.output_gate_activation = output_gate_activation;
    a.output_activation = output_activation;
    return a;
}
/* lstm.c:110-118 */
LSTMActivations LSTMActivationsCreateDefault(int size) {
    return LSTMActivationsCreate(ActivationFunctionCreateSigmoid(size), ActivationFunctionCreateSigmoid(size),
                                 ActivationFunctionCreateTanh(size), ActivationFunctionCreateSigmoid(size),
                                 ActivationFunctionCreateTanh(size));
}
void LSTMActivationsDestroy(LSTMActivations activations) {
    ActivationFunctionDestroy(activations.input_gate_activation);
    ActivationFunctionDestroy(activations.forget_gate_activation);
    ActivationFunctionDestroy(activations.candidate_gate_activation);
    ActivationFunctionDestroy(activations.output_gate_activation);
    ActivationFunctionDestroy(activations.output_activation);
}
/* lstm.c:132-138 */
LSTMConfig LSTMConfigCreate(int input_feature_channels, int output_feature_channels, bool return_sequences,
                            int timesteps, bool v2, LSTMActivations activations) {
    LSTMConfig c;
    memset(&c, 0, sizeof(c));
    c.base = RecurrentConfigCreate(input_feature_channels, output_feature_channels, return_sequences, timesteps);
    c.v2 = v2;
    c.activations = activations;
    return c;
}

LSTM LSTMCreateForInference(LSTMConfig config) {
    nntk_shim_clear_error();
    LSTM f = (LSTM)calloc(1, sizeof(struct LSTMStruct));
    if (!f) return NULL;
    f->config = config;
    if (core_init(&f->core, 4, config.base)) { free(f); return NULL; }
    return f;
}
LSTMWeights *LSTMGetWeights(LSTM filter) { return filter->core.weights; }
void LSTMDestroy(LSTM filter) {
    if (!filter) return;
    core_free(&filter->core);
    train_free(&filter->train);
    free(filter);
}

static int lstm_acts(LSTM f, int acts[5], float scales[5]) {
    const LSTMActivations *a = &f->config.activations;
    if (gate_kind(a->input_gate_activation, &acts[0], &scales[0])) return -1;
    if (gate_kind(a->forget_gate_activation, &acts[1], &scales[1])) return -1;
    if (gate_kind(a->candidate_gate_activation, &acts[2], &scales[2])) return -1;
    if (gate_kind(a->output_gate_activation, &acts[3], &scales[3])) return -1;
    if (gate_kind(a->output_activation, &acts[4], &scales[4])) return -1;
    return 0;
}

int LSTMBroadcastWeights(LSTM filter, int root) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("LSTMBroadcastWeights: NULL handle");
    return core_broadcast(&filter->core, root);
}
int LSTMSyncWeights(LSTM filter) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("LSTMSyncWeights: NULL handle");
    nntk_shim_synchronize();
    return core_upload(&filter->core);
}

/* lstm.c:241-268 */
/* ---- training: lstm.c:418-475 (forward keeping zifgo, c, h), :294-416 + :477-556 (BPTT); same
 *      structure and caveats as the GRU path ---- */
LSTM LSTMCreateForTraining(LSTMConfig config, LSTMTrainingConfig training_config) {
    LSTM f = LSTMCreateForInference(config);
    if (!f) return NULL;
    f->train.on = 1;
    f->train.mini_batch = training_config.mini_batch_size;
    return f;
}
LSTMGradient *LSTMGradientCreate(LSTMConfig config, LSTMTrainingConfig training_config) {
    LSTMGradient *g = (LSTMGradient *)malloc(sizeof(LSTMGradient));
    if (!g) return NULL;
    size_t in = (size_t)config.base.input_feature_channels, H = (size_t)config.base.output_feature_channels;
    size_t w = in * 4 * H, u = H * 4 * H, b = 4 * H;
    size_t x = (size_t)training_config.mini_batch_size * in * config.base.timesteps;
    g->d_W = (float *)calloc(w + u + 2 * b + x + 1, sizeof(float));
    if (!g->d_W) { free(g); return NULL; }
    g->d_U = g->d_W + w;
    g->d_b_i = g->d_U + u;
    g->d_b_h = g->d_b_i + b;
    g->d_X = g->d_b_h + b;
    return g;
}

static int lstm_train_forward_dev(LSTM filter, const float *d_x) {
    int acts[5];
    float sc[5];
    if (lstm_acts(filter, acts, sc)) return -1;
    rec_core *c = &filter->core;
    rec_train *t = &filter->train;
    const int B = t->mini_batch, T = c->T, in = c->in, H = c->H;
    const size_t nw = (size_t)in * 4 * H + (size_t)H * 4 * H + 8 * (size_t)H;
    float *d_h = nntk_devbuf_reserve(&t->d_h, (size_t)B * T * H);
    float *d_z = nntk_devbuf_reserve(&t->d_Zg, (size_t)B * T * 8 * H);
    float *d_c = nntk_devbuf_reserve(&t->d_hU, (size_t)B * T * H);
    float *d_raw = nntk_devbuf_reserve(&t->d_raw, nw);
    if (!d_h || !d_z || !d_c || !d_raw) return -1;
    if (nntk_shim_upload(d_raw, c->wb.host, nw * sizeof(float))) return -1;          /* W | U | b_i | b_h, caller layout */
    const float *dW = d_raw, *dU = dW + (size_t)in * 4 * H, *dbi = dU + (size_t)H * 4 * H, *dbh = dbi + 4 * (size_t)H;
    /* standard activations, any mini-batch: the register-resident inference kernel with the caches written
     * from its gate phase (recurrent_rr.hip, TRAIN): ONE launch instead of T; its weight images are packed on the device from
     * the block just uploaded (the weights change with every optimiser step).  Otherwise: one launch per timestep. */
    int ran = 0, rr_on = -1;
    (void)nntk_shim_get_option("rec_rr", &rr_on);
    size_t img = nntk_shim_lstm_rr_image_floats(H, in);
    if (rr_on != 0 && img && lstm_std_acts(acts) && !rr_unsplittable(c->wb.host, (size_t)in * 4 * H + (size_t)H * 4 * H)) {
        if (!c->d_rr_train && !(c->d_rr_train = (float *)nntk_shim_malloc(img * sizeof(float)))) return -1;
        float *d_wk = nntk_devbuf_reserve(&c->d_work_rr, nntk_shim_lstm_rr_work_floats(B, H));
        if (!d_wk) return -1;
        if (nntk_shim_lstm_rr_pack_raw(dU, dW, c->d_rr_train, H, in)) return -1;
        float *d_hs = nntk_devbuf_reserve(&c->d_hseq, nntk_shim_rr_hseq_floats(B, T, H));
        if (!d_hs) return -1;
        int rc = nntk_shim_lstm_rr_train_forward(d_x, c->d_rr_train, dbi, filter->config.v2 ? dbh : NULL, d_h, d_c, d_z, d_hs, d_wk, B, T, in, H);
        if (rc < 0) return -1;
        ran = rc == 0;
    }
    if (!ran && nntk_shim_lstm_train_forward(d_x, dW, dU, dbi, dbh, d_h, d_c, d_z, B, T, in, H, filter->config.v2 ? 1 : 0, acts, sc)) return -1;
    t->d_x_cur = d_x;
    t->have_batch = 1;
    return 0;
}
int LSTMApplyTrainingBatch(LSTM filter, const float *input, float *output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("LSTMApplyTrainingBatch: NULL handle");
    if (!filter->train.on) NNTK_FAIL("LSTMApplyTrainingBatch: the handle was created for inference");    /* lstm.c:419-421 */
    rec_core *c = &filter->core;
    rec_train *t = &filter->train;
    const int B = t->mini_batch, T = c->T, in = c->in, H = c->H;
    if (B <= 0 || T <= 0) return 0;
    float *d_x = nntk_devbuf_reserve(&t->d_x, (size_t)B * T * in);
    if (!d_x) return -1;
    if (nntk_shim_upload(d_x, input, (size_t)B * T * in * sizeof(float))) return -1;
    if (lstm_train_forward_dev(filter, d_x)) return -1;
    const float *d_h = t->d_h.p;
    if (c->return_sequences) return nntk_shim_download(output, d_h, (size_t)B * T * H * sizeof(float));
    /* the last step of every sequence (gru.c:286-291, lstm.c:466-471, rnn.c:283-288): one strided copy */
    return nntk_shim_download_rows(output, d_h + (size_t)(T - 1) * H, (size_t)T * H * sizeof(float), (size_t)H * sizeof(float), (size_t)B);
}
/* device-pointer form: d_input [B][T][in] must stay valid until LSTMCalculateGradientDevice; d_output [B][T][H] or [B][H] */
int LSTMApplyTrainingBatchDevice(LSTM filter, const float *d_input, float *d_output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("LSTMApplyTrainingBatchDevice: NULL handle");
    if (!filter->train.on) NNTK_FAIL("LSTMApplyTrainingBatchDevice: the handle was created for inference");
    rec_core *c = &filter->core;
    rec_train *t = &filter->train;
    const int B = t->mini_batch, T = c->T, H = c->H;
    if (B <= 0 || T <= 0) return 0;
    if (lstm_train_forward_dev(filter, d_input)) return -1;
    if (!d_output) return 0;
    if (c->return_sequences) return nntk_shim_copy_d2d(d_output, t->d_h.p, (size_t)B * T * H * sizeof(float));
    return nntk_shim_copy_rows_d2d(d_output, t->d_h.p + (size_t)(T - 1) * H, (size_t)T * H * sizeof(float), (size_t)H * sizeof(float), (size_t)B);
}

static int lstm_train_gradient_dev(LSTM filter, const float *d_dout, float *d_grad, float *d_dX) {
    rec_core *c = &filter->core;
    rec_train *t = &filter->train;
    int acts[5];
    float sc[5];
    if (lstm_acts(filter, acts, sc)) return -1;
    const int B = t->mini_batch, T = c->T, in = c->in, H = c->H;
    const size_t w = (size_t)in * 4 * H, u = (size_t)H * 4 * H, b4 = 4 * (size_t)H, rows = (size_t)B * T;
    float *d_dG = nntk_devbuf_reserve(&t->d_dxW, rows * 4 * H);
    float *d_work = nntk_devbuf_reserve(&t->d_work, (size_t)B * 6 * H);
    float *d_UT = nntk_devbuf_reserve(&t->d_scr, u);
    if (!d_dG || !d_work || !d_UT) return -1;
    const float *dW = t->d_raw.p, *dU = dW + w;
    if (nntk_shim_transpose(dU, d_UT, H, 4 * H, 0)) return -1;
    if (nntk_shim_lstm_train_backward(d_dout, d_UT, t->d_hU.p, t->d_Zg.p, d_dG, d_work, B, T, H, c->return_sequences ? 1 : 0, acts, sc)) return -1;
    /* d_W += x^T dgates, d_U += h_prev^T dgates, d_b_i += colsum, d_b_h += colsum (lstm.c:412-415), d_X = dgates W^T */
    if (nntk_train_outer_accumulate(t->d_x_cur, d_dG, d_grad, d_grad + w + u, (long)rows, in, 4 * H, 0)) return -1;
    if (nntk_train_outer_accumulate(t->d_h.p, d_dG, d_grad + w, d_grad + w + u + b4, (long)rows, H, 4 * H, T)) return -1;
    return nntk_train_rows_times_rowmat(d_dG, dW, d_dX, (long)rows, in, 4 * H);
}
void LSTMCalculateGradient(LSTM filter, LSTMGradient *gradient, float *d_out) {
    nntk_shim_clear_error();
    if (!filter || !gradient || !d_out) { nntk_set_error("LSTMCalculateGradient: NULL argument"); return; }
    rec_core *c = &filter->core;
    rec_train *t = &filter->train;
    if (!t->on || !t->have_batch) { nntk_set_error("LSTMCalculateGradient: run LSTMApplyTrainingBatch on a training handle first"); return; }
    const int B = t->mini_batch, T = c->T, in = c->in, H = c->H;
    const size_t w = (size_t)in * 4 * H, u = (size_t)H * 4 * H, b4 = 4 * (size_t)H, rows = (size_t)B * T;
    const size_t n_do = c->return_sequences ? rows * H : (size_t)B * H;
    float *d_dout = nntk_devbuf_reserve(&t->d_dout, n_do);
    float *d_grad = nntk_devbuf_reserve(&t->d_grad, w + u + 2 * b4);
    float *d_dX = nntk_devbuf_reserve(&t->d_dX, rows * in);
    if (!d_dout || !d_grad || !d_dX) return;
    if (nntk_shim_upload(d_dout, d_out, n_do * sizeof(float))) return;
    if (nntk_shim_upload(d_grad, gradient->d_W, (w + u + 2 * b4) * sizeof(float))) return;
    if (lstm_train_gradient_dev(filter, d_dout, d_grad, d_dX)) return;
    if (nntk_shim_download(gradient->d_W, d_grad, (w + u + 2 * b4) * sizeof(float))) return;
    nntk_shim_download(gradient->d_X, d_dX, rows * in * sizeof(float));
}
/* device-pointer form: d_grad = W [in][4H] | U [H][4H] | b_i [4H] | b_h [4H] is ADDED to, d_dX [B][T][in] overwritten */
int LSTMCalculateGradientDevice(LSTM filter, float *d_grad, float *d_dX, const float *d_dout) {
    nntk_shim_clear_error();
    if (!filter || !d_grad || !d_dX || !d_dout) NNTK_FAIL("LSTMCalculateGradientDevice: NULL argument");
    if (!filter->train.on || !filter->train.have_batch) NNTK_FAIL("LSTMCalculateGradientDevice: run LSTMApplyTrainingBatch[Device] on a training handle first");
    return lstm_train_gradient_dev(filter, d_dout, d_grad, d_dX);
}

int LSTMApplyInference(LSTM filter, const float *input, float *output) {
    nntk_shim_clear_error();
    int acts[5];
    float sc[5];
    if (!filter) NNTK_FAIL("LSTMApplyInference: NULL handle");
    if (filter->train.on) NNTK_FAIL("LSTMApplyInference: the handle was created for training");         /* lstm.c:242-244 */
    if (lstm_acts(filter, acts, sc)) return -1;
    return core_apply_host(&filter->core, 1, filter->config.v2, acts, sc, input, output, 1, 1);
}
/* lstm.c:426-475 forward semantics */
int LSTMApplyInferenceBatch(LSTM filter, const float *input, float *output, int batch) {
    nntk_shim_clear_error();
    int acts[5];
    float sc[5];
    if (!filter) NNTK_FAIL("LSTMApplyInferenceBatch: NULL handle");
    if (lstm_acts(filter, acts, sc)) return -1;
    return core_apply_host(&filter->core, 1, filter->config.v2, acts, sc, input, output, batch, 0);
}
int LSTMApplyDevice(LSTM filter, const float *d_input, float *d_output, int batch) {
    nntk_shim_clear_error();
    int acts[5];
    float sc[5];
    if (!filter) NNTK_FAIL("LSTMApplyDevice: NULL handle");
    if (lstm_acts(filter, acts, sc)) return -1;
    if (core_ensure(&filter->core, 0)) return -1;
    return core_apply_device(&filter->core, 1, filter->config.v2, acts, sc, d_input, d_output, batch, 0);
}
int LSTMApplyDeviceFrag3(LSTM filter, const float *d_input, const float *d_input_frag3, float *d_output, float *d_output_frag3, int batch) {
    nntk_shim_clear_error();
    int acts[5];
    float sc[5];
    if (!filter) NNTK_FAIL("LSTMApplyDeviceFrag3: NULL handle");
    if (lstm_acts(filter, acts, sc)) return -1;
    if (core_ensure(&filter->core, 0)) return -1;
    return core_apply_device_f3(&filter->core, 1, filter->config.v2, acts, sc, d_input, d_input_frag3, d_output, d_output_frag3, batch);
}
/* The sequence output as a FRAG2H tensor (frag3.hip: two f16 images of h * 2^15 -- the dense GEMM's three-product operand form).  The form
 * holds magnitudes below 2, so it exists for the standard activations only (|h| = |o tanh(c)| < 1).  The split-K register-resident kernels
 * write it themselves (the wave that would write the f32 rows); every other kernel goes through an f32 scratch tensor and the pack pass:
 * the same bits, because the form is a function of the f32 value.  0 = done; 1 = not available for this layer; -1 = error. */
static int lstm_apply_device_h2(LSTM filter, const float *d_in, const float *d_in_f3, float *d_out_h2, int B) {
    rec_core *c = &filter->core;
    int acts[5];
    float sc[5];
    if (lstm_acts(filter, acts, sc)) return -1;
    if (!lstm_std_acts(acts) || !c->return_sequences) return 1;
    if (B <= 0 || c->T <= 0) return 0;
    if (core_ensure(c, 0)) return -1;
    /* H > 256: the HF instantiation of the register-resident kernel -- its recurrence itself runs on two f16 images of h (three products per
     * k step), and its hand-off buffer IS the FRAG2H output.  Another contraction than the bf16 x 3 kernels' (same tolerance): the choice
     * depends on the layer only (shape, weights, option rec_hf), never on the call. */
    {
        int on = -1;
        (void)nntk_shim_get_option("rec_rr", &on);
        if (on != 0 && !c->rr_exact_only && nntk_shim_lstm_rr_hf_ok(c->H, c->in)) {
            if (!c->rr_hf_valid) {
                const size_t nW = (size_t)c->in * 4 * c->H, nU = (size_t)c->H * 4 * c->H;
                float us = nntk_f16_scale(c->weights->U, nU), mw = 0.f;
                for (size_t i = 0; i < nW; ++i) { float a = fabsf(c->weights->W[i]); if (a > mw) mw = a; }
                /* W's bf16 images are packed times 2^15 us: they must stay inside bf16's range (and the sums inside f32's) */
                c->hf_wscale = (us > 0.f && us < 1e15f && us > 1e-15f && mw * 32768.f * us < 1e30f) ? 32768.f * us : 0.f;
                if (c->hf_wscale > 0.f) {
                    size_t img = nntk_shim_lstm_rr_hf_image_floats(c->H, c->in);
                    if (!c->d_rr_hf && !(c->d_rr_hf = (float *)nntk_shim_malloc(img * sizeof(float)))) return -1;
                    if (nntk_shim_lstm_rr_pack_hf(c->d_ut, c->d_wp, c->d_rr_hf, c->H, c->in, us, c->hf_wscale)) return -1;
                }
                c->rr_hf_valid = 1;
            }
            if (c->hf_wscale > 0.f) {
                const float *xf3 = d_in_f3;
                if (!xf3) {
                    float *buf = nntk_devbuf_reserve(&c->d_xf3, nntk_shim_frag3_floats(B, c->T, c->in));
                    if (!buf || nntk_shim_frag3_pack(d_in, buf, B, c->T, c->in)) return -1;
                    xf3 = buf;
                }
                float *d_work = nntk_devbuf_reserve(&c->d_work_rr, nntk_shim_lstm_rr_work_floats(B, c->H));
                if (!d_work) return -1;
                int rc = nntk_shim_lstm_rr_hf(xf3, c->d_rr_hf, c->d_bi, filter->config.v2 ? c->d_bh : NULL, d_out_h2, d_work, B, c->T, c->in, c->H,
                                              1.f / c->hf_wscale);
                if (rc <= 0) return rc;
            }
        }
    }
    if (!nntk_shim_fk_image_floats(c->H, c->in)) {       /* (the full-K family has no FRAG2H store; its shapes take the f32 way below) */
        const rr_io io = { d_in, d_in_f3, NULL, NULL, d_out_h2 };
        int rc = core_try_lstm_rr(c, filter->config.v2, acts, &io, B, 0);
        if (rc <= 0) return rc;
    }
    float *o = nntk_devbuf_reserve(&c->d_out, (size_t)B * c->T * c->H);
    if (!o) return -1;
    if (core_apply_device_f3(c, 1, filter->config.v2, acts, sc, d_in, d_in_f3, o, NULL, B)) return -1;
    return nntk_shim_frag2h_pack(o, d_out_h2, B, c->T, c->H);
}
int nntk_lstm_apply_device_h2(LSTM filter, const float *d_in, const float *d_in_f3, float *d_out_h2, int B) {
    return lstm_apply_device_h2(filter, d_in, d_in_f3, d_out_h2, B);
}
float *nntk_lstm_frag2h_scratch(LSTM f, int batch) {
    return nntk_devbuf_reserve(&f->core.d_h2, nntk_shim_frag2h_floats(batch, f->core.T, f->core.H));
}
int LSTMApplyDeviceFrag2h(LSTM filter, const float *d_input, const float *d_input_frag3, float *d_output_frag2h, int batch) {
    nntk_shim_clear_error();
    if (!filter || (!d_input && !d_input_frag3) || !d_output_frag2h) NNTK_FAIL("LSTMApplyDeviceFrag2h: NULL argument");
    int rc = lstm_apply_device_h2(filter, d_input, d_input_frag3, d_output_frag2h, batch);
    if (rc == 1) NNTK_FAIL("LSTMApplyDeviceFrag2h: the frag2h form holds |h| < 2 -- standard activations and return_sequences only");
    return rc;
}
/* accessors for the fused LSTM -> TimeDistributedDense call (dense.c) */
void nntk_lstm_dims(LSTM f, int *T, int *in, int *H, int *return_sequences) {
    *T = f->core.T; *in = f->core.in; *H = f->core.H; *return_sequences = f->core.return_sequences ? 1 : 0;
}
float *nntk_lstm_frag3_scratch(LSTM f, int batch) {
    /* the register-resident kernels' hand-off size counts H / 16 k steps (they take H % 16 == 0 only); a layer they do not take writes a
     * frag3 tensor of ceil(H / 16) k steps into this buffer through the pack pass: the larger of the two (found by the FRAG2H soak: for
     * H % 16 != 0 the fused LSTM -> dense call wrote one k step per row block past the end of the smaller one) */
    size_t a = nntk_shim_rr_hseq_floats(batch, f->core.T, f->core.H), b = nntk_shim_frag3_floats(batch, f->core.T, f->core.H);
    return nntk_devbuf_reserve(&f->core.d_hseq, a > b ? a : b);
}
const char *LSTMKernelPlan(LSTM filter) {
    static _Thread_local char buf[320];
    if (!filter) return "";
    int a[5]; float sc[5];
    if (lstm_acts(filter, a, sc)) return "lstm: invalid gate activations";
    return core_plan(&filter->core, lstm_std_acts(a), buf, sizeof buf);
}
/* lstm.c:270-274 (lstm_zero_state) */
int LSTMResetState(LSTM filter) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("LSTMResetState: NULL handle");
    return core_reset_state(&filter->core);
}
int LSTMGetState(LSTM filter, float *h_host, float *c_host) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("LSTMGetState: NULL handle");
    const rec_core *c = &filter->core;
    if (h_host && nntk_shim_download(h_host, c->d_h[c->cur], (size_t)c->H * sizeof(float))) return -1;
    if (c_host && nntk_shim_download(c_host, c->d_c[c->cur], (size_t)c->H * sizeof(float))) return -1;
    return 0;
}

/* ================================= RNN ==================================== */
/* layers/rnn.c forward.  One gate; the activation is the layer's single
 * ActivationFunction (rnn.h:20-24), which must be one of the built-ins to run inside the step kernel. */

struct RNNStruct {
    RNNConfig config;
    rec_core core;
    rec_train train;            /* d_Zg = gate [B][T][H] */
};

/* rnn.c:48-61 */
RNNConfig RNNConfigCreate(int input_feature_channels, int output_feature_channels, bool return_sequences,
                          int timesteps, bool v2, ActivationFunction activation) {
    RNNConfig c;
    memset(&c, 0, sizeof(c));
    c.base = RecurrentConfigCreate(input_feature_channels, output_feature_channels, return_sequences, timesteps);
    c.v2 = v2;
    c.activation = activation;
    return c;
}

RNN RNNCreateForInference(RNNConfig config) {
    nntk_shim_clear_error();
    RNN f = (RNN)calloc(1, sizeof(struct RNNStruct));
    if (!f) return NULL;
    f->config = config;
    if (core_init(&f->core, 1, config.base)) { free(f); return NULL; }
    return f;
}
RNNWeights *RNNGetWeights(RNN filter) { return filter->core.weights; }
void RNNDestroy(RNN filter) {
    if (!filter) return;
    core_free(&filter->core);   /* the activation stays with the caller, as in GRU/LSTM */
    train_free(&filter->train);
    free(filter);
}

int RNNBroadcastWeights(RNN filter, int root) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("RNNBroadcastWeights: NULL handle");
    return core_broadcast(&filter->core, root);
}
int RNNSyncWeights(RNN filter) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("RNNSyncWeights: NULL handle");
    nntk_shim_synchronize();
    return core_upload(&filter->core);
}

/* rnn.c:228-247 (intended semantics, see the header) */
/* ---- training: rnn.c:249-291 (forward keeping gate, h), :184-221 + :293-351 (BPTT) ---- */
RNN RNNCreateForTraining(RNNConfig config, RNNTrainingConfig training_config) {
    RNN f = RNNCreateForInference(config);
    if (!f) return NULL;
    f->train.on = 1;
    f->train.mini_batch = training_config.mini_batch_size;
    return f;
}
RNNGradient *RNNGradientCreate(RNNConfig config, RNNTrainingConfig training_config) {
    RNNGradient *g = (RNNGradient *)malloc(sizeof(RNNGradient));
    if (!g) return NULL;
    size_t in = (size_t)config.base.input_feature_channels, H = (size_t)config.base.output_feature_channels;
    size_t x = (size_t)training_config.mini_batch_size * in * config.base.timesteps;
    g->d_W = (float *)calloc(in * H + H * H + 2 * H + x + 1, sizeof(float));
    if (!g->d_W) { free(g); return NULL; }
    g->d_U = g->d_W + in * H;
    g->d_b_i = g->d_U + H * H;
    g->d_b_h = g->d_b_i + H;
    g->d_X = g->d_b_h + H;
    return g;
}
int RNNApplyTrainingBatch(RNN filter, const float *input, float *output) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("RNNApplyTrainingBatch: NULL handle");
    if (!filter->train.on) NNTK_FAIL("RNNApplyTrainingBatch: the handle was created for inference");      /* rnn.c:250-252 */
    int act;
    float sc;
    if (gate_kind(filter->config.activation, &act, &sc)) return -1;
    rec_core *c = &filter->core;
    rec_train *t = &filter->train;
    const int B = t->mini_batch, T = c->T, in = c->in, H = c->H;
    if (B <= 0 || T <= 0) return 0;
    const size_t nw = (size_t)in * H + (size_t)H * H + 2 * (size_t)H;
    float *d_x = nntk_devbuf_reserve(&t->d_x, (size_t)B * T * in);
    float *d_h = nntk_devbuf_reserve(&t->d_h, (size_t)B * T * H);
    float *d_g = nntk_devbuf_reserve(&t->d_Zg, (size_t)B * T * H);
    float *d_raw = nntk_devbuf_reserve(&t->d_raw, nw);
    if (!d_x || !d_h || !d_g || !d_raw) return -1;
    if (nntk_shim_upload(d_x, input, (size_t)B * T * in * sizeof(float))) return -1;
    if (nntk_shim_upload(d_raw, c->wb.host, nw * sizeof(float))) return -1;
    const float *dW = d_raw, *dU = dW + (size_t)in * H, *dbi = dU + (size_t)H * H, *dbh = dbi + H;
    if (nntk_shim_rnn_train_forward(d_x, dW, dU, dbi, dbh, d_h, d_g, B, T, in, H, filter->config.v2 ? 1 : 0, act, sc)) return -1;
    t->have_batch = 1;
    if (c->return_sequences) return nntk_shim_download(output, d_h, (size_t)B * T * H * sizeof(float));
    /* the last step of every sequence (gru.c:286-291, lstm.c:466-471, rnn.c:283-288): one strided copy */
    return nntk_shim_download_rows(output, d_h + (size_t)(T - 1) * H, (size_t)T * H * sizeof(float), (size_t)H * sizeof(float), (size_t)B);
}
void RNNCalculateGradient(RNN filter, RNNGradient *gradient, float *d_out) {
    nntk_shim_clear_error();
    if (!filter || !gradient || !d_out) { nntk_set_error("RNNCalculateGradient: NULL argument"); return; }
    rec_core *c = &filter->core;
    rec_train *t = &filter->train;
    if (!t->on || !t->have_batch) { nntk_set_error("RNNCalculateGradient: run RNNApplyTrainingBatch on a training handle first"); return; }
    int act;
    float sc;
    if (gate_kind(filter->config.activation, &act, &sc)) return;
    const int B = t->mini_batch, T = c->T, in = c->in, H = c->H;
    const size_t w = (size_t)in * H, u = (size_t)H * H, rows = (size_t)B * T;
    const size_t n_do = c->return_sequences ? rows * H : (size_t)B * H;
    float *d_dout = nntk_devbuf_reserve(&t->d_dout, n_do);
    float *d_dG = nntk_devbuf_reserve(&t->d_dxW, rows * H);
    float *d_work = nntk_devbuf_reserve(&t->d_work, (size_t)B * 2 * H);
    float *d_grad = nntk_devbuf_reserve(&t->d_grad, w + u + 2 * (size_t)H);
    float *d_UT = nntk_devbuf_reserve(&t->d_scr, u);
    float *d_dX = nntk_devbuf_reserve(&t->d_dX, rows * in);
    if (!d_dout || !d_dG || !d_work || !d_grad || !d_UT || !d_dX) return;
    const float *dW = t->d_raw.p, *dU = dW + w;
    if (nntk_shim_upload(d_dout, d_out, n_do * sizeof(float))) return;
    if (nntk_shim_upload(d_grad, gradient->d_W, (w + u + 2 * (size_t)H) * sizeof(float))) return;
    if (nntk_shim_transpose(dU, d_UT, H, H, 0)) return;
    if (nntk_shim_rnn_train_backward(d_dout, d_UT, t->d_h.p, t->d_Zg.p, d_dG, d_work, B, T, H, c->return_sequences ? 1 : 0, act)) return;
    if (nntk_train_outer_accumulate(t->d_x.p, d_dG, d_grad, d_grad + w + u, (long)rows, in, H, 0)) return;
    if (nntk_train_outer_accumulate(t->d_h.p, d_dG, d_grad + w, d_grad + w + u + H, (long)rows, H, H, T)) return;
    if (nntk_train_rows_times_rowmat(d_dG, dW, d_dX, (long)rows, in, H)) return;
    if (nntk_shim_download(gradient->d_W, d_grad, (w + u + 2 * (size_t)H) * sizeof(float))) return;
    nntk_shim_download(gradient->d_X, d_dX, rows * in * sizeof(float));
}

int RNNApplyInference(RNN filter, const float *input, float *output) {
    nntk_shim_clear_error();
    int act;
    float sc;
    if (!filter) NNTK_FAIL("RNNApplyInference: NULL handle");
    if (filter->train.on) NNTK_FAIL("RNNApplyInference: the handle was created for training");           /* rnn.c:223-225 */
    if (gate_kind(filter->config.activation, &act, &sc)) return -1;
    return core_apply_host(&filter->core, 0, filter->config.v2, &act, &sc, input, output, 1, 1);
}
/* rnn.c:249-291 forward semantics */
int RNNApplyInferenceBatch(RNN filter, const float *input, float *output, int batch) {
    nntk_shim_clear_error();
    int act;
    float sc;
    if (!filter) NNTK_FAIL("RNNApplyInferenceBatch: NULL handle");
    if (gate_kind(filter->config.activation, &act, &sc)) return -1;
    return core_apply_host(&filter->core, 0, filter->config.v2, &act, &sc, input, output, batch, 0);
}
int RNNApplyDevice(RNN filter, const float *d_input, float *d_output, int batch) {
    nntk_shim_clear_error();
    int act;
    float sc;
    if (!filter) NNTK_FAIL("RNNApplyDevice: NULL handle");
    if (gate_kind(filter->config.activation, &act, &sc)) return -1;
    if (core_ensure(&filter->core, 0)) return -1;
    return core_apply_device(&filter->core, 0, filter->config.v2, &act, &sc, d_input, d_output, batch, 0);
}
int RNNResetState(RNN filter) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("RNNResetState: NULL handle");
    return core_reset_state(&filter->core);
}
int RNNGetState(RNN filter, float *h_host) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("RNNGetState: NULL handle");
    return nntk_shim_download(h_host, filter->core.d_h[filter->core.cur], (size_t)filter->core.H * sizeof(float));
}

/* ============================ bidirectional helpers ======================== */
/* layers/bidirectional.c forward helpers.  The device forms are the product; the host-pointer forms keep the
 * reference's signatures (void, caller-owned host buffers) and stage through device scratch. */

/* scratch of the host-pointer helpers: per calling thread (they have no handle to own it) */
#define g_bd_a (*nntk_thread_scratch(NNTK_TS_BD_A))
#define g_bd_b (*nntk_thread_scratch(NNTK_TS_BD_B))
#define g_bd_out (*nntk_thread_scratch(NNTK_TS_BD_OUT))

int bd_reverse_input_batch_device(const float *d_input, float *d_output, RecurrentConfig config, int batch) {
    nntk_shim_clear_error();
    return nntk_shim_reverse_time(d_input, d_output, batch, config.timesteps, config.input_feature_channels);
}
int bd_reverse_backward_batch_device(const float *d_input, float *d_output, RecurrentConfig config, int batch) {
    nntk_shim_clear_error();
    return nntk_shim_reverse_time(d_input, d_output, batch, config.timesteps, config.output_feature_channels);
}
int bd_merge_concat_device(const float *d_forward, const float *d_backward, float *d_output, RecurrentConfig config, int batch) {
    nntk_shim_clear_error();
    long rows = (long)batch * (config.return_sequences ? config.timesteps : 1);
    return nntk_shim_concat2(d_forward, d_backward, d_output, rows, config.output_feature_channels);
}
int bd_merge_sum_device(const float *d_forward, const float *d_backward, float *d_output, RecurrentConfig config, int batch) {
    nntk_shim_clear_error();
    long rows = (long)batch * (config.return_sequences ? config.timesteps : 1);
    return nntk_shim_add2(d_forward, d_backward, d_output, rows * config.output_feature_channels);
}

static void bd_reverse_host(const float *input, float *output, int batch, int T, int F) {
    nntk_shim_clear_error();
    size_t n = (size_t)batch * T * F;
    if (!n) return;
    float *d_in = nntk_devbuf_reserve(&g_bd_a, n), *d_out = nntk_devbuf_reserve(&g_bd_out, n);
    if (!d_in || !d_out) return;
    if (nntk_shim_upload(d_in, input, n * sizeof(float))) return;
    if (nntk_shim_reverse_time(d_in, d_out, batch, T, F)) return;
    (void)nntk_shim_download(output, d_out, n * sizeof(float));
}
/* bidirectional.c:25-35 */
void bd_reverse_input_batch(const float *input, float *output, RecurrentConfig config, int batch) {
    bd_reverse_host(input, output, batch, config.timesteps, config.input_feature_channels);
}
void bd_reverse_backward_batch(const float *input, float *output, RecurrentConfig config, int batch) {
    bd_reverse_host(input, output, batch, config.timesteps, config.output_feature_channels);
}
/* bidirectional.c:37-40 */
int bd_merge_concat_buffer_size(RecurrentConfig config) {
    int rows = config.return_sequences ? config.timesteps : 1;
    return 2 * rows * config.output_feature_channels;
}
static void bd_merge_host(const float *fwd, const float *bwd, float *output, RecurrentConfig config, int batch, int concat) {
    nntk_shim_clear_error();
    size_t n = (size_t)batch * (config.return_sequences ? config.timesteps : 1) * config.output_feature_channels;
    if (!n) return;
    float *d_a = nntk_devbuf_reserve(&g_bd_a, n), *d_b = nntk_devbuf_reserve(&g_bd_b, n);
    float *d_out = nntk_devbuf_reserve(&g_bd_out, concat ? 2 * n : n);
    if (!d_a || !d_b || !d_out) return;
    if (nntk_shim_upload(d_a, fwd, n * sizeof(float)) || nntk_shim_upload(d_b, bwd, n * sizeof(float))) return;
    int rc = concat ? bd_merge_concat_device(d_a, d_b, d_out, config, batch) : bd_merge_sum_device(d_a, d_b, d_out, config, batch);
    if (rc) return;
    (void)nntk_shim_download(output, d_out, (concat ? 2 * n : n) * sizeof(float));
}
/* bidirectional.c:42-58: per sequence, [rows, out] | [rows, out] -> [rows, 2*out] (the reference does it with three transposes) */
void bd_merge_concat(const float *forward_result, const float *backward_result, float *output,
                     RecurrentConfig config, int batch, float *buffer) {
    (void)buffer;
    bd_merge_host(forward_result, backward_result, output, config, batch, 1);
}
/* ---- gradient helpers (bidirectional.c:58-74, :87-108): pure data movement, as their forward counterparts ---- */
void bd_merge_concat_gradient(const float *d_out, float *d_forward_out, float *d_backward_out, RecurrentConfig config,
                              int batch, float *buffer) {
    (void)buffer;
    nntk_shim_clear_error();
    int rows = config.return_sequences ? config.timesteps : 1, C = config.output_feature_channels;
    size_t n = (size_t)batch * rows * C;
    if (!n) return;
    float *d_in = nntk_devbuf_reserve(&g_bd_out, 2 * n);
    float *d_a = nntk_devbuf_reserve(&g_bd_a, n), *d_b = nntk_devbuf_reserve(&g_bd_b, n);
    if (!d_in || !d_a || !d_b) return;
    if (nntk_shim_upload(d_in, d_out, 2 * n * sizeof(float))) return;
    if (nntk_shim_split2(d_in, d_a, d_b, (long)batch * rows, C)) return;
    if (nntk_shim_download(d_forward_out, d_a, n * sizeof(float))) return;
    (void)nntk_shim_download(d_backward_out, d_b, n * sizeof(float));
}
void bd_merge_sum_gradient(const float *d_out, float *d_forward_out, float *d_backward_out, RecurrentConfig config, int batch) {
    size_t n = (size_t)batch * (config.return_sequences ? config.timesteps : 1) * config.output_feature_channels;
    memcpy(d_forward_out, d_out, n * sizeof(float));        /* f_copy twice (bidirectional.c:95-96): no arithmetic */
    memcpy(d_backward_out, d_out, n * sizeof(float));
}
/* output = forward_dx + time-reversed backward_dx (bidirectional.c:99-108) */
void bd_accumulate_d_x(const float *forward_dx, const float *backward_dx, float *output, RecurrentConfig config, int batch) {
    nntk_shim_clear_error();
    size_t n = (size_t)batch * config.timesteps * config.input_feature_channels;
    if (!n) return;
    float *d_a = nntk_devbuf_reserve(&g_bd_a, n), *d_b = nntk_devbuf_reserve(&g_bd_b, n), *d_o = nntk_devbuf_reserve(&g_bd_out, n);
    if (!d_a || !d_b || !d_o) return;
    if (nntk_shim_upload(d_a, forward_dx, n * sizeof(float)) || nntk_shim_upload(d_b, backward_dx, n * sizeof(float))) return;
    if (nntk_shim_reverse_time(d_b, d_o, batch, config.timesteps, config.input_feature_channels)) return;
    if (nntk_shim_add2(d_a, d_o, d_o, (long)n)) return;
    (void)nntk_shim_download(output, d_o, n * sizeof(float));
}
/* bidirectional.c:76-85 */
void bd_merge_sum(const float *forward_result, const float *backward_result, float *output,
                  RecurrentConfig config, int batch) {
    bd_merge_host(forward_result, backward_result, output, config, batch, 0);
}
