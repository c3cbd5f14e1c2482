// recurrent.hip -- K4: GRU / LSTM time loop.  The input projection x_t*W + b_i for
// ALL timesteps is hoisted out of the loop into one MFMA GEMM (conv1d.hip, k = 1,
// written time-major).  What remains per step is the strictly sequential part:
//     hU = h_{t-1} [B,H] x U [H, G*H]      (exact-f32 MFMA 16x16x4)
// with the whole gate fusion (bias, sigmoid/tanh, state update) done in registers
// in the MFMA epilogue -- each lane ends up holding all G gate pre-activations of
// the same (batch row, hidden unit), so no cross-lane traffic is needed.
//
// Gate semantics:
//   GRU  (layers/gru.c:129-187): blocks [z | r | h], reset-after form, two biases:
//        z = s(xW_z + hU_z + b_hz); r = s(xW_r + hU_r + b_hr);
//        h~ = tanh(r * (hU_h + b_hh) + xW_h);  h' = (1 - z) * h~ + z * h
//   LSTM (layers/lstm.c:185-239): blocks [i | f | g | o]; b_h only when v2:
//        Z = xW + hU (+ b_h); c' = s(Z_f)*c + s(Z_i)*tanh(Z_g); h' = s(Z_o)*tanh(c')
//
// Work decomposition per step: workgroup = 64 batch rows x 16 hidden units x G
// gates (4 wavefronts, one 16-row slab each; all four share the U^T tile through
// LDS).  K = H is walked in chunks of 32 through double-buffered LDS.  State lives
// in two ping-pong [B,H] buffers (compact rows, L2 resident); c is updated in
// place.  One launch per timestep: the kernel boundary is the grid-wide
// dependency between steps (every hidden unit of step t needs all of h_{t-1}).
#include "nntk_common.hpp"

#define REC_KC 32
#define REC_LS 36      // LDS row stride in floats (16-B aligned rows, skewed banks)
#define REC_BM 64
#define REC_HN 16

struct RecParams {
    const float *xw;      // [T, B, G*H] (this step's slab is xw + t*B*G*H)
    const float *ut;      // [G, Hj_p, Hk_p]  U^T, zero padded
    const float *bh;      // [G*H] or NULL
    const float *h_prev;  // [B, H]
    float *h_next;        // [B, H]
    float *c;             // [B, H] (LSTM, in place)
    float *out;           // this step's output row base or NULL; element (b, j) at out[b*out_ld + j]
    long out_ld;
    int B, H, Hj_p, Hk_p;
    int a0, a1, a2, a3, a4;   // activation kinds
};

template <int G, bool IS_LSTM>
__global__ __launch_bounds__(256) void rec_step_kernel(RecParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][REC_BM * REC_LS];
    __shared__ __attribute__((aligned(16))) float Bs[2][G * REC_HN * REC_LS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int b0 = blockIdx.x * REC_BM;
    const int j0 = blockIdx.y * REC_HN;
    const bool vec4 = (p.H & 3) == 0;

    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: A tile 64 rows x 8 float4, B tile 16G rows x 8 float4
    constexpr int A_F4 = REC_BM * (REC_KC / 4);         // 512
    constexpr int B_F4 = G * REC_HN * (REC_KC / 4);     // 384 or 512
    constexpr int A_PT = A_F4 / 256;                    // 2
    constexpr int B_PT = (B_F4 + 255) / 256;            // 2
    float4 areg[A_PT], breg[B_PT];

    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int s = 0; s < A_PT; ++s) {
            int e = tid + s * 256;
            int r = e >> 3, c4 = e & 7;
            int b = b0 + r, k = k0 + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < p.B) {
                const float *src = p.h_prev + (size_t)b * p.H + k;
                if (vec4) {
                    if (k < p.H) v = *reinterpret_cast<const float4 *>(src);
                } else {
                    if (k + 0 < p.H) v.x = src[0];
                    if (k + 1 < p.H) v.y = src[1];
                    if (k + 2 < p.H) v.z = src[2];
                    if (k + 3 < p.H) v.w = src[3];
                }
            }
            areg[s] = v;
        }
#pragma unroll
        for (int s = 0; s < B_PT; ++s) {
            int e = tid + s * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < B_F4) {
                int r = e >> 3, c4 = e & 7;          // r = g*16 + jj
                int g = r >> 4, jj = r & 15;
                v = *reinterpret_cast<const float4 *>(p.ut + ((size_t)g * p.Hj_p + j0 + jj) * p.Hk_p + k0 + c4 * 4);
            }
            breg[s] = v;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int s = 0; s < A_PT; ++s) {
            int e = tid + s * 256;
            int r = e >> 3, c4 = e & 7;
            *reinterpret_cast<float4 *>(&As[buf][r * REC_LS + c4 * 4]) = areg[s];
        }
#pragma unroll
        for (int s = 0; s < B_PT; ++s) {
            int e = tid + s * 256;
            if (e < B_F4) {
                int r = e >> 3, c4 = e & 7;
                *reinterpret_cast<float4 *>(&Bs[buf][r * REC_LS + c4 * 4]) = breg[s];
            }
        }
    };

    const int nchunks = p.Hk_p / REC_KC;
    load_chunk(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        store_chunk(buf);
        __syncthreads();
        if (ch + 1 < nchunks) load_chunk((ch + 1) * REC_KC);
        // lane group q owns k = q*8 + s (s = 0..7) of this chunk: any bijection of
        // k onto (group, step) is a valid MFMA K order as long as A and B agree.
        const float *arow = &As[buf][(wave * 16 + l15) * REC_LS + q * 8];
        const float4 a_lo = *reinterpret_cast<const float4 *>(arow);
        const float4 a_hi = *reinterpret_cast<const float4 *>(arow + 4);
        const float av[8] = {a_lo.x, a_lo.y, a_lo.z, a_lo.w, a_hi.x, a_hi.y, a_hi.z, a_hi.w};
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float *brow = &Bs[buf][(g * 16 + l15) * REC_LS + q * 8];
            const float4 b_lo = *reinterpret_cast<const float4 *>(brow);
            const float4 b_hi = *reinterpret_cast<const float4 *>(brow + 4);
            const float bv[8] = {b_lo.x, b_lo.y, b_lo.z, b_lo.w, b_hi.x, b_hi.y, b_hi.z, b_hi.w};
#pragma unroll
            for (int s = 0; s < 8; ++s)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc[g], 0, 0, 0);
        }
    }

    // ---- fused gate epilogue: lane holds (b = b0 + 16*wave + 4*q + r, j = j0 + l15) ----
    const int j = j0 + l15;
    if (j >= p.H) return;
    const int GH = G * p.H;
    float bh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) bh[g] = p.bh ? p.bh[g * p.H + j] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + wave * 16 + q * 4 + r;
        if (b >= p.B) continue;
        const float *xw = p.xw + (size_t)b * GH + j;
        float hn;
        if (!IS_LSTM) {
            // gru.c:144-186
            const float hz = acc[0][r] + bh[0], hr = acc[1][r] + bh[1], hh = acc[2][r] + bh[2];
            const float z = nntk_act(p.a0, xw[0] + hz, 1.0f);
            const float rg = nntk_act(p.a2, xw[p.H] + hr, 1.0f);
            const float ht = nntk_act(p.a1, rg * hh + xw[2 * p.H], 1.0f);
            const float hp = p.h_prev[(size_t)b * p.H + j];
            hn = (-z + 1.0f) * ht + z * hp;
        } else {
            // lstm.c:201-238
            const float zi = xw[0] + (acc[0][r] + bh[0]);
            const float zf = xw[p.H] + (acc[1][r] + bh[1]);
            const float zg = xw[2 * p.H] + (acc[2][r] + bh[2]);
            const float zo = xw[3 * p.H] + (acc[G - 1][r] + bh[G - 1]);
            const float ig = nntk_act(p.a0, zi, 1.0f);
            const float fg = nntk_act(p.a1, zf, 1.0f);
            const float gg = nntk_act(p.a2, zg, 1.0f);
            const float og = nntk_act(p.a3, zo, 1.0f);
            const size_t ci = (size_t)b * p.H + j;
            const float cn = fg * p.c[ci] + ig * gg;
            p.c[ci] = cn;
            hn = og * nntk_act(p.a4, cn, 1.0f);
        }
        p.h_next[(size_t)b * p.H + j] = hn;
        if (p.out) p.out[(size_t)b * p.out_ld + j] = hn;
    }
}

extern "C" size_t nntk_shim_recurrent_work_floats(int B, int H) {
    return (size_t)3 * B * H;     // h ping, h pong, c
}

static int act_ok(int a) {
    return a == NNTK_ACT_IDENTITY || a == NNTK_ACT_SIGMOID || a == NNTK_ACT_TANH || a == NNTK_ACT_RELU;
}

template <int G, bool IS_LSTM>
static int run_recurrent(const float *d_xw, const float *d_ut, const float *d_bh, const float *d_h0,
                         const float *d_c0, float *d_out, float *d_hT, float *d_cT, float *d_work,
                         int B, int T, int H, int return_sequences, const int *acts, int nacts) {
    for (int i = 0; i < nacts; ++i)
        if (!act_ok(acts[i]))
            return nntk_fail_msg("recurrent: gate activation must be one of the built-in identity/sigmoid/tanh/relu");
    const size_t BH = (size_t)B * H;
    float *hbuf[2] = {d_work, d_work + BH};
    float *cbuf = d_work + 2 * BH;
    if (d_h0) { if (nntk_shim_copy_d2d(hbuf[0], d_h0, BH * 4)) return -1; }
    else      { if (nntk_shim_memset(hbuf[0], 0, BH * 4)) return -1; }
    if (IS_LSTM) {
        if (d_c0) { if (nntk_shim_copy_d2d(cbuf, d_c0, BH * 4)) return -1; }
        else      { if (nntk_shim_memset(cbuf, 0, BH * 4)) return -1; }
    }
    RecParams p;
    p.ut = d_ut; p.bh = d_bh; p.c = cbuf;
    p.B = B; p.H = H;
    p.Hj_p = (H + 15) & ~15;
    p.Hk_p = (H + 31) & ~31;
    p.a0 = acts[0]; p.a1 = acts[1]; p.a2 = acts[2];
    p.a3 = nacts > 3 ? acts[3] : 0; p.a4 = nacts > 4 ? acts[4] : 0;
    dim3 grid((unsigned)((B + REC_BM - 1) / REC_BM), (unsigned)(p.Hj_p / REC_HN));
    const int span = nntk_prof_span_begin();
    for (int t = 0; t < T; ++t) {
        p.xw = d_xw + (size_t)t * B * G * H;
        p.h_prev = hbuf[t & 1];
        p.h_next = hbuf[(t + 1) & 1];
        if (return_sequences) { p.out = d_out + (size_t)t * H; p.out_ld = (long)T * H; }
        else if (t == T - 1)  { p.out = d_out; p.out_ld = H; }
        else                  { p.out = nullptr; p.out_ld = 0; }
        hipLaunchKernelGGL((rec_step_kernel<G, IS_LSTM>), grid, dim3(256), 0, nntk_stream(), p);
    }
    nntk_prof_span_end(span, T);
    NNTK_LAUNCH_CHECK("rec_step_kernel");
    if (d_hT) { if (nntk_shim_copy_d2d(d_hT, hbuf[T & 1], BH * 4)) return -1; }
    if (IS_LSTM && d_cT) { if (nntk_shim_copy_d2d(d_cT, cbuf, BH * 4)) return -1; }
    return 0;
}

extern "C" int nntk_shim_gru(const float *d_xw, const float *d_ut, const float *d_bh, const float *d_h0,
                             float *d_out, float *d_hT, float *d_work, int B, int T, int H,
                             int return_sequences, const int acts[3]) {
    if (B <= 0 || T <= 0) return 0;
    return run_recurrent<3, false>(d_xw, d_ut, d_bh, d_h0, nullptr, d_out, d_hT, nullptr, d_work, B, T, H,
                                   return_sequences, acts, 3);
}

extern "C" int nntk_shim_lstm(const float *d_xw, const float *d_ut, const float *d_bh, const float *d_h0,
                              const float *d_c0, float *d_out, float *d_hT, float *d_cT, float *d_work,
                              int B, int T, int H, int return_sequences, const int acts[5]) {
    if (B <= 0 || T <= 0) return 0;
    return run_recurrent<4, true>(d_xw, d_ut, d_bh, d_h0, d_c0, d_out, d_hT, d_cT, d_work, B, T, H,
                                  return_sequences, acts, 5);
}
