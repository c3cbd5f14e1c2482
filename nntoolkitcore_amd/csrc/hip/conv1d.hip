// conv1d.hip -- K2/K3: channels-last Conv1d as an implicit GEMM on the exact-f32
// MFMA (v_mfma_f32_32x32x2_f32), with bias + BatchNorm(inference) + activation
// fused into the epilogue.  The k = 1 case is the dense GEMM used by Dense,
// TimeDistributedDense and the GRU/LSTM input projections.
//
// Reference semantics (layers/conv_1d.c:122-147): valid cross-correlation,
//   y[x, o] = b[o] + sum_i sum_kk in[x*stride + kk, i] * W[o, i, kk]
// on a [T, Cin] channels-last sequence.  Because the layout is channels-last, the
// im2col row of output x is the CONTIGUOUS span in[x*stride .. x*stride+k-1, :],
// so the sliding window is staged once per tile in LDS and every tap reads it at a
// row offset -- no im2col buffer, no transpose pass (the reference's op_mat_transp).
//
// Tile: BM = 128 output positions x BN in {32, 64, 128} output channels per
// 256-thread workgroup (4 wavefronts of 64).  K is walked as (channel chunk of
// KC = 16) x (tap kk); the window chunk is reused by all k taps.  LDS is double
// buffered and the next chunk's global loads are issued before the MFMA block of
// the current one (register staging, write-after-barrier).
//
// Roofline: exact-f32 MFMA is 64 FLOP/clk/SIMD = 157 TFLOP/s, so config 3
// (52.2 GFLOP over 686 MB) is MFMA-bound at 332 us, not HBM-bound (86 us).
#include "nntk_common.hpp"
#include <stdlib.h>

#define CONV_BM 128

extern "C" void nntk_shim_conv_pack_sizes(int Cin, int Cout, int k, int *Cin_p, int *Cout_p) {
    (void)k;
    *Cin_p = (Cin + 7) & ~7;          // the MFMA loop consumes 8 channels (4 K-steps of 2) per trip
    *Cout_p = (Cout + 31) & ~31;      // whole 32-wide MFMA column tiles
}

struct ConvParams {
    const float *in;     // [B, T, Cin]
    const float *wp;     // [k * Cin_p, Cout_p]
    const float *bias;   // [Cout]
    const float *bn;     // NULL or gamma|beta|mean|var, each [Cout]
    float *out;
    float bn_eps;
    float relu_a;
    int act_kind;
    int B, T, Cin, Cin_p, Cout, Cout_p, k, stride, Tout;
    int tiles_per_seq;   // ceil(Tout / BM)
    int m_tiles, n_tiles; // B * tiles_per_seq row tiles x Cout_p / BN column tiles
    int out_mode;        // 0: row b*Tout+x ; 1: row x*B+b
    int rows_a;          // (BM-1)*stride + k window rows per tile
    int bn_fast;         // A/B: multiply by 1/sd instead of the reference's divide
};

// WM x WN wavefronts, each computing TM x TN MFMA tiles of 32x32.
// A4 = the window can be fetched with 16-byte loads (Cin % 4 == 0, 16-B aligned base).
template <int WM, int WN, int TM, int TN, bool A4, int KC>
__global__ __launch_bounds__(256, 2) void conv1d_mfma_kernel(ConvParams p) {
    constexpr int BN = WN * TN * 32;
    constexpr int AS = KC + 1;                       // LDS row stride of the window chunk (odd: conflict-free ds_read_b32)
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    static_assert(WM * TM * 32 == CONV_BM, "BM = 128");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS carve (offsets from the one dynamic array, so every access stays a ds_* op):
    //   window chunk [2][rows_a][AS] | weight chunk [2][KC][BN]
    const int a_elems = p.rows_a * AS;
    constexpr int w_elems = KC * BN;
    const int w_base = 2 * a_elems;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;

    // XCD-aware tile order (blocks are dealt round-robin over the 8 XCDs): XCD c walks the
    // row tiles c, c+8, ... and, for each, ALL column tiles back to back, so a row tile's
    // window is fetched from HBM once and re-read from that XCD's L2 by its other column
    // tiles.  (x-major order re-streamed the whole input once per column tile: PMC showed
    // 16x / 8x the algorithmic reads on the LSTM input projection / TDD GEMMs.)
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = (local / p.n_tiles) * 8 + xcd;
    if (tile >= p.m_tiles) return;
    const int b = tile / p.tiles_per_seq;
    const int x0 = (tile % p.tiles_per_seq) * CONV_BM;
    const int n0 = (local % p.n_tiles) * BN;
    const float *in_b = p.in + (size_t)b * p.T * p.Cin;
    const int t0 = x0 * p.stride;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- register staging (global -> VGPR now, VGPR -> LDS after the next barrier) ----
    constexpr int W_F4 = KC * BN / 4;                // float4 per weight chunk
    constexpr int W_PT = (W_F4 + 255) / 256;         // float4 per thread per weight chunk
    constexpr int A_TPR = A4 ? KC / 4 : KC;          // threads per window row
    constexpr int AR_STEP = 256 / A_TPR;             // rows covered per pass of the workgroup
    constexpr int A_PT = (192 + AR_STEP - 1) / AR_STEP;   // window: rows_a <= 192 (host checks)
    float4 wreg[W_PT];
    float4 areg4[A4 ? A_PT : 1];
    float areg[A4 ? 1 : A_PT];
    // thread -> window element mapping
    const int ac = A4 ? (tid % A_TPR) * 4 : (tid % A_TPR);   // first channel inside the chunk
    const int ar = tid / A_TPR;                              // first row; rows advance by AR_STEP

    const int n_cchunks = (p.Cin_p + KC - 1) / KC;
    const int n_chunks = n_cchunks * p.k;

    auto load_w = [&](int cc, int kk) {
        const int i0 = cc * KC;
        const int len = min(KC, p.Cin_p - i0);
#pragma unroll
        for (int q = 0; q < W_PT; ++q) {
            const int e = tid + q * 256;             // float4 index inside [KC, BN/4]
            const int r = e / (BN / 4), c4 = e % (BN / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < len && (W_F4 % 256 == 0 || e < W_F4))
                v = *reinterpret_cast<const float4 *>(p.wp + (size_t)(kk * p.Cin_p + i0 + r) * p.Cout_p + n0 + c4 * 4);
            wreg[q] = v;
        }
    };
    auto load_a = [&](int cc) {
        const int ch = cc * KC + ac;
#pragma unroll
        for (int q = 0; q < A_PT; ++q) {
            const int r = ar + q * AR_STEP;
            const int t = t0 + r;
            const bool ok = r < p.rows_a && t < p.T && ch < p.Cin;
            if (A4) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) v = *reinterpret_cast<const float4 *>(in_b + (size_t)t * p.Cin + ch);
                areg4[q] = v;
            } else {
                areg[q] = ok ? in_b[(size_t)t * p.Cin + ch] : 0.0f;
            }
        }
    };

    load_a(0);
    load_w(0, 0);
    int cc = 0, kk = 0;
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int wbuf = chunk & 1;
        const int abuf = cc & 1;
        if (kk == 0) {
            float *As = smem + abuf * a_elems;
#pragma unroll
            for (int q = 0; q < A_PT; ++q) {
                const int r = ar + q * AR_STEP;
                if (r < p.rows_a) {
                    if (A4) {
                        float *d = As + r * AS + ac;
                        d[0] = areg4[q].x; d[1] = areg4[q].y; d[2] = areg4[q].z; d[3] = areg4[q].w;
                    } else {
                        As[r * AS + ac] = areg[q];
                    }
                }
            }
        }
        {
            float *Ws = smem + w_base + wbuf * w_elems;
#pragma unroll
            for (int q = 0; q < W_PT; ++q)
                if (W_F4 % 256 == 0 || tid + q * 256 < W_F4)
                    *reinterpret_cast<float4 *>(Ws + (size_t)(tid + q * 256) * 4) = wreg[q];
        }
        __syncthreads();
        // next chunk's coordinates; prefetch it while this one is multiplied
        int ncc = cc, nkk = kk + 1;
        if (nkk == p.k) { nkk = 0; ncc = cc + 1; }
        if (chunk + 1 < n_chunks) {
            load_w(ncc, nkk);
            if (nkk == 0) load_a(ncc);
        }
        const int len = min(KC, p.Cin_p - cc * KC);       // multiple of 8
        const float *A = smem + abuf * a_elems + ((wm * TM * 32 + l31) * p.stride + kk) * AS + kh;
        const float *W = smem + w_base + wbuf * w_elems + kh * BN + wn * TN * 32 + l31;
        const int a_tile = 32 * p.stride * AS;
        for (int s = 0; s < len; s += 8) {
            float a[4][TM], w[4][TN];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[u][i] = A[i * a_tile + s + 2 * u];
#pragma unroll
                for (int j = 0; j < TN; ++j) w[u][j] = W[(s + 2 * u) * BN + j * 32];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], w[u][j], acc[i][j], 0, 0, 0);
        }
        cc = ncc; kk = nkk;
    }

    // ---- epilogue: bias, BatchNorm (batch_norm.c:140-163 op order), activation ----
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int o = n0 + wn * TN * 32 + j * 32 + l31;
        if (o >= p.Cout) continue;
        const float bias = p.bias ? p.bias[o] : 0.0f;
        float g = 1.f, be = 0.f, mu = 0.f, sd = 1.f;
        if (p.bn) {
            g = p.bn[o]; be = p.bn[p.Cout + o]; mu = p.bn[2 * p.Cout + o];
            sd = sqrtf(p.bn[3 * p.Cout + o] + p.bn_eps);
        }
        const bool fast_bn = p.bn_fast != 0;
        const float rsd = 1.0f / sd;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int x = x0 + wm * TM * 32 + i * 32 + row;
                if (x >= p.Tout) continue;
                float v = acc[i][j][r] + bias;
                if (p.bn) v = fast_bn ? ((v - mu) * rsd) * g + be : ((v - mu) / sd) * g + be;
                v = nntk_act(p.act_kind, v, p.relu_a);
                const size_t orow = p.out_mode ? ((size_t)x * p.B + b) : ((size_t)b * p.Tout + x);
                p.out[orow * p.Cout + o] = v;
            }
        }
    }
}

// Generic VALU kernel for shapes the MFMA tile does not cover (tiny K such as
// config 1's Conv1d(1->16, k=9), or huge stride*k windows).  One thread per output
// element, accumulation in the reference's order (dot over taps inside, channels
// outside: conv_1d.c:136-140).
__global__ __launch_bounds__(256) void conv1d_valu_kernel(ConvParams p) {
    const long total = (long)p.B * p.Tout * p.Cout;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int o = (int)(e % p.Cout);
        const long bx = e / p.Cout;
        const int x = (int)(bx % p.Tout);
        const int b = (int)(bx / p.Tout);
        const float *in_b = p.in + (size_t)b * p.T * p.Cin + (size_t)x * p.stride * p.Cin;
        float result = 0.0f;
        for (int i = 0; i < p.Cin; ++i) {
            float dot = 0.0f;
            for (int kk = 0; kk < p.k; ++kk)
                dot += in_b[(size_t)kk * p.Cin + i] * p.wp[(size_t)(kk * p.Cin_p + i) * p.Cout_p + o];
            result += dot;
        }
        if (p.bias) result += p.bias[o];
        if (p.bn) {
            float sd = sqrtf(p.bn[3 * p.Cout + o] + p.bn_eps);
            result = ((result - p.bn[2 * p.Cout + o]) / sd) * p.bn[o] + p.bn[p.Cout + o];
        }
        result = nntk_act(p.act_kind, result, p.relu_a);
        const size_t orow = p.out_mode ? ((size_t)x * p.B + b) : ((size_t)b * p.Tout + x);
        p.out[orow * p.Cout + o] = result;
    }
}

template <int WM, int WN, int TM, int TN, bool A4, int KC>
static int launch_mfma_kc(const ConvParams &p) {
    constexpr int BN = WN * TN * 32;
    size_t lds = (size_t)(2 * p.rows_a * (KC + 1) + 2 * KC * BN) * sizeof(float);
    ConvParams q = p;
    q.m_tiles = p.B * p.tiles_per_seq;
    q.n_tiles = p.Cout_p / BN;
    const long blocks = (long)((q.m_tiles + 7) / 8) * 8 * q.n_tiles;
    if ((long)p.B * p.tiles_per_seq > 0x7fffffffL / 8 || blocks > 0x7fffffffL)
        return nntk_fail_msg("conv1d: too many tiles for one launch");
    dim3 grid((unsigned)blocks);
    auto kern = conv1d_mfma_kernel<WM, WN, TM, TN, A4, KC>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return nntk_fail("hipFuncSetAttribute(conv1d)", e);
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, nntk_stream(), q);
    NNTK_LAUNCH_CHECK("conv1d_mfma_kernel");
    return 0;
}

template <int WM, int WN, int TM, int TN, bool A4>
static int launch_mfma(const ConvParams &p) {
    const char *e = getenv("NNTK_CONV_KC");
    // Chunk depth 16 halves the LDS footprint (33 KB at BN = 128): 3 workgroups per CU instead of
    // 2, which is what hides the barrier / staging latency.  Measured vs depth 32: config 3
    // 0.79 -> 0.62 ms, LSTM input projection 3.6 -> 2.6 ms, TDD 5.08 -> 4.67 ms.  NNTK_CONV_KC=32 restores it.
    if constexpr (WN * TN * 32 >= 64) {
        if (!(e && e[0] == '3')) return launch_mfma_kc<WM, WN, TM, TN, A4, 16>(p);
    }
    return launch_mfma_kc<WM, WN, TM, TN, A4, 32>(p);
}

extern "C" int nntk_shim_conv1d(const float *d_in, const float *d_wp, const float *d_bias, const float *d_bn,
                                float bn_eps, int act_kind, float relu_a, float *d_out,
                                int B, int T, int Cin, int Cout, int k, int stride, int Tout, int out_mode) {
    if (B <= 0 || Tout <= 0) return 0;
    if (act_kind == NNTK_ACT_SOFTMAX || act_kind == NNTK_ACT_CUSTOM)
        return nntk_fail_msg("conv1d epilogue: softmax/custom activations are not fusable");
    if (act_kind == NNTK_ACT_NONE) act_kind = NNTK_ACT_IDENTITY;
    ConvParams p;
    p.in = d_in; p.wp = d_wp; p.bias = d_bias; p.bn = d_bn; p.out = d_out;
    p.bn_eps = bn_eps; p.relu_a = relu_a; p.act_kind = act_kind;
    p.B = B; p.T = T; p.Cin = Cin; p.Cout = Cout; p.k = k; p.stride = stride; p.Tout = Tout;
    nntk_shim_conv_pack_sizes(Cin, Cout, k, &p.Cin_p, &p.Cout_p);
    p.tiles_per_seq = (Tout + CONV_BM - 1) / CONV_BM;
    p.out_mode = out_mode;
    p.rows_a = (CONV_BM - 1) * stride + k;
    { const char *e = getenv("NNTK_BN_FAST"); p.bn_fast = (e && e[0] == '1') ? 1 : 0; }

    const bool window_fits = p.rows_a <= 192;                          // register staging budget (A_PT)
    const long Kdim = (long)Cin * k;
    if (!window_fits || Kdim < 16 || Cout < 32) {
        const long total = (long)B * Tout * Cout;
        long g = (total + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(conv1d_valu_kernel, dim3((unsigned)g), dim3(256), 0, nntk_stream(), p);
        NNTK_LAUNCH_CHECK("conv1d_valu_kernel");
        return 0;
    }
    const bool a4 = (Cin % 4 == 0) && ((size_t)d_in % 16 == 0);
    { const char *e = getenv("NNTK_CONV_LOWLDS");
      if (e && e[0] == '1' && a4 && p.Cout_p % 64 == 0 && k == 1) return launch_mfma_kc<4, 1, 1, 2, true, 8>(p); }
    if (p.Cout_p % 128 == 0) return a4 ? launch_mfma<2, 2, 2, 2, true>(p) : launch_mfma<2, 2, 2, 2, false>(p);
    if (p.Cout_p % 64 == 0)  return a4 ? launch_mfma<4, 1, 1, 2, true>(p) : launch_mfma<4, 1, 1, 2, false>(p);
    return a4 ? launch_mfma<4, 1, 1, 1, true>(p) : launch_mfma<4, 1, 1, 1, false>(p);
}
