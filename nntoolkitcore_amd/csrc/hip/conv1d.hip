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
// KC = 32) x (tap kk); the window chunk is reused by all k taps.  LDS is double
// buffered and the next chunk's global loads are issued before the MFMA block of
// the current one (register staging, write-after-barrier).
//
// Roofline: exact-f32 MFMA is 64 FLOP/clk/SIMD = 157 TFLOP/s, so config 3
// (52.2 GFLOP over 686 MB) is MFMA-bound at 332 us, not HBM-bound (86 us).
#include "nntk_common.hpp"

#define CONV_BM 128
#define CONV_KC 32
#define CONV_AS (CONV_KC + 1)   // LDS row stride of the window chunk (odd: conflict-free ds_read_b32)

extern "C" void nntk_shim_conv_pack_sizes(int Cin, int Cout, int k, int *Cin_p, int *Cout_p) {
    (void)k;
    *Cin_p = (Cin + 1) & ~1;          // K-steps of 2 never straddle a tap
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
    int out_mode;        // 0: row b*Tout+x ; 1: row x*B+b
    int rows_a;          // (BM-1)*stride + k window rows per tile
};

// WM x WN wavefronts, each computing TM x TN MFMA tiles of 32x32.
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(ConvParams p) {
    constexpr int BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    static_assert(WM * TM * 32 == CONV_BM, "BM = 128");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int a_elems = p.rows_a * CONV_AS;
    float *As[2] = {smem, smem + a_elems};
    float *Ws[2] = {smem + 2 * a_elems, smem + 2 * a_elems + CONV_KC * BN};

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;

    const int tile = blockIdx.x;
    const int b = tile / p.tiles_per_seq;
    const int x0 = (tile % p.tiles_per_seq) * CONV_BM;
    const int n0 = blockIdx.y * BN;
    const float *in_b = p.in + (size_t)b * p.T * p.Cin;
    const int t0 = x0 * p.stride;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // register staging
    constexpr int W_PER_T = CONV_KC * BN / 4 / 256;           // float4 per thread for one W chunk
    const int a_total = p.rows_a * CONV_KC;
    constexpr int A_MAX = 24;                                  // rows_a*KC/256 must be <= A_MAX (host checks)
    float4 wreg[W_PER_T];
    float areg[A_MAX];

    const int n_cchunks = (p.Cin_p + CONV_KC - 1) / CONV_KC;
    const int n_chunks = n_cchunks * p.k;

    auto load_w = [&](int chunk) {
        const int cc = chunk / p.k, kk = chunk % p.k;
        const int i0 = cc * CONV_KC;
        const int len = min(CONV_KC, p.Cin_p - i0);
#pragma unroll
        for (int q = 0; q < W_PER_T; ++q) {
            int e = tid + q * 256;                 // float4 index inside [KC, BN/4]
            int r = e / (BN / 4), c4 = e % (BN / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < len)
                v = *reinterpret_cast<const float4 *>(p.wp + (size_t)(kk * p.Cin_p + i0 + r) * p.Cout_p + n0 + c4 * 4);
            wreg[q] = v;
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int q = 0; q < W_PER_T; ++q) {
            int e = tid + q * 256;
            *reinterpret_cast<float4 *>(Ws[buf] + (size_t)e * 4) = wreg[q];
        }
    };
    auto load_a = [&](int cc) {
        const int i0 = cc * CONV_KC;
#pragma unroll
        for (int q = 0; q < A_MAX; ++q) {
            int e = tid + q * 256;
            float v = 0.0f;
            if (e < a_total) {
                int r = e / CONV_KC, c = e % CONV_KC;
                int t = t0 + r, ch = i0 + c;
                if (t < p.T && ch < p.Cin) v = in_b[(size_t)t * p.Cin + ch];
            }
            areg[q] = v;
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int q = 0; q < A_MAX; ++q) {
            int e = tid + q * 256;
            if (e < a_total) {
                int r = e / CONV_KC, c = e % CONV_KC;
                As[buf][r * CONV_AS + c] = areg[q];
            }
        }
    };

    load_a(0);
    load_w(0);
    int abuf = 0;
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int cc = chunk / p.k, kk = chunk % p.k;
        const int wbuf = chunk & 1;
        if (kk == 0) { abuf = cc & 1; store_a(abuf); }
        store_w(wbuf);
        __syncthreads();
        if (chunk + 1 < n_chunks) {
            load_w(chunk + 1);
            if ((chunk + 1) % p.k == 0) load_a((chunk + 1) / p.k);
        }
        const int len = min(CONV_KC, p.Cin_p - cc * CONV_KC);
        const float *A = As[abuf] + ((wm * TM * 32 + l31) * p.stride + kk) * CONV_AS + kh;
        const float *W = Ws[wbuf] + kh * BN + wn * TN * 32 + l31;
        for (int s = 0; s < len; s += 2) {
            float a[TM], w[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = A[i * 32 * p.stride * CONV_AS + s];
#pragma unroll
            for (int j = 0; j < TN; ++j) w[j] = W[s * BN + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], w[j], acc[i][j], 0, 0, 0);
        }
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
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int x = x0 + wm * TM * 32 + i * 32 + row;
                if (x >= p.Tout) continue;
                float v = acc[i][j][r] + bias;
                if (p.bn) v = ((v - mu) / sd) * g + be;
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

template <int WM, int WN, int TM, int TN>
static int launch_mfma(const ConvParams &p) {
    constexpr int BN = WN * TN * 32;
    size_t lds = (size_t)(2 * p.rows_a * CONV_AS + 2 * CONV_KC * BN) * sizeof(float);
    dim3 grid((unsigned)((long)p.B * p.tiles_per_seq), (unsigned)(p.Cout_p / BN));
    auto kern = conv1d_mfma_kernel<WM, WN, TM, TN>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return nntk_fail("hipFuncSetAttribute(conv1d)", e);
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, nntk_stream(), p);
    NNTK_LAUNCH_CHECK("conv1d_mfma_kernel");
    return 0;
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

    const bool window_fits = (long)p.rows_a * CONV_KC <= 24L * 256;     // register staging budget (A_MAX)
    const long Kdim = (long)Cin * k;
    if (!window_fits || Kdim < 16 || Cout < 32) {
        const long total = (long)B * Tout * Cout;
        long g = (total + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(conv1d_valu_kernel, dim3((unsigned)g), dim3(256), 0, nntk_stream(), p);
        NNTK_LAUNCH_CHECK("conv1d_valu_kernel");
        return 0;
    }
    if (p.Cout_p % 128 == 0) return launch_mfma<2, 2, 2, 2>(p);
    if (p.Cout_p % 64 == 0)  return launch_mfma<4, 1, 1, 2>(p);
    return launch_mfma<4, 1, 1, 1>(p);
}
