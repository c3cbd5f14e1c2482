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
// Everything around the MFMAs is written to need (almost) no VALU instructions, because on
// gfx950 an f32 MFMA and a VALU instruction never overlap -- not inside a wave, not between
// the waves of a SIMD (tools/micro/coissue_bench.hip: 3200 MFMAs = 205k cycles alone, + 4.1
// cycles for every v_fma added in their shadow).  Every VALU instruction is therefore paid
// in full out of the 157 TFLOP/s:
//   * both LDS tiles are K-contiguous ([row][KC + 4]; the weights are packed [Cout_p][k * Cin_p]
//     in HBM for that) and the K order inside an 8-deep block is permuted so that a lane's four
//     operands are ONE ds_read_b128 (lane half kh takes k = 4 kh .. 4 kh + 3): 4 LDS reads and
//     no address arithmetic per 16 MFMAs (was 8 ds_read2_b32 + 6 v_add);
//   * global -> register staging uses buffer loads whose per-thread offset never changes; the
//     chunk advance moves the descriptor's base (scalar ALU), the tensor's end is its range check
//     (only the VECTOR offset is range-checked, so nothing that must be clipped rides in soffset);
//   * the epilogue stores through a buffer descriptor with the row advance in the scalar offset;
//     padded columns get an out-of-range vector offset, and only the last row tile of a sequence
//     takes a compare-per-element path.
//
// Roofline: exact-f32 MFMA is 64 FLOP/clk/SIMD = 157 TFLOP/s, so config 3
// (52.2 GFLOP over 686 MB) is MFMA-bound at 332 us, not HBM-bound (86 us).
#include "nntk_common.hpp"
#include <stdlib.h>
#include <type_traits>

#define CONV_BM 128
#define CONV_KC 16
#define CONV_LS (CONV_KC + 4)     // LDS row stride in floats: 16-B aligned rows, conflict-free b128 reads

extern "C" void nntk_shim_conv_pack_sizes(int Cin, int Cout, int k, int *Cin_p, int *Cout_p) {
    (void)k;
    *Cin_p = (Cin + CONV_KC - 1) & ~(CONV_KC - 1);   // whole K chunks: no guard in the weight staging
    *Cout_p = (Cout + 31) & ~31;                     // whole 32-wide MFMA column tiles
}

struct ConvParams {
    const float *in;     // [B, T, Cin]: row t of sequence b at in + b * in_seq + t * in_row
    long in_seq, in_row; // element strides (T * Cin, Cin for the plain layout)
    const float *wp;     // [Cout_p][k * Cin_p]   (K-contiguous)
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
#ifdef NNTK_CONV_DBG
    int dbg;             // timing experiments only: 1 no stores, 2 no MFMAs, 4 no global loads in the loop
#endif
};
#ifdef NNTK_CONV_DBG
#define CONV_DBG(bit) (p.dbg & (bit))
#else
#define CONV_DBG(bit) 0
#endif

typedef unsigned v4u32_t __attribute__((ext_vector_type(4)));

// Out-of-range sentinel for a lane's VECTOR offset (the only part of a buffer address the hardware range-checks).
// Every descriptor here is clamped to at most CONV_OOB bytes, so the sentinel is out of range whatever the
// tensor's size: a load returns 0, a store is dropped.  Legitimate vector offsets stay far below it: descriptors
// are based at the tile (or weight matrix) they serve, and the host rejects shapes whose in-tile offsets would not fit.
#define CONV_OOB 0x7ffffff0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_rsrc(const void *base, size_t bytes) {
    const unsigned n = bytes > (size_t)CONV_OOB ? (unsigned)CONV_OOB : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)n, 0x00020000);
}

// WM x WN wavefronts, each computing TM x TN MFMA tiles of 32x32.
// A4 = the window can be fetched with 16-byte loads (Cin % 4 == 0, 16-B aligned base).
template <int WM, int WN, int TM, int TN, bool A4>
__global__ __launch_bounds__(256, 2) void conv1d_mfma_kernel(ConvParams p) {
    constexpr int BN = WN * TN * 32;
    constexpr int KC = CONV_KC, LS = CONV_LS;
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    static_assert(WM * TM * 32 == CONV_BM, "BM = 128");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS carve: window chunk [2][rows_a][LS] | weight chunk [2][BN][LS]
    const int a_elems = p.rows_a * LS;
    constexpr int w_elems = BN * LS;
    const int w_base = 2 * a_elems;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int Ktot = p.k * p.Cin_p;

    // (Measured and NOT kept, see DESIGN.md: persistent workgroups with the next tile's first chunk in
    // flight during the epilogue, an LDS-transposed epilogue with 16-byte stores, a start stagger of the
    // co-resident workgroups -- each within +-3 % of this simpler form on the stack's GEMMs.)

    f32x16 acc[TM][TN];

    // ---- global -> register staging through buffer descriptors ----
    constexpr int A_TPR = A4 ? KC / 4 : KC;          // threads per window row
    constexpr int AR_STEP = 256 / A_TPR;             // rows covered per pass of the workgroup
    constexpr int A_PT = (192 + AR_STEP - 1) / AR_STEP;   // window: rows_a <= 192 (host checks)
    constexpr int W_PT = (BN * (KC / 4) + 255) / 256;     // 16-byte pieces per thread per weight chunk
    constexpr bool W_ALL = (BN * (KC / 4)) % 256 == 0;
    const int ac = A4 ? (tid % A_TPR) * 4 : (tid % A_TPR);   // first channel inside the chunk
    const int ar = tid / A_TPR;                              // first row; rows advance by AR_STEP
    const int wr = tid >> 2, wc4 = tid & 3;                  // weight row (output channel) / 16-byte piece
    v4u32_t areg4[A4 ? A_PT : 1];
    unsigned areg[A4 ? 1 : A_PT];
    v4u32_t wreg[W_PT];
    int a_voff[A_PT], w_voff[W_PT];
#pragma unroll
    for (int q = 0; q < A_PT; ++q) a_voff[q] = (int)(((long)(ar + q * AR_STEP) * p.in_row + ac) * 4);
#pragma unroll
    for (int q = 0; q < W_PT; ++q) w_voff[q] = ((wr + q * 64) * Ktot + wc4 * 4) * 4;
    const bool cin_ragged = (p.Cin % KC) != 0;       // the last channel chunk runs past the row: mask it

    const int n_cchunks = p.Cin_p / KC;
    const int n_chunks = n_cchunks * p.k;
    const int last_blocks = (((p.Cin + 7) & ~7) - (n_cchunks - 1) * KC + 7) / 8;      // 1 or 2

    // weights: ONE descriptor for the kernel; tile column block and chunk ride in the scalar offset
    // (always in range: the packed matrix is whole chunks x whole column tiles)
    const __amdgpu_buffer_rsrc_t rs_w = conv_rsrc(p.wp, (size_t)p.Cout_p * Ktot * 4);
    auto load_w = [&](int tn0, int cc, int kk) {
        const int soff = (tn0 * Ktot + kk * p.Cin_p + cc * KC) * 4;
#pragma unroll
        for (int q = 0; q < W_PT; ++q)
            if (W_ALL || tid + q * 256 < BN * (KC / 4))
                wreg[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_voff[q], soff, 0);
    };
    // window: the descriptor starts at the tile's first window row and ends with the tensor, so the
    // (range-checked) per-thread row offset drops rows past the tensor's end; rows past this
    // sequence's end read the next sequence and only feed outputs x >= Tout, which are never stored.
    // The chunk's channel offset rides in the unchecked scalar offset: a row that exists contains all
    // its chunks -- except the ragged last chunk, whose out-of-row lanes get an out-of-range offset.
    const size_t in_total = (size_t)p.B * p.T * p.Cin;
    auto load_a = [&](int tb, int tx0, int cc) {
        const size_t in_off = (size_t)tb * p.in_seq + (size_t)tx0 * p.stride * p.in_row;
        const __amdgpu_buffer_rsrc_t rs_in = conv_rsrc(p.in + in_off, (in_total - in_off) * 4);
        const int soff = cc * KC * 4;
        if (cin_ragged && cc == n_cchunks - 1) {     // uniform branch, last chunk only
            const bool ch_ok = cc * KC + ac < p.Cin;
#pragma unroll
            for (int q = 0; q < A_PT; ++q) {
                const int vo = ch_ok ? a_voff[q] : CONV_OOB;
                if (A4) areg4[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, soff, 0);
                else    areg[q] = __builtin_amdgcn_raw_buffer_load_b32(rs_in, vo, soff, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < A_PT; ++q) {
                if (A4) areg4[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_voff[q], soff, 0);
                else    areg[q] = __builtin_amdgcn_raw_buffer_load_b32(rs_in, a_voff[q], soff, 0);
            }
        }
    };

    // per-lane LDS read bases (floats): row l31 of the wave's first tile, k offset 4*kh
    const int a_rd = ((wm * TM * 32 + l31) * p.stride) * LS + 4 * kh;
    const int w_rd = w_base + (wn * TN * 32 + l31) * LS + 4 * kh;
    const int a_tile = 32 * p.stride * LS;

    load_a(b, x0, 0);
    load_w(n0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
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
                    if (A4) *reinterpret_cast<v4u32_t *>(As + r * LS + ac) = areg4[q];
                    else    *reinterpret_cast<unsigned *>(As + r * LS + ac) = areg[q];
                }
            }
        }
        {
            float *Ws = smem + w_base + wbuf * w_elems;
#pragma unroll
            for (int q = 0; q < W_PT; ++q)
                if (W_ALL || tid + q * 256 < BN * (KC / 4))
                    *reinterpret_cast<v4u32_t *>(Ws + (wr + q * 64) * LS + wc4 * 4) = wreg[q];
        }
        __syncthreads();
        // next chunk's coordinates; prefetch it while this one is multiplied
        int ncc = cc, nkk = kk + 1;
        if (nkk == p.k) { nkk = 0; ncc = cc + 1; }
        if (chunk + 1 < n_chunks && !CONV_DBG(4)) {
            load_w(n0, ncc, nkk);
            if (nkk == 0) load_a(b, x0, ncc);
        }
        const float *A = smem + abuf * a_elems + kk * LS + a_rd;
        const float *W = smem + wbuf * w_elems + w_rd;
        auto block8 = [&](int s) {
            float4 a[TM], w[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4 *>(A + i * a_tile + s);
#pragma unroll
            for (int j = 0; j < TN; ++j) w[j] = *reinterpret_cast<const float4 *>(W + j * 32 * LS + s);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float av = u == 0 ? a[i].x : u == 1 ? a[i].y : u == 2 ? a[i].z : a[i].w;
                        const float wv = u == 0 ? w[j].x : u == 1 ? w[j].y : u == 2 ? w[j].z : w[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wv, acc[i][j], 0, 0, 0);
                    }
        };
        if (!CONV_DBG(2)) {
        block8(0);
        // the last channel chunk may hold only one 8-deep block of real channels (the rest is zero padding)
        if (cc != n_cchunks - 1 || last_blocks > 1) block8(8);
        }
        cc = ncc; kk = nkk;
    }

    // ---- epilogue: bias, BatchNorm (batch_norm.c:140-163 op order), activation, buffer stores ----
    // descriptor = THIS TILE's output rows: based at the tile's first row (so offsets inside it stay small however
    // large the tensor is -- a TimeDistributedDense output passes 2 GiB at ~540 k rows of 1000) and never longer than
    // CONV_OOB bytes (so the padded columns' sentinel offset is out of range by construction).  Interior row tiles
    // store with the row advance in the (unchecked) scalar offset and the padded columns on the out-of-range vector
    // offset; the last row tile of a sequence compares every element's row instead.  The uniform choices (BatchNorm
    // or not, which activation, which store path) are taken ONCE, outside the 64-element loops: inside them every
    // instruction is VALU time taken from the MFMAs.
    const size_t row_elems = (size_t)(p.out_mode ? p.B : 1) * p.Cout;           // distance between output rows x, x+1
    const size_t row_bytes = row_elems * 4;
    const size_t obase = (p.out_mode ? (size_t)b * p.Cout : (size_t)b * p.Tout * p.Cout) + (size_t)x0 * row_elems;
    const size_t oend = p.out_mode ? (size_t)p.Tout * p.B * p.Cout : ((size_t)b + 1) * p.Tout * p.Cout;
    const __amdgpu_buffer_rsrc_t rs_out = conv_rsrc(p.out + obase, (oend - obase) * 4);
    const bool fast_store = x0 + CONV_BM <= p.Tout &&                              // every row of the tile exists
                            row_bytes * (size_t)(CONV_BM + 4) < (size_t)CONV_OOB;  // and is reachable by 32-bit offsets
    const int rb = (int)row_bytes;

    auto epilogue = [&](auto bn_tag, auto act_tag, auto fast_tag) {
        constexpr bool HAS_BN = decltype(bn_tag)::value;
        constexpr int ACT = decltype(act_tag)::value;            // -1: run-time kind
        constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int o = n0 + wn * TN * 32 + j * 32 + l31;
            const bool col_ok = o < p.Cout;
            const float bias = (p.bias && col_ok) ? p.bias[o] : 0.0f;
            float g = 1.f, be = 0.f, mu = 0.f, sd = 1.f;
            if (HAS_BN && col_ok) {
                g = p.bn[o]; be = p.bn[p.Cout + o]; mu = p.bn[2 * p.Cout + o];
                sd = sqrtf(p.bn[3 * p.Cout + o] + p.bn_eps);
            }
            const float rsd = 1.0f / sd;
            // per-lane part of the address: column, and the +4 rows of the upper lane half
            const int voff = col_ok ? o * 4 + kh * 4 * rb : CONV_OOB;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int xr = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2);          // row inside the tile, wave-uniform
                    const int xs = x0 + xr;
                    float v = acc[i][j][r] + bias;
                    if (HAS_BN) {
                        // (v - mu) / sd with the quotient refined by one FMA step: the correctly rounded
                        // quotient without the 11-instruction IEEE division sequence
                        const float d = v - mu;
                        float qn = d * rsd;
                        qn = fmaf(fmaf(-qn, sd, d), rsd, qn);
                        v = p.bn_fast ? (d * rsd) * g + be : qn * g + be;
                    }
                    v = ACT == -1 ? nntk_act(p.act_kind, v, p.relu_a)
                      : ACT == NNTK_ACT_RELU ? nntk_act(NNTK_ACT_RELU, v, p.relu_a)
                      : v;
                    if (FAST) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, voff, xr * rb, 0);
                    } else {
                        const int x = xs + 4 * kh;
                        if (x < p.Tout && col_ok) {
                            const size_t orow = p.out_mode ? ((size_t)x * p.B + b) : ((size_t)b * p.Tout + x);
                            p.out[orow * p.Cout + o] = v;
                        }
                    }
                }
            }
        }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    using AId = std::integral_constant<int, NNTK_ACT_IDENTITY>;
    using ARelu = std::integral_constant<int, NNTK_ACT_RELU>;
    using AAny = std::integral_constant<int, -1>;
    if (CONV_DBG(1) && acc[0][0][0] != 12345.678f) {
    } else if (fast_store) {
        if (!p.bn && p.act_kind == NNTK_ACT_IDENTITY) epilogue(F_{}, AId{}, T_{});
        else if (p.bn && p.act_kind == NNTK_ACT_RELU) epilogue(T_{}, ARelu{}, T_{});
        else if (p.bn)                                epilogue(T_{}, AAny{}, T_{});
        else                                          epilogue(F_{}, AAny{}, T_{});
    } else {
        if (p.bn) epilogue(T_{}, AAny{}, F_{});
        else      epilogue(F_{}, AAny{}, F_{});
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
                dot += in_b[(size_t)kk * p.Cin + i] * p.wp[(size_t)o * p.k * p.Cin_p + (size_t)kk * p.Cin_p + i];
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

template <int WM, int WN, int TM, int TN, bool A4>
static int launch_mfma(const ConvParams &p) {
    constexpr int BN = WN * TN * 32;
    // 41 KB at BN = 128: three workgroups per CU, which is what hides the barrier / staging latency
    size_t lds = (size_t)(2 * p.rows_a * CONV_LS + 2 * BN * CONV_LS) * sizeof(float);
    ConvParams q = p;
    q.m_tiles = p.B * p.tiles_per_seq;
    q.n_tiles = p.Cout_p / BN;
    const long blocks = (long)((q.m_tiles + 7) / 8) * 8 * q.n_tiles;
    if ((long)p.B * p.tiles_per_seq > 0x7fffffffL / 8 || blocks > 0x7fffffffL)
        return nntk_fail_msg("conv1d: too many tiles for one launch");
    auto kern = conv1d_mfma_kernel<WM, WN, TM, TN, A4>;
    if (lds > 64 * 1024) {
        if (nntk_set_max_dynamic_lds((const void *)kern, lds)) return -1;
    }
    dim3 grid((unsigned)blocks);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, nntk_stream(), q);
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
    p.in_seq = (long)T * Cin; p.in_row = Cin;
    nntk_shim_conv_pack_sizes(Cin, Cout, k, &p.Cin_p, &p.Cout_p);
    p.tiles_per_seq = (Tout + CONV_BM - 1) / CONV_BM;
    p.out_mode = out_mode;
    p.rows_a = (CONV_BM - 1) * stride + k;
    const NntkOptions &opt = nntk_options();
    p.bn_fast = opt.bn_fast == 1 ? 1 : 0;
#ifdef NNTK_CONV_DBG
    p.dbg = opt.conv_dbg;
#endif

    if ((long)p.Cout_p * k * p.Cin_p * 4 >= (long)CONV_OOB)
        return nntk_fail_msg("conv1d: packed weights must stay below 2 GiB (32-bit buffer offsets)");
    if ((long)(192 + 1) * Cin * 4 >= (long)CONV_OOB)
        return nntk_fail_msg("conv1d: a window of 192 input rows must stay below 2 GiB (32-bit buffer offsets)");
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
    // Time-major output of a dense GEMM (the recurrent input projection): tile over the BATCH at a
    // fixed timestep instead of over time within a sequence.  A tile then writes 128 rows of one
    // [B, Cout] slab (8 KB apart for LSTM-512) instead of 128 rows that are B * Cout * 4 bytes = 4 MB
    // apart; its input rows become the strided side (T * Cin apart, each still Cin contiguous floats).
    // Same arithmetic per output element, so the results are bit-identical.
    // (small batches keep the time tiling: a batch tile would be mostly padding -- one sequence of 1000 steps is
    // 8 time tiles but 1000 batch tiles of one valid row each)
    const bool tm_batch = opt.gemm_tm_batch >= 0 ? opt.gemm_tm_batch != 0 : B >= 64;
    if (out_mode == 1 && k == 1 && stride == 1 && tm_batch &&
        (long)(CONV_BM + 64) * T * Cin * 4 < (long)CONV_OOB) {
        p.B = T; p.T = B; p.Tout = B;                 // "sequences" = timesteps, "rows" = batch entries
        p.in_seq = Cin; p.in_row = (long)T * Cin;
        p.out_mode = 0;                               // row (t, b) -> t * B + b: exactly the time-major layout
        p.tiles_per_seq = (p.Tout + CONV_BM - 1) / CONV_BM;
    }
    if (p.Cout_p % 128 == 0) return a4 ? launch_mfma<2, 2, 2, 2, true>(p) : launch_mfma<2, 2, 2, 2, false>(p);
    if (p.Cout_p % 64 == 0)  return a4 ? launch_mfma<4, 1, 1, 2, true>(p) : launch_mfma<4, 1, 1, 2, false>(p);
    return a4 ? launch_mfma<4, 1, 1, 1, true>(p) : launch_mfma<4, 1, 1, 1, false>(p);
}
