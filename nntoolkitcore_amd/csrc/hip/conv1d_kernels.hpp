// conv1d_kernels.hpp -- the implicit-GEMM Conv1d / dense kernels (exact-f32 MFMA and split-bf16 x 3) and their launcher
// template, shared by conv1d.hip (stride 1) and conv1d_s2.hip (the stride-2 instantiations: a translation unit of their
// own, so that the two sets compile in parallel).  See conv1d.hip for the design notes.
#pragma once
#include "nntk_common.hpp"
#include <stdlib.h>
#include <type_traits>

#define CONV_BM 128
#define CONV_KC 16
#define CONV_LS (CONV_KC + 4)     // LDS row stride in floats: 16-B aligned rows, conflict-free b128 reads

struct ConvParams {
    const float *in;     // [B, T, Cin]: row t of sequence b at in + b * in_seq + t * in_row
    long in_seq, in_row; // element strides (T * Cin, Cin for the plain layout)
    const float *wp;     // [Cout_p][k * Cin_p]   (K-contiguous)
    const float *bias;   // [Cout]
    const float *bn;     // NULL or gamma|beta|mean|var|sd|1/sd, each [Cout] (sd = sqrtf(var + eps), nntk_shim_bn_derive)
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
    int store16;         // Cout % 4 == 0 and out 16-byte aligned: 16-byte stores of channel quads
    int quad;            // which accumulator orientation / epilogue (host choice, see conv_epilogue_rows)
    int f3_nht, f3_nks;  // frag3 output (conv1d_mfma_bf16x3_kernel<.., F3OUT>): row blocks 2 ceil(B / 64), k steps ceil(Cout / 16)
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

// compile-time loop ('#pragma unroll' gives up silently on large bodies; a loop that survives indexes the accumulators dynamically)
template <int LO, int HI, class F>
__device__ __forceinline__ void conv_for(F &&f) {
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        conv_for<LO + 1, HI>(f);
    }
}
#define CONV_EPI_LDS_BYTES 16384      // four wavefronts x 4 KB: the LDS the whole-line epilogue borrows from the (dead) window buffer

// Out-of-range sentinel for a lane's VECTOR offset (the only part of a buffer address the hardware range-checks).
// Every descriptor here is clamped to at most CONV_OOB bytes, so the sentinel is out of range whatever the
// tensor's size: a load returns 0, a store is dropped.  Legitimate vector offsets stay far below it: descriptors
// are based at the tile (or weight matrix) they serve, and the host rejects shapes whose in-tile offsets would not fit.
#define CONV_OOB 0x7ffffff0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_rsrc(const void *base, size_t bytes) {
    const unsigned n = bytes > (size_t)CONV_OOB ? (unsigned)CONV_OOB : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)n, 0x00020000);
}

// ---- epilogue shared by the exact-f32 and the split-bf16 kernels (the C/D layout of a 32x32 MFMA is dtype-independent) ----
// Both kernels feed the WEIGHTS as the MFMA's A operand and the window as B, so the accumulator tile is TRANSPOSED:
// a lane owns ONE output position (tile row l31) and register r holds channel 8 (r >> 2) + 4 kh + (r & 3) of the
// 32-channel tile -- four consecutive channels per register quad, i.e. 16 contiguous bytes of the channels-last output
// row.  The tile leaves with 16 buffer_store_b128 per wavefront instead of 64 buffer_store_b32 (a vector-memory
// instruction costs the same ~25 address-unit cycles whatever its width: the 4-byte stores were 150 us of config 3's
// 490 us on the split path).  Swapping the operands does not change a single product or the order they are summed in.
//
// bias, BatchNorm (batch_norm.c:140-163 op order; sd and 1/sd come precomputed behind the four BatchNorm vectors,
// nntk_shim_bn_derive) and the activation are applied per register quad; the uniform choices (BatchNorm or not, which
// activation, which store path) are taken ONCE outside the loops.  The descriptor is THIS TILE's output rows (based at
// the tile's first row, never longer than CONV_OOB bytes): positions past the sequence's end and padded channels get
// the out-of-range sentinel as their vector offset, so the last row tile needs no separate path.
//
// The per-channel constants come from LDS (conv_stage_constants, written before the K loop's first barrier): fetched
// from global memory inside the epilogue they were eight serial load -> use round trips per tile, a quarter of the
// fused Conv+BN+ReLU time.
// zero the components of a 4-channel piece that lie past the row's last channel (ragged last chunk, 16-byte loads)
__device__ __forceinline__ v4u32_t mask_channels(v4u32_t v, int nvalid) {
    if (nvalid < 4) v.w = 0;
    if (nvalid < 3) v.z = 0;
    if (nvalid < 2) v.y = 0;
    if (nvalid < 1) v.x = 0;
    return v;
}

// a * b + c with TWO roundings (the instructions keep this mode when the helper is inlined)
#pragma clang fp contract(off)
__device__ __forceinline__ float nofma_muladd(float a, float b, float c) { return a * b + c; }
#pragma clang fp contract(fast)

template <int BN>
__device__ __forceinline__ void conv_load_constants(const ConvParams &p, int n0, int tid, float (&c)[6]) {
    c[0] = 0.f; c[1] = 1.f; c[2] = 0.f; c[3] = 0.f; c[4] = 1.f; c[5] = 1.f;      // bias | gamma | beta | mean | sd | 1/sd
    const int col = n0 + tid;
    if (tid < BN && col < p.Cout) {
        if (p.bias) c[0] = p.bias[col];
        if (p.bn) {
            c[1] = p.bn[col]; c[2] = p.bn[p.Cout + col]; c[3] = p.bn[2 * p.Cout + col];
            c[4] = p.bn[4 * p.Cout + col]; c[5] = p.bn[5 * p.Cout + col];
        }
    }
}
template <int BN>
__device__ __forceinline__ void conv_stage_constants(float *cst, int tid, const float (&c)[6]) {
    if (tid < BN) {
#pragma unroll
        for (int a = 0; a < 6; ++a) cst[a * BN + tid] = c[a];
    }
}

// epi != NULL (16 KB of LDS nobody reads any more; the caller has passed a barrier): WHOLE-LINE stores.  A lane owns 16 channels of one
// position, so the direct form's instructions write 32 bytes of 32 different lines -- 12.5 B/clk and CU on this chip against 41 when an
// instruction writes 8 whole lines (tools/micro/tdd_store_pattern.hip); at configs[2] the stores were a third of the kernel
// (profiles/r05_conv_store_ablation.log).  Each 32 x 32 tile takes a trip through 4 KB of the wavefront's own LDS region (row-major, the
// 16-byte quad index XORed with (row >> 1) & 7: writes and reads conflict-free; a wavefront's LDS operations execute in order, no barrier)
// and comes back as lane -> (row 8 s + lane / 8, quad lane % 8).  Same values, same bits.
template <int TM, int TN, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvParams &p, f32x16 (&acc)[TM][TN], int b, int x0, int n0,
                                              int wm, int wn, int l31, int kh, const float *cst, v4u32_t *epi = nullptr) {
    constexpr int BN = WN * TN * 32;
    const size_t row_elems = (size_t)(p.out_mode ? p.B : 1) * p.Cout;           // distance between output rows x, x+1
    const size_t row_bytes = row_elems * 4;
    const size_t obase = (p.out_mode ? (size_t)b * p.Cout : (size_t)b * p.Tout * p.Cout) + (size_t)x0 * row_elems;
    const size_t oend = p.out_mode ? (size_t)p.Tout * p.B * p.Cout : ((size_t)b + 1) * p.Tout * p.Cout;
    const __amdgpu_buffer_rsrc_t rs_out = conv_rsrc(p.out + obase, (oend - obase) * 4);
    const bool vec_store = p.store16 && row_bytes * (size_t)(CONV_BM + 4) < (size_t)CONV_OOB;   // 32-bit offsets reach the tile
    const int rb = (int)row_bytes;

    auto epilogue = [&](auto bn_tag, auto act_tag, auto vec_tag) {
        constexpr bool HAS_BN = decltype(bn_tag)::value;
        constexpr int ACT = decltype(act_tag)::value;            // -1: run-time kind
        constexpr bool VEC = decltype(vec_tag)::value;
        int row_voff[TM];                                        // per-lane: this lane's position inside the tile
        bool row_ok[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int xr = wm * TM * 32 + i * 32 + l31;
            row_ok[i] = x0 + xr < p.Tout;
            row_voff[i] = row_ok[i] ? xr * rb + 16 * kh : CONV_OOB;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cu = n0 + wn * TN * 32 + j * 32 + 8 * g;           // wave-uniform first channel of the quad pair
                const int cb = cu + 4 * kh;                                  // this lane's four channels cb .. cb + 3
                const int cl = (wn * TN + j) * 32 + 8 * g + 4 * kh;           // column inside the workgroup's BN
                const float4 bias4 = *reinterpret_cast<const float4 *>(cst + cl);
                const float bias[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
                float ga[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f}, mu[4] = {0.f, 0.f, 0.f, 0.f},
                      sd[4] = {1.f, 1.f, 1.f, 1.f}, rsd[4] = {1.f, 1.f, 1.f, 1.f};
                if (HAS_BN) {
                    const float4 t1 = *reinterpret_cast<const float4 *>(cst + BN + cl);
                    const float4 t2 = *reinterpret_cast<const float4 *>(cst + 2 * BN + cl);
                    const float4 t3 = *reinterpret_cast<const float4 *>(cst + 3 * BN + cl);
                    const float4 t4 = *reinterpret_cast<const float4 *>(cst + 4 * BN + cl);
                    const float4 t5 = *reinterpret_cast<const float4 *>(cst + 5 * BN + cl);
                    ga[0] = t1.x; ga[1] = t1.y; ga[2] = t1.z; ga[3] = t1.w;
                    be[0] = t2.x; be[1] = t2.y; be[2] = t2.z; be[3] = t2.w;
                    mu[0] = t3.x; mu[1] = t3.y; mu[2] = t3.z; mu[3] = t3.w;
                    sd[0] = t4.x; sd[1] = t4.y; sd[2] = t4.z; sd[3] = t4.w;
                    rsd[0] = t5.x; rsd[1] = t5.y; rsd[2] = t5.z; rsd[3] = t5.w;
                }
                const bool col_ok = cb < p.Cout;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j][4 * g + e] + bias[e];
                        if (HAS_BN) {
                            // (v - mu) / sd with the quotient refined by one FMA step: the correctly rounded
                            // quotient without the 11-instruction IEEE division sequence
                            const float d = v[e] - mu[e];
                            float qn = d * rsd[e];
                            qn = fmaf(fmaf(-qn, sd[e], d), rsd[e], qn);
                            // separately rounded multiply and add, as the reference's op_vec_mul / op_vec_add (and the
                            // standalone BatchNorm kernel): the fused result equals the conv -> BatchNorm chain bit for bit
                            v[e] = p.bn_fast ? (d * rsd[e]) * ga[e] + be[e] : nofma_muladd(qn, ga[e], be[e]);
                        }
                        v[e] = ACT == -1 ? nntk_act(p.act_kind, v[e], p.relu_a)
                             : ACT == NNTK_ACT_RELU ? nntk_act(NNTK_ACT_RELU, v[e], p.relu_a)
                             : v[e];
                    }
                    if (VEC) {
                        const v4u32_t pk = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                        // soffset stays IMMEDIATE on purpose: with an SGPR soffset the compiler inserts no wait state
                        // before the next VALU write of these data registers (it believes that form has no hazard) and
                        // MI355X then stores the overwritten value in some lanes (tools/check_store_hazard.py)
#ifdef CONV_EPI_NOSTORE      // timing experiment (WRONG results): the epilogue's arithmetic without its stores
                        if (v[0] == 12345.678f)
#endif
                        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, col_ok ? row_voff[i] + cu * 4 : CONV_OOB, 0, 0);
                    } else if (row_ok[i]) {
                        const int x = x0 + wm * TM * 32 + i * 32 + l31;
                        const size_t orow = p.out_mode ? ((size_t)x * p.B + b) : ((size_t)b * p.Tout + x);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (cb + e < p.Cout) p.out[orow * p.Cout + cb + e] = v[e];
                    }
                }
            }
        }
    };
    auto epilogue_lds = [&](auto bn_tag, auto act_tag) {
        constexpr bool HAS_BN = decltype(bn_tag)::value;
        constexpr int ACT = decltype(act_tag)::value;
        int le = kh * 32 + l31;
        asm volatile("" : "+v"(le));                             // (lane-derived values are derived HERE, not above the K loop)
        v4u32_t *buf = epi + (wm * WN + wn) * 256;
        const int wr = (le & 31) * 8, wr_sw = ((le & 31) >> 1) & 7, khe = le >> 5;
        const int rd_r = le >> 3, rd_q = le & 7;
        conv_for<0, TM>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            conv_for<0, TN>([&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = (wn * TN + j) * 32 + 8 * g + 4 * khe;
                    const float4 bias4 = *reinterpret_cast<const float4 *>(cst + cl);
                    const float bias[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
                    float ga[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f}, mu[4] = {0.f, 0.f, 0.f, 0.f},
                          sd[4] = {1.f, 1.f, 1.f, 1.f}, rsd[4] = {1.f, 1.f, 1.f, 1.f};
                    if (HAS_BN) {
                        const float4 t1 = *reinterpret_cast<const float4 *>(cst + BN + cl);
                        const float4 t2 = *reinterpret_cast<const float4 *>(cst + 2 * BN + cl);
                        const float4 t3 = *reinterpret_cast<const float4 *>(cst + 3 * BN + cl);
                        const float4 t4 = *reinterpret_cast<const float4 *>(cst + 4 * BN + cl);
                        const float4 t5 = *reinterpret_cast<const float4 *>(cst + 5 * BN + cl);
                        ga[0] = t1.x; ga[1] = t1.y; ga[2] = t1.z; ga[3] = t1.w;
                        be[0] = t2.x; be[1] = t2.y; be[2] = t2.z; be[3] = t2.w;
                        mu[0] = t3.x; mu[1] = t3.y; mu[2] = t3.z; mu[3] = t3.w;
                        sd[0] = t4.x; sd[1] = t4.y; sd[2] = t4.z; sd[3] = t4.w;
                        rsd[0] = t5.x; rsd[1] = t5.y; rsd[2] = t5.z; rsd[3] = t5.w;
                    }
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j][4 * g + e] + bias[e];
                        if (HAS_BN) {                            // (the expressions of the direct form)
                            const float d = v[e] - mu[e];
                            float qn = d * rsd[e];
                            qn = fmaf(fmaf(-qn, sd[e], d), rsd[e], qn);
                            v[e] = p.bn_fast ? (d * rsd[e]) * ga[e] + be[e] : nofma_muladd(qn, ga[e], be[e]);
                        }
                        v[e] = ACT == -1 ? nntk_act(p.act_kind, v[e], p.relu_a)
                             : ACT == NNTK_ACT_RELU ? nntk_act(NNTK_ACT_RELU, v[e], p.relu_a)
                             : v[e];
                    }
                    buf[wr + ((2 * g + khe) ^ wr_sw)] = (v4u32_t){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int row = 8 * s + rd_r;
                    const v4u32_t x = buf[row * 8 + (rd_q ^ ((row >> 1) & 7))];
                    const int xr = wm * TM * 32 + i * 32 + row;
                    const int c = n0 + (wn * TN + j) * 32 + 4 * rd_q;
                    __builtin_amdgcn_raw_buffer_store_b128(x, rs_out, (x0 + xr < p.Tout && c < p.Cout) ? xr * rb + c * 4 : CONV_OOB, 0, 0);
                }
            });
        });
    };
    using T_ = std::true_type; using F_ = std::false_type;
    using AId = std::integral_constant<int, NNTK_ACT_IDENTITY>;
    using ARelu = std::integral_constant<int, NNTK_ACT_RELU>;
    using AAny = std::integral_constant<int, -1>;
    if (vec_store && epi) {
        if (!p.bn && p.act_kind == NNTK_ACT_IDENTITY) epilogue_lds(F_{}, AId{});
        else if (p.bn && p.act_kind == NNTK_ACT_RELU) epilogue_lds(T_{}, ARelu{});
        else if (p.bn)                                epilogue_lds(T_{}, AAny{});
        else                                          epilogue_lds(F_{}, AAny{});
    } else if (vec_store) {
        if (!p.bn && p.act_kind == NNTK_ACT_IDENTITY) epilogue(F_{}, AId{}, T_{});
        else if (p.bn && p.act_kind == NNTK_ACT_RELU) epilogue(T_{}, ARelu{}, T_{});
        else if (p.bn)                                epilogue(T_{}, AAny{}, T_{});
        else                                          epilogue(F_{}, AAny{}, T_{});
    } else {
        if (p.bn) epilogue(T_{}, AAny{}, F_{});
        else      epilogue(F_{}, AAny{}, F_{});
    }
}

// The other orientation (window as the MFMA's A operand): a lane owns one CHANNEL and register r holds row
// (r & 3) + 8 (r >> 2) + 4 kh, stored with 64 buffer_store_b32 per wavefront, each writing two rows x 128 contiguous
// bytes.  Kept for the epilogue-heavy time-major input projection (K = 128, rows 8 KB apart), where the quad form's 16
// stores touch 32 rows x 32 bytes each (four times as many cache lines per tile) and measured 1 % slower (LSTM phase
// 10.94 vs 11.05 ms, tools/ab.py stack conv_store=0,1); the quad form is the faster one everywhere else (config 3:
// 0.62 -> 0.55 ms exact, 0.50 -> 0.41 ms split; TimeDistributedDense 2.82 -> 2.74 ms).  Option conv_store forces either.
// Identical values either way: swapping the MFMA's operands changes no product and no summation order.
template <int TM, int TN, int WN>
__device__ __forceinline__ void conv_epilogue_rows(const ConvParams &p, f32x16 (&acc)[TM][TN], int b, int x0, int n0,
                                                   int wm, int wn, int l31, int kh, const float *cst) {
    constexpr int BN = WN * TN * 32;
    const size_t row_elems = (size_t)(p.out_mode ? p.B : 1) * p.Cout;
    const size_t row_bytes = row_elems * 4;
    const size_t obase = (p.out_mode ? (size_t)b * p.Cout : (size_t)b * p.Tout * p.Cout) + (size_t)x0 * row_elems;
    const size_t oend = p.out_mode ? (size_t)p.Tout * p.B * p.Cout : ((size_t)b + 1) * p.Tout * p.Cout;
    const __amdgpu_buffer_rsrc_t rs_out = conv_rsrc(p.out + obase, (oend - obase) * 4);
    const bool fast_store = x0 + CONV_BM <= p.Tout && row_bytes * (size_t)(CONV_BM + 4) < (size_t)CONV_OOB;
    const int rb = (int)row_bytes;
    auto epilogue = [&](auto bn_tag, auto act_tag, auto fast_tag) {
        constexpr bool HAS_BN = decltype(bn_tag)::value;
        constexpr int ACT = decltype(act_tag)::value;
        constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cl = (wn * TN + j) * 32 + l31;
            const int o = n0 + cl;
            const bool col_ok = o < p.Cout;
            const float bias = cst[cl];
            const float g = cst[BN + cl], be = cst[2 * BN + cl], mu = cst[3 * BN + cl], sd = cst[4 * BN + cl], rsd = cst[5 * BN + cl];
            const int voff = col_ok ? o * 4 + kh * 4 * rb : CONV_OOB;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int xr = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][r] + bias;
                    if (HAS_BN) {
                        const float d = v - mu;
                        float qn = d * rsd;
                        qn = fmaf(fmaf(-qn, sd, d), rsd, qn);
                        v = p.bn_fast ? (d * rsd) * g + be : nofma_muladd(qn, g, be);
                    }
                    v = ACT == -1 ? nntk_act(p.act_kind, v, p.relu_a)
                      : ACT == NNTK_ACT_RELU ? nntk_act(NNTK_ACT_RELU, v, p.relu_a)
                      : v;
                    if (FAST) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, voff, xr * rb, 0);
                    } else {
                        const int x = x0 + xr + 4 * kh;
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
    if (fast_store) {
        if (!p.bn && p.act_kind == NNTK_ACT_IDENTITY) epilogue(F_{}, AId{}, T_{});
        else if (p.bn && p.act_kind == NNTK_ACT_RELU) epilogue(T_{}, ARelu{}, T_{});
        else if (p.bn)                                epilogue(T_{}, AAny{}, T_{});
        else                                          epilogue(F_{}, AAny{}, T_{});
    } else {
        if (p.bn) epilogue(T_{}, AAny{}, F_{});
        else      epilogue(F_{}, AAny{}, F_{});
    }
}

// ---- frag3 epilogue (F3OUT): the layer output leaves as the NEXT layer's MFMA operand ---------------------------------------------
// out = frag3 tensor [Tout][2 ceil(B / 64)][ceil(Cout / 16)][3] blocks of 1 KB (frag3.hip): block = 32 batch rows x 16 channels, lane
// 32 kh' + n = 8 consecutive bf16 channels of batch row n.  A row block needs 32 UTTERANCES of one timestep, so the tile of this mode is
// 8 utterances x 16 timesteps (tile row m = 8 tl + bl) instead of 128 timesteps of one utterance: eight rows of one block are then 128
// contiguous bytes = one cache line, filled by this workgroup alone.  Quad orientation: a lane owns position (bl, tl) and, per 16-channel
// k step, channels 4 kh .. 4 kh + 3 of BOTH 8-channel slots; v_permlane32_swap (gfx950) trades the half it does not store against the
// half it lacks -- upper lanes' slot-0 registers <-> lower lanes' slot-1 registers -- after which lane 32 kh' + l31 holds exactly the 16
// bytes of fragment lane (kh', row): one 16-byte store per k step and image, no LDS.  Values are bias + BatchNorm + activation as in
// conv_epilogue, split as frag3_pack_kernel splits them (same instructions, fp contract off): the tensor equals
// conv -> nntk_frag3_pack_device BIT FOR BIT (tests/test_gpu_conv_frag3.py).  Padding channels are zeros; padding rows are not written.
typedef __bf16 cf3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float cf3_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cf3_cvt_pk(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((cf3_f32x2){a, b}, cf3_bf16x2));
}
__device__ __forceinline__ void cf3_split_pair(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
#pragma clang fp contract(off)    // residuals of the ROUNDED value (frag3.hip f3_split_pair)
    hi = cf3_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    mid = cf3_cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
    lo = cf3_cvt_pk(s0, s1);
}

#define CONV_F3_BB 8              // utterances per tile
#define CONV_F3_TT 16             // timesteps per tile
// tile row m -> (utterance, timestep): the utterance is the FAST index, so the 64 lanes of a store instruction (after the half-slot swap:
// 2 slots x 4 timesteps x 8 utterances x 16 bytes) write 8 WHOLE 128-byte lines of the frag3 tensor instead of 32 bytes of 32 lines
#ifndef CONV_F3_TMAJOR
#define CONV_F3_BL(m) ((m) & 7)
#define CONV_F3_TL(m) ((m) >> 3)
#else                             // A/B: the first version (16 timesteps of one utterance per half wavefront)
#define CONV_F3_BL(m) ((m) >> 4)
#define CONV_F3_TL(m) ((m) & 15)
#endif

template <int TM, int TN, int WN>
__device__ __forceinline__ void conv_epilogue_frag3(const ConvParams &p, f32x16 (&acc)[TM][TN], int b0, int x0, int n0,
                                                    int wm, int wn, int l31, int kh, const float *cst) {
    constexpr int BN = WN * TN * 32;
    const size_t total = (size_t)p.Tout * p.f3_nht * p.f3_nks * 3 * 1024;          // host: < CONV_OOB
    const __amdgpu_buffer_rsrc_t rs_out = conv_rsrc(p.out, total);
    auto epilogue = [&](auto bn_tag, auto act_tag) {
        constexpr bool HAS_BN = decltype(bn_tag)::value;
        constexpr int ACT = decltype(act_tag)::value;
        int row_voff[TM];                        // byte offset of this lane's 16-byte slot inside block (t, row block, k step 0, image 0)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = (wm * TM + i) * 32 + l31;
            const int b = b0 + CONV_F3_BL(m), x = x0 + CONV_F3_TL(m);
            row_voff[i] = (b < p.B && x < p.Tout) ? (int)((((size_t)x * p.f3_nht + (b >> 5)) * p.f3_nks * 3) * 1024) + (32 * kh + (b & 31)) * 16
                                                  : CONV_OOB;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {                                     // 16-channel k step of the 32-channel tile
                const int cu = n0 + (wn * TN + j) * 32 + 16 * gp;                // wave-uniform first channel of the k step
                const bool ks_ok = cu < p.Cout;                                  // (k steps past ceil(Cout / 16) are not in the tensor)
                const int ks_off = (cu >> 4) * 3 * 1024;
                unsigned img[TM][2][3][2];                                       // [tile][slot][image][channel pair]
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int g = 2 * gp + s;
                    const int cb = cu + 8 * s + 4 * kh;
                    const int cl = (wn * TN + j) * 32 + 8 * g + 4 * kh;
                    const float4 bias4 = *reinterpret_cast<const float4 *>(cst + cl);
                    const float bias[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
                    float ga[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f}, mu[4] = {0.f, 0.f, 0.f, 0.f},
                          sd[4] = {1.f, 1.f, 1.f, 1.f}, rsd[4] = {1.f, 1.f, 1.f, 1.f};
                    if (HAS_BN) {
                        const float4 t1 = *reinterpret_cast<const float4 *>(cst + BN + cl);
                        const float4 t2 = *reinterpret_cast<const float4 *>(cst + 2 * BN + cl);
                        const float4 t3 = *reinterpret_cast<const float4 *>(cst + 3 * BN + cl);
                        const float4 t4 = *reinterpret_cast<const float4 *>(cst + 4 * BN + cl);
                        const float4 t5 = *reinterpret_cast<const float4 *>(cst + 5 * BN + cl);
                        ga[0] = t1.x; ga[1] = t1.y; ga[2] = t1.z; ga[3] = t1.w;
                        be[0] = t2.x; be[1] = t2.y; be[2] = t2.z; be[3] = t2.w;
                        mu[0] = t3.x; mu[1] = t3.y; mu[2] = t3.z; mu[3] = t3.w;
                        sd[0] = t4.x; sd[1] = t4.y; sd[2] = t4.z; sd[3] = t4.w;
                        rsd[0] = t5.x; rsd[1] = t5.y; rsd[2] = t5.z; rsd[3] = t5.w;
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[i][j][4 * g + e] + bias[e];
                            if (HAS_BN) {                                        // (same expressions as conv_epilogue)
                                const float d = v[e] - mu[e];
                                float qn = d * rsd[e];
                                qn = fmaf(fmaf(-qn, sd[e], d), rsd[e], qn);
                                v[e] = p.bn_fast ? (d * rsd[e]) * ga[e] + be[e] : nofma_muladd(qn, ga[e], be[e]);
                            }
                            v[e] = ACT == -1 ? nntk_act(p.act_kind, v[e], p.relu_a)
                                 : ACT == NNTK_ACT_RELU ? nntk_act(NNTK_ACT_RELU, v[e], p.relu_a)
                                 : v[e];
                            if (cb + e >= p.Cout) v[e] = 0.0f;                   // padding channels of the last k step
                        }
                        cf3_split_pair(v[0], v[1], img[i][s][0][0], img[i][s][1][0], img[i][s][2][0]);
                        cf3_split_pair(v[2], v[3], img[i][s][0][1], img[i][s][1][1], img[i][s][2][1]);
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int im = 0; im < 3; ++im) {
                        // upper lanes' slot-0 halves <-> lower lanes' slot-1 halves: lane 32 kh' + l31 then holds slot kh' whole
                        const auto s01 = __builtin_amdgcn_permlane32_swap(img[i][0][im][0], img[i][1][im][0], false, false);
                        const auto s23 = __builtin_amdgcn_permlane32_swap(img[i][0][im][1], img[i][1][im][1], false, false);
                        const v4u32_t pk = {s01[0], s23[0], s01[1], s23[1]};
                        // (offsets ride in the VECTOR offset: the immediate field is too short and an SGPR soffset trips the
                        // store hazard of conv_epilogue's note)
                        const int vo = (ks_ok && row_voff[i] != CONV_OOB) ? row_voff[i] + ks_off + im * 1024 : CONV_OOB;
                        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, vo, 0, 0);
                    }
                }
            }
        }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    using AId = std::integral_constant<int, NNTK_ACT_IDENTITY>;
    using ARelu = std::integral_constant<int, NNTK_ACT_RELU>;
    using AAny = std::integral_constant<int, -1>;
    if (!p.bn && p.act_kind == NNTK_ACT_IDENTITY) epilogue(F_{}, AId{});
    else if (p.bn && p.act_kind == NNTK_ACT_RELU) epilogue(T_{}, ARelu{});
    else if (p.bn)                                epilogue(T_{}, AAny{});
    else                                          epilogue(F_{}, AAny{});
}

// WM x WN wavefronts, each computing TM x TN MFMA tiles of 32x32.
// A4 = the window is fetched with 16-byte loads (always, unless option conv_a4 = 0 and the rows are not 16-byte multiples).
// QUAD: weights as the MFMA's A operand and the 16-byte quad epilogue; else window as A and the row epilogue.
// AMAX = window rows the register staging covers: 192 (stride 1, k <= 65) or 320 (stride 2: (128 - 1) * 2 + k rows).  Its own
// instantiation because the extra, mostly idle load passes of a 320-row budget cost the stride-1 shapes 4-12 % (measured in r02).
template <int WM, int WN, int TM, int TN, bool A4, bool QUAD, int AMAX = 192>
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
    constexpr int A_PT = (AMAX + AR_STEP - 1) / AR_STEP;   // window: rows_a <= AMAX (host checks)
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
    float *cst = smem + 2 * a_elems + 2 * w_elems;     // [6][BN] epilogue constants
    {
        float c[6];
        conv_load_constants<BN>(p, n0, tid, c);
        conv_stage_constants<BN>(cst, tid, c);           // visible to the epilogue: at least one barrier follows
    }
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
                    // 16-byte loads of a row whose channel count is not a multiple of 4 pick up the next row's first
                    // channels in the ragged last chunk: zeroed here (their weights are zero padding, but 0 x NaN is not 0)
                    if (A4) *reinterpret_cast<v4u32_t *>(As + r * LS + ac) =
                                (cin_ragged && cc == n_cchunks - 1) ? mask_channels(areg4[q], p.Cin - (cc * KC + ac)) : areg4[q];
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
                        acc[i][j] = QUAD ? __builtin_amdgcn_mfma_f32_32x32x2f32(wv, av, acc[i][j], 0, 0, 0)    // D = W x window^T
                                         : __builtin_amdgcn_mfma_f32_32x32x2f32(av, wv, acc[i][j], 0, 0, 0);
                    }
        };
        if (!CONV_DBG(2)) {
        block8(0);
        // the last channel chunk may hold only one 8-deep block of real channels (the rest is zero padding)
        if (cc != n_cchunks - 1 || last_blocks > 1) block8(8);
        }
        cc = ncc; kk = nkk;
    }

    if (CONV_DBG(1) && acc[0][0][0] != 12345.678f) return;
    if constexpr (QUAD) conv_epilogue<TM, TN, WN>(p, acc, b, x0, n0, wm, wn, l31, kh, cst);
    else conv_epilogue_rows<TM, TN, WN>(p, acc, b, x0, n0, wm, wn, l31, kh, cst);
}

// ---------------------------------------------------------------------------------------------------------------
// Option gemm_split_bf16: the same implicit GEMM on the bf16 MFMA with every f32 operand split into three bf16
// terms, x = hi + mid + lo EXACTLY (8 + 8 + 8 significand bits; each remainder is exact in f32), and the six
// products of weight <= 2 accumulated in f32:  x*y ~ hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi.  The dropped
// terms (mid*lo, lo*mid, lo*lo) are <= 2^-23 |x*y|, the size of ONE f32 rounding, and bf16 x bf16 products are exact in
// the f32 accumulator -- so this is an f32-accuracy contraction, but NOT the k-ordered fmaf chain of the exact path
// (results differ in the last bits; against a float64 contraction it measures slightly CLOSER than the exact chain,
// tools/split_error.py).  6 x v_mfma_f32_32x32x16_bf16 (32 cycles each) replace 8 x v_mfma_f32_32x32x2_f32
// (64 cycles each) per 16-deep block: 2.67x less MFMA time.
//
// Weights are split ONCE at upload (nntk_shim_split_bf16x3 writes the three images behind the packed f32 matrix) and
// stored in MFMA FRAGMENT ORDER: one 1 KB block per (32-column tile, 16-deep k step) holding [k half][column][8 bf16],
// so a block moves HBM/L2 -> LDS -> registers as 64 lanes x 16 bytes in lane order: coalesced, conflict-free, no
// address arithmetic.  The window is split while it is staged (11 VALU per two elements, once per channel chunk and
// reused by all k taps).  Window LDS rows are [row][image 0..2][16 bf16] = 96 bytes with the two 16-byte halves of an
// image swapped on rows with bit 3 set: row stride 6 slots (of 16 B) + that swap makes the 16 lanes of a ds_read_b128
// group hit 16 distinct slots (rows {0-3, 12-15, 20-27} -> slots {0,6,12,2, 9,15,5,11, 8,14,4,10, 1,7,13,3}).
//
// What bounds it (tools/conv_probe.py on the -DNNTK_CONV_DBG build): with the MFMA time cut to a third the
// kernel is bound by the weight re-reads out of L2 -- every 128-row tile streams its whole [BN x K] weight slab again
// (config 3: 7,969 tiles x 184 KB = 1.5 GB against 0.69 GB of HBM traffic), and L2 serves shared data at ~30 B/clk/CU
// (MI355X_MICROARCH.md "Indexed rows").  Measured and NOT kept: weight fragments loaded straight from L2 into
// registers by every wave (no LDS round trip, one barrier per window chunk instead of one per tap: twice the L2
// traffic, 3-25 % slower on the dense GEMMs) and persistent workgroups with the next tile's first loads under the
// epilogue (+7 %: launch latency was not the limit); 256 x 128 tiles (4 wavefronts of 64 x 128, 243 VGPRs, two per SIMD:
// half the weight stream per output, and 2-5 % SLOWER on all four shapes -- so the L2 stream is not the whole story);
// weight loads issued TWO steps ahead through a second register set (150 VGPRs: 343 vs 338-352 us, no change).  PMC
// (tools/conv_pmc.sh): MFMA busy 44 %, no LDS bank conflicts, VALU 10 % of wave-cycles.  What is left untried is a
// deeper K step per barrier (32-64 deep instead of 16: MI355X guide section 5 prices that at +7-16 % on a bf16 GEMM).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {     // RNE, a in the low half
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
    hi = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    mid = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
    lo = cvt_pk_bf16(s0, s1);
}

#define SPLIT_ROW 96              // LDS bytes per window row: 3 images x 16 bf16

// F3OUT: the tile is 8 utterances x 16 timesteps (stride 1; the window = 8 runs of 16 + k - 1 rows) and the epilogue writes a frag3 tensor
// (conv_epilogue_frag3); everything between staging and epilogue is the same instruction stream.
template <int WM, int WN, int TM, int TN, bool A4, bool QUAD, int AMAX = 192, bool F3OUT = false>
__global__ __launch_bounds__(256, 2) void conv1d_mfma_bf16x3_kernel(ConvParams p) {
    constexpr int BN = WN * TN * 32;
    constexpr int KC = CONV_KC;
    static_assert(WM * WN == 4 && WM * TM * 32 == CONV_BM, "4 wavefronts, BM = 128");
    static_assert(!F3OUT || QUAD, "the frag3 epilogue is written for the quad orientation");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char *lds = reinterpret_cast<char *>(smem);
    // LDS carve (bytes): window [2][rows_a][96] | weights [2][BN / 32 column tiles][3 images][1 KB fragment block]
    const int a_bytes = p.rows_a * SPLIT_ROW;
    constexpr int w_bytes = (BN / 32) * 3 * 1024;
    const int w_base = 2 * a_bytes;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;

    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;          // same XCD-aware tile order as the f32 kernel
#ifndef CONV_F3_INTERLEAVED
    // F3OUT: an XCD walks a CONTIGUOUS run of tiles -- neighbours in time are neighbouring 16-timestep blocks of the same eight utterances,
    // whose windows share k - 1 of 15 + k rows (a quarter of the input at k = 5): those re-reads hit the XCD's own L2
    const int tile = F3OUT ? xcd * ((p.m_tiles + 7) >> 3) + local / p.n_tiles : (local / p.n_tiles) * 8 + xcd;
#else
    const int tile = (local / p.n_tiles) * 8 + xcd;
#endif
    if (tile >= p.m_tiles) return;
    // F3OUT: b = first utterance of the tile's eight, x0 = first of its 16 timesteps
    const int b = F3OUT ? (tile / p.tiles_per_seq) * CONV_F3_BB : tile / p.tiles_per_seq;
    const int x0 = (tile % p.tiles_per_seq) * (F3OUT ? CONV_F3_TT : CONV_BM);
    const int n0 = (local % p.n_tiles) * BN;
    const int Ktot = p.k * p.Cin_p;
    const int ksteps = Ktot >> 4, cin_steps = p.Cin_p >> 4;
    const int wrows = CONV_F3_TT + p.k - 1;               // F3OUT: window rows per utterance

    f32x16 acc[TM][TN];

    constexpr int A_TPR = A4 ? KC / 4 : KC;
    constexpr int AR_STEP = 256 / A_TPR;
    constexpr int A_PT = (AMAX + AR_STEP - 1) / AR_STEP;
    const int ac = A4 ? (tid % A_TPR) * 4 : (tid % A_TPR);
    const int ar = tid / A_TPR;
    v4u32_t areg4[A4 ? A_PT : 1];
    unsigned areg[A4 ? 1 : A_PT];
    constexpr int W_CT = (BN / 32 + 3) / 4;              // column tiles each wavefront stages (2 for the 256-wide tile)
    v4u32_t wreg[W_CT][3];
    int a_voff[A_PT];
#pragma unroll
    for (int q = 0; q < A_PT; ++q) {
        const int r = ar + q * AR_STEP;
        if (F3OUT) {      // window row r = utterance r / wrows, its row r % wrows (rows past the utterance's end feed unstored positions only)
            const int bl = r / wrows, tr = r - bl * wrows;
            a_voff[q] = (r < p.rows_a && b + bl < p.B) ? (int)(((long)bl * p.in_seq + (long)tr * p.in_row + ac) * 4) : CONV_OOB;
        } else {
            a_voff[q] = (int)(((long)r * p.in_row + ac) * 4);
        }
    }
    const bool cin_ragged = (p.Cin % KC) != 0;
    const int n_cchunks = p.Cin_p / KC;
    const int n_chunks = n_cchunks * p.k;

    // weights: wavefront w stages column tile w of the workgroup's BN / 32 (one 1 KB block per image and step)
    const size_t w_elems_total = (size_t)p.Cout_p * Ktot;
    const int img_bytes = (int)(w_elems_total * 2);
    const __amdgpu_buffer_rsrc_t rs_w = conv_rsrc(p.wp + w_elems_total, (size_t)3 * img_bytes);
    const bool w_thread = wave < BN / 32;
    int w_voff[W_CT];
#pragma unroll
    for (int c = 0; c < W_CT; ++c)
        w_voff[c] = wave + 4 * c < BN / 32 ? (((n0 >> 5) + wave + 4 * c) * ksteps) * 1024 + lane * 16 : CONV_OOB;
    auto load_w = [&](int cc, int kk) {
        const int soff = (kk * cin_steps + cc) * 1024;
#pragma unroll
        for (int c = 0; c < W_CT; ++c)
#pragma unroll
            for (int m = 0; m < 3; ++m) wreg[c][m] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_voff[c], soff + m * img_bytes, 0);
    };
    const size_t in_total = (size_t)p.B * p.T * p.Cin;
    const size_t in_off = (size_t)b * p.in_seq + (size_t)x0 * p.stride * p.in_row;
    const __amdgpu_buffer_rsrc_t rs_in = conv_rsrc(p.in + in_off, (in_total - in_off) * 4);
    auto load_a = [&](int cc) {
        const bool ch_ok = !(cin_ragged && cc == n_cchunks - 1) || cc * KC + ac < p.Cin;
#pragma unroll
        for (int q = 0; q < A_PT; ++q) {
            const int vo = ch_ok ? a_voff[q] : CONV_OOB;
            if (A4) areg4[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, cc * KC * 4, 0);
            else    areg[q] = __builtin_amdgcn_raw_buffer_load_b32(rs_in, vo, cc * KC * 4, 0);
        }
    };
    auto stage_a = [&](char *As, int nvalid) {        // registers -> three bf16 images in LDS (nvalid: channels left in the row)
#pragma unroll
        for (int q = 0; q < A_PT; q += (A4 ? 1 : 2)) {
            const int r = ar + q * AR_STEP;
            if (A4) {
                if (r < p.rows_a) {
                    const v4u32_t x = nvalid < 4 ? mask_channels(areg4[q], nvalid) : areg4[q];
                    unsigned h0, m0, l0, h1, m1, l1;
                    split3_pair(__uint_as_float(x.x), __uint_as_float(x.y), h0, m0, l0);
                    split3_pair(__uint_as_float(x.z), __uint_as_float(x.w), h1, m1, l1);
                    char *dst = As + r * SPLIT_ROW + 16 * ((ac >> 3) ^ ((r >> 3) & 1)) + (ac & 7) * 2;
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(dst + 32) = make_uint2(m0, m1);
                    *reinterpret_cast<uint2 *>(dst + 64) = make_uint2(l0, l1);
                }
            } else {
                // one element per thread and pass: split two passes' elements together (rows r and r + AR_STEP)
                unsigned h, m, l;
                split3_pair(__uint_as_float(areg[q]), __uint_as_float(areg[q + 1 < A_PT ? q + 1 : q]), h, m, l);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int re = r + e * AR_STEP;
                    if (q + e < A_PT && re < p.rows_a) {
                        char *dst = As + re * SPLIT_ROW + 16 * ((ac >> 3) ^ ((re >> 3) & 1)) + (ac & 7) * 2;
                        *reinterpret_cast<unsigned short *>(dst) = (unsigned short)(e ? h >> 16 : h);
                        *reinterpret_cast<unsigned short *>(dst + 32) = (unsigned short)(e ? m >> 16 : m);
                        *reinterpret_cast<unsigned short *>(dst + 64) = (unsigned short)(e ? l >> 16 : l);
                    }
                }
            }
        }
    };

    const int a_row0 = (wm * TM * 32 + l31) * p.stride;       // window row of tile i at tap kk: a_row0 + i * 32 * stride + kk
    int a_row_f3[TM];                                         // F3OUT: tile row m = 16 bl + tl sits at window row bl * wrows + tl (+ kk)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = (wm * TM + i) * 32 + l31;
        a_row_f3[i] = CONV_F3_BL(m) * wrows + CONV_F3_TL(m);
    }
    const int w_rd = w_base + (wn * TN * 3) * 1024 + lane * 16;     // this wave's first column tile, image 0
    const int w_wr = w_base + (wave * 3) * 1024 + lane * 16;

    load_a(0);
    load_w(0, 0);
    float *cst = reinterpret_cast<float *>(lds + 2 * a_bytes + 2 * w_bytes);     // [6][BN] epilogue constants
    {
        float c[6];
        conv_load_constants<BN>(p, n0, tid, c);
        conv_stage_constants<BN>(cst, tid, c);           // visible to the epilogue: at least one barrier follows
    }
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
        if (kk == 0 && !CONV_DBG(16)) stage_a(lds + abuf * a_bytes, (cin_ragged && cc == n_cchunks - 1) ? p.Cin - (cc * KC + ac) : 4);
        if (w_thread) {
#pragma unroll
            for (int c = 0; c < W_CT; ++c) {
                char *dst = lds + wbuf * w_bytes + w_wr + c * 4 * 3 * 1024;
#pragma unroll
                for (int m = 0; m < 3; ++m) *reinterpret_cast<v4u32_t *>(dst + 1024 * m) = wreg[c][m];
            }
        }
        __syncthreads();
        int ncc = cc, nkk = kk + 1;
        if (nkk == p.k) { nkk = 0; ncc = cc + 1; }
        if (chunk + 1 < n_chunks) {
            if (!CONV_DBG(4)) load_w(ncc, nkk);
            if (nkk == 0 && !CONV_DBG(32)) load_a(ncc);
        }
        const char *Ab = lds + abuf * a_bytes;
        const char *Wb = lds + wbuf * w_bytes + w_rd;
        bf16x8_t a[3][TM], w[3][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = F3OUT ? a_row_f3[i] + kk : a_row0 + i * 32 * p.stride + kk;
            const char *src = Ab + row * SPLIT_ROW + 16 * (kh ^ ((row >> 3) & 1));
#pragma unroll
            for (int m = 0; m < 3; ++m)
                if (!CONV_DBG(8)) a[m][i] = *reinterpret_cast<const bf16x8_t *>(src + 32 * m);
                else a[m][i] = __builtin_bit_cast(bf16x8_t, wreg[0][m]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m)
                if (!CONV_DBG(8)) w[m][j] = *reinterpret_cast<const bf16x8_t *>(Wb + (j * 3 + m) * 1024);
                else w[m][j] = __builtin_bit_cast(bf16x8_t, wreg[0][m]);
        // smallest terms first; the TM x TN accumulators interleave so dependent MFMAs are TM * TN apart
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};      // window image (0 hi, 1 mid, 2 lo)
        constexpr int PW[6] = {0, 2, 1, 0, 1, 0};      // weight image
        if (!CONV_DBG(2)) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = QUAD ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PW[t]][j], a[PA[t]][i], acc[i][j], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][i], w[PW[t]][j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int m = 0; m < 3; ++m)
                        acc[i][j][m] += __builtin_bit_cast(v4u32_t, a[m][i]).x + __builtin_bit_cast(v4u32_t, w[m][j]).x;
        }
        cc = ncc; kk = nkk;
    }
    if (CONV_DBG(1) && acc[0][0][0] != 12345.678f) return;
    if constexpr (F3OUT) conv_epilogue_frag3<TM, TN, WN>(p, acc, b, x0, n0, wm, wn, l31, kh, cst);
    else if constexpr (QUAD) {
        // whole-line stores through the window buffer, which nobody reads once every wavefront has left the K loop (barrier); the
        // epilogue constants sit behind it
        v4u32_t *epi = nullptr;
#ifndef CONV_EPI_DIRECT
        if (2 * a_bytes + 2 * w_bytes >= CONV_EPI_LDS_BYTES && p.out_mode == 0) {
            __syncthreads();
            epi = reinterpret_cast<v4u32_t *>(lds);
        }
#endif
        conv_epilogue<TM, TN, WN>(p, acc, b, x0, n0, wm, wn, l31, kh, cst, epi);
    }
    else conv_epilogue_rows<TM, TN, WN>(p, acc, b, x0, n0, wm, wn, l31, kh, cst);
}

// frag3 output: m_tiles = groups of eight utterances x blocks of 16 timesteps (p.tiles_per_seq = ceil(Tout / 16), set by the caller)
template <int WM, int WN, int TM, int TN, bool A4>
static int launch_mfma_f3(const ConvParams &p) {
    constexpr int BN = WN * TN * 32;
    size_t lds = (size_t)2 * (p.rows_a * SPLIT_ROW + (BN / 32) * 3 * 1024) + 6 * BN * sizeof(float);
    ConvParams q = p;
    q.m_tiles = ((p.B + CONV_F3_BB - 1) / CONV_F3_BB) * p.tiles_per_seq;
    q.n_tiles = p.Cout_p / BN;
    const long blocks = (long)((q.m_tiles + 7) / 8) * 8 * q.n_tiles;
    if ((long)((p.B + CONV_F3_BB - 1) / CONV_F3_BB) * p.tiles_per_seq > 0x7fffffffL / 8 || blocks > 0x7fffffffL)
        return nntk_fail_msg("conv1d: too many tiles for one launch");
    auto kern = conv1d_mfma_bf16x3_kernel<WM, WN, TM, TN, A4, true, 192, true>;
    if (lds > 64 * 1024) {
        if (nntk_set_max_dynamic_lds((const void *)kern, lds)) return -1;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, nntk_stream(), q);
    NNTK_LAUNCH_CHECK("conv1d_mfma_bf16x3_kernel<frag3>");
    nntk_set_last_conv_kernel("conv1d_mfma_bf16x3_kernel<frag3>");
    return 0;
}

template <int WM, int WN, int TM, int TN, bool A4, bool SPLIT, bool QUAD, int AMAX = 192>
static int launch_mfma_o(const ConvParams &p) {
    constexpr int BN = WN * TN * 32;
    // 41 KB at BN = 128 (50 KB split): three workgroups per CU, which is what hides the barrier / staging latency
    size_t lds = SPLIT ? (size_t)2 * (p.rows_a * SPLIT_ROW + (BN / 32) * 3 * 1024)
                       : (size_t)(2 * p.rows_a * CONV_LS + 2 * BN * CONV_LS) * sizeof(float);
    lds += 6 * BN * sizeof(float);                    // epilogue constants
    ConvParams q = p;
    q.m_tiles = p.B * p.tiles_per_seq;
    q.n_tiles = p.Cout_p / BN;
    const long blocks = (long)((q.m_tiles + 7) / 8) * 8 * q.n_tiles;
    if ((long)p.B * p.tiles_per_seq > 0x7fffffffL / 8 || blocks > 0x7fffffffL)
        return nntk_fail_msg("conv1d: too many tiles for one launch");
    auto kern = SPLIT ? conv1d_mfma_bf16x3_kernel<WM, WN, TM, TN, A4, QUAD, AMAX> : conv1d_mfma_kernel<WM, WN, TM, TN, A4, QUAD, AMAX>;
    if (lds > 64 * 1024) {
        if (nntk_set_max_dynamic_lds((const void *)kern, lds)) return -1;
    }
    dim3 grid((unsigned)blocks);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, nntk_stream(), q);
    NNTK_LAUNCH_CHECK("conv1d_mfma_kernel");
    nntk_set_last_conv_kernel(SPLIT ? "conv1d_mfma_bf16x3_kernel" : "conv1d_mfma_kernel");
    return 0;
}


// stride-2 launches live in conv1d_s2.hip
int nntk_conv1d_launch_s2(const ConvParams &p, bool a4, bool split);
