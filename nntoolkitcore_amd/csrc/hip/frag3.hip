// frag3.hip -- activation tensors in FRAG3 form, and the dense GEMM that consumes them.
//
// The split-bf16 x 3 contraction (conv1d_kernels.hpp) multiplies f32 operands as three bf16 images, x = hi + mid + lo exactly.
// Weights are split once at upload and stored in MFMA fragment order.  Round 3's kernels split the ACTIVATIONS inside every
// consumer: the conv / dense kernels while staging the window through LDS (11 VALU instructions per two elements), the
// register-resident recurrent kernels (recurrent_rr.hip) in their x path -- whose f32 requests touch 32 rows of 16 bytes each per
// instruction and were 0.5 us of a 6.7 us LSTM step -- while the recurrent kernels' own hand-off already travelled pre-split in
// fragment order.  A FRAG3 tensor is that hand-off format as a tensor in its own right:
//
//   logical [B][T][C] f32  ->  [T][NHT = 2 ceil(B / 64) row blocks][NKS = ceil(C / 16) k steps][3 images hi, mid, lo] blocks of 1 KB,
//   block = the B fragment of v_mfma_f32_32x32x16_bf16 for 32 batch rows x 16 channels: lane l = 32 kh + n holds, as 8 consecutive
//   bf16, channels 16 ks + 8 kh .. + 7 of batch row 32 ht + n at timestep t.  Rows past B and channels past C are zeros.
//
// 6 bytes per element instead of 4, and in exchange every consumer's operand fetch is a run of coalesced 1 KB loads (64 lanes x
// 16 bytes in lane order) straight into MFMA operand registers: no LDS staging, no split, no per-row requests.
//   producers: frag3_pack_kernel (any f32 tensor), lstm_rr_kernel / gru_rr_kernel (their T-deep hand-off IS the layer output);
//   consumers: lstm_rr_kernel / gru_rr_kernel <.., XF> (x operand), dense_frag3_kernel below (TimeDistributedDense / Dense).
//
// dense_frag3_kernel: out[(b, t), :] = act(h[(b, t), :] . W + bias)  (layers/dense.c:122-133, time_distributed_dense.c:52-58) with
// BOTH operands global -> registers: the A side is the frag3 tensor, the W side the split images the upload already keeps in
// fragment order.  No LDS, no barriers: four independent wavefronts (one per SIMD, 512 registers each) own 128 x 128 outputs each
// (16 accumulator tiles = 256 registers) of a 256 x 256 workgroup tile and stream 24 coalesced 1 KB loads per 16-deep k step
// against 96 MFMAs.  Same products in the same order as conv1d_mfma_bf16x3_kernel: bit-identical to TimeDistributedDenseApplyDevice
// on the f32 tensor the frag3 tensor was split from.
//
// FRAG2H (round 5, late): the same tensor as TWO f16 images of a power-of-two multiple, x * 2^15 = hi + lo with hi = f16(x 2^15), lo = f16(x 2^15 - hi):
//   [T][NHT][NKS][2 images hi, lo] blocks of 1 KB in the same lane order (the B fragment of v_mfma_f32_32x32x16_f16 is that of the bf16 form).
//   11 + 11 significand bits plus the residual's sign: |x - (hi + lo) 2^-15| <= 2^-23 |x| -- at worst ONE f32 ulp (just above a power of two),
//   0.3 ulp rms -- instead of exact; and the contraction needs THREE products (lo.hi, hi.lo, hi.hi; dropped: lo.lo at 2^-24, where the bf16 form drops mid.lo, lo.mid,
//   lo.lo) instead of six at the same MFMA rate.  The step is power-limited (DESIGN status, round 5): MFMA work is the currency.
//   Range: f16 ends at 65 504, so the form is for tensors of magnitude < 2 -- the outputs of the recurrent layers with their standard
//   activations (|h| <= 1), which is who produces it (lstm_rr_kernel / gru_rr_kernel's output wave, recurrent_rr.hip) -- and the scale keeps
//   the low image out of f16's subnormals down to |x| ~ 2^-17.  The weights get their own scale 2^q, the largest power of two with
//   max |W| 2^q <= 32 768 (nntk_shim_split_f16x2 at upload); the epilogue multiplies the sums by 2^-(15 + q), exactly.
//   Measured at the stack's TimeDistributedDense (tools/micro/gemm_f16x2.hip, profiles/r05_gemm_f16x2_micro.log): 1.51 ms against 2.36, error
//   against an f64 dot product of the same f32 operands rms 5.6e-8 / max 4.8e-7 -- below the six-product kernel's (7.6e-8 / 7.7e-7) and
//   below the reference's own left-to-right f32 accumulation (8.8e-8 / 8.4e-7, core/default_ops.cc:224-231).
#include "nntk_common.hpp"
#include <type_traits>

typedef unsigned f3_v4u __attribute__((ext_vector_type(4)));
typedef __bf16 f3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 f3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float f3_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f3_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f3_f16x2 __attribute__((ext_vector_type(2)));

#ifndef DF3_MFMA_PER_LOAD
#define DF3_MFMA_PER_LOAD 2
#endif
#ifndef DF3_LDS_EPI
#define DF3_LDS_EPI 1             // dense_frag3_kernel: whole-line stores through an LDS transposition (0: a lane stores its own quads)
#endif
#ifndef DF3_STAGGER
#define DF3_STAGGER 0             // ... first round of workgroups delayed by (CU index) x DF3_STAGGER x 10 ns (s_memrealtime ticks)
#endif
#define F3_OOB 0x70000000          // out-of-range vector offset (every descriptor here is shorter; + a few KB of immediates does not wrap)

// compile-time loop: '#pragma unroll' gives up silently on large bodies ("unrolled size is too large"), and a loop that survives indexes the
// accumulator array dynamically -- hipcc then keeps all 256 accumulators in scratch (1 088 bytes in the first build of the LDS epilogue)
template <int LO, int HI, class F>
__device__ __forceinline__ void f3_for(F &&f) {
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        f3_for<LO + 1, HI>(f);
    }
}

__device__ __forceinline__ unsigned f3_cvt_pk(float a, float b) {       // RNE, a in the low half
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f3_f32x2){a, b}, f3_bf16x2));
}
__device__ __forceinline__ void f3_split_pair(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
#pragma clang fp contract(off)    // the residuals of the ROUNDED x (an inlined x = a * b must not turn x - hi into fma(a, b, -hi); tools/check_rr_waits.py)
    hi = f3_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    mid = f3_cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
    lo = f3_cvt_pk(s0, s1);
}

extern "C" size_t nntk_shim_frag3_floats(int B, int T, int C) {
    if (B <= 0 || T <= 0 || C <= 0) return 0;
    return (size_t)T * ((size_t)(B + 63) / 64 * 2) * (size_t)((C + 15) / 16) * 3 * 256;
}

// ---- f32 [B][T][C] (row b, t at x + b * seq_pitch + t * row_pitch floats) -> frag3 ---------------------------------
// One workgroup per (t, row block): its four wavefronts walk the k steps; a lane reads its 8 channels (two 16-byte pieces when the
// row allows, else element by element) and writes its 16 bytes of each image: three coalesced 1 KB stores per wave and k step.
__global__ __launch_bounds__(256) void frag3_pack_kernel(const float *__restrict__ x, f3_v4u *__restrict__ dst, int B, int T, int C,
                                                         long seq_pitch, long row_pitch, int NHT, int NKS, int vec_ok) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const long rb = blockIdx.x;                       // t * NHT + ht
    const int t = (int)(rb / NHT), ht = (int)(rb % NHT);
    const int b = ht * 32 + n;
    const float *row = x + (size_t)b * seq_pitch + (size_t)t * row_pitch;
    for (int ks = w; ks < NKS; ks += 4) {
        const int c0 = 16 * ks + 8 * kh;
        float v[8];
        if (b < B && vec_ok && c0 + 8 <= C) {
            const float4 p0 = *reinterpret_cast<const float4 *>(row + c0), p1 = *reinterpret_cast<const float4 *>(row + c0 + 4);
            v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w; v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (b < B && c0 + q < C) ? row[c0 + q] : 0.0f;
        }
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f3_split_pair(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
        f3_v4u *d = dst + (((size_t)rb * NKS + ks) * 3) * 64 + lane;
        d[0] = (f3_v4u){h[0], h[1], h[2], h[3]};
        d[64] = (f3_v4u){m[0], m[1], m[2], m[3]};
        d[128] = (f3_v4u){l[0], l[1], l[2], l[3]};
    }
}
extern "C" int nntk_shim_frag3_pack(const float *d_x, void *d_frag, int B, int T, int C) {
    if (B <= 0 || T <= 0 || C <= 0) return 0;
    const int NHT = (B + 63) / 64 * 2, NKS = (C + 15) / 16;
    const long blocks = (long)T * NHT;
    if (blocks > 0x7fffffffL) return nntk_fail_msg("frag3_pack: too many row blocks");
    const int vec_ok = (C % 4) == 0 && (((size_t)d_x) & 15) == 0;
    hipLaunchKernelGGL(frag3_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, nntk_stream(), d_x, (f3_v4u *)d_frag, B, T, C,
                       (long)T * C, (long)C, NHT, NKS, vec_ok);
    NNTK_LAUNCH_CHECK("frag3_pack_kernel");
    return 0;
}

// ---- frag3 -> f32 [B][T][C]: x = (hi + mid) + lo, exact (tests; fallback of consumers that do not take the format) ----
__global__ __launch_bounds__(256) void frag3_unpack_kernel(const f3_v4u *__restrict__ src, float *__restrict__ x, int B, int T, int C,
                                                           int NHT, int NKS) {
    const long total = (long)T * NHT * NKS * 64;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        long r = e >> 6;
        const int ks = (int)(r % NKS); r /= NKS;
        const int ht = (int)(r % NHT);
        const int t = (int)(r / NHT);
        const int b = ht * 32 + (lane & 31);
        if (b >= B) continue;
        const f3_v4u *s = src + ((((size_t)t * NHT + ht) * NKS + ks) * 3) * 64 + lane;
        const f3_v4u hi = s[0], mid = s[64], lo = s[128];
        const unsigned hh[4] = {hi.x, hi.y, hi.z, hi.w}, mm[4] = {mid.x, mid.y, mid.z, mid.w}, ll[4] = {lo.x, lo.y, lo.z, lo.w};
        const int c0 = 16 * ks + 8 * (lane >> 5);
        float *row = x + ((size_t)b * T + t) * C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a0 = (__uint_as_float(hh[i] << 16) + __uint_as_float(mm[i] << 16)) + __uint_as_float(ll[i] << 16);
            const float a1 = (__uint_as_float(hh[i] & 0xffff0000u) + __uint_as_float(mm[i] & 0xffff0000u)) + __uint_as_float(ll[i] & 0xffff0000u);
            if (c0 + 2 * i < C) row[c0 + 2 * i] = a0;
            if (c0 + 2 * i + 1 < C) row[c0 + 2 * i + 1] = a1;
        }
    }
}
extern "C" int nntk_shim_frag3_unpack(const void *d_frag, float *d_x, int B, int T, int C) {
    if (B <= 0 || T <= 0 || C <= 0) return 0;
    const int NHT = (B + 63) / 64 * 2, NKS = (C + 15) / 16;
    long g = ((long)T * NHT * NKS * 64 + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(frag3_unpack_kernel, dim3((unsigned)g), dim3(256), 0, nntk_stream(), (const f3_v4u *)d_frag, d_x, B, T, C, NHT, NKS);
    NNTK_LAUNCH_CHECK("frag3_unpack_kernel");
    return 0;
}

// ---- dense GEMM with a frag3 A operand ----------------------------------------------------------------------------
struct DF3Params {
    const char *a;        // frag3 tensor: row block rb = t * NHT + ht at a + rb * NKS * 3072
    const char *w;        // the weight's three split images (behind the packed f32 matrix): image m at w + m * img_bytes,
                          // block (column tile ct, k step ks) at (ct * NKS + ks) * 1024
    const float *bias;    // [N] or NULL
    float *out;           // [B][T][N]
    size_t img_bytes;
    long NRB;             // row blocks = T * NHT
    int NHT, NKS, B, T, N, act_kind;
    float relu_a;
    int m_tiles, n_tiles;
    int dbg;              // unused by the product kernels (the probes are compile-time instantiations)
    float out_scale;      // FMT 1: 2^-(15 + q), the operands' power-of-two scales taken back out of the sums
};

// WM x WN wavefronts, each TM row blocks x TN column tiles of 32 x 32.  D = W x h^T, so a lane owns ONE output row (batch row n of
// its row block) and register r of a tile holds channel 8 (r >> 2) + 4 kh + (r & 3): four consecutive channels per register quad =
// one 16-byte store (the orientation of conv_epilogue).
// (Tried: the A fragments requested TWO k steps ahead through a ring of three register sets -- 240 operand registers next to the 256
// accumulators: 2 784 bytes of scratch, not measured.  The deeper A prefetch lives in dense_frag3_hybrid_kernel below instead.)
// FMT 1: the FRAG2H form of both operands (see below) -- two f16 images, three products, v_mfma_f32_32x32x16_f16; p.a / p.w / p.img_bytes
// describe that form's blocks, everything else is the same kernel.
template <int WM, int WN, int TM, int TN, int FMT = 0>
__global__ __launch_bounds__(256) void dense_frag3_kernel(DF3Params p) {
    static_assert(WM * WN == 4, "4 wavefronts");
    constexpr int NIMG = FMT == 0 ? 3 : 2, NPROD = FMT == 0 ? 6 : 3;
    constexpr int BM_RB = WM * TM;                   // row blocks per workgroup tile
    constexpr int BN = WN * TN * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    // XCD-aware order (conv1d_kernels.hpp): XCD c walks row tiles c, c + 8, ... and all column tiles of a row tile back to back, so a
    // row tile's A blocks come from HBM once and from that XCD's L2 afterwards
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = (local / p.n_tiles) * 8 + xcd;
    if (tile >= p.m_tiles) return;
    // (Measured and not kept: staggering the first round of workgroups by 2.5 - 20 us per phase so that the CUs' epilogues -- 64 MB of
    // stores in a burst while every MFMA pipe idles -- fall into each other's main loops: 2.60 ms with and without, tools/r04j.sh.)
    const int n0 = (local % p.n_tiles) * BN;
    const long rb0 = (long)tile * BM_RB + wm * TM;    // this wave's first row block
#if DF3_LDS_EPI
    // the tile's bias values wait in LDS: fetched from global memory inside the epilogue they are sixteen load -> use round trips per
    // wavefront at the moment the MFMA pipe has just gone idle
    __shared__ __attribute__((aligned(16))) float epi_bias[BN];
    for (int c = threadIdx.x; c < BN; c += 256) epi_bias[c] = (p.bias && n0 + c < p.N) ? p.bias[n0 + c] : 0.0f;
    __syncthreads();
#endif
#if DF3_STAGGER
    if (blockIdx.x < 256) {                           // the first round (one workgroup per CU): spread the CUs' epilogues over one tile period
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)(blockIdx.x & 255) * DF3_STAGGER;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif
    const int NKS = p.NKS;

    // A: one descriptor at the wave's first row block; row blocks past the tensor read zeros (range-checked VECTOR offset)
    const size_t rb_bytes = (size_t)NKS * NIMG * 1024;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)(p.a + (size_t)rb0 * rb_bytes), 0, (int)(TM * rb_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)(NIMG * p.img_bytes), 0x00020000);
    const int lane16 = lane * 16;
    int a_vo[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_vo[i] = rb0 + i < p.NRB ? lane16 + i * (int)rb_bytes : F3_OOB;
    int w_vo[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) w_vo[j] = lane16 + (((n0 >> 5) + wn * TN + j) * NKS) * 1024;
    const int img = (int)p.img_bytes;

    // FRAG2H: THREE operand sets, requested two k steps ahead -- a k step is 48 MFMAs per wavefront there, less than an L2 round trip under
    // load, and a set is 64 registers instead of 96: the same 192 operand registers (1.57 -> 1.49 ms, tools/micro/gemm_f16x2.hip)
    constexpr int NSETS = FMT == 0 ? 2 : 3;
    f3_v4u av[NSETS][TM][NIMG], wv[NSETS][TN][NIMG];
    // requested in the order the products need them (lo of A and hi of W first, see PA / PW below), so the consumer's counted waits
    // release its first MFMAs before the whole set has landed
    constexpr int MA[3] = {NIMG - 1, 0, 1}, MW[3] = {0, NIMG - 1, 1};
    auto load = [&](auto buf_tag, int ks) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_tag)::value;
#pragma unroll
        for (int q = 0; q < NIMG; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) av[buf][i][MA[q]] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_vo[i] + MA[q] * 1024, ks * NIMG * 1024, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) wv[buf][j][MW[q]] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_vo[j], ks * 1024 + MW[q] * img, 0);
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // the 16 accumulator tiles fill the accumulator half of the register file exactly; pinned there, or hipcc budgets them as ordinary
    // registers and spills the operand sets around them (first build: 888 bytes of scratch, 1 269 v_accvgpr copies)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+a"(acc[i][j]));
    // smallest terms first, the order of conv1d_mfma_bf16x3_kernel (bit-identical sums): (A image, W image); FRAG2H: lo.hi, hi.lo, hi.hi
    constexpr int PA[6] = {NIMG - 1, 0, FMT == 0 ? 1 : 0, 1, 0, 0}, PW[6] = {0, NIMG - 1, FMT == 0 ? 1 : 0, 0, 1, 0};
    auto mma2 = [&](auto abuf_tag, auto wbuf_tag) __attribute__((always_inline)) {
        constexpr int ab = decltype(abuf_tag)::value, wb = decltype(wbuf_tag)::value;
#pragma unroll
        for (int t = 0; t < NPROD; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (FMT == 0)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(f3_bf16x8, wv[wb][j][PW[t]]),
                                                                            __builtin_bit_cast(f3_bf16x8, av[ab][i][PA[t]]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f3_f16x8, wv[wb][j][PW[t]]),
                                                                           __builtin_bit_cast(f3_f16x8, av[ab][i][PA[t]]), acc[i][j], 0, 0, 0);
                }
    };
    auto mma = [&](auto buf_tag) __attribute__((always_inline)) { mma2(buf_tag, buf_tag); };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // two operand sets, no branch inside the pair: the loads of one set are in flight under the other set's 96 MFMAs and the compiler
    // can COUNT them (vmcnt(24)); with a conditional load in the loop it waited vmcnt(0) at the join -- one exposed round trip per pair
    int ks = 0;
    {
    load(I0{}, 0);
    // one request per two MFMAs over the first half of a set's MFMAs (four lock-stepped waves share the CU's address path: a burst of
    // 24 stalls the issuing wave and its MFMAs), the second half is the cover for the requests' latency
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NIMG * (TM + TN); ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, DF3_MFMA_PER_LOAD, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (NSETS == 2) {
    for (; ks + 1 < NKS; ks += 2) {
        load(I1{}, ks + 1);
        mma(I0{});
        interleave();
        load(I0{}, ks + 2 < NKS ? ks + 2 : NKS - 1);      // (the last pair re-requests a k step nobody reads: no branch)
        mma(I1{});
        interleave();
    }
    if (ks < NKS) mma(I0{});                              // odd K / 16
    } else {
    using I2 = std::integral_constant<int, 2>;
    load(I1{}, NKS > 1 ? 1 : 0);
    for (; ks + 2 < NKS; ks += 3) {                       // sets 0, 1 hold k steps ks, ks + 1
        load(I2{}, ks + 2);
        mma(I0{});
        interleave();
        load(I0{}, ks + 3 < NKS ? ks + 3 : NKS - 1);      // (past the end: re-requests a k step nobody reads -- no branch)
        mma(I1{});
        interleave();
        load(I1{}, ks + 4 < NKS ? ks + 4 : NKS - 1);
        mma(I2{});
        interleave();
    }
    if (ks < NKS) mma(I0{});
    if (ks + 1 < NKS) mma(I1{});
    }
    }

    // ---- epilogue: bias + activation (chosen once, outside the loops), 16-byte stores of channel quads into out[(b, t), :] ----
    auto epilogue = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long rb = rb0 + i;
            if (rb >= p.NRB) continue;                    // wave-uniform
            const int t = (int)(rb / p.NHT), ht = (int)(rb % p.NHT);
            const int b = ht * 32 + l31;
#if defined(DF3_DBG_STORES) && DF3_DBG_STORES == 2      // timing experiment (WRONG results): every store lands in a 4 MB window that stays in L2
            float *orow = p.out + ((size_t)(b & 31) * p.T + (t & 31)) * p.N;
#elif defined(DF3_DBG_STORES) && DF3_DBG_STORES == 3    // ... the same 4 MB, but contiguous: a store instruction's 32 rows share a page
            float *orow = p.out + ((size_t)(b & 31) * 32 + (t & 31)) * p.N;
#else
            float *orow = p.out + ((size_t)b * p.T + t) * p.N;
#endif
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = n0 + (wn * TN + j) * 32 + 8 * g + 4 * kh;
                    if (b >= p.B || c >= p.N) continue;   // N % 4 == 0 (host): a quad is whole or absent
#if defined(DF3_DBG_STORES) && DF3_DBG_STORES == 1      // timing experiment (WRONG results): the epilogue's arithmetic without its stores
                    if (acc[i][j][4 * g] != 12345.678f) continue;
#endif
                    float4 bi = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.bias) bi = *reinterpret_cast<const float4 *>(p.bias + c);
                    float v[4] = {acc[i][j][4 * g + 0] + bi.x, acc[i][j][4 * g + 1] + bi.y, acc[i][j][4 * g + 2] + bi.z, acc[i][j][4 * g + 3] + bi.w};
                    if constexpr (FMT != 0) {
                        v[0] = fmaf(acc[i][j][4 * g + 0], p.out_scale, bi.x); v[1] = fmaf(acc[i][j][4 * g + 1], p.out_scale, bi.y);
                        v[2] = fmaf(acc[i][j][4 * g + 2], p.out_scale, bi.z); v[3] = fmaf(acc[i][j][4 * g + 3], p.out_scale, bi.w);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = ACT == -1 ? nntk_act(p.act_kind, v[e], p.relu_a) : ACT == NNTK_ACT_RELU ? nntk_act(NNTK_ACT_RELU, v[e], p.relu_a) : v[e];
#if defined(DF3_NT_STORE)
                    __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4 *>(orow + c));
#else
                    *reinterpret_cast<float4 *>(orow + c) = make_float4(v[0], v[1], v[2], v[3]);
#endif
                }
        }
    };
#if DF3_LDS_EPI
    // ---- the same epilogue with WHOLE-LINE stores: a lane owns 16 channels of ONE row (an utterance; rows lie T * N * 4 bytes apart), so the
    // direct form's instructions write 32 bytes of 32 different lines -- 12.5 B/clk and CU (tools/micro/tdd_store_pattern.hip: 8 CUs reach
    // 0.21 TB/s that way and 0.69 TB/s when an instruction writes 8 whole lines): ~10 us per tile and CU that no other CU's work can hide.
    // Each 32 x 32 tile goes through 4 KB of the wavefront's own LDS (row-major, the 16-byte quad index XORed with (row >> 1) & 7: writes
    // and reads conflict-free), comes back as lane -> (row 8 s + lane / 8, quad lane % 8) and leaves as four instructions of 8 x 128
    // bytes.  No barrier: a wavefront's LDS operations execute in order.  Same values (bias and activation applied before the trip).
    __shared__ f3_v4u epi_lds[4][2][256];                // [wavefront][parity][32 rows x 8 quads]
    auto epilogue_lds = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
        // (the kernel sits at 512 registers: whatever the epilogue derives from the lane id must not be hoisted above the K loop, or the
        // loop spills -- 1 088 bytes of scratch in the first build; the empty asm pins the derivation down here)
        int le = lane;
        asm volatile("" : "+v"(le));
        const int wr_row = (le & 31) * 8, wr_sw = ((le & 31) >> 1) & 7;
        const int rd_r = le >> 3, rd_q = le & 7;
        const float osc = p.out_scale;
        (void)osc;
        f3_for<0, TM>([&](auto i_tag) __attribute__((always_inline)) {
            constexpr int i = decltype(i_tag)::value;
            const long rb = rb0 + i;
            if (rb >= p.NRB) return;                      // wave-uniform
            const int t = (int)(rb / p.NHT), ht = (int)(rb % p.NHT);
            f3_for<0, TN>([&](auto j_tag) __attribute__((always_inline)) {
                constexpr int j = decltype(j_tag)::value;
                f3_v4u *buf = epi_lds[wave][(i * TN + j) & 1];
                const int cw = n0 + (wn * TN + j) * 32;
                f32x16 tile = acc[i][j];
                asm volatile("" : "+v"(tile));            // one accumulator tile at a time out of the accumulator half (else hipcc dumps all 256 to scratch at the loop's exit)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bi = *reinterpret_cast<const float4 *>(epi_bias + (wn * TN + j) * 32 + 8 * g + 4 * (le >> 5));
                    float v[4] = {tile[4 * g + 0] + bi.x, tile[4 * g + 1] + bi.y, tile[4 * g + 2] + bi.z, tile[4 * g + 3] + bi.w};
                    if constexpr (FMT != 0) {      // the sums carry the operands' scales 2^15 (h) and 2^q (W): one exact multiplication, fused with the bias
                        v[0] = fmaf(tile[4 * g + 0], osc, bi.x); v[1] = fmaf(tile[4 * g + 1], osc, bi.y);
                        v[2] = fmaf(tile[4 * g + 2], osc, bi.z); v[3] = fmaf(tile[4 * g + 3], osc, bi.w);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = ACT == -1 ? nntk_act(p.act_kind, v[e], p.relu_a) : ACT == NNTK_ACT_RELU ? nntk_act(NNTK_ACT_RELU, v[e], p.relu_a) : v[e];
                    buf[wr_row + ((2 * g + (le >> 5)) ^ wr_sw)] = (f3_v4u){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int row = 8 * s + rd_r;
                    const f3_v4u x = buf[row * 8 + (rd_q ^ ((row >> 1) & 7))];
                    const int b = ht * 32 + row, c = cw + 4 * rd_q;
                    if (b < p.B && c < p.N) *reinterpret_cast<f3_v4u *>(p.out + ((size_t)b * p.T + t) * p.N + c) = x;
                }
            });
        });
    };
    if (p.act_kind == NNTK_ACT_IDENTITY) epilogue_lds(std::integral_constant<int, NNTK_ACT_IDENTITY>{});
    else if (p.act_kind == NNTK_ACT_RELU) epilogue_lds(std::integral_constant<int, NNTK_ACT_RELU>{});
    else epilogue_lds(std::integral_constant<int, -1>{});
#else
    if (p.act_kind == NNTK_ACT_IDENTITY) epilogue(std::integral_constant<int, NNTK_ACT_IDENTITY>{});
    else if (p.act_kind == NNTK_ACT_RELU) epilogue(std::integral_constant<int, NNTK_ACT_RELU>{});
    else epilogue(std::integral_constant<int, -1>{});
#endif
}

#ifdef NNTK_VARIANT_DENSE_RING      // A/B variant only (tools/build_variant.py ... frag3.hip -DNNTK_VARIANT_DENSE_RING; option dense_frag3 = 3): measured slower
// ---- the same GEMM with both operands staged through an LDS ring by LDS-DMA ----------------------------------------
// First measurement of the register-direct kernel above at the stack's TimeDistributedDense (510 k x 512 x 1000): 3.43 ms against
// 2.74 ms for the LDS-staged f32-input GEMM it was meant to beat.  Its two wn (wm) waves fetch every A (W) block twice and a wave can
// keep only one k step (24 KB) in flight next to its 256 accumulator registers: 96 KB per k step and CU at ~1.5 us of cover.
// Here a 256 x 256 tile's 48 blocks of a k step are fetched ONCE per workgroup, global -> LDS by buffer_load ... lds (no registers
// on the way; a block's lane order is its LDS image), into a ring of three 48 KB stages: two k steps (96 KB) are in flight under the
// MFMAs of a third.  Per k step: counted wait for the wave's own 12 DMAs of the stage, ONE barrier (stage k landed everywhere, and
// everybody is done reading stage k - 1), 12 DMAs for stage k + 2 into the slot just freed, 24 conflict-free ds_read_b128, 96 MFMAs.
// Same products, same order: bit-identical to the kernel above and to the f32-input GEMM.
#define DF3_STAGES 3
#define DF3_STAGE_BYTES (48 * 1024)
// DBG (timing experiments, tools/tdd_probe.py; WRONG results; compile-time on purpose -- a run-time mask made hipcc spill the
// accumulators: 1 648 bytes of scratch, 10 x slower): 1 no output stores, 2 operand fetch of k step 0 only, 4 no MFMAs
template <int DBG>
__global__ __launch_bounds__(256) void dense_frag3_lds_kernel(DF3Params p) {
    constexpr int TM = 4, TN = 4, WN = 2;
    extern __shared__ __attribute__((aligned(16))) char f3_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = (local / p.n_tiles) * 8 + xcd;
    if (tile >= p.m_tiles) return;
    const int n0 = (local % p.n_tiles) * 256;
    // A tile = ONE 32-row batch block (ht) x EIGHT consecutive timesteps: its output rows are 32 runs of 8 adjacent rows of
    // out [B][T][N] (32 pages) instead of 256 rows that are T * N floats apart.  (First version: 8 batch blocks of one timestep --
    // its output stores alone were 1.36 ms of a 3.74 ms launch, tools/tdd_probe.py.)
    const int ht = tile % p.NHT, t0 = (tile / p.NHT) * 8;
    const int NKS = p.NKS;
    const size_t rb_bytes = (size_t)NKS * 3072;
    const size_t t_bytes = (size_t)p.NHT * rb_bytes;               // distance between the tile's row blocks (one timestep)
    const int lane16 = lane * 16;
    // DMA duty of this wave: 12 of a stage's 48 blocks.  Waves 0, 1: the A blocks of row blocks 4 w .. 4 w + 3 (3 images each, 3 KB
    // contiguous per row block and k step); waves 2, 3: the W blocks of column tiles 4 (w - 2) .. + 3.  One descriptor and one pair of
    // strides per wave, chosen on the scalar unit: the DMA sequence itself is branch-free.
    const bool dma_a = wave < 2;
    const char *d_base = dma_a ? p.a + ((size_t)t0 * p.NHT + ht) * rb_bytes : p.w;
    const int d_range = dma_a ? (int)(7 * t_bytes + rb_bytes) : (int)(3 * p.img_bytes);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void *)d_base, 0, d_range, 0x00020000);
    const int d_ks = dma_a ? 3072 : 1024, d_m = dma_a ? 1024 : (int)p.img_bytes;
    int d_vo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int vo_a = t0 + wave * 4 + q < p.T ? lane16 + (wave * 4 + q) * (int)t_bytes : F3_OOB;
        const int vo_w = lane16 + (((n0 >> 5) + (wave - 2) * 4 + q) * NKS) * 1024;
        d_vo[q] = dma_a ? vo_a : vo_w;
    }
    auto dma = [&](int slot, int ks) __attribute__((always_inline)) {
        char *base = f3_smem + slot * DF3_STAGE_BYTES + wave * 12 * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m = 0; m < 3; ++m)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (__attribute__((address_space(3))) void *)(base + (q * 3 + m) * 1024), 16,
                                                         d_vo[q], ks * d_ks + m * d_m, 0, 0);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+a"(acc[i][j]));
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PW[6] = {0, 2, 1, 0, 1, 0};
    const int a_rd = (wm * 4 * 3) * 1024 + lane16;              // this wave's first A block inside a stage
    const int w_rd = (24 + wn * 4 * 3) * 1024 + lane16;

    dma(0, 0);
    dma(1, NKS > 1 ? 1 : 0);
    int slot = 0;
    for (int ks = 0; ks < NKS; ++ks) {
        // this wave's 12 DMAs of stage ks have landed once only the 12 of stage ks + 1 are outstanding
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int nslot = slot == 0 ? 2 : slot - 1;               // (ks + 2) % 3: the slot stage ks - 1 has just left
        dma(nslot, (DBG & 2) ? 0 : ks + 2 < NKS ? ks + 2 : NKS - 1);     // (past the end: re-requests a k step nobody reads -- no branch)
        const char *st = f3_smem + slot * DF3_STAGE_BYTES;
        f3_bf16x8 av[TM][3], wv[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int m = 0; m < 3; ++m) av[i][m] = *reinterpret_cast<const f3_bf16x8 *>(st + a_rd + (i * 3 + m) * 1024);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m) wv[j][m] = *reinterpret_cast<const f3_bf16x8 *>(st + w_rd + (j * 3 + m) * 1024);
        if (!(DBG & 4)) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[j][PW[t]], av[i][PA[t]], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int m = 0; m < 3; ++m) acc[i][j][m] += (float)__builtin_bit_cast(f3_v4u, av[i][m]).x + (float)__builtin_bit_cast(f3_v4u, wv[j][m]).y;
        }
        // the fragments are in registers (the MFMAs above consumed them): the next barrier may release this slot
        slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the tail's spare DMAs must not outlive the ring
    if ((DBG & 1) && acc[0][0][0] != 12345.678f) return;

    // ---- epilogue: bias + activation, then the tile leaves ROW BY ROW.  A lane owns one output row and 4-channel quads of it, so a
    // direct store instruction touched 32 rows x 32 bytes; the accumulators go through the (now idle) LDS ring instead -- pass q moves
    // row block q of every wave, [2 wm][32 rows][256 columns] f32 with a 16-byte row pad (conflict-free both ways) -- and leave as one
    // 1 KB contiguous store per row (64 lanes x 16 bytes): 64 store instructions per wave as before, each one cache-line run.
    // The activation is chosen ONCE, outside the loops (identity / ReLU compiled in; anything else takes the per-element switch):
    // with the run-time switch inside, the 256 values of a lane became 256 chains of scalar branches -- 39 k lines of ISA and
    // 1.37 ms of a 3.72 ms launch, all of it exposed because one workgroup owns the CU (tools/tdd_probe.py).
    constexpr int EP_PITCH = 1024 + 16, EP_BUF = 2 * 32 * EP_PITCH;       // 2 buffers = 133 KB of the 144 KB ring
    __builtin_amdgcn_s_barrier();                                         // every wave has left the ring
    auto epilogue = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;                     // -1: run-time kind
#pragma unroll
        for (int q = 0; q < TM; ++q) {
            char *buf = f3_smem + (q & 1) * EP_BUF;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = wn * 128 + j * 32 + 8 * g + 4 * kh;       // column inside the tile
                    const int c = n0 + cl;
                    float4 bi = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.bias && c < p.N) bi = *reinterpret_cast<const float4 *>(p.bias + c);
                    float v[4] = {acc[q][j][4 * g + 0] + bi.x, acc[q][j][4 * g + 1] + bi.y, acc[q][j][4 * g + 2] + bi.z, acc[q][j][4 * g + 3] + bi.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = ACT == -1 ? nntk_act(p.act_kind, v[e], p.relu_a) : ACT == NNTK_ACT_RELU ? nntk_act(NNTK_ACT_RELU, v[e], p.relu_a) : v[e];
                    *reinterpret_cast<float4 *>(buf + (wm * 32 + l31) * EP_PITCH + cl * 4) = make_float4(v[0], v[1], v[2], v[3]);
                }
            __builtin_amdgcn_s_barrier();
            // this wave stores rows 16 wave .. + 15 of the pass's 64 (block wm' = row >> 5 holds timestep t0 + 4 wm' + q)
            const int c = n0 + lane * 4;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = wave * 16 + rr;
                const int t = t0 + (row >> 5) * 4 + q;
                const int b = ht * 32 + (row & 31);
                if (t >= p.T || b >= p.B) continue;                           // wave-uniform
                const float4 v = *reinterpret_cast<const float4 *>(buf + row * EP_PITCH + lane16);
                if (c < p.N) *reinterpret_cast<float4 *>(p.out + ((size_t)b * p.T + t) * p.N + c) = v;
            }
        }
    };
    if (p.act_kind == NNTK_ACT_IDENTITY) epilogue(std::integral_constant<int, NNTK_ACT_IDENTITY>{});
    else if (p.act_kind == NNTK_ACT_RELU) epilogue(std::integral_constant<int, NNTK_ACT_RELU>{});
    else epilogue(std::integral_constant<int, -1>{});
}

#endif

// (Measured and not kept -- profiles/r04_tdd_three_kernels.log, same box, ms at the stack's shape: register-direct 2.54, this LDS
// ring 2.63, and a hybrid that took A through a three-stage LDS ring requested TWO k steps ahead (one fetch per workgroup, 12 ds_read
// per wave and k step under the MFMAs) with W register-direct: 2.60.  Three operand paths, one result: the operand fetch is not what
// holds the kernels at 0.61 MFMA-busy; the exposed epilogue (0.35-0.4 ms: one workgroup owns the CU) and the clock the bf16 pipe is
// allowed (1.89 GHz here) are.  The hybrid kernel was removed again; this ring kernel stays as the probe harness, tools/tdd_probe.py.)


// ---- FRAG2H: pack / unpack (tests, fallbacks) and the weights' two f16 images -----------------------------------------
#define F3_H2_SCALE 32768.0f
__device__ __forceinline__ void f3_split_pair_h2(float x0, float x1, float scale, unsigned &hi, unsigned &lo) {
#pragma clang fp contract(off)    // the residual is that of the ROUNDED product
    const float y0 = x0 * scale, y1 = x1 * scale;                         // power of two: exact
    const f3_f16x2 hh = __builtin_convertvector((f3_f32x2){y0, y1}, f3_f16x2);
    const f3_f16x2 ll = __builtin_convertvector((f3_f32x2){y0 - (float)hh[0], y1 - (float)hh[1]}, f3_f16x2);
    hi = __builtin_bit_cast(unsigned, hh);
    lo = __builtin_bit_cast(unsigned, ll);
}
extern "C" size_t nntk_shim_frag2h_floats(int B, int T, int C) {
    if (B <= 0 || T <= 0 || C <= 0) return 0;
    return (size_t)T * ((size_t)(B + 63) / 64 * 2) * (size_t)((C + 15) / 16) * 2 * 256;
}
// f32 [B][T][C] (|x| < 2) -> frag2h; the recurrent kernels write the same bits for the same values (rr_split8_h2)
__global__ __launch_bounds__(256) void frag2h_pack_kernel(const float *__restrict__ x, f3_v4u *__restrict__ dst, int B, int T, int C, int NHT, int NKS) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const long rb = blockIdx.x;
    const int t = (int)(rb / NHT), ht = (int)(rb % NHT);
    const int b = ht * 32 + n;
    const float *row = x + ((size_t)b * T + t) * C;
    for (int ks = w; ks < NKS; ks += 4) {
        const int c0 = 16 * ks + 8 * kh;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (b < B && c0 + q < C) ? row[c0 + q] : 0.0f;
        unsigned h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f3_split_pair_h2(v[2 * i], v[2 * i + 1], F3_H2_SCALE, h[i], l[i]);
        f3_v4u *d = dst + (((size_t)rb * NKS + ks) * 2) * 64 + lane;
        d[0] = (f3_v4u){h[0], h[1], h[2], h[3]};
        d[64] = (f3_v4u){l[0], l[1], l[2], l[3]};
    }
}
extern "C" int nntk_shim_frag2h_pack(const float *d_x, void *d_frag, int B, int T, int C) {
    if (B <= 0 || T <= 0 || C <= 0) return 0;
    const int NHT = (B + 63) / 64 * 2, NKS = (C + 15) / 16;
    const long blocks = (long)T * NHT;
    if (blocks > 0x7fffffffL) return nntk_fail_msg("frag2h_pack: too many row blocks");
    hipLaunchKernelGGL(frag2h_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, nntk_stream(), d_x, (f3_v4u *)d_frag, B, T, C, NHT, NKS);
    NNTK_LAUNCH_CHECK("frag2h_pack_kernel");
    return 0;
}
// frag2h -> f32: x = (hi + lo) 2^-15 (the sum is exact in f32: lo lies within half an ulp of hi's 11 bits and has 11 bits itself)
__global__ __launch_bounds__(256) void frag2h_unpack_kernel(const f3_v4u *__restrict__ src, float *__restrict__ x, int B, int T, int C, int NHT, int NKS) {
    const long total = (long)T * NHT * NKS * 64;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        long r = e >> 6;
        const int ks = (int)(r % NKS); r /= NKS;
        const int ht = (int)(r % NHT);
        const int t = (int)(r / NHT);
        const int b = ht * 32 + (lane & 31);
        if (b >= B) continue;
        const f3_v4u *s = src + ((((size_t)t * NHT + ht) * NKS + ks) * 2) * 64 + lane;
        const f3_v4u hi = s[0], lo = s[64];
        const unsigned hh[4] = {hi.x, hi.y, hi.z, hi.w}, ll[4] = {lo.x, lo.y, lo.z, lo.w};
        const int c0 = 16 * ks + 8 * (lane >> 5);
        float *row = x + ((size_t)b * T + t) * C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f3_f16x2 h2 = __builtin_bit_cast(f3_f16x2, hh[i]), l2 = __builtin_bit_cast(f3_f16x2, ll[i]);
            if (c0 + 2 * i < C) row[c0 + 2 * i] = ((float)h2[0] + (float)l2[0]) * (1.0f / F3_H2_SCALE);
            if (c0 + 2 * i + 1 < C) row[c0 + 2 * i + 1] = ((float)h2[1] + (float)l2[1]) * (1.0f / F3_H2_SCALE);
        }
    }
}
extern "C" int nntk_shim_frag2h_unpack(const void *d_frag, float *d_x, int B, int T, int C) {
    if (B <= 0 || T <= 0 || C <= 0) return 0;
    const int NHT = (B + 63) / 64 * 2, NKS = (C + 15) / 16;
    long g = ((long)T * NHT * NKS * 64 + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(frag2h_unpack_kernel, dim3((unsigned)g), dim3(256), 0, nntk_stream(), (const f3_v4u *)d_frag, d_x, B, T, C, NHT, NKS);
    NNTK_LAUNCH_CHECK("frag2h_unpack_kernel");
    return 0;
}
// packed weights src [rows][ktot] f32 (rows % 32 == 0, ktot % 16 == 0: nntk_upload_gemm_weights) -> dst [2 images][rows / 32][ktot / 16][2][32][8] f16
// of src * scale -- the order of split_bf16x3_kernel (conv1d.hip), i.e. the A fragment of column tile rows / 32, k step ktot / 16
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float *__restrict__ src, unsigned *__restrict__ dst, int rows, int ktot, float scale) {
    const size_t n_pairs = (size_t)rows * ktot / 2;
    const int ksteps = ktot >> 4;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n_pairs; e += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(e / (ktot / 2)), kt = (int)(e % (ktot / 2)) * 2;
        unsigned h, l;
        f3_split_pair_h2(src[2 * e], src[2 * e + 1], scale, h, l);
        const size_t d = ((((size_t)(row >> 5) * ksteps + (kt >> 4)) * 2 + ((kt >> 3) & 1)) * 32 + (row & 31)) * 4 + ((kt & 7) >> 1);
        dst[d] = h; dst[n_pairs + d] = l;
    }
}
extern "C" int nntk_shim_split_f16x2(const float *d_src, void *d_dst, int rows, int ktot, float scale) {
    if (rows <= 0 || ktot <= 0) return 0;
    if ((rows & 31) || (ktot & 15)) return nntk_fail_msg("split_f16x2: rows must be a multiple of 32 and ktot of 16");
    size_t g = ((size_t)rows * ktot / 2 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)g), dim3(256), 0, nntk_stream(), d_src, (unsigned *)d_dst, rows, ktot, scale);
    NNTK_LAUNCH_CHECK("split_f16x2_kernel");
    return 0;
}

// The GEMM on FRAG2H operands.  d_wh2: the weights' two f16 images of W * w_scale (nntk_shim_split_f16x2 over the packed f32 matrix, N_p * K_p
// pairs each); w_scale a power of two.  0 = launched; 1 = shape not taken (probe = 1: nothing is launched, the answer only); -1 = error.
extern "C" int nntk_shim_dense_frag2h(const void *d_frag, const void *d_wh2, float w_scale, const float *d_bias, int act_kind, float relu_a,
                                      float *d_out, int B, int T, int K, int N, int probe) {
    if (B <= 0 || T <= 0) return 0;
    if (act_kind == NNTK_ACT_NONE) act_kind = NNTK_ACT_IDENTITY;
    if (act_kind == NNTK_ACT_SOFTMAX || act_kind == NNTK_ACT_CUSTOM) return 1;
    const NntkOptions &opt = nntk_options();
    if (opt.gemm_split_bf16 == 0 || opt.gemm_split_bf16 == 2 || opt.dense_frag3 == 0 || opt.dense_f16x2 == 0) return 1;
    if (!d_wh2 || !(w_scale > 0.0f)) return 1;
    int K_p, N_p;
    nntk_shim_conv_pack_sizes(K, N, 1, &K_p, &N_p);
    if ((N % 4) != 0 || (((size_t)d_out) & 15) != 0 || (d_bias && (((size_t)d_bias) & 15) != 0)) return 1;
    if (N_p % 128 != 0 || K_p != ((K + 15) / 16) * 16) return 1;
    const int NHT = (B + 63) / 64 * 2, NKS = K_p / 16;
    const size_t n_w = (size_t)N_p * K_p;
    if (n_w * 4 >= (size_t)F3_OOB || (size_t)8 * NKS * 2048 >= (size_t)F3_OOB) return 1;
    DF3Params p;
    p.a = (const char *)d_frag;
    p.w = (const char *)d_wh2;
    p.img_bytes = n_w * 2;
    p.bias = d_bias; p.out = d_out;
    p.NRB = (long)T * NHT; p.NHT = NHT; p.NKS = NKS; p.B = B; p.T = T; p.N = N;
    p.act_kind = act_kind; p.relu_a = relu_a;
    p.dbg = 0;
    p.out_scale = 1.0f / (F3_H2_SCALE * w_scale);           // powers of two: exact
    const bool wide = N_p % 256 == 0;
    p.m_tiles = (int)((p.NRB + 7) / 8);
    p.n_tiles = N_p / (wide ? 256 : 128);
    const long blocks = (long)((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
    if (blocks > 0x7fffffffL) return 1;
    if (probe) return 0;
    if (wide) hipLaunchKernelGGL((dense_frag3_kernel<2, 2, 4, 4, 1>), dim3((unsigned)blocks), dim3(256), 0, nntk_stream(), p);
    else hipLaunchKernelGGL((dense_frag3_kernel<4, 1, 2, 4, 1>), dim3((unsigned)blocks), dim3(256), 0, nntk_stream(), p);
    NNTK_LAUNCH_CHECK("dense_frag3_kernel<f16x2>");
    return 0;
}

// 0 = launched; 1 = shape not taken (the caller unpacks and runs the f32 GEMM); -1 = error.
// d_wp: the packed weights of nntk_upload_gemm_weights ([N_p][K_p] f32 followed by the three split images); K = the frag3 tensor's C.
extern "C" int nntk_shim_dense_frag3(const void *d_frag, const float *d_wp, const float *d_bias, int act_kind, float relu_a,
                                     float *d_out, int B, int T, int K, int N) {
    if (B <= 0 || T <= 0) return 0;
    if (act_kind == NNTK_ACT_NONE) act_kind = NNTK_ACT_IDENTITY;
    if (act_kind == NNTK_ACT_SOFTMAX || act_kind == NNTK_ACT_CUSTOM) return 1;
    const NntkOptions &opt = nntk_options();
    if (opt.gemm_split_bf16 == 0 || opt.gemm_split_bf16 == 2 || opt.dense_frag3 == 0) return 1;      // (2: the split contraction for convolutions only)
    // a weight block the bf16 split cannot hold (non-finite, > 3.39e38, denormal: marked at upload) keeps the exact-f32 GEMM, exactly as
    // nntk_shim_conv1d decides for the f32 call -- or this route would give NaN where that one stays finite
    if (opt.gemm_split_bf16 < 0 && nntk_weights_exact_only(d_wp)) return 1;
    int K_p, N_p;
    nntk_shim_conv_pack_sizes(K, N, 1, &K_p, &N_p);
    if ((N % 4) != 0 || (((size_t)d_out) & 15) != 0 || (d_bias && (((size_t)d_bias) & 15) != 0)) return 1;
    if (N_p % 128 != 0 || K_p != ((K + 15) / 16) * 16) return 1;
    const int NHT = (B + 63) / 64 * 2, NKS = K_p / 16;
    const size_t n_w = (size_t)N_p * K_p;
    if (n_w * 6 >= (size_t)F3_OOB || (size_t)8 * NKS * 3072 >= (size_t)F3_OOB) return 1;
    DF3Params p;
    p.a = (const char *)d_frag;
    p.w = (const char *)(d_wp + n_w);
    p.img_bytes = n_w * 2;
    p.bias = d_bias; p.out = d_out;
    p.NRB = (long)T * NHT; p.NHT = NHT; p.NKS = NKS; p.B = B; p.T = T; p.N = N;
    p.act_kind = act_kind; p.relu_a = relu_a;
    p.dbg = opt.conv_dbg;
    p.out_scale = 1.0f;
    const bool wide = N_p % 256 == 0;
    // the register-direct kernel (2.44 vs 2.54 ms for the LDS-ring variant at the stack's shape, same box: profiles/r04_tdd_three_kernels.log)
#ifdef NNTK_VARIANT_DENSE_RING
    const bool ring = wide && opt.dense_frag3 == 3;
#else
    const bool ring = false;
#endif
    // ring kernel: a tile is one batch block x 8 timesteps; register-direct kernel: 8 consecutive row blocks (t-major)
    p.m_tiles = ring ? NHT * ((T + 7) / 8) : (int)((p.NRB + 7) / 8);
    if (ring && (size_t)8 * NHT * NKS * 3072 >= (size_t)F3_OOB) return 1;
    (void)ring;
    p.n_tiles = N_p / (wide ? 256 : 128);
    const long blocks = (long)((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
    if (blocks > 0x7fffffffL) return 1;
#ifdef NNTK_VARIANT_DENSE_RING
    if (ring) {
        const size_t lds = (size_t)DF3_STAGES * DF3_STAGE_BYTES;
        void (*kern)(DF3Params) = dense_frag3_lds_kernel<0>;
#ifdef NNTK_CONV_DBG
        switch (opt.conv_dbg) {     // diagnostics build only (tools/tdd_probe.py)
        case 1: kern = dense_frag3_lds_kernel<1>; break;
        case 2: kern = dense_frag3_lds_kernel<2>; break;
        case 3: kern = dense_frag3_lds_kernel<3>; break;
        case 4: kern = dense_frag3_lds_kernel<4>; break;
        case 7: kern = dense_frag3_lds_kernel<7>; break;
        default: break;
        }
#endif
        if (nntk_set_max_dynamic_lds((const void *)kern, lds)) return -1;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, nntk_stream(), p);
    } else
#endif
#ifdef DF3_TILE      // A/B: another tile shape, -DDF3_TILE=WM,WN,TM,TN (smaller tiles leave room for a second workgroup per CU)
    if (wide) {
        constexpr int sh[4] = {DF3_TILE};
        p.m_tiles = (int)((p.NRB + sh[0] * sh[2] - 1) / (sh[0] * sh[2]));
        p.n_tiles = N_p / (sh[1] * sh[3] * 32);
        const long blocks2 = (long)((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
        hipLaunchKernelGGL((dense_frag3_kernel<DF3_TILE>), dim3((unsigned)blocks2), dim3(256), 0, nntk_stream(), p);
    } else
#else
    if (wide) hipLaunchKernelGGL((dense_frag3_kernel<2, 2, 4, 4>), dim3((unsigned)blocks), dim3(256), 0, nntk_stream(), p);
    else
#endif
                     hipLaunchKernelGGL((dense_frag3_kernel<4, 1, 2, 4>), dim3((unsigned)blocks), dim3(256), 0, nntk_stream(), p);
    NNTK_LAUNCH_CHECK("dense_frag3_kernel");
    return 0;
}
