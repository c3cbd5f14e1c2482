// recurrent_rr_common.hpp -- what the two register-resident recurrent kernel families share (recurrent_rr.hip: two half-streams per
// workgroup, 16 hidden units; recurrent_rr4.hip: four half-streams, 8 hidden units): operand types, the exact 3-way bf16 split, the
// launch parameters, the diagnostics hooks.
#pragma once
#include "nntk_common.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

typedef __bf16 rr_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rr_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rr_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned rr_v4u __attribute__((ext_vector_type(4)));

#ifndef RR_ST_AUX
#define RR_ST_AUX 16              // cache policy of the hand-off STORES: 16 = sc1 (write-through, placement-independent)
#endif
#define RR_FLAGS 64               // flag words per (batch tile, half): one per column tile (<= 32), padded to one wave-wide load
#ifndef RR_POLL_LEAD
#define RR_POLL_LEAD 1           // the flags are requested this many k steps before they are looked at
#endif
#ifndef RR_NPRE
#define RR_NPRE 2                // operand k steps requested ahead, at the end of the other half's sequence (the rest: own sequence)
#endif
#ifndef RR_S_E1
#define RR_S_E1 3                // k step at which a half's publication is taken to have drained (tools/rr_stamps.py)
#endif

// "not yet written" pattern of the frag3 hand-off (recurrent_rr.hip): a published word never equals it
#define RR_PENDING 0xffffffffu
__device__ __forceinline__ rr_v4u rr_not_pending(rr_v4u v) {
    return (rr_v4u){min(v.x, 0xfffffffeu), min(v.y, 0xfffffffeu), min(v.z, 0xfffffffeu), min(v.w, 0xfffffffeu)};
}

#define RR_HS_LD 12               // dwords per row of a split h exchange image (8 used; 16-byte aligned rows, two-way bank spread)
#define RR_HX_LD 20               // floats per row of the h exchange image (16-byte aligned rows)

__device__ __forceinline__ unsigned rr_cvt_pk(float a, float b) {       // RNE, a in the low half
    return __builtin_bit_cast(unsigned, __builtin_convertvector((rr_f32x2){a, b}, rr_bf16x2));
}
// x = hi + mid + lo exactly (8 + 8 + 8 significand bits), two elements at a time
__device__ __forceinline__ void rr_split_pair(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
#ifndef RR_LINT_SELFTEST          // (-DRR_LINT_SELFTEST: what tools/check_rr_waits.py's split check must catch)
#pragma clang fp contract(off)    // the residuals are those of the ROUNDED x (inlined behind x = a * b, x - hi must not become fma(a, b, -hi))
#endif
    hi = rr_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    mid = rr_cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
    lo = rr_cvt_pk(s0, s1);
}
// eight consecutive f32 -> the three 16-byte bf16 fragments
__device__ __forceinline__ void rr_split8(const float (&v)[8], rr_v4u &hi, rr_v4u &mid, rr_v4u &lo) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rr_split_pair(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
    hi = (rr_v4u){h[0], h[1], h[2], h[3]};
    mid = (rr_v4u){m[0], m[1], m[2], m[3]};
    lo = (rr_v4u){l[0], l[1], l[2], l[3]};
}

// eight consecutive f32 of magnitude < 2 -> the two 16-byte f16 fragments of x * 2^15: hi = f16(x 2^15), lo = f16(x 2^15 - hi) (frag3.hip FRAG2H)
typedef _Float16 rr_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 rr_f16x8 __attribute__((ext_vector_type(8)));
#define RR_H2_SCALE 32768.0f
// two f32 already multiplied by their power-of-two scale -> hi = f16(y), lo = f16(y - hi), packed
__device__ __forceinline__ void rr_split_pair_f16(float y0, float y1, unsigned &hi, unsigned &lo) {
#pragma clang fp contract(off)    // the residual is that of the ROUNDED value
    const rr_f16x2 hh = __builtin_convertvector((rr_f32x2){y0, y1}, rr_f16x2);
    const rr_f16x2 ll = __builtin_convertvector((rr_f32x2){y0 - (float)hh[0], y1 - (float)hh[1]}, rr_f16x2);
    hi = __builtin_bit_cast(unsigned, hh);
    lo = __builtin_bit_cast(unsigned, ll);
}
__device__ __forceinline__ void rr_split_pair_h2(float v0, float v1, unsigned &hi, unsigned &lo) {
#pragma clang fp contract(off)
    rr_split_pair_f16(v0 * RR_H2_SCALE, v1 * RR_H2_SCALE, hi, lo);       // power of two: exact
}
__device__ __forceinline__ void rr_split8_h2(const float (&v)[8], rr_v4u &hi, rr_v4u &lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rr_split_pair_h2(v[2 * i], v[2 * i + 1], h[i], l[i]);
    hi = (rr_v4u){h[0], h[1], h[2], h[3]};
    lo = (rr_v4u){l[0], l[1], l[2], l[3]};
}

struct RRParams {
    const float *x;            // [B][T][in]
    const rr_v4u *img;         // weight images (rr_pack_kernel)
    const float *bi, *bh;      // [4H]; bh NULL when !v2
    // split hand-off = the layer output in FRAG3 form (frag3.hip): h0f [NHT][NKS][3] blocks of 1 KB holds h_0; hseq [T][NHT][NKS][3]
    // receives h_t of every step (step t reads h_{t-1}: t == 0 from h0f, else from hseq + (t - 1) * hstep) -- T-deep, so the
    // published fragments ARE a tensor the next layer can consume (a stacked GRU's x operand, the dense GEMM's A operand)
    char *h0f, *hseq;
    size_t hstep;              // bytes per timestep of hseq = NHT_total * NKS * 3072, NKS = H / 16
    const char *xf3;           // XF: x as a frag3 tensor [T][NHT][NKSx][3] blocks, NKSx = ceil(in / 16) (instead of p.x)
    size_t xstep;
    const float *c0;           // [B][H] or NULL (zeros)
    float *cT, *hT;            // [B][H] or NULL
    float *out;                // [B][T][H] or [B][H]
    unsigned *flags;           // [NBT][2 halves][RR_FLAGS], zeroed before the launch
    int x_tm, out_tm;          // x / the sequence output in time-major layout ([T][B][.]: the tensor between two stacked layers)
    float *c_cache;            // training forward (TRAIN): cell state of every step [B][T][H] ...
    float *z_cache;            // ... and pre-activations | activations [B][T][8H] (lstm.c:426-475 keeps them for BPTT)
    unsigned *fault;
    unsigned long long spin_ticks;
    int B, T, H, in, NBT, NCT, b_base, return_sequences;
    int NKSx;                  // k steps of 16 the x frag3 tensor stores per row block
    int NHT;                   // half-tiles (32 rows) the frag3 tensors hold: 2 ceil(B / 64) (recurrent_rr4.hip masks streams past it)
    // the sequence output once more as a FRAG2H tensor (frag3.hip: two f16 images of h * 2^15, [T][NHT][NKS][2] blocks of 1 KB) for the dense
    // GEMM's three-product contraction, or NULL; written by the wave that would write the f32 rows (p.out must be NULL then).  recurrent_rr.hip only
    char *out_h2;
    size_t h2step;             // bytes per timestep = NHT * NKS * 2048
    float z_scale;             // HF instantiations: 2^-(15 + q), the operands' scales taken out of Z (recurrent_rr.hip)
#ifdef NNTK_REC_STAMPS
    unsigned long long *stamp; // [T + 1][4 streams][16] s_memtime of workgroup 0, wave 0 (diagnostics build only; the two-stream kernels use streams 0, 1)
#endif
#ifdef NNTK_RR_BOUNDS
    unsigned long long *bounds; // [8]: see RR_BOUND below (diagnostics build only)
#endif
};

// -DNNTK_RR_BOUNDS (tools/rr_bounds_check.py, tests/test_gpu_lstm_rr.py): every request the kernel sends towards a CALLER-visible tensor
// records the last byte it really touches -- lanes whose vector offset falls outside the descriptor's range touch nothing and are
// skipped, exactly as the hardware skips them -- as an offset from the tensor's base: word 0 x (f32 rows), 1 x (frag3), 2 f32 output,
// 3 frag3 hand-off / output (hseq), 4 the h_0 slot, 5 the FRAG2H output.  The host compares them with the tensors' sizes.  Round 3 closed a read past the
// end of x that the buffer range check could not see (the half-tile rode in the scalar offset); this makes such a read visible.
#ifdef NNTK_RR_BOUNDS
#define RR_BOUND(word, base_off, vo, so, range, bytes) do { \
        if (p.bounds && (unsigned)(vo) < (unsigned)(range)) \
            atomicMax(p.bounds + (word), (unsigned long long)(base_off) + (unsigned long long)(unsigned)(vo) + (unsigned long long)(so) + (bytes)); } while (0)
#else
#define RR_BOUND(word, base_off, vo, so, range, bytes) do {} while (0)
#endif

#ifdef NNTK_REC_STAMPS
#define RR_STAMP(half, t, i) do { if (p.stamp && blockIdx.x == 0 && w == 0 && lane == 0) \
        p.stamp[((size_t)(t) * 4 + (half)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RR_STAMP(half, t, i) do {} while (0)
#endif

// timing ablations only (WRONG results; tools/rr_ablate.sh builds one library per mask): -DNNTK_RR_DBG=<mask>,
// 1 no operand loads, 2 no finish, 4 no arrive / poll, 8 no x, 16 no MFMAs, 32 no partial-sum writes,
// 64 no gate arithmetic, 128 no publication (split + stores), 256 no partial-sum reads, 512 no workgroup barriers, 1024 no output stores.  Compile-time on
// purpose: a run-time mask changed the register allocation of the whole kernel (2x slower with the mask at 0).
#ifdef NNTK_RR_DBG
#define RR_DBG(bit) ((NNTK_RR_DBG) & (bit))
#else
#define RR_DBG(bit) 0
#endif

// raw barrier: LDS traffic ordered, vector-memory operations (the operand prefetch!) left in flight
#define RR_BARRIER() do { if (!RR_DBG(512)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
#ifndef RR_NO_PIN
#define RR_PIN_A(v) asm volatile("" : "+a"(v))      // accumulator-file registers: MFMA operands only, never copied about
#else
#define RR_PIN_A(v) do {} while (0)
#endif
#define RR_OOB 0x7ffffff0          // out-of-range vector offset: a buffer load returns 0, a buffer store is dropped
#define RR_OOB_F 0x7f000000        // the same for the frag3 blocks, whose instructions add up to 3 KB of immediate offset (no wrap); steps < this

