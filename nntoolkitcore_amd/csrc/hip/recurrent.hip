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
#include <stdio.h>
#include <stdlib.h>

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
    float s0, s1, s2, s3, s4; // ReLU output scales of those activations (activation_default.c:123-129), 1 otherwise
};

// NG = number of intra-workgroup split-K groups (256 threads each).  NG = 2 puts two
// wavefronts on every SIMD: while one issues its 32 MFMAs of a K-chunk, the other's
// LDS reads / barrier wait / global prefetch are hidden behind them.  The two partial
// accumulators are summed through LDS in a fixed order (group 0 + group 1), so results
// do not depend on scheduling or on how the batch is sharded.
template <int G, bool IS_LSTM, int NG>
__global__ __launch_bounds__(256 * NG) void rec_step_kernel(RecParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int A_SZ = REC_BM * REC_LS;
    constexpr int B_SZ = G * REC_HN * REC_LS;
    constexpr int BUF = A_SZ + B_SZ;
    constexpr int RPW = 4 / NG;                  // accumulator rows finished by each wave in the epilogue

    const int tid = threadIdx.x & 255;
    const int grp = threadIdx.x >> 8;
    const int lane = tid & 63;
    const int wave = tid >> 6;                   // row slab 0..3
    const int l15 = lane & 15, q = lane >> 4;
    // blockIdx.x walks the hidden-unit tiles: blocks are dealt round-robin over the 8
    // XCDs, so XCD c keeps U^T tiles {c, c+8, ...} (a few hundred KB) resident in its
    // private L2 for every batch tile, instead of each XCD streaming all of U^T.
    // blockIdx.x = batch tile: blocks are dealt round-robin over the XCDs, so a batch tile's
    // workgroups share an XCD and its h rows stay in that L2 (hidden-tile-major was 7 % slower)
    const int j0 = blockIdx.y * REC_HN;
    const int b0 = blockIdx.x * REC_BM;
    const int j = j0 + l15;
    const bool vec4 = (p.H & 3) == 0;
    float *my = smem + grp * 2 * BUF;

    // ---- epilogue operands do not depend on the GEMM: fetch them first so their
    //      latency hides behind the K loop ----
    const int GH = G * p.H;
    float xwv[RPW][G], prev[RPW], bh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) bh[g] = (p.bh && j < p.H) ? p.bh[g * p.H + j] : 0.0f;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int b = b0 + wave * 16 + q * 4 + grp * RPW + rr;
        const bool ok = b < p.B && j < p.H;
#pragma unroll
        for (int g = 0; g < G; ++g) xwv[rr][g] = ok ? p.xw[(size_t)b * GH + g * p.H + j] : 0.0f;
        prev[rr] = ok ? (IS_LSTM ? p.c[(size_t)b * p.H + j] : p.h_prev[(size_t)b * p.H + j]) : 0.0f;
    }

    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: A tile 64 rows x 8 float4, B tile 16G rows x 8 float4 (per group)
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
        float *As = my + buf * BUF, *Bs = As + A_SZ;
#pragma unroll
        for (int s = 0; s < A_PT; ++s) {
            int e = tid + s * 256;
            int r = e >> 3, c4 = e & 7;
            *reinterpret_cast<float4 *>(&As[r * REC_LS + c4 * 4]) = areg[s];
        }
#pragma unroll
        for (int s = 0; s < B_PT; ++s) {
            int e = tid + s * 256;
            if (e < B_F4) {
                int r = e >> 3, c4 = e & 7;
                *reinterpret_cast<float4 *>(&Bs[r * REC_LS + c4 * 4]) = breg[s];
            }
        }
    };

    const int nchunks = p.Hk_p / REC_KC;
    const int iters = (nchunks + NG - 1) / NG;       // same barrier count for every group
    if (grp < nchunks) load_chunk(grp * REC_KC);
    for (int it = 0; it < iters; ++it) {
        const int ch = it * NG + grp;
        const int buf = it & 1;
        const bool live = ch < nchunks;
        if (live) store_chunk(buf);
        __syncthreads();
        if (ch + NG < nchunks) load_chunk((ch + NG) * REC_KC);
        if (live) {
            // lane group q owns k = q*8 + s (s = 0..7) of this chunk: any bijection of
            // k onto (group, step) is a valid MFMA K order as long as A and B agree.
            const float *As = my + buf * BUF, *Bs = As + A_SZ;
            const float *arow = &As[(wave * 16 + l15) * REC_LS + q * 8];
            const float4 a_lo = *reinterpret_cast<const float4 *>(arow);
            const float4 a_hi = *reinterpret_cast<const float4 *>(arow + 4);
            const float av[8] = {a_lo.x, a_lo.y, a_lo.z, a_lo.w, a_hi.x, a_hi.y, a_hi.z, a_hi.w};
            float bv[G][8];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float *brow = &Bs[(g * 16 + l15) * REC_LS + q * 8];
                const float4 b_lo = *reinterpret_cast<const float4 *>(brow);
                const float4 b_hi = *reinterpret_cast<const float4 *>(brow + 4);
                bv[g][0] = b_lo.x; bv[g][1] = b_lo.y; bv[g][2] = b_lo.z; bv[g][3] = b_lo.w;
                bv[g][4] = b_hi.x; bv[g][5] = b_hi.y; bv[g][6] = b_hi.z; bv[g][7] = b_hi.w;
            }
            // gates innermost: consecutive MFMAs hit different accumulators, so the
            // 40-cycle dependent-accumulator latency of 16x16x4 never stalls the pipe
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int g = 0; g < G; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[g][s], acc[g], 0, 0, 0);
        }
    }

    // ---- split-K reduction: every wave publishes its partial tile, then finishes
    //      rows [grp*RPW, grp*RPW+RPW) of the sum (fixed order: group 0 + group 1) ----
    float fin[RPW][G];
    if (NG == 1) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int g = 0; g < G; ++g) fin[rr][g] = acc[g][rr];
    } else {
        __syncthreads();                              // staging buffers are dead
        float *red = smem;                            // [grp][wave][g][r][lane]
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((grp * 4 + wave) * G + g) * 4 + r) * 64 + lane] = acc[g][r];
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = grp * RPW + rr;
#pragma unroll
            for (int g = 0; g < G; ++g)
                fin[rr][g] = red[(((0 * 4 + wave) * G + g) * 4 + r) * 64 + lane] +
                             red[(((1 * 4 + wave) * G + g) * 4 + r) * 64 + lane];
        }
    }

    // ---- fused gate epilogue: lane holds (b = b0 + 16*wave + 4*q + r, j = j0 + l15) ----
    if (j >= p.H) return;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int b = b0 + wave * 16 + q * 4 + grp * RPW + rr;
        if (b >= p.B) continue;
        float hn;
        if constexpr (G == 1) {
            // rnn.c:144-166: gate = (h U [+ b_h]) + (x W + b_i); h' = act(gate)
            hn = nntk_gate_act(p.a0, xwv[rr][0] + (fin[rr][0] + bh[0]), p.s0);
        } else if constexpr (!IS_LSTM) {
            // gru.c:144-186
            const float hz = fin[rr][0] + bh[0], hr = fin[rr][1] + bh[1], hh = fin[rr][2] + bh[2];
            const float z = nntk_gate_act(p.a0, xwv[rr][0] + hz, p.s0);
            const float rg = nntk_gate_act(p.a2, xwv[rr][1] + hr, p.s2);
            const float ht = nntk_gate_act(p.a1, fmaf(rg, hh, xwv[rr][2]), p.s1);
            hn = fmaf(-z + 1.0f, ht, z * prev[rr]);
        } else {
            // lstm.c:201-238
            const float zi = xwv[rr][0] + (fin[rr][0] + bh[0]);
            const float zf = xwv[rr][1] + (fin[rr][1] + bh[1]);
            const float zg = xwv[rr][2] + (fin[rr][2] + bh[2]);
            const float zo = xwv[rr][G - 1] + (fin[rr][G - 1] + bh[G - 1]);
            const float ig = nntk_gate_act(p.a0, zi, p.s0);
            const float fg = nntk_gate_act(p.a1, zf, p.s1);
            const float gg = nntk_gate_act(p.a2, zg, p.s2);
            const float og = nntk_gate_act(p.a3, zo, p.s3);
            const float cn = fmaf(fg, prev[rr], ig * gg);
            p.c[(size_t)b * p.H + j] = cn;
            hn = og * nntk_gate_act(p.a4, cn, p.s4);
        }
        p.h_next[(size_t)b * p.H + j] = hn;
        if (p.out) p.out[(size_t)b * p.out_ld + j] = hn;
    }
}

// ============================================================================
// Persistent variant: ONE launch runs all T timesteps.
//
// Each workgroup owns (batch tile of 64 rows) x (16 hidden units x G gates) for the
// whole sequence and keeps its U^T tile RESIDENT in LDS (LSTM-512: 64 x 512 f32 =
// 128 KiB of the CU's 160 KiB), so the only per-step operand traffic is the h tile.
// The grid-wide dependency between steps is local to a batch tile: the NCT
// workgroups of one batch tile exchange h through memory each step.  Block ids
// are dealt so that one batch tile's workgroups share an XCD (bt = id % NBT, with
// NBT a multiple of 8 at the BASELINE shapes): their h traffic stays in that XCD's
// L2.  Placement is a speed matter only: the hand-off below is the
// placement-independent recipe of the CDNA4 guide (Guideline 16, R1 / first row
// of the sc1 table): payload stored write-through (sc1), every storing wave
// drains vmcnt, workgroup barrier, ONE lane adds to an agent-scope counter; the
// consumer polls that counter from one lane (relaxed, s_sleep), joins a
// workgroup barrier, and EVERY load of handed-off bytes is an sc1 load.
//
// MFMA orientation is swapped relative to rec_step_kernel: D[hidden][batch] =
// U^T tile (A operand) x h^T (B operand), so a lane ends up with consecutive
// hidden units of ONE batch row: xW loads, h/out stores are contiguous per lane.
//
// Residency: the grid is sized to at most one workgroup per CU (LDS use forces
// that) and never exceeds the CU count; every spin is bounded (1 s of
// s_memrealtime) and poisons the counter's top bit, which the host checks.
// ============================================================================
#define RECP_CNT_STRIDE 64  // uints between arrival counters (256 B)
#define RECP_MAXCH 16     // H <= 512 (16 chunks of 32): the whole h tile is held in registers per step
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));

struct RecPParams {
    const float *xw;      // [T, B, G*H]
    const float *ut;      // [G, Hj_p, Hk_p]
    const float *bh;      // [G*H] or NULL
    float *hbuf;          // [2][hb_floats] ping-pong h in the TILED hand-off layout (see below); parity 0 holds h_0
    const float *h0;      // [B][H] initial state in the caller's layout, or NULL (zeros): GRU's own-h term
    float *hT;            // [B][H] final state in the caller's layout, or NULL
    size_t hb_floats;     // floats per parity: B rounded up to 64, times the padded K
    float *c;             // [B][H] LSTM cell state (read at start, written at end)
    float *out;           // [B, T, H] or [B, H]
    unsigned *cnt;        // [NBT] arrival counters, zeroed before the launch
    int B, T, H, Hj_p, Hk_p, NBT, NCT, b_base;
    int return_sequences;
    int a0, a1, a2, a3, a4;
    float s0, s1, s2, s3, s4;       // ReLU output scales (generic-activation variant only)
    unsigned *fault;                // sticky device fault word (runtime.hip): OR-ed on a spin timeout, never reset per launch
    unsigned long long spin_ticks;  // budget of every spin in s_memrealtime ticks (100 MHz)
#ifdef NNTK_REC_STAMPS
    unsigned long long *stamp;   // [2 halves][T][8] s_memtime of workgroup 0's leader waves (diagnostics build only)
#endif
};
#ifdef NNTK_REC_STAMPS
#define REC_STAMP(i) do { if (p.stamp && blockIdx.x == 0 && lane == 0) {                                               \
        if (PP ? wih == 0 : (w8 & 3) == 0)                                                                            \
            p.stamp[((size_t)(PP ? half : (w8 >> 2)) * p.T + t) * 8 + (i)] = __builtin_amdgcn_s_memtime();           \
        if ((i) == 2 || (i) == 3)      /* K-loop start / end of EVERY wave, after the leaders' block */              \
            p.stamp[(size_t)2 * p.T * 8 + ((size_t)w8 * p.T + t) * 2 + ((i) - 2)] = __builtin_amdgcn_s_memtime();    \
    } } while (0)
#else
#define REC_STAMP(i) do {} while (0)
#endif

// LDS spin on a monotonic word written by another wavefront of the same workgroup.  Relaxed
// loads only: an acquire at workgroup scope would make hipcc drain vmcnt, i.e. wait for the
// h / xW loads in flight.  LDS operations of one wave execute in order, so "data, then flag"
// on the producer side and "flag, then data" on the consumer side is enough.
__device__ __forceinline__ void recp_lds_wait_ge(const unsigned *w, unsigned target) {
    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

template <bool VEC>
__device__ __forceinline__ float4 load4_sc1(__amdgpu_buffer_rsrc_t rsrc, int soff, const float *base, size_t idx,
                                            int n_valid) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n_valid <= 0) return v;
    if (VEC) {     // H % 4 == 0: k < H implies the whole 16-byte piece is inside the row
        v4u32 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(idx * 4), soff, 16 /* sc1 */);
        v.x = __uint_as_float(r.x); v.y = __uint_as_float(r.y); v.z = __uint_as_float(r.z); v.w = __uint_as_float(r.w);
        return v;
    }
    const unsigned *u = reinterpret_cast<const unsigned *>(base + idx);
    v.x = __uint_as_float(__hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (n_valid > 1) v.y = __uint_as_float(__hip_atomic_load(u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (n_valid > 2) v.z = __uint_as_float(__hip_atomic_load(u + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (n_valid > 3) v.w = __uint_as_float(__hip_atomic_load(u + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return v;
}

// NCH = K chunks of 32 (compile-time: the h loads and the MFMA loop must be straight-line code --
// a runtime bound made hipcc guard every load with a branch and wait vmcnt(0) for the WHOLE tile
// before the first MFMA, ~6k cycles per step).  H is padded up to 32*NCH with zero U^T columns
// and out-of-range (zero-returning) h loads.
// XW = where this step's xW loads and the previous step's output store are issued:
//   0 before the poll (prefetch), 1 right after the h loads.
// Vector-memory results return in issue order, so with XW = 0 the poll's load waits behind HBM
// loads / stores: global timestamps showed the poll passing 0.2 .. 2.6 us after the last arrival
// (the workgroup that signalled last stays last) against 0.3 .. 1.0 us with XW = 1.  That was with
// eight 4-byte xW loads per wave; with one 16-byte load per gate the two are within 2 % of each other
// in the classic kernel (default XW = 1) and XW = 0 is 10 % faster in the ping-pong kernel, where the
// poll is off the critical path; NNTK_REC_XW overrides.
// Hand-off layout of h (hbuf): [16-row slab][16-float k block][k group q = 0..3][row n = 0..15][4 floats],
//   i.e. exactly the order in which the 64 lanes (lane = 16 q + n) of one wavefront consume a
//   (slab, k block) as the MFMA B operand -- one b128 load instruction reads 1 KB CONTIGUOUS.
//   With h in the caller's row-major [B][H] layout the same instruction touched 16 rows x 64 B
//   with neighbouring lanes 2 KB apart; a microbenchmark of just these loads (tools/micro/
//   hload_bench.hip) delivers the 128 KB tile in 3.6 us that way and in 1.15 us contiguously,
//   and the in-kernel stamps showed the second split-K wave receiving its data 9k cycles late.
//   Padding (rows >= B, k >= H) stays zero: the buffers are cleared before the launch and only
//   valid elements are ever written.
// PP = ping-pong: the workgroup's 64 batch rows are two independent 32-row halves, each with its
//   own arrival counter and hand-off chain, worked by one wavefront per SIMD (2 slabs x 2 split-K
//   halves: wavefronts 0-3 and 4-7) in strict alternation: half A multiplies, half A runs its
//   exchange + gates and publishes, half B multiplies, ... so one half's drain / arrival /
//   propagation / poll / first-bytes latency (about 7k cycles of a 27k-cycle step) hides behind
//   the other half's MFMA phase.  The token passes only AFTER the gate phase: a wave issuing
//   back-to-back f32 MFMAs starves the VALU of the other wave on its SIMD completely, whatever
//   s_setprio says (tools/micro/coissue_bench.hip: 410k + 81k cycles alone, 490k together), so
//   gates that overlap the other half's K loop simply wait for its end (measured: +9k cycles).
//   There is no workgroup barrier in the step loop; wavefronts meet through monotonic LDS words.
//   LSTM-512: 27.2k -> 21.7k cycles per step.  Not for short K loops (GRU-256: the hand-off
//   chain, not the SIMD, is the bound there and the extra sync costs 5 %).
// STD = the gate activations are the standard ones (GRU: sigmoid / tanh / sigmoid, LSTM: sigmoid x3 +
//   tanh x2): the kinds become compile-time constants and the gate phase -- pure VALU work that
//   cannot overlap the MFMAs -- loses its 10 scalar branch ladders.
template <int G, bool IS_LSTM, int NCH, int XW, bool PP, bool STD>
__global__ __launch_bounds__(512, 2) void rec_persistent_kernel(RecPParams p) {
    const int A0 = STD ? (G == 1 ? NNTK_ACT_TANH : NNTK_ACT_SIGMOID) : p.a0;
    const int A1 = STD ? (IS_LSTM ? NNTK_ACT_SIGMOID : NNTK_ACT_TANH) : p.a1;
    const int A2 = STD ? (IS_LSTM ? NNTK_ACT_TANH : NNTK_ACT_SIGMOID) : p.a2;
    const int A3 = STD ? NNTK_ACT_SIGMOID : p.a3;
    const int A4 = STD ? NNTK_ACT_TANH : p.a4;
    const float S0 = STD ? 1.0f : p.s0, S1 = STD ? 1.0f : p.s1, S2 = STD ? 1.0f : p.s2, S3 = STD ? 1.0f : p.s3,
                S4 = STD ? 1.0f : p.s4;
    constexpr bool VEC = true;                    // launcher guarantees H % 4 == 0
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KP = NCH * REC_KC;              // padded K
    constexpr int US = KP + 8;                    // U^T row stride: 16-B aligned, conflict-free b128 slots
    float *Us = smem;                             // [G*16][US]   resident for the whole sequence
    float *red = smem + G * 16 * US;              // [4 slabs][2 groups][G][2][64] split-K exchange
    // ping-pong sync words (monotonic): [0..1] poll passed (step) | [2..3] waves drained (4 per step)
    //   | [4..5] waves past their gate phase (4 per step) | [8..15] split-K partials written (step+1), per (slab, K half)
    unsigned *syncw = reinterpret_cast<unsigned *>(red + 4 * 2 * G * 2 * 64);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // wavefront 0..7, as a SCALAR: everything derived from it (slab, split-K half, buffer soffsets) stays
    // in SGPRs -- with a VGPR-derived index hipcc wrapped each of the 16 h loads in a waterfall loop
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 8 wavefronts = 4 batch slabs of 16 rows x 2 split-K halves.  (A variant in which every
    // slab was its own sync domain with per-wave polling was measured 40 % SLOWER: polling
    // traffic from 2048 waves outweighed the overlap -- one poller per workgroup it is.)
    // ping-pong: half = which 32-row half this wave works on; one wave of each half per SIMD
    const int half = PP ? (w8 >> 2) : 0;
    const int wih = PP ? (w8 & 3) : w8;           // wave index inside the half, 0..3
    const int grp = PP ? (wih >> 1) : (w8 >> 2);  // split-K half: k sub-range [16*grp, 16*grp+16) of each chunk
    const int slab = PP ? (half * 2 + (wih & 1)) : (w8 & 3);
    const bool leader = PP ? (wih == 0) : (w8 == 0);
    const int l15 = lane & 15, q = lane >> 4;
    const int bt = blockIdx.x % p.NBT;
    const int ct = blockIdx.x / p.NBT;
    const int b0 = p.b_base + bt * REC_BM;
    const int j0 = ct * REC_HN;
    constexpr bool pair8 = VEC;
    const int GH = G * p.H;
    // one arrival counter per batch tile, each on its own 256-B line: atomics execute at the
    // memory side, so counters sharing a line serialise on one channel (measured: 3.5x slower)
    unsigned *cnt = p.cnt + ((size_t)bt * (PP ? 2 : 1) + half) * RECP_CNT_STRIDE;

    // this lane finishes hidden units j, j+1 of batch row b (after the split-K exchange)
    const int b = b0 + slab * 16 + l15;
    const int j = j0 + q * 4 + grp * 2;
    const bool row_ok = b < p.B;
    const bool ok0 = row_ok && j < p.H, ok1 = row_ok && j + 1 < p.H;

    // ---- resident U^T tile (columns >= Hk_p are zero) ----
    {
        constexpr int f4_per_row = KP / 4;
        constexpr int total = G * 16 * f4_per_row;
        for (int e = tid; e < total; e += 512) {
            const int r = e / f4_per_row, c4 = e % f4_per_row;
            const int g = r >> 4, jj = r & 15;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c4 * 4 < p.Hk_p)
                v = *reinterpret_cast<const float4 *>(p.ut + ((size_t)g * p.Hj_p + j0 + jj) * p.Hk_p + c4 * 4);
            *reinterpret_cast<float4 *>(&Us[r * US + c4 * 4]) = v;
        }
    }
    float bh[2][G];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int g = 0; g < G; ++g) bh[e][g] = (p.bh && j + e < p.H) ? p.bh[g * p.H + j + e] : 0.0f;
    // own previous state: c (LSTM) or h (GRU) for the two elements this lane finishes
    float prev[2];
    {
        const float *src = IS_LSTM ? p.c : p.h0;
        prev[0] = (ok0 && src) ? src[(size_t)b * p.H + j] : 0.0f;
        prev[1] = (ok1 && src) ? src[(size_t)b * p.H + j + 1] : 0.0f;
    }
    // tiled hand-off buffers (ping / pong): this wave's (slab, k block) pieces are 1 KB apart
    constexpr int KB = KP / 16;                   // 16-float k blocks per row
    const int slab_abs = (b0 >> 4) + slab;
    const int hb_bytes = (int)(p.hb_floats * 4);
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)p.hbuf, 0, hb_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.hbuf + p.hb_floats), 0, hb_bytes, 0x00020000);
    const int h_lane_off = lane * 16;
    const int h_wave_off = (slab_abs * KB + grp) * 1024;      // + ch * 2048
    // where this lane publishes its two finished elements (k = j, j + 1) of row b
    const size_t h_pub_off = ((((size_t)slab_abs * KB + ct) * 4 + q) * 16 + l15) * 4 + grp * 2;
    float *my_red = red + ((slab * 2 + grp) * G * 2) * 64;
    const float *peer_red = red + ((slab * 2 + (1 - grp)) * G * 2) * 64;
    if (PP && tid < 16) syncw[tid] = 0u;
    __syncthreads();

    float hn_prev[2] = {0.f, 0.f};
    for (int t = 0; t < p.T; ++t) {
        float xwv[2][G];
#define REC_XW_ISSUE() do {                                                                          \
            /* one 16-byte load per gate: the 4 hidden units of this lane's quad (both split-K waves  */ \
            /* fetch the quad, each keeps its two) -- a quarter of the VMEM instructions of 2 x b32   */ \
            const float *xw = p.xw + ((size_t)t * p.B + b) * GH + (j & ~3);                            \
            _Pragma("unroll") for (int g = 0; g < G; ++g) {                                            \
                float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);                                           \
                if (ok0) x4 = *reinterpret_cast<const float4 *>(xw + g * p.H);                         \
                xwv[0][g] = grp ? x4.z : x4.x;                                                         \
                xwv[1][g] = grp ? x4.w : x4.y;                                                         \
            }                                                                                          \
            if (XW != 0 && t > 0 && p.return_sequences) {                                              \
                float *o = p.out + ((size_t)b * p.T + (t - 1)) * p.H + j;                              \
                if (pair8 && ok1) *reinterpret_cast<float2 *>(o) = make_float2(hn_prev[0], hn_prev[1]); \
                else { if (ok0) o[0] = hn_prev[0]; if (ok1) o[1] = hn_prev[1]; }                       \
            }                                                                                          \
        } while (0)
        REC_STAMP(0);
        if (XW == 0) REC_XW_ISSUE();
        // ---- wait until every workgroup of this batch tile has published h_{t-1}:
        //      ONE lane polls (relaxed, s_sleep), the workgroup joins a barrier, and every
        //      load of handed-off bytes below is an sc1 load ----
        if (t > 0) {
            if (leader && lane == 0) {
                const unsigned target = (unsigned)p.NCT * (unsigned)t;
                const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                // spin_ticks == 0 is fault injection (tests): behave as if the first poll had run out of budget
                bool expired = p.spin_ticks == 0;
                while (!expired && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    expired = __builtin_amdgcn_s_memrealtime() - t_start > p.spin_ticks;     // default 1 s at 100 MHz
                }
                if (expired) {
                    // give up: raise the sticky fault word for the host, and poison the counter so that
                    // every other poller of this batch tile falls through as well (all spins stay bounded)
                    __hip_atomic_fetch_or(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_or(cnt, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (PP) __hip_atomic_store(&syncw[half], (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (PP) { if (!leader) recp_lds_wait_ge(&syncw[half], (unsigned)t); }
            else __syncthreads();
        }
        REC_STAMP(1);
        f32x4 acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // The h operand needs no LDS: in this orientation lane (batch row l15, k-group q)
        // consumes h[b][32*ch + 16*grp + 4*q .. +3] -- 16 contiguous bytes per chunk that
        // only this lane uses.  All NCH pieces are requested back to back (sc1 loads); the
        // compiler's counted vmcnt lets chunk c start as soon as piece c has landed.
        v4u32 hreg[NCH];
        if (t & 1) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                hreg[ch] = __builtin_amdgcn_raw_buffer_load_b128(rs1, h_lane_off, h_wave_off + ch * 2048, 16 /* sc1 */);
        } else {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                hreg[ch] = __builtin_amdgcn_raw_buffer_load_b128(rs0, h_lane_off, h_wave_off + ch * 2048, 16 /* sc1 */);
        }
        if (XW == 1) REC_XW_ISSUE();
        // token: half 0 multiplies step t after half 1 has finished the gates of step t-1, half 1 after half 0's of step t
        if (PP) recp_lds_wait_ge(&syncw[4 + (1 - half)], 4u * (unsigned)(t + half));
        REC_STAMP(2);
        // U^T fragments: software-pipelined one chunk ahead of the MFMAs that use them.  (Pinning that
        // order with sched_barrier cost 12 %, a sched_group_barrier interleave serialised the gates'
        // accumulators: the scheduler's own placement is the best measured.)
        // two base addresses so that every fragment read is base + immediate (ds offsets stop at 64 KB)
        const int uoff01 = l15 * US + grp * 16 + q * 4;
        int uoff23 = uoff01 + 2 * 16 * US;
        asm volatile("" : "+v"(uoff23));        // opaque: otherwise hipcc folds it back into uoff01 + (too large an) immediate
#define REC_UFRAG(g, ch) (*reinterpret_cast<const float4 *>(&Us[((g) < 2 ? uoff01 + (g) * 16 * US : uoff23 + ((g) - 2) * 16 * US) + (ch) * REC_KC]))
        float4 un[G];
#pragma unroll
        for (int g = 0; g < G; ++g) un[g] = REC_UFRAG(g, 0);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            float4 uc[G];
#pragma unroll
            for (int g = 0; g < G; ++g) uc[g] = un[g];
            if (ch + 1 < NCH) {
#pragma unroll
                for (int g = 0; g < G; ++g) un[g] = REC_UFRAG(g, ch + 1);
            }
            // keep the next chunk's LDS reads ABOVE this chunk's 4 G MFMAs: left alone, the scheduler
            // sinks each read to just before its first use and every chunk eats the LDS latency
            const float hv[4] = {__uint_as_float(hreg[ch].x), __uint_as_float(hreg[ch].y),
                                 __uint_as_float(hreg[ch].z), __uint_as_float(hreg[ch].w)};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float u = s == 0 ? uc[g].x : s == 1 ? uc[g].y : s == 2 ? uc[g].z : uc[g].w;
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(u, hv[s], acc[g], 0, 0, 0);
                }
        }

        REC_STAMP(3);
        // ---- split-K exchange through LDS: send the half the partner wave finishes ----
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int e = 0; e < 2; ++e) my_red[(g * 2 + e) * 64 + lane] = grp ? acc[g][e] : acc[g][2 + e];
        if (PP) {
            asm volatile("" ::: "memory");
            if (lane == 0) {
                __hip_atomic_store(&syncw[8 + slab * 2 + grp], (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            recp_lds_wait_ge(&syncw[8 + slab * 2 + (1 - grp)], (unsigned)(t + 1));
        } else {
            __syncthreads();
        }
        REC_STAMP(4);
        float fin[2][G];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float other = peer_red[(g * 2 + e) * 64 + lane];
                const float mine = grp ? acc[g][2 + e] : acc[g][e];
                fin[e][g] = grp == 0 ? mine + other : other + mine;      // always (group 0) + (group 1)
            }

        // ---- gates (same formulas as rec_step_kernel) ----
        float hn[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if constexpr (G == 1) {
                hn[e] = nntk_gate_act(A0, xwv[e][0] + (fin[e][0] + bh[e][0]), S0);      // rnn.c:144-166
            } else if constexpr (!IS_LSTM) {
                const float hz = fin[e][0] + bh[e][0], hr = fin[e][1] + bh[e][1], hh = fin[e][2] + bh[e][2];
                const float z = nntk_gate_act(A0, xwv[e][0] + hz, S0);
                const float rg = nntk_gate_act(A2, xwv[e][1] + hr, S2);
                const float ht = nntk_gate_act(A1, fmaf(rg, hh, xwv[e][2]), S1);
                hn[e] = fmaf(-z + 1.0f, ht, z * prev[e]);
                prev[e] = hn[e];
            } else {
                const float zi = xwv[e][0] + (fin[e][0] + bh[e][0]);
                const float zf = xwv[e][1] + (fin[e][1] + bh[e][1]);
                const float zg = xwv[e][2] + (fin[e][2] + bh[e][2]);
                const float zo = xwv[e][G - 1] + (fin[e][G - 1] + bh[e][G - 1]);
                const float ig = nntk_gate_act(A0, zi, S0);
                const float fg = nntk_gate_act(A1, zf, S1);
                const float gg = nntk_gate_act(A2, zg, S2);
                const float og = nntk_gate_act(A3, zo, S3);
                const float cn = fmaf(fg, prev[e], ig * gg);
                prev[e] = cn;
                hn[e] = og * nntk_gate_act(A4, cn, S4);
            }
        }
        // ---- publish h_t (write-through), then this wave's arrival; the layer output
        //      (never read in this launch) is stored after the arrival ----
        float *hdst = p.hbuf + (size_t)((t + 1) & 1) * p.hb_floats + h_pub_off;
        if (pair8 && ok1) {
            const unsigned long long pk = ((unsigned long long)__float_as_uint(hn[1]) << 32) | __float_as_uint(hn[0]);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(hdst), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (ok0) __hip_atomic_store(reinterpret_cast<unsigned *>(hdst), __float_as_uint(hn[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ok1) __hip_atomic_store(reinterpret_cast<unsigned *>(hdst + 1), __float_as_uint(hn[1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the other half may start multiplying now (not earlier: its MFMAs would starve these gates)
        if (PP && lane == 0) __hip_atomic_fetch_add(&syncw[4 + half], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        REC_STAMP(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // R1: every storing wave drains ...
        REC_STAMP(6);
        if (PP) {
            // ... the half's four waves meet on an LDS word (`red` is not rewritten before the next
            // step's poll has passed, i.e. after the partner has drained too) ...
            if (lane == 0) {
                __hip_atomic_fetch_add(&syncw[2 + half], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (leader) {
                    while (__hip_atomic_load(&syncw[2 + half], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * (unsigned)(t + 1)) {}
                    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ... ONE lane arrives
                }
            }
        } else {
            __syncthreads();                                  // ... the workgroup meets (this also frees `red`) ...
            if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... ONE lane arrives
        }
        REC_STAMP(7);
        if (XW == 0) {
            if (p.return_sequences || t == p.T - 1) {
                float *o = p.return_sequences ? p.out + ((size_t)b * p.T + t) * p.H + j : p.out + (size_t)b * p.H + j;
                if (pair8 && ok1) *reinterpret_cast<float2 *>(o) = make_float2(hn[0], hn[1]);
                else { if (ok0) o[0] = hn[0]; if (ok1) o[1] = hn[1]; }
            }
        }
        hn_prev[0] = hn[0]; hn_prev[1] = hn[1];
    }
    if (XW != 0) {   // output of the last step (the only one when !return_sequences)
        float *o = p.return_sequences ? p.out + ((size_t)b * p.T + (p.T - 1)) * p.H + j : p.out + (size_t)b * p.H + j;
        if (pair8 && ok1) *reinterpret_cast<float2 *>(o) = make_float2(hn_prev[0], hn_prev[1]);
        else { if (ok0) o[0] = hn_prev[0]; if (ok1) o[1] = hn_prev[1]; }
    }
    if (IS_LSTM) {
        if (ok0) p.c[(size_t)b * p.H + j] = prev[0];
        if (ok1) p.c[(size_t)b * p.H + j + 1] = prev[1];
    }
    if (p.hT) {
        if (ok0) p.hT[(size_t)b * p.H + j] = hn_prev[0];
        if (ok1) p.hT[(size_t)b * p.H + j + 1] = hn_prev[1];
    }
}

static size_t rec_hb_floats_fwd(int B, int H);
static size_t rec_cnt_words_fwd(int B);

// ============================================================================
// Fused two-layer GRU (BASELINE configs[3]: GRU 128 -> 256 -> 256): ONE persistent launch runs both layers, layer 2
// one step behind layer 1.  In iteration i a workgroup computes, for its 64 batch rows x 16 hidden units,
//     layer 1, step i     :  U1^T . h1_{i-1}                       (split-K over the two waves of a slab, as above)
//     layer 2, step i - 1 :  W2^T . h1_{i-1}   and   U2^T . h2_{i-2}   (one full-K product per wave of the slab)
// -- everything it needs was published in iteration i-1.  Each layer keeps its own hand-off chain (own arrival counter):
// while the workgroup multiplies for one layer, the other layer's publication -> arrival -> poll latency elapses, which
// is what bounds the single-layer GRU-256 kernel.  Layer 2's input projection (a 403 GFLOP GEMM launch and its
// [T, B, 3H] tensor) disappears into the resident W2^T tile, and the inter-layer [B, T, H] tensor never reaches HBM.
// Three 48 x K tiles stay in LDS (H = 256: 2 x 50.7 KB + 49.9 KB + the 12 KB exchange = 163.6 of 160 KiB = 163.8 KB;
// W2^T takes the 260-float row stride -- one 2-way bank conflict per b128 group -- because three 264-float tiles do
// not fit).  Layer 1's numbers are bit-identical to the single-layer kernel; layer 2's input projection is summed in
// the MFMA's k order here instead of the GEMM kernel's (same tolerance, not the same bits).
// Standard GRU activations, zero initial state, both layers H hidden units (H % 16 == 0, H <= 256).
// ============================================================================
struct Gru2Params {
    const float *xw1;     // [T, B, 3H] layer-1 input projections incl. b_i1 (projection GEMM)
    const float *ut1, *bh1;          // layer 1: U1^T [3][Hj_p][Hk_p], b_h1 [3H]
    const float *wt2, *bi2;          // layer 2: W2^T in the same layout, b_i2 [3H]
    const float *ut2, *bh2;          // layer 2: U2^T, b_h2
    float *hbuf1, *hbuf2;            // tiled hand-off buffers (zeroed before the launch): layer 1 THREE slots [3][hb_floats], layer 2 two.
                                     // h1_k lives in slot (k + 1) % 3: layer 2 reads h1_{i-1} one phase AFTER layer 1 did, so with two slots
                                     // a workgroup that has already run ahead into layer 1 of iteration i + 1 (all it waits for is every
                                     // peer's h1_i) could publish h1_{i+1} over the h1_{i-1} a slower peer's layer-2 loads are still reading
    size_t hb_floats;
    float *out;                      // [B, T, H] layer-2 outputs (or [B, H] when !return_sequences)
    float *out1;                     // [B, T, H] layer-1 outputs, or NULL (not needed by the stack itself)
    unsigned *cnt;                   // [NBT] arrival counters
    unsigned *fault;
    unsigned long long spin_ticks;
    int B, T, H, Hj_p, Hk_p, NBT, NCT, b_base, return_sequences;
};

template <int NCH>
__global__ __launch_bounds__(512, 2) void gru2_persistent_kernel(Gru2Params p) {
    constexpr int G = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KP = NCH * REC_KC;
    constexpr int US = KP + 8, WS = KP + 4;       // row strides: U1^T / U2^T conflict-free, W2^T one 2-way conflict (LDS budget)
    float *U1s = smem;                            // [48][US]
    float *U2s = U1s + G * 16 * US;               // [48][US]
    float *W2s = U2s + G * 16 * US;               // [48][WS]
    float *red = W2s + G * 16 * WS;               // [4 slabs][2 waves][G][2][64] exchange

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w8 >> 2, slab = w8 & 3;       // grp: split-K half for layer 1; layer 2: 0 = the W2 product, 1 = the U2 product
    const int l15 = lane & 15, q = lane >> 4;
    const int bt = blockIdx.x % p.NBT, ct = blockIdx.x / p.NBT;
    const int b0 = p.b_base + bt * REC_BM, j0 = ct * REC_HN;
    const int GH = G * p.H;
    unsigned *cnt = p.cnt + (size_t)bt * 2 * RECP_CNT_STRIDE;       // two counters per batch tile: layer 1, layer 2
    const int b = b0 + slab * 16 + l15;
    const int j = j0 + q * 4 + grp * 2;           // this lane finishes hidden units j, j + 1 of row b, in both layers
    const bool row_ok = b < p.B;
    const bool ok0 = row_ok && j < p.H, ok1 = row_ok && j + 1 < p.H;

    {   // resident tiles
        constexpr int f4_per_row = KP / 4;
        constexpr int total = G * 16 * f4_per_row;
        for (int e = tid; e < total; e += 512) {
            const int r = e / f4_per_row, c4 = e % f4_per_row;
            const int g = r >> 4, jj = r & 15;
            float4 u1 = make_float4(0.f, 0.f, 0.f, 0.f), u2 = u1, w2 = u1;
            if (c4 * 4 < p.Hk_p) {
                const size_t off = ((size_t)g * p.Hj_p + j0 + jj) * p.Hk_p + c4 * 4;
                u1 = *reinterpret_cast<const float4 *>(p.ut1 + off);
                u2 = *reinterpret_cast<const float4 *>(p.ut2 + off);
                w2 = *reinterpret_cast<const float4 *>(p.wt2 + off);
            }
            *reinterpret_cast<float4 *>(&U1s[r * US + c4 * 4]) = u1;
            *reinterpret_cast<float4 *>(&U2s[r * US + c4 * 4]) = u2;
            *reinterpret_cast<float4 *>(&W2s[r * WS + c4 * 4]) = w2;
        }
    }
    float bh1[2][G], bh2[2][G], bi2[2][G];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bool in = j + e < p.H;
            bh1[e][g] = in ? p.bh1[g * p.H + j + e] : 0.0f;
            bh2[e][g] = in ? p.bh2[g * p.H + j + e] : 0.0f;
            bi2[e][g] = in ? p.bi2[g * p.H + j + e] : 0.0f;
        }
    float prev1[2] = {0.f, 0.f}, prev2[2] = {0.f, 0.f};      // own h of each layer (zero initial state)

    constexpr int KB = KP / 16;
    const int slab_abs = (b0 >> 4) + slab;
    const int hb_bytes = (int)(p.hb_floats * 4);
    // ONE descriptor over layer 1's three slots; the slot rides in the (scalar) offset
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void *)p.hbuf1, 0, 3 * hb_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2a = __builtin_amdgcn_make_buffer_rsrc((void *)p.hbuf2, 0, hb_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2b = __builtin_amdgcn_make_buffer_rsrc((void *)(p.hbuf2 + p.hb_floats), 0, hb_bytes, 0x00020000);
    const int h_lane_off = lane * 16;
    const int h_slab_off = slab_abs * KB * 1024;              // + (2 ch + half) * 1024
    const size_t h_pub_off = ((((size_t)slab_abs * KB + ct) * 4 + q) * 16 + l15) * 4 + grp * 2;
    float *my_red = red + ((slab * 2 + grp) * G * 2) * 64;
    const float *peer_red = red + ((slab * 2 + (1 - grp)) * G * 2) * 64;
    const int uoff = l15 * US + q * 4, woff = l15 * WS + q * 4;
    __syncthreads();

    // Two hand-off chains, one per layer, each with its own arrival counter: while the workgroup multiplies for one
    // layer, the other layer's publication -> arrival -> poll latency elapses (the two layers play the roles of the
    // ping-pong halves of the LSTM kernel, in one instruction stream).
    auto wait_for = [&](unsigned *c, unsigned target) {
        if (w8 == 0 && lane == 0) {
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            bool expired = p.spin_ticks == 0;
            while (!expired && __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                expired = __builtin_amdgcn_s_memrealtime() - t_start > p.spin_ticks;
            }
            if (expired) {
                __hip_atomic_fetch_or(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_or(c, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    };
    unsigned *cnt1 = cnt, *cnt2 = cnt + RECP_CNT_STRIDE;

    for (int i = 0; i <= p.T; ++i) {
        const bool do1 = i < p.T, do2 = i > 0;                // layer 1 runs step i, layer 2 step i - 1
        // =============================== layer 1, step i ===============================
        if (do1) {
            if (i > 0) wait_for(cnt1, (unsigned)p.NCT * (unsigned)i);      // h1_{i-1} is published
            v4u32 h1own[NCH];
            const int slot1 = (i % 3) * hb_bytes;                           // h1_{i-1}
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int so = slot1 + h_slab_off + (2 * ch + grp) * 1024;
                h1own[ch] = __builtin_amdgcn_raw_buffer_load_b128(r1, h_lane_off, so, 16);
            }
            // xW for this step: one 16-byte load per gate (the quad of hidden units this lane and its partner share)
            float xw1[2][G];
            {
                const float *xw = p.xw1 + ((size_t)i * p.B + b) * GH + (j & ~3);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok0) x4 = *reinterpret_cast<const float4 *>(xw + g * p.H);
                    xw1[0][g] = grp ? x4.z : x4.x;
                    xw1[1][g] = grp ? x4.w : x4.y;
                }
            }
            f32x4 acc1[G];
#pragma unroll
            for (int g = 0; g < G; ++g) acc1[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const float hv[4] = {__uint_as_float(h1own[ch].x), __uint_as_float(h1own[ch].y),
                                     __uint_as_float(h1own[ch].z), __uint_as_float(h1own[ch].w)};
                float4 uf[G];
#pragma unroll
                for (int g = 0; g < G; ++g) uf[g] = *reinterpret_cast<const float4 *>(&U1s[uoff + g * 16 * US + ch * REC_KC + grp * 16]);
#pragma unroll
                for (int sI = 0; sI < 4; ++sI)
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float u = sI == 0 ? uf[g].x : sI == 1 ? uf[g].y : sI == 2 ? uf[g].z : uf[g].w;
                        acc1[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(u, hv[sI], acc1[g], 0, 0, 0);
                    }
            }
            // split-K exchange: send the half the partner wave finishes
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int e = 0; e < 2; ++e) my_red[(g * 2 + e) * 64 + lane] = grp ? acc1[g][e] : acc1[g][2 + e];
            __syncthreads();
            float h1n[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float fin[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float other = peer_red[(g * 2 + e) * 64 + lane];
                    const float mine = grp ? acc1[g][2 + e] : acc1[g][e];
                    fin[g] = grp == 0 ? mine + other : other + mine;          // always (half 0) + (half 1)
                }
                // gru.c:144-186, same code as the single-layer kernel
                const float hz = fin[0] + bh1[e][0], hr = fin[1] + bh1[e][1], hh = fin[2] + bh1[e][2];
                const float z = nntk_fast_sigmoid(xw1[e][0] + hz);
                const float rg = nntk_fast_sigmoid(xw1[e][1] + hr);
                const float ht = nntk_fast_tanh(fmaf(rg, hh, xw1[e][2]));
                h1n[e] = fmaf(-z + 1.0f, ht, z * prev1[e]);
                prev1[e] = h1n[e];
            }
            float *hdst = p.hbuf1 + (size_t)((i + 1) % 3) * p.hb_floats + h_pub_off;       // h1_i -> slot (i + 1) % 3
            if (ok1) {
                const unsigned long long pk = ((unsigned long long)__float_as_uint(h1n[1]) << 32) | __float_as_uint(h1n[0]);
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(hdst), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (ok0) {
                __hip_atomic_store(reinterpret_cast<unsigned *>(hdst), __float_as_uint(h1n[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // R1: every storing wave drains ...
            __syncthreads();                                      // ... the workgroup meets (this also frees `red`) ...
            if (tid == 0) __hip_atomic_fetch_add(cnt1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... ONE lane arrives
            if (p.out1) {
                float *o = p.out1 + ((size_t)b * p.T + i) * p.H + j;
                if (ok1) *reinterpret_cast<float2 *>(o) = make_float2(h1n[0], h1n[1]);
                else if (ok0) o[0] = h1n[0];
            }
        }
        // =============================== layer 2, step i - 1 ===============================
        if (do2) {
            // its inputs: h1_{i-1} (arrival i on layer 1's counter: seen by this workgroup's own poll above, or, in the
            // last iteration, polled here) and h2_{i-2} (arrival i - 1 on layer 2's counter)
            if (!do1) wait_for(cnt1, (unsigned)p.NCT * (unsigned)i);
            if (i > 1) wait_for(cnt2, (unsigned)p.NCT * (unsigned)(i - 1));
            // one full-K product per wave of the slab: grp 0 multiplies W2^T . h1_{i-1}, grp 1 multiplies U2^T . h2_{i-2}
            v4u32 hfull[2 * NCH];
#pragma unroll
            for (int pc = 0; pc < 2 * NCH; ++pc) {
                const int so = h_slab_off + pc * 1024;
                if (grp == 0) hfull[pc] = __builtin_amdgcn_raw_buffer_load_b128(r1, h_lane_off, (i % 3) * hb_bytes + so, 16);      // h1_{i-1}
                else          hfull[pc] = ((i - 1) & 1) ? __builtin_amdgcn_raw_buffer_load_b128(r2b, h_lane_off, so, 16)
                                                        : __builtin_amdgcn_raw_buffer_load_b128(r2a, h_lane_off, so, 16);
            }
            f32x4 acc2[G];
#pragma unroll
            for (int g = 0; g < G; ++g) acc2[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pc = 0; pc < 2 * NCH; ++pc) {
                const float hv[4] = {__uint_as_float(hfull[pc].x), __uint_as_float(hfull[pc].y),
                                     __uint_as_float(hfull[pc].z), __uint_as_float(hfull[pc].w)};
                float4 wf[G];
#pragma unroll
                for (int g = 0; g < G; ++g)
                    wf[g] = grp == 0 ? *reinterpret_cast<const float4 *>(&W2s[woff + g * 16 * WS + pc * 16])
                                     : *reinterpret_cast<const float4 *>(&U2s[uoff + g * 16 * US + pc * 16]);
#pragma unroll
                for (int sI = 0; sI < 4; ++sI)
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float u = sI == 0 ? wf[g].x : sI == 1 ? wf[g].y : sI == 2 ? wf[g].z : wf[g].w;
                        acc2[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(u, hv[sI], acc2[g], 0, 0, 0);
                    }
            }
            // the W2 wave and the U2 wave swap the halves the other one finishes
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int e = 0; e < 2; ++e) my_red[(g * 2 + e) * 64 + lane] = grp ? acc2[g][e] : acc2[g][2 + e];
            __syncthreads();
            float h2n[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float xw2[G], hu2[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float other = peer_red[(g * 2 + e) * 64 + lane];
                    const float mine = grp ? acc2[g][2 + e] : acc2[g][e];
                    xw2[g] = (grp == 0 ? mine : other) + bi2[e][g];          // x W2 + b_i2, as the projection GEMM's epilogue
                    hu2[g] = grp == 0 ? other : mine;
                }
                const float hz = hu2[0] + bh2[e][0], hr = hu2[1] + bh2[e][1], hh = hu2[2] + bh2[e][2];
                const float z = nntk_fast_sigmoid(xw2[0] + hz);
                const float rg = nntk_fast_sigmoid(xw2[1] + hr);
                const float ht = nntk_fast_tanh(fmaf(rg, hh, xw2[2]));
                h2n[e] = fmaf(-z + 1.0f, ht, z * prev2[e]);
                prev2[e] = h2n[e];
            }
            float *hdst = p.hbuf2 + (size_t)(i & 1) * p.hb_floats + h_pub_off;      // h2_{i-1}: read next iteration at parity ((i + 1) - 1) & 1
            if (ok1) {
                const unsigned long long pk = ((unsigned long long)__float_as_uint(h2n[1]) << 32) | __float_as_uint(h2n[0]);
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(hdst), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (ok0) {
                __hip_atomic_store(reinterpret_cast<unsigned *>(hdst), __float_as_uint(h2n[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // layer-2 output (never read inside this launch) goes out after the arrival
            if (p.return_sequences || i == p.T) {
                float *o = p.return_sequences ? p.out + ((size_t)b * p.T + (i - 1)) * p.H + j : p.out + (size_t)b * p.H + j;
                if (ok1) *reinterpret_cast<float2 *>(o) = make_float2(h2n[0], h2n[1]);
                else if (ok0) o[0] = h2n[0];
            }
        }
    }
}

// Both layers' work areas: [hbuf1 x 3 slots][hbuf2 ping | pong][counters]
extern "C" size_t nntk_shim_gru2_work_floats(int B, int H) {
    return 5 * rec_hb_floats_fwd(B, H) + rec_cnt_words_fwd(B);
}

// 0 = launched; 1 = this shape / configuration is not taken by the fused kernel (caller runs the two layers one after
// the other); -1 = error.  d_xw1 [T, B, 3H] from the projection GEMM; d_wt2 = W2^T packed like U^T.
extern "C" int nntk_shim_gru2(const float *d_xw1, const float *d_ut1, const float *d_bh1, const float *d_wt2,
                              const float *d_bi2, const float *d_ut2, const float *d_bh2, float *d_out, float *d_out1,
                              float *d_work, int B, int T, int H, int return_sequences) {
    if (B <= 0 || T <= 0) return 0;
    const NntkOptions &opt = nntk_options();
    if (opt.rec_fused2 == 0 || opt.rec_persistent == 0 || nntk_persistent_disabled()) return 1;
    const int Hj_p = (H + 15) & ~15, Hk_p = (H + 31) & ~31;
    const int nch = Hk_p / REC_KC;
    if ((H % 4) != 0 || nch > 8) return 1;
    const int nch_p = nch <= 4 ? 4 : 8;
    const int KP = nch_p * REC_KC;
    const size_t lds = ((size_t)3 * 16 * (2 * (KP + 8) + (KP + 4)) + (size_t)4 * 2 * 3 * 2 * 64) * sizeof(float);
    if (lds > 160 * 1024) return 1;
    const int NCT = Hj_p / REC_HN;
    unsigned *fault = nntk_fault_word();
    if (!fault) return 1;
    const size_t hbmax = rec_hb_floats_fwd(B, H);
    const size_t hb_floats = (size_t)((B + 63) & ~63) * (size_t)KP;
    if (hb_floats * 4 * 3 >= 0x7ffffff0ULL) return 1;       // one descriptor spans layer 1's three slots
    auto kern = nch_p == 4 ? gru2_persistent_kernel<4> : gru2_persistent_kernel<8>;
    if (nntk_set_max_dynamic_lds((const void *)kern, lds)) return -1;
    // every workgroup of a launch must be resident: the grid comes from the runtime's occupancy answer for THIS kernel
    // at THIS LDS size (one per CU by design), not from the bare CU count
    const int resident = nntk_resident_blocks((const void *)kern, 512, lds, 1);
    const int tiles_per_launch = NCT > 0 ? resident / NCT : 0;
    if (tiles_per_launch < 1) return 1;
    unsigned *cnt = reinterpret_cast<unsigned *>(d_work + 5 * hbmax);
    const int nbt_total = (B + REC_BM - 1) / REC_BM;
    if (nntk_shim_memset(d_work, 0, 5 * hbmax * 4 + (size_t)nbt_total * 2 * RECP_CNT_STRIDE * sizeof(unsigned))) return -1;
    Gru2Params q;
    q.xw1 = d_xw1; q.ut1 = d_ut1; q.bh1 = d_bh1; q.wt2 = d_wt2; q.bi2 = d_bi2; q.ut2 = d_ut2; q.bh2 = d_bh2;
    q.hbuf1 = d_work; q.hbuf2 = d_work + 3 * hbmax; q.hb_floats = hb_floats;      // layer 1's slots are hb_floats (<= hbmax) apart
    q.out = d_out; q.out1 = d_out1; q.fault = fault;
    q.spin_ticks = (unsigned long long)(opt.rec_spin_us > 0 ? opt.rec_spin_us : 0) * 100ull;
    q.B = B; q.T = T; q.H = H; q.Hj_p = Hj_p; q.Hk_p = Hk_p; q.NCT = NCT; q.return_sequences = return_sequences;
    const int span = nntk_prof_span_begin(NNTK_SPAN_REC);
    nntk_persistent_launch_begin();
    for (int bt0 = 0; bt0 < nbt_total; bt0 += tiles_per_launch) {
        const int nbt = nbt_total - bt0 < tiles_per_launch ? nbt_total - bt0 : tiles_per_launch;
        q.NBT = nbt; q.b_base = bt0 * REC_BM;
        q.cnt = cnt + (size_t)bt0 * 2 * RECP_CNT_STRIDE;
        hipLaunchKernelGGL(kern, dim3((unsigned)(nbt * NCT)), dim3(512), lds, nntk_stream(), q);
    }
    const int copy_rc = nntk_fault_enqueue_copy();
    nntk_persistent_launch_end();
    if (copy_rc) return -1;
    nntk_prof_span_end(span, (nbt_total + tiles_per_launch - 1) / tiles_per_launch, T + 1);
    NNTK_LAUNCH_CHECK("gru2_persistent_kernel");
    nntk_set_last_rec_kernel(nch_p == 4 ? "gru2_persistent_kernel<4>" : "gru2_persistent_kernel<8>");
    return 0;
}

// h_0 from the caller's [B][H] layout into the tiled hand-off layout (parity 0)
__global__ __launch_bounds__(256) void rec_tile_h0_kernel(const float *src, float *dst, int B, int H, int KB) {
    const long total = (long)B * H;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int b = (int)(e / H), k = (int)(e % H);
        dst[(((size_t)(b >> 4) * KB + (k >> 4)) * 4 + ((k >> 2) & 3)) * 64 + (b & 15) * 4 + (k & 3)] = src[e];
    }
}

static int act_ok(int a) {
    return a == NNTK_ACT_IDENTITY || a == NNTK_ACT_SIGMOID || a == NNTK_ACT_TANH || a == NNTK_ACT_RELU;
}

// ============================================================================
// Streaming variant: ONE sequence (the reference's own call shape: GRUApplyInference / LSTMApplyInference with carried
// state, gru.c:189-204, lstm.c:241-268), a few timesteps per call.  At batch 1 the step is a pair of GEMVs over 1..5 MB
// of weights -- nothing for an MFMA tile to do, and every fixed cost of the batch path (projection GEMM launch, two
// memsets, the 128 KB U^T -> LDS reload of 32 co-resident workgroups, counters, the blocking upload and download)
// is pure latency.  Here a call is T launches of ONE small kernel and no copies: x_t is read from, and h_t written
// to, pinned host memory directly; state ping-pongs in device memory.
//
// Bit-compatible with the batch path by construction: each (hidden unit j, gate g) is three sequential fmaf chains
// that visit k in exactly the order the MFMA kernels do --
//   xW  : the projection GEMM's order (conv1d.hip): 8-deep blocks, inside a block k = 0,4,1,5,2,6,3,7, then + b_i
//   hU  : the persistent kernel's two split-K halves: for chunk, for s, for q: k = 32 chunk + 16 half + 4 q + s;
//         then (half 0) + (half 1), then xW + (hU + b_h)
// (an f32 MFMA is bit for bit a k-ordered fmaf chain), followed by the same gate code.  One lane per chain, 16 lanes
// per hidden unit (4 gates x {xW, hU half 0, hU half 1, idle}), 16 hidden units per 256-thread workgroup.
// ============================================================================
struct RecStreamParams {
    const float *x;       // [in] this step's input row (pinned host or device memory)
    const float *wp;      // [G*H (padded)][Kin_p] packed input weights, K-contiguous (nntk_upload_gemm_weights)
    const float *bi;      // [G*H]
    const float *ut;      // [G][Hj_p][Hk_p] U^T
    const float *bh;      // [G*H] or NULL
    const float *h_prev;  // [H]
    const float *c_prev;  // [H] (LSTM)
    float *h_next, *c_next;
    float *out;           // [H] this step's output row or NULL
    int in, Kin_p, H, Hj_p, Hk_p, nch_p;
    int a0, a1, a2, a3, a4;
    float s0, s1, s2, s3, s4;
    unsigned *done_cnt;   // device counter of finished workgroups (last launch of a call only, else NULL)
    unsigned *flag;       // pinned host word: the last workgroup to finish writes `seq` into it
    unsigned seq;
};

template <int G, bool IS_LSTM>
__global__ __launch_bounds__(256) void rec_stream_step_kernel(RecStreamParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp16 = lane >> 4, l = lane & 15;
    const int g = l >> 2, role = l & 3;
    const int j = blockIdx.x * 16 + wave * 4 + grp16;
    const bool live = j < p.H && g < G && role < 3;
    // x_t (possibly in pinned host memory: one coalesced read over PCIe instead of one per chain element) and h_{t-1}
    // are staged once in LDS; every chain then reads them as LDS broadcasts
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *xs = sm, *hs = sm + p.Kin_p;
    for (int k = threadIdx.x; k < p.Kin_p; k += 256) xs[k] = k < p.in ? p.x[k] : 0.0f;
    for (int k = threadIdx.x; k < p.Hk_p; k += 256) hs[k] = k < p.H ? p.h_prev[k] : 0.0f;
    __syncthreads();
    float acc = 0.0f;
    if (live && role == 0) {
        const float4 *w = reinterpret_cast<const float4 *>(p.wp + (size_t)(g * p.H + j) * p.Kin_p);
        for (int kb = 0; kb < p.Kin_p; kb += 8) {
            const float4 w0 = w[kb >> 2], w1 = w[(kb >> 2) + 1];
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            const float4 x0 = *reinterpret_cast<const float4 *>(xs + kb), x1 = *reinterpret_cast<const float4 *>(xs + kb + 4);
            const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {                   // the GEMM's MFMA u multiplies k = u (lane half 0), then k = 4 + u
                acc = fmaf(xv[u], wv[u], acc);
                acc = fmaf(xv[4 + u], wv[4 + u], acc);
            }
        }
        acc += p.bi[g * p.H + j];
    } else if (live) {
        const int half = role - 1;
        const float4 *u4 = reinterpret_cast<const float4 *>(p.ut + ((size_t)g * p.Hj_p + j) * p.Hk_p);
        const int nch = p.Hk_p / REC_KC;                    // chunks that exist in memory; the kernel's padding chunks add + 0
        for (int ch = 0; ch < nch; ++ch) {
            const int k0 = ch * REC_KC + half * 16;
            float4 uq[4];
            float hq[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uq[q] = u4[(k0 >> 2) + q];
                const float4 h4 = *reinterpret_cast<const float4 *>(hs + k0 + 4 * q);
                hq[q][0] = h4.x; hq[q][1] = h4.y; hq[q][2] = h4.z; hq[q][3] = h4.w;
            }
#pragma unroll
            for (int sI = 0; sI < 4; ++sI)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float uv = sI == 0 ? uq[q].x : sI == 1 ? uq[q].y : sI == 2 ? uq[q].z : uq[q].w;
                    acc = fmaf(uv, hq[q][sI], acc);
                }
        }
    }
    // gather the three chains of every gate on the hidden unit's first lane
    const int base = lane & ~15;
    float xw[G], fin[G];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
        xw[gg] = __shfl(acc, base + 4 * gg);
        const float h0 = __shfl(acc, base + 4 * gg + 1), h1 = __shfl(acc, base + 4 * gg + 2);
        fin[gg] = h0 + h1;                                   // always (half 0) + (half 1)
    }
    if (l == 0 && j < p.H) {
        float bh[G];
#pragma unroll
        for (int gg = 0; gg < G; ++gg) bh[gg] = p.bh ? p.bh[gg * p.H + j] : 0.0f;
        float hn;
        if constexpr (G == 1) {
            hn = nntk_gate_act(p.a0, xw[0] + (fin[0] + bh[0]), p.s0);
        } else if constexpr (!IS_LSTM) {
            const float prev = hs[j];
            const float hz = fin[0] + bh[0], hr = fin[1] + bh[1], hh = fin[2] + bh[2];
            const float z = nntk_gate_act(p.a0, xw[0] + hz, p.s0);
            const float rg = nntk_gate_act(p.a2, xw[1] + hr, p.s2);
            const float ht = nntk_gate_act(p.a1, fmaf(rg, hh, xw[2]), p.s1);
            hn = fmaf(-z + 1.0f, ht, z * prev);
        } else {
            const float prev = p.c_prev[j];
            const float zi = xw[0] + (fin[0] + bh[0]);
            const float zf = xw[1] + (fin[1] + bh[1]);
            const float zg = xw[2] + (fin[2] + bh[2]);
            const float zo = xw[G - 1] + (fin[G - 1] + bh[G - 1]);
            const float ig = nntk_gate_act(p.a0, zi, p.s0);
            const float fg = nntk_gate_act(p.a1, zf, p.s1);
            const float gg_ = nntk_gate_act(p.a2, zg, p.s2);
            const float og = nntk_gate_act(p.a3, zo, p.s3);
            const float cn = fmaf(fg, prev, ig * gg_);
            p.c_next[j] = cn;
            hn = og * nntk_gate_act(p.a4, cn, p.s4);
        }
        p.h_next[j] = hn;
        if (p.out) p.out[j] = hn;
    }
    // completion signal for the host (last launch of a call): outputs first, system-wide, then count in; the
    // workgroup that counts in last raises the flag the host is spinning on
    if (p.done_cnt) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned n = __hip_atomic_fetch_add(p.done_cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (n == gridDim.x - 1) {
                __hip_atomic_store(p.done_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                __hip_atomic_store(p.flag, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

template <int G, bool IS_LSTM>
static int run_stream(const float *x, const float *d_wp, const float *d_bi, const float *d_ut, const float *d_bh,
                      float *const h[2], float *const c[2], int cur, float *out, int T, int in, int H,
                      int return_sequences, const int *acts, const float *scales, int nacts,
                      unsigned *d_done, unsigned *flag, unsigned seq) {
    for (int i = 0; i < nacts; ++i)
        if (!act_ok(acts[i]))
            return nntk_fail_msg("recurrent: gate activation must be one of the built-in identity/sigmoid/tanh/relu");
    RecStreamParams p;
    p.wp = d_wp; p.bi = d_bi; p.ut = d_ut; p.bh = d_bh;
    p.in = in; p.H = H;
    int cin_p, cout_p;
    nntk_shim_conv_pack_sizes(in, G * H, 1, &cin_p, &cout_p);
    p.Kin_p = cin_p;
    p.Hj_p = (H + 15) & ~15;
    p.Hk_p = (H + 31) & ~31;
    p.nch_p = p.Hk_p / REC_KC;
    p.a0 = acts[0]; p.a1 = nacts > 1 ? acts[1] : 0; p.a2 = nacts > 2 ? acts[2] : 0;
    p.a3 = nacts > 3 ? acts[3] : 0; p.a4 = nacts > 4 ? acts[4] : 0;
    auto sc = [&](int i) { return (scales && i < nacts && acts[i] == NNTK_ACT_RELU) ? scales[i] : 1.0f; };
    p.s0 = sc(0); p.s1 = sc(1); p.s2 = sc(2); p.s3 = sc(3); p.s4 = sc(4);
    p.flag = flag; p.seq = seq;
    const unsigned grid = (unsigned)((H + 15) / 16);
    for (int t = 0; t < T; ++t) {
        p.x = x + (size_t)t * in;
        p.h_prev = h[(cur + t) & 1]; p.h_next = h[(cur + t + 1) & 1];
        p.c_prev = c[(cur + t) & 1]; p.c_next = c[(cur + t + 1) & 1];
        p.out = return_sequences ? out + (size_t)t * H : (t == T - 1 ? out : nullptr);
        p.done_cnt = t == T - 1 ? d_done : nullptr;
        hipLaunchKernelGGL((rec_stream_step_kernel<G, IS_LSTM>), dim3(grid), dim3(256), (size_t)(p.Kin_p + p.Hk_p) * sizeof(float), nntk_stream(), p);
    }
    NNTK_LAUNCH_CHECK("rec_stream_step_kernel");
    return 0;
}

// One sequence, T steps, state h[cur] (c[cur]) -> h[(cur + T) & 1]; x and out may be pinned host memory.  The last
// launch stores `seq` to *flag (pinned host word) once every output byte is visible system-wide.
extern "C" int nntk_shim_rec_stream(int G, int is_lstm, const float *x, const float *d_wp, const float *d_bi,
                                    const float *d_ut, const float *d_bh, float *d_h0, float *d_h1, float *d_c0, float *d_c1,
                                    int cur, float *out, int T, int in, int H, int return_sequences,
                                    const int *acts, const float *scales, unsigned *d_done, unsigned *flag, unsigned seq) {
    if (T <= 0) return 0;
    float *const h[2] = {d_h0, d_h1};
    float *const c[2] = {d_c0, d_c1};
    if (G == 1) return run_stream<1, false>(x, d_wp, d_bi, d_ut, d_bh, h, c, cur, out, T, in, H, return_sequences, acts, scales, 1, d_done, flag, seq);
    if (G == 3) return run_stream<3, false>(x, d_wp, d_bi, d_ut, d_bh, h, c, cur, out, T, in, H, return_sequences, acts, scales, 3, d_done, flag, seq);
    if (G == 4 && is_lstm) return run_stream<4, true>(x, d_wp, d_bi, d_ut, d_bh, h, c, cur, out, T, in, H, return_sequences, acts, scales, 5, d_done, flag, seq);
    return nntk_fail_msg("rec_stream: unsupported cell");
}

// floats per parity of the h ping-pong area: the persistent kernel's tiled layout pads rows to 64
// and k to a multiple of 128 (4 chunks); the per-step kernels use the first B*H floats of each half
static size_t rec_hb_floats(int B, int H) {
    return (size_t)((B + 63) & ~63) * (size_t)((H + 127) & ~127);
}

// arrival counters: up to two per batch tile (ping-pong halves), each on its own 256-B line
static size_t rec_cnt_words(int B) {
    const size_t nbt = (size_t)(B + REC_BM - 1) / REC_BM;
    return (2 * nbt < 256 ? 256 : 2 * nbt) * RECP_CNT_STRIDE;
}

static size_t rec_hb_floats_fwd(int B, int H) { return rec_hb_floats(B, H); }
static size_t rec_cnt_words_fwd(int B) { return rec_cnt_words(B); }

extern "C" size_t nntk_shim_recurrent_work_floats(int B, int H) {
    // h ping | h pong | c | arrival counters of the persistent kernel
    return 2 * rec_hb_floats(B, H) + (size_t)B * H + rec_cnt_words(B);
}

template <int G, bool IS_LSTM>
static int run_recurrent(const float *d_xw, const float *d_ut, const float *d_bh, const float *d_h0,
                         const float *d_c0, float *d_out, float *d_hT, float *d_cT, float *d_work,
                         int B, int T, int H, int return_sequences, const int *acts, const float *scales, int nacts) {
    for (int i = 0; i < nacts; ++i)
        if (!act_ok(acts[i]))
            return nntk_fail_msg("recurrent: gate activation must be one of the built-in identity/sigmoid/tanh/relu");
    const NntkOptions &opt = nntk_options();
    const size_t BH = (size_t)B * H;
    const size_t hbmax = rec_hb_floats(B, H);
    float *hbuf[2] = {d_work, d_work + hbmax};
    float *cbuf = d_work + 2 * hbmax;
    if (IS_LSTM) {
        if (d_c0) { if (nntk_shim_copy_d2d(cbuf, d_c0, BH * 4)) return -1; }
        else      { if (nntk_shim_memset(cbuf, 0, BH * 4)) return -1; }
    }
    RecParams p;
    p.ut = d_ut; p.bh = d_bh; p.c = cbuf;
    p.B = B; p.H = H;
    p.Hj_p = (H + 15) & ~15;
    p.Hk_p = (H + 31) & ~31;
    p.a0 = acts[0]; p.a1 = nacts > 1 ? acts[1] : 0; p.a2 = nacts > 2 ? acts[2] : 0;
    p.a3 = nacts > 3 ? acts[3] : 0; p.a4 = nacts > 4 ? acts[4] : 0;
    auto sc = [&](int i) { return (scales && i < nacts && acts[i] == NNTK_ACT_RELU) ? scales[i] : 1.0f; };
    p.s0 = sc(0); p.s1 = sc(1); p.s2 = sc(2); p.s3 = sc(3); p.s4 = sc(4);
    dim3 grid((unsigned)((B + REC_BM - 1) / REC_BM), (unsigned)(p.Hj_p / REC_HN));
    // ---- persistent path: one launch for the whole sequence when U^T fits in LDS ----
    {
        const bool want = opt.rec_persistent != 0 && !nntk_persistent_disabled();
        const int NCT = p.Hj_p / REC_HN;
        const int nch = p.Hk_p / REC_KC;
        const int nch_p = nch <= 4 ? 4 : nch <= 8 ? 8 : nch <= 12 ? 12 : 16;      // compiled K depths (x32)
        const size_t lds = ((size_t)G * 16 * (nch_p * REC_KC + 8) + (size_t)4 * 2 * G * 2 * 64 + 16) * sizeof(float);
        const size_t hb_floats = (size_t)((B + 63) & ~63) * (size_t)(nch_p * REC_KC);      // <= hbmax
        unsigned *fault = want ? nntk_fault_word() : nullptr;
        if (want && fault && (H % 4) == 0 && lds <= 160 * 1024 && p.Hk_p <= RECP_MAXCH * REC_KC && NCT >= 1 &&
            NCT <= nntk_cu_count() && hb_floats * 4 < 0x3ffffff0ULL) {
            void (*kern)(RecPParams);
            int xwm = opt.rec_xw;
            const bool std_acts = G == 1 ? p.a0 == NNTK_ACT_TANH
                                 : IS_LSTM ? (p.a0 == NNTK_ACT_SIGMOID && p.a1 == NNTK_ACT_SIGMOID && p.a2 == NNTK_ACT_TANH &&
                                              p.a3 == NNTK_ACT_SIGMOID && p.a4 == NNTK_ACT_TANH)
                                           : (p.a0 == NNTK_ACT_SIGMOID && p.a1 == NNTK_ACT_TANH && p.a2 == NNTK_ACT_SIGMOID);
            // ping-pong halves pay when the K loop is long (see the kernel's header); option rec_pingpong = 0/1 overrides
            // measured (ms, classic -> ping-pong): LSTM H=320/384/448/512 5.5->4.3 / 5.8->4.5 / 6.1->5.5 / 14.1->11.8,
            // GRU H=320/384/512 4.7->4.4 / 9.0->8.2 / 5.8->4.5; LSTM-256 +-0, GRU-256 +5..20 %, RNN-512 +24 %: the K loop
            // must carry >= 36 groups of 4 MFMAs per wave for the alternation to pay
            const bool pp = std_acts && nch_p >= 12 && G * nch_p >= 36 && opt.rec_pingpong != 0;
            if (xwm < 0) xwm = pp ? 0 : 1;      // measured: ping-pong LSTM-512 11.8 (XW 0) vs 13.1 ms; classic GRU-256 7.82 vs 7.63 (XW 1)
#define REC_PICK(N) (!std_acts ? rec_persistent_kernel<G, IS_LSTM, N, 0, false, false> \
                     : xwm == 1 ? rec_persistent_kernel<G, IS_LSTM, N, 1, false, true> : rec_persistent_kernel<G, IS_LSTM, N, 0, false, true>)
            if (pp && nch_p == 12) kern = xwm == 1 ? rec_persistent_kernel<G, IS_LSTM, 12, 1, true, true> : rec_persistent_kernel<G, IS_LSTM, 12, 0, true, true>;
            else if (pp)          kern = xwm == 1 ? rec_persistent_kernel<G, IS_LSTM, 16, 1, true, true> : rec_persistent_kernel<G, IS_LSTM, 16, 0, true, true>;
            else if (nch_p == 4)  kern = REC_PICK(4);
            else if (nch_p == 8)  kern = REC_PICK(8);
            else if (nch_p == 12) kern = REC_PICK(12);
            else                  kern = REC_PICK(16);
            if (nntk_set_max_dynamic_lds((const void *)kern, lds)) return -1;
            // every workgroup of a launch must be resident: the grid comes from the runtime's occupancy answer for THIS
            // kernel at THIS LDS size (one per CU by design), not from the bare CU count
            const int tiles_per_launch = nntk_resident_blocks((const void *)kern, 512, lds, 1) / NCT;
            if (tiles_per_launch >= 1) {
            unsigned *cnt = reinterpret_cast<unsigned *>(d_work + 2 * hbmax + BH);
            // tiled hand-off buffers: both parities cleared (the padding must stay zero), h_0 tiled into parity 0
            if (nntk_shim_memset(d_work, 0, 2 * hbmax * 4)) return -1;
            if (d_h0) {
                long g = (long)((BH + 255) / 256);
                if (g > 2048) g = 2048;
                hipLaunchKernelGGL(rec_tile_h0_kernel, dim3((unsigned)g), dim3(256), 0, nntk_stream(), d_h0, d_work, B, H,
                                   nch_p * REC_KC / 16);
            }
            RecPParams q;
            q.xw = d_xw; q.ut = d_ut; q.bh = d_bh; q.hbuf = d_work; q.c = cbuf; q.out = d_out; q.cnt = cnt;
            q.h0 = d_h0; q.hT = d_hT; q.hb_floats = hb_floats;
            q.B = B; q.T = T; q.H = H; q.Hj_p = p.Hj_p; q.Hk_p = p.Hk_p; q.NCT = NCT;
            q.return_sequences = return_sequences;
            q.a0 = p.a0; q.a1 = p.a1; q.a2 = p.a2; q.a3 = p.a3; q.a4 = p.a4;
            q.s0 = p.s0; q.s1 = p.s1; q.s2 = p.s2; q.s3 = p.s3; q.s4 = p.s4;
            q.fault = fault;
            q.spin_ticks = (unsigned long long)(opt.rec_spin_us > 0 ? opt.rec_spin_us : 0) * 100ull;     // s_memrealtime: 100 MHz; 0 = inject a fault
            const int ncnt = pp ? 2 : 1;      // arrival counters per batch tile
#ifdef NNTK_REC_STAMPS
            q.stamp = nullptr;
            const char *stamp_path = getenv("NNTK_REC_STAMP_FILE");
            if (stamp_path) {
                if (hipMalloc((void **)&q.stamp, (size_t)(2 * T * 8 + 8 * T * 2) * 8) != hipSuccess) return nntk_fail_msg("stamp alloc");
                (void)hipMemset(q.stamp, 0, (size_t)(2 * T * 8 + 8 * T * 2) * 8);
            }
#endif
            const int nbt_total = (B + REC_BM - 1) / REC_BM;
            const int span = nntk_prof_span_begin(NNTK_SPAN_REC);
            // all arrival counters of all launches are cleared by ONE memset (each launch gets its own slice)
            if (nntk_shim_memset(cnt, 0, (size_t)nbt_total * ncnt * RECP_CNT_STRIDE * sizeof(unsigned))) return -1;
            nntk_persistent_launch_begin();
            for (int bt0 = 0; bt0 < nbt_total; bt0 += tiles_per_launch) {
                const int nbt = nbt_total - bt0 < tiles_per_launch ? nbt_total - bt0 : tiles_per_launch;
                q.NBT = nbt; q.b_base = bt0 * REC_BM;
                q.cnt = cnt + (size_t)bt0 * ncnt * RECP_CNT_STRIDE;
                hipLaunchKernelGGL(kern, dim3((unsigned)(nbt * NCT)), dim3(512), lds, nntk_stream(), q);
            }
            const int copy_rc = nntk_fault_enqueue_copy();
            nntk_persistent_launch_end();
            if (copy_rc) return -1;
            nntk_prof_span_end(span, (nbt_total + tiles_per_launch - 1) / tiles_per_launch, T);
#ifdef NNTK_REC_STAMPS
            if (q.stamp) {
                (void)hipStreamSynchronize(nntk_stream());
                unsigned long long *hs = (unsigned long long *)malloc((size_t)(2 * T * 8 + 8 * T * 2) * 8);
                (void)hipMemcpy(hs, q.stamp, (size_t)(2 * T * 8 + 8 * T * 2) * 8, hipMemcpyDeviceToHost);
                FILE *f = fopen(stamp_path, "wb");
                if (f) { fwrite(hs, 8, (size_t)(2 * T * 8 + 8 * T * 2), f); fclose(f); }
                free(hs); (void)hipFree(q.stamp);
            }
#endif
            NNTK_LAUNCH_CHECK("rec_persistent_kernel");
            nntk_set_last_rec_kernel(G == 1 ? "rec_persistent_kernel<1,RNN>" : IS_LSTM ? "rec_persistent_kernel<4,LSTM>" : "rec_persistent_kernel<3,GRU>");
            if (IS_LSTM && d_cT) { if (nntk_shim_copy_d2d(d_cT, cbuf, BH * 4)) return -1; }
            return 0;
            }
        }
    }
    if (d_h0) { if (nntk_shim_copy_d2d(hbuf[0], d_h0, BH * 4)) return -1; }
    else      { if (nntk_shim_memset(hbuf[0], 0, BH * 4)) return -1; }
    // split-K groups per workgroup: 2 (two waves per SIMD) unless overridden for A/B runs
    const int ng = opt.rec_groups == 1 ? 1 : 2;
    constexpr size_t buf_floats = (size_t)REC_BM * REC_LS + (size_t)G * REC_HN * REC_LS;
    const size_t lds1 = 2 * buf_floats * sizeof(float);
    size_t lds2 = 4 * buf_floats * sizeof(float);
    const size_t red2 = (size_t)8 * G * 4 * 64 * sizeof(float);
    if (red2 > lds2) lds2 = red2;
    if (ng == 2 && lds2 > 64 * 1024) {
        if (nntk_set_max_dynamic_lds((const void *)rec_step_kernel<G, IS_LSTM, 2>, lds2)) return -1;
    }
    const int span = nntk_prof_span_begin(NNTK_SPAN_REC);
    for (int t = 0; t < T; ++t) {
        p.xw = d_xw + (size_t)t * B * G * H;
        p.h_prev = hbuf[t & 1];
        p.h_next = hbuf[(t + 1) & 1];
        if (return_sequences) { p.out = d_out + (size_t)t * H; p.out_ld = (long)T * H; }
        else if (t == T - 1)  { p.out = d_out; p.out_ld = H; }
        else                  { p.out = nullptr; p.out_ld = 0; }
        if (ng == 2) hipLaunchKernelGGL((rec_step_kernel<G, IS_LSTM, 2>), grid, dim3(512), lds2, nntk_stream(), p);
        else         hipLaunchKernelGGL((rec_step_kernel<G, IS_LSTM, 1>), grid, dim3(256), lds1, nntk_stream(), p);
    }
    nntk_prof_span_end(span, T, T);
    NNTK_LAUNCH_CHECK("rec_step_kernel");
    nntk_set_last_rec_kernel(G == 1 ? "rec_step_kernel<1,RNN>" : IS_LSTM ? "rec_step_kernel<4,LSTM>" : "rec_step_kernel<3,GRU>");
    if (d_hT) { if (nntk_shim_copy_d2d(d_hT, hbuf[T & 1], BH * 4)) return -1; }
    if (IS_LSTM && d_cT) { if (nntk_shim_copy_d2d(d_cT, cbuf, BH * 4)) return -1; }
    return 0;
}

extern "C" int nntk_shim_gru(const float *d_xw, const float *d_ut, const float *d_bh, const float *d_h0,
                             float *d_out, float *d_hT, float *d_work, int B, int T, int H,
                             int return_sequences, const int acts[3], const float act_scales[3]) {
    if (B <= 0 || T <= 0) return 0;
    return run_recurrent<3, false>(d_xw, d_ut, d_bh, d_h0, nullptr, d_out, d_hT, nullptr, d_work, B, T, H,
                                   return_sequences, acts, act_scales, 3);
}

extern "C" int nntk_shim_rnn(const float *d_xw, const float *d_ut, const float *d_bh, const float *d_h0,
                             float *d_out, float *d_hT, float *d_work, int B, int T, int H,
                             int return_sequences, int act, float act_scale) {
    if (B <= 0 || T <= 0) return 0;
    const int acts[1] = {act};
    const float scales[1] = {act_scale};
    return run_recurrent<1, false>(d_xw, d_ut, d_bh, d_h0, nullptr, d_out, d_hT, nullptr, d_work, B, T, H,
                                   return_sequences, acts, scales, 1);
}

extern "C" int nntk_shim_lstm(const float *d_xw, const float *d_ut, const float *d_bh, const float *d_h0,
                              const float *d_c0, float *d_out, float *d_hT, float *d_cT, float *d_work,
                              int B, int T, int H, int return_sequences, const int acts[5], const float act_scales[5]) {
    if (B <= 0 || T <= 0) return 0;
    return run_recurrent<4, true>(d_xw, d_ut, d_bh, d_h0, d_c0, d_out, d_hT, d_cT, d_work, B, T, H,
                                  return_sequences, acts, act_scales, 5);
}
