// recurrent_fk.hip -- K4d: persistent GRU / LSTM on the bf16 MFMA, weights register-resident, input projection fused, and
// NO SPLIT-K ("full K"): every wavefront owns ONE 32-row MFMA tile (8 hidden units x 4 gate slots) of ONE 32-row half of the
// batch tile and multiplies the whole K = [x_t (in) | h_{t-1} (H)] of it.  Reference semantics: layers/gru.c:129-187 (cell),
// :246-293 (batch forward); layers/lstm.c:185-239, :426-475.
//
// Why another family.  recurrent_rr.hip splits K over the four wavefronts of a workgroup; its stamps at the H <= 256 shapes
// (profiles/r04_rr_pending_stamps.log: 4.4 k cycles per half-step against 2.3 k of MFMA time) show no single stall left but the
// split-K frame itself: two workgroup barriers, 64 KB of partial sums through LDS, a reduce slice and a publication slice per
// half-step, spread over every k step.  Here there is no partial sum to exchange:
//
//   * workgroup = 64 batch rows x 16 hidden units, as before (grid = batch tiles x H / 16 <= resident workgroups, one per CU);
//     wavefront w = (half hf = w >> 1, unit group ug = w & 1): tile row 8 q + e <-> gate slot q, hidden unit 16 ct + 8 ug + e.  With
//     the 32x32 MFMA's D layout lane (n, kh) then holds all four gate slots of FOUR CONSECUTIVE hidden units 4 kh .. 4 kh + 3 of
//     batch row n: the gate arithmetic, the 3-way bf16 split and the stores of a step all happen in the accumulator's own lanes --
//     no reduce, no LDS exchange, no barrier anywhere in the step loop.
//   * U's three images of the tile (H = 256: 192 registers, pinned to the accumulator half of the file) stay in registers for the
//     whole launch; W's images sit in LDS (48 KB at in = 128, 96 KB at in = 256), shared by the two halves.
//   * the operand (h_{t-1} and x_t in frag3 form: three 1 KB fragments per 16-deep k step) is the same for the two wavefronts of a
//     half, and fetching it twice would double the L2 -> CU traffic (290-390 KB per CU and step: the L1 fill rate alone would take
//     as long as the MFMAs).  So the two wavefronts of a half share the fetch: k steps go in PAIRS, each wavefront requests the
//     fragments of ONE k step of every pair global -> registers (counted waits, three pairs ahead), multiplies them from those
//     registers and leaves a copy in a small LDS ring for its partner, which multiplies its own k step first and the partner's
//     second.  Partners meet through a sequence word per ring slot (LDS is in-order per wavefront: data, then the word; the reader
//     takes the word first) -- never through s_barrier, so the two halves of a workgroup are independent streams.
//     Ring reuse needs no back-pressure (see FK_RING).
//   * one stream per wavefront: nothing hides the inter-workgroup hand-off chain (gates -> publication -> visible -> fetched) except
//     the x part of the NEXT step, which does not depend on it: a step is [h k steps of t] -> [x k steps of t + 1 || gates,
//     publication of t], on two accumulator sets.
//
// Hand-off: the pending-pattern protocol of recurrent_rr.hip (KH = 4 shapes), unchanged: the T-deep frag3 tensor that is also the layer
// output; an unwritten block reads 0xffffffff in every word; a consumer looks at every word of the fragments it fetched and requests
// them again while one is pending; the owner of a block marks step t + 2 while it publishes step t (here the SAME wavefront issues the
// marks of t + 2, then -- a whole step of consumed operand loads, whose in-order vmcnt waits retire them, later -- the data of t + 1).
// Every wavefront publishes its own 8 units: three 8-byte write-through stores per lane (a row's 16-byte fragment slot is filled by
// the two lanes (n, 0), (n, 1)).
//
// Numerics: the split-bf16 x 3 contraction, ONE accumulation chain per output: x k steps in ascending order, then h k steps
// pairwise (own k step of the pair first).  Not bit-identical to recurrent_rr.hip (another summation order), same tolerance; a row's
// bits do not depend on the batch tile or shard it arrives in.
#include "recurrent_rr_common.hpp"

typedef unsigned fk_v2u __attribute__((ext_vector_type(2)));
// f(integral_constant<int, LO>), .., f(integral_constant<int, HI - 1>): the groups of a step as separate instantiations ("#pragma unroll"
// gives up on a body this large -- "unrolled size is too large" -- and the staging registers, indexed by the group, then live in scratch)
template <int LO, int HI, class F>
__device__ __forceinline__ void fk_for(F &&f) {
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        fk_for<LO + 1, HI>(f);
    }
}
// Ring slots per wavefront.  4 (default): a wavefront writes group g + 1's k step (slot of group g - 3) after it has checked its partners' words of
// group g; each partner wrote that word after ITS check of group g - 1, i.e. after all its reads of group g - 2 had been issued (LDS runs a
// wavefront's operations in order): three slots would do, four keep the slot index a compile-time constant, and the ring reads of a group may
// be spread over the group.  2 (32 units x 256 inputs: W's mid / low images take 128 KB of LDS): every ring read of a group is issued in the
// group's first block, BEFORE the hand-over of the next group's k step -- the same argument then holds one group shorter.  (Two slots with spread
// reads and the hand-over at the END of a group were measured slower: the partners find the word missing and spin, profiles/r05_fk_ring.log.)
__host__ __device__ constexpr int fk_ring(int NKX, int NW) { return NW == 4 && NKX > 8 ? 2 : 4; }
#ifndef FK_ND_4
#define FK_ND_4 3                  // staging sets (groups of four k steps requested ahead) of the NW = 4 kernels
#endif
#ifndef FK_ND_4W
#define FK_ND_4W 4                 // ... of the NW = 4 kernels with a 256-wide input (eight groups per step)
#endif
#ifndef FK_LEAD_CAP
#define FK_LEAD_CAP 0              // 1: the h groups of a step are requested behind the finish of the step before, and the step's last group waits for the
                                   // next step's head operand at its END; 0: everything ND groups ahead, the look always in a group's first block.
                                   // (1 measured slower at in = 128 -- 4.66 vs 4.44 ms -- and equal at in = 256: profiles/r05_fk_ablate.log)
#endif
#ifndef FK_FIN_SLICED
#define FK_FIN_SLICED 0
#endif
#ifndef FK_FIN_VALU
#define FK_FIN_VALU 5              // vector-ALU instructions of the finish per MFMA of the x groups that carry it
#endif
#ifndef FK_ND_2
#define FK_ND_2 8                  // ... (pairs of k steps) of the NW = 2 kernels
#endif

#ifdef NNTK_REC_STAMPS
// fine stamps inside ONE group (FK_FINE_GS): s_memtime by inline asm into SGPR pairs, no wait (hipcc does not know it is a scalar-memory
// operation; its own lgkmcnt waits only get stricter), read back after the group
#ifndef FK_FINE_GS
#define FK_FINE_GS 2
#endif
#define FK_FINE(gs, k) do { if ((gs) == FK_FINE_GS) asm volatile("s_memtime %0" : "=s"(fine[k])); } while (0)
#define FK_FINE_OUT(gs, t) do { if ((gs) == FK_FINE_GS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        if (p.stamp && blockIdx.x == 0 && w == 0 && lane == 0 && (t) >= 0) { for (int k_ = 0; k_ < 10; ++k_) p.stamp[(size_t)(t) * 64 + 48 + k_] = fine[k_]; } } } while (0)
#define FK_STAMP(t, i) do { if (p.stamp && blockIdx.x == 0 && w == 0 && lane == 0 && (t) >= 0) \
        p.stamp[(size_t)(t) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FK_STAMP(t, i) do {} while (0)
#define FK_FINE(gs, k) do {} while (0)
#define FK_FINE_OUT(gs, t) do {} while (0)
#endif

// ---- weight images ------------------------------------------------------------------------------------------------
// NW = wavefronts that share one 32-row operand (2: workgroup = 64 rows x 16 units; 4: workgroup = 32 rows x 32 units).
// Per column tile ct (8 NW hidden units), 1 KB blocks = the MFMA A fragment of (32-row tile, 16-deep k step): lane l holds tile row
// l & 31, k = 8 (l >> 5) .. + 7.  Tile row c of unit group ug  <->  gate slot c >> 3, hidden unit 8 (NW ct + ug) + (c & 7).
// A wavefront walks the k steps of a part in ITS consumption order: position NW g + j is k step NW g + (ug ^ j) -- j = 0 its own k
// step of group g, then its partners'.  Regions per ct:
//   UR [ug][pos < NKH][m < NMR]        -> registers (NMR = 3: all of U's images)
//   WR [ug][pos < NKX][m < NMW]        -> registers (NMW = 1: W's high image, when 32 units x 256 inputs x 3 images do not fit LDS)
//   UL [ug][pos < NKH][m = NMR .. 2]   -> LDS (empty when NMR = 3)
//   WX [ug][pos < NKX][m = NMW .. 2]   -> LDS
__host__ __device__ inline int fk_blocks_per_ct(int NKH, int NKX, int NW) { return NW * (NKH * 3 + NKX * 3); }
__host__ __device__ inline int fk_nmr(int NKH) { return NKH <= 16 ? 3 : 2; }                         // images of U kept in registers
__host__ __device__ inline int fk_nmw(int NKX, int NW) { return NW == 4 && NKX > 8 ? 1 : 0; }       // images of W kept in registers (32 units x 256 inputs do not fit LDS)
// The K WALK of a column tile.  Every workgroup of a row block needs the same operand blocks, and workgroups that walked K in the same
// order would request the same few L2 lines at the same moment (measured: the L2 then serves ~8 TB/s chip-wide whatever the kernel does
// -- a handful of channels busy, the others idle).  So column tile ct starts its walk of a part's NGP groups at group ct mod NGP and
// assigns the k steps of a group to its wavefronts rotated by ct / NGP: consumption position NW g + j of wavefront ug is k step
// NW ((g + ct) mod NGP) + ((ug ^ j ^ (ct / NGP)) mod NW).  The weight images are packed in that order; sums differ between column
// tiles in the last bits only by their order, and never with the batch.
__host__ __device__ inline int fk_walk(int NGP, int NW, int ct, int ug, int pos) {
    const int g = pos / NW, j = pos % NW;
    return NW * ((g + ct) % NGP) + ((ug ^ j ^ (ct / NGP)) & (NW - 1));
}

template <bool RAW>
__global__ __launch_bounds__(256) void fk_pack_kernel(const float *__restrict__ ut, const float *__restrict__ wp,
                                                      rr_v4u *__restrict__ img, int H, int in, int Hj_p, int Hk_p, int Kin_p,
                                                      int NKH, int NKX, int NMR, int NW, int NCT) {
    const int NMW = fk_nmw(NKX, NW);
    const int n_ur = NW * NKH * NMR, n_wr = NW * NKX * NMW, n_ul = NW * NKH * (3 - NMR), bpc = fk_blocks_per_ct(NKH, NKX, NW);
    const long total = (long)NCT * bpc * 64;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int l = (int)(e & 63);
        const long blk = e >> 6;
        const int ct = (int)(blk / bpc);
        int r = (int)(blk % bpc);
        int part, ug, pos, m;
        if (r < n_ur) { part = 0; m = r % NMR; pos = (r / NMR) % NKH; ug = r / (NMR * NKH); }
        else if (r < n_ur + n_wr) { r -= n_ur; part = 2; m = r % NMW; pos = (r / NMW) % NKX; ug = r / (NMW * NKX); }
        else if (r < n_ur + n_wr + n_ul) { r -= n_ur + n_wr; part = 0; const int nl = 3 - NMR; m = NMR + r % nl; pos = (r / nl) % NKH; ug = r / (nl * NKH); }
        else { r -= n_ur + n_wr + n_ul; part = 2; const int nl = 3 - NMW; m = NMW + r % nl; pos = (r / nl) % NKX; ug = r / (nl * NKX); }
        const int ks = fk_walk(part == 2 ? NKX / NW : NKH / NW, NW, ct, ug, pos);
        const int c = l & 31;
        const int g = c >> 3;
        const int j = 8 * (NW * ct + ug) + (c & 7);
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = 16 * ks + 8 * (l >> 5) + q;
            float val = 0.0f;
            if (j < H) {
                if (RAW) {
                    if (part == 2) { if (k < in) val = wp[(size_t)k * 4 * H + (size_t)g * H + j]; }
                    else if (k < H) val = ut[(size_t)k * 4 * H + (size_t)g * H + j];
                } else {
                    if (part == 2) { if (k < in) val = wp[((size_t)g * H + j) * Kin_p + k]; }
                    else if (k < H) val = ut[((size_t)g * Hj_p + j) * Hk_p + k];
                }
            }
            v[q] = val;
        }
        rr_v4u hi, mid, lo;
        rr_split8(v, hi, mid, lo);
        img[e] = m == 0 ? hi : m == 1 ? mid : lo;
    }
}

// NKH / NKX: k steps (of 16) of the h / x part the kernel is compiled for (multiples of NW; H <= 16 NKH, in <= 16 NKX, zero padded).
// CELL 0: LSTM (slots i | f | g | o); 1: GRU (slots z | r | h.U_h | x.W_h, recurrent.c gru_rr_build_image).
// NW: wavefronts sharing an operand (see above); ND: staging sets = groups of k steps requested ahead.
template <int NKH, int NKX, int CELL, int NW, int ND>
__device__ __forceinline__ void fk_body(const RRParams &p) {
    constexpr int NMR = NKH <= 16 ? 3 : 2;            // images of U kept in registers
    constexpr int NMW = NW == 4 && NKX > 8 ? 1 : 0;   // images of W kept in registers
    constexpr int FK_RING = fk_ring(NKX, NW);
    constexpr bool BURST = FK_RING == 2;              // every ring read of a group in the group's first block
    constexpr int NGH = NKH / NW, NGX = NKX / NW, NG = NGH + NGX;          // groups of NW k steps per step: h part, x part
    constexpr int NHF = 4 / NW;                       // 32-row halves per workgroup
    static_assert((NW == 2 || NW == 4) && NKH % NW == 0 && NKX % NW == 0 && NGX >= 1, "group schedule");
    static_assert((2 * NG) % ND == 0 && (2 * NG) % fk_ring(NKX, NW) == 0 && ND <= NGH, "staging / ring indices repeat every two steps at most");
    // REQUEST SCHEDULE: the own k step of group q (of its step) is requested fk_lead(q) groups before the group that multiplies it -- as
    // early as the staging sets allow (ND groups), except the h groups at the head of a step: their data is published by the finish of
    // the step before, so a request issued before the x part of that step can only find the pending pattern (and be repeated).
    constexpr auto fk_lead = [](int q) constexpr {
        if (q >= NGH) return ND;                        // x part: nothing to wait for
        if (!FK_LEAD_CAP) return ND;
        // FK_LEAD_CAP: the h groups of a step are requested in the LAST TWO groups of the step before (2, 3, 3, 4, 4, .. groups ahead) --
        // behind its finish, when a request reaches memory about as its peers' stores do -- and never before that step's x part
        const int want = 2 + (q + 1) / 2, cap = NGX + q;
        const int l = want < cap ? want : cap;
        return l < ND ? l : ND;
    };
    constexpr int FIN_G = NW == 4 ? 1 : (NGX >= 4 ? 4 : NGX);     // x groups that carry the finish of the previous step (a hidden unit or more each)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    rr_v4u *ULs = reinterpret_cast<rr_v4u *>(smem);                       // [ug][pos][3 - NMR] blocks
    rr_v4u *WXs = ULs + NW * NKH * (3 - NMR) * 64;                        // [ug][pos][3 - NMW] blocks
    rr_v4u *ring = WXs + NW * NKX * (3 - NMW) * 64;                               // [w][slot][3] blocks: a wavefront's own k steps, for its partners
    unsigned *flg = reinterpret_cast<unsigned *>(ring + 4 * FK_RING * 3 * 64);      // [w][slot][64] sequence words
    // (plain accesses fenced by compiler barriers: `volatile` makes hipcc drain vmcnt around every access)
#define FK_FENCE() asm volatile("" ::: "memory")

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = w / NW, ug = w % NW;
    const int n = lane & 31, kh = lane >> 5;
    const int bt = blockIdx.x % p.NBT;               // row block of 32 NHF rows
    const int ct = blockIdx.x / p.NBT;
    const int ht_abs = p.b_base / 32 + bt * NHF + hf;
    const int b0 = p.b_base + bt * 32 * NHF;
    const int H = p.H, T = p.T;
    const int NKS = H >> 4;
    const int rows_valid = p.B - b0 < 32 * NHF ? p.B - b0 : 32 * NHF;
    const int gi = NW * ct + ug;                     // this wavefront's group of 8 hidden units
    const bool grp_ok = 8 * gi < H;                  // (H % 16 == 0: a group is inside or outside as a whole)

    // ---- resident operands ----
    const rr_v4u *img = p.img + (size_t)ct * fk_blocks_per_ct(NKH, NKX, NW) * 64;
    rr_bf16x8 uh[NKH][NMR];
#pragma unroll
    for (int pos = 0; pos < NKH; ++pos)
#pragma unroll
        for (int m = 0; m < NMR; ++m) {
            uh[pos][m] = __builtin_bit_cast(rr_bf16x8, img[(((size_t)ug * NKH + pos) * NMR + m) * 64 + lane]);
            RR_PIN_A(uh[pos][m]);
        }
    rr_bf16x8 wr[NMW ? NKX : 1][NMW ? NMW : 1];
    if (NMW) {
#pragma unroll
        for (int pos = 0; pos < NKX; ++pos)
#pragma unroll
            for (int m = 0; m < NMW; ++m) {
                wr[pos][m] = __builtin_bit_cast(rr_bf16x8, img[(size_t)NW * NKH * NMR * 64 + (((size_t)ug * NKX + pos) * NMW + m) * 64 + lane]);
                // (not pinned to the accumulator half: U's 192 registers + these 64 would leave the step's own working set only 256 ordinary
                // registers, and it spills)
            }
    }
    {
        const rr_v4u *src = img + (size_t)(NW * NKH * NMR + NW * NKX * NMW) * 64;       // UL then WX, contiguous, same order as in LDS
        constexpr int n16 = (NW * NKH * (3 - NMR) + NW * NKX * (3 - NMW)) * 64;
        for (int e = tid; e < n16; e += 256) ULs[e] = src[e];
        for (int e = tid; e < 4 * FK_RING * 64; e += 256) flg[e] = 0u;
    }
    // this lane finishes hidden units jf .. jf + 3 (all four gate slots) of batch row n of its half
    const int jf = 8 * gi + 4 * kh;
    float bsum[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            bsum[g][e] = jf + e < H ? p.bi[g * H + jf + e] + (p.bh ? p.bh[g * H + jf + e] : 0.0f) : 0.0f;
    float cst[4];
    {
        const int row = b0 + hf * 32 + n;
#pragma unroll
        for (int e = 0; e < 4; ++e) cst[e] = (p.c0 && row < p.B && jf + e < H) ? p.c0[(size_t)row * H + jf + e] : 0.0f;
    }
    // ---- descriptors and offsets ----
    const int hb_bytes = (int)p.hstep;
    // the hand-off of step t: read h_{t-1} (t == 0: the h_0 slot; t >= T: nothing -- zeros), write h_t
    auto rs_rd = [&](int t) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(t ? p.hseq + (size_t)(t - 1) * p.hstep : p.h0f), 0, t < T ? hb_bytes : 0, 0x00020000);
    };
    auto rs_wr = [&](int t) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(p.hseq + (size_t)t * p.hstep), 0, hb_bytes, 0x00020000);
    };
    auto rs_x = [&](int t) __attribute__((always_inline)) {          // x_t as frag3 blocks; t >= T: nothing
#ifdef FK_X_HOT     // (timing ablation, WRONG results: every step reads x_0 -- an L2-resident x)
        return __builtin_amdgcn_make_buffer_rsrc((void *)(p.xf3), 0, t < T ? (int)p.xstep : 0, 0x00020000);
#endif
        return __builtin_amdgcn_make_buffer_rsrc((void *)(p.xf3 + (size_t)(t < T ? t : 0) * p.xstep), 0, t < T ? (int)p.xstep : 0, 0x00020000);
    };
    const int lane16 = lane * 16;
    // own k step of group g of a part: fk_walk(.., pos = NW g).  k steps the tensors do not store read as zeros through an out-of-range
    // vector offset
    const int ugh = (ug ^ (ct / NGH)) & (NW - 1), ugx = (ug ^ (ct / NGX)) & (NW - 1);
    const int hso = ht_abs * NKS * 3 * 1024;
    const int xso = ht_abs * p.NKSx * 3 * 1024;
    // f32 output rows of this half (rows past the batch masked per lane)
    const long o_row_bytes = (long)(p.return_sequences ? p.T : 1) * H * 4;
    const long o_range = rows_valid * o_row_bytes;
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.out + (size_t)b0 * (o_row_bytes / 4)), 0, (int)(o_range < 0x7fffffffL ? o_range : 0x7fffffffL), 0x00020000);
    const bool row_ok = hf * 32 + n < rows_valid && grp_ok;
    const int out_vo = row_ok ? (int)((hf * 32 + n) * o_row_bytes) + jf * 4 : 0x7fff0000;
    // publication: this lane's 8 bytes of the row's 16-byte fragment slot: block (half-tile, k step = gi / 2, image), k half = gi & 1
    const int pub_vo = grp_ok ? ((ht_abs * NKS + (gi >> 1)) * 3) * 1024 + (32 * (gi & 1) + n) * 16 + 8 * kh : RR_OOB_F;
    RR_BARRIER();

    using Tt = std::true_type;
    using Ff = std::false_type;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    rr_v4u S[ND][3];               // own k steps: one being multiplied / handed over, the others in flight
    rr_v4u P[NW - 1][3];           // the partners' k steps of a group (from their rings)
    unsigned fl[NW - 1];           // ... and their sequence words
    rr_bf16x8 wa[2][3];            // W's fragments of an x k step (LDS -> registers one k step ahead)
    rr_bf16x8 ulo[2];              // NMR == 2: U's low image of an h k step, likewise

    // requests the three fragments of this wavefront's own k step of group g (h part: g_h, x part: g_x) of step tq into staging set si
    auto issue_own = [&](int g_h, int g_x, bool is_h, int si, int tq, int m0 = 0, int m1 = 3) __attribute__((always_inline)) {
        if (RR_DBG(1)) return;
        if (is_h) {
            const __amdgpu_buffer_rsrc_t rs = rs_rd(tq);
            const int ks = NW * ((g_h + ct) % NGH) + ugh;
            const int so = hso + ks * 3072;
            const int vo = lane16 + (ks < NKS ? 0 : RR_OOB_F);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (m < m0 || m >= m1) continue;
                S[si][m] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + m * 1024, so, 16 /* sc1 */);
                RR_BOUND(tq ? 3 : 4, tq ? (size_t)(tq - 1) * p.hstep : 0, vo + m * 1024, so, tq < T ? hb_bytes : 0, 16);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs = rs_x(tq + 1);               // the x part of step tq multiplies x_{tq + 1}
            const int ks = NW * ((g_x + ct) % NGX) + ugx;
            const int so = xso + ks * 3072;
            const int vo = lane16 + (ks < p.NKSx ? 0 : RR_OOB_F);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (m < m0 || m >= m1) continue;
                S[si][m] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + m * 1024, so, 0);
                RR_BOUND(1, (size_t)(tq + 1) * p.xstep, vo + m * 1024, so, tq + 1 < T ? p.xstep : 0, 16);
            }
        }
    };
    auto probe = [&](int si) __attribute__((always_inline)) {
        unsigned mx = 0;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const rr_v4u v = S[si][m];
            mx = max(mx, max(max(v.x, v.y), max(v.z, v.w)));
        }
        return __builtin_amdgcn_ballot_w64(mx == RR_PENDING);
    };
    const unsigned long long inject = p.spin_ticks == 0 ? ~0ull : 0ull;     // fault injection (tests): the first look "finds nothing" and the budget has run out
    // own k step of group g (h part, step tq >= 1: published inside this launch): every word of its fragments must have been written
    auto settle = [&](int g_h, int si, int tq) __attribute__((always_inline)) {
        {
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            bool expired = p.spin_ticks == 0;
            while (!expired) {
                issue_own(g_h, 0, true, si, tq);
                if (probe(si) == 0) break;
                if (__hip_atomic_load(p.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { expired = true; break; }
                __builtin_amdgcn_s_sleep(1);
                expired = __builtin_amdgcn_s_memrealtime() - t_start > p.spin_ticks;
            }
            if (expired && lane == 0) __hip_atomic_fetch_or(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // hand the staged k step to the partners: data, then the sequence word (LDS executes a wavefront's operations in order)
    // (piece m of 3; the word goes with the last piece; piece -1: all at once)
    auto ring_put = [&](int si, int slot, unsigned seq, int piece) __attribute__((always_inline)) {
        rr_v4u *d = ring + ((w * FK_RING + slot) * 3) * 64 + lane;
        FK_FENCE();
#pragma unroll
        for (int m = 0; m < 3; ++m)
            if (piece < 0 || piece == m) d[m * 64] = S[si][m];
        FK_FENCE();
        if (piece < 0 || piece == 2) flg[(w * FK_RING + slot) * 64 + lane] = seq;
        FK_FENCE();
    };
    // partner j's k step (wavefront ug ^ j of this half): the sequence word first (ring_word), the data after it (ring_data: fragments
    // [m0, m1) -- spread over the group: four wavefronts fetching 9 KB each at the same moment is 150 cycles of the LDS array)
    auto ring_word = [&](int j, int slot) __attribute__((always_inline)) {
        const int pw = hf * NW + (ug ^ j);
        FK_FENCE();
        fl[j - 1] = flg[(pw * FK_RING + slot) * 64 + lane];
        FK_FENCE();
    };
    auto ring_data = [&](int j, int slot, int m0, int m1) __attribute__((always_inline)) {
        const int pw = hf * NW + (ug ^ j);
        const rr_v4u *s = ring + ((pw * FK_RING + slot) * 3) * 64 + lane;
        FK_FENCE();
#pragma unroll
        for (int m = 0; m < 3; ++m)
            if (m >= m0 && m < m1) P[j - 1][m] = s[m * 64];
        FK_FENCE();
    };
    // a word was not there: look again until every partner's is, then fetch (again) what block a had fetched ahead of the check
    auto ring_check = [&](int slot, unsigned seq) __attribute__((always_inline)) {
        bool bad = false;
#pragma unroll
        for (int j = 1; j < NW; ++j) bad |= fl[j - 1] != seq;
        if (__builtin_amdgcn_ballot_w64(bad) != 0) {
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            bool expired = false;
            while (!expired) {
                bad = false;
#pragma unroll
                for (int j = 1; j < NW; ++j) { ring_word(j, slot); bad |= fl[j - 1] != seq; }
                if (__builtin_amdgcn_ballot_w64(bad) == 0) break;
                if (__hip_atomic_load(p.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { expired = true; break; }
                __builtin_amdgcn_s_sleep(1);
                expired = __builtin_amdgcn_s_memrealtime() - t_start > p.spin_ticks;
            }
            if (expired && lane == 0) __hip_atomic_fetch_or(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 1; j < (BURST ? NW : 2); ++j) ring_data(j, slot, 0, 3);
        }
    };
    // ---- the finish of step t, straight from the accumulator lanes, in two kinds of pieces: the gate arithmetic of one hidden unit (fin_unit),
    //      and the split + stores of the four (fin_store).  They run BETWEEN the MFMAs of the next step's x part (which does not depend on
    //      them): FIN_G x groups carry one or more units each, the last of them the stores ----
    float hn[4];
    rr_v4u olast = {0u, 0u, 0u, 0u};
    auto fin_unit = [&](int e, const f32x16 &acc) __attribute__((always_inline)) {
        if (RR_DBG(2)) return;
        float zc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) zc[g] = acc[4 * g + e] + bsum[g][e];
        if (RR_DBG(64)) { hn[e] = zc[0] + zc[1] + zc[2] + zc[3] + cst[e]; return; }
        if (CELL == 1) {
            // gru.c:144-186 (the expressions of recurrent_rr.hip / rec_persistent_kernel): slots z | r | h.U_h + b_h | x.W_h + b_i
            const float zg = nntk_fast_sigmoid(zc[0]);
            const float rg = nntk_fast_sigmoid(zc[1]);
            const float ht = nntk_fast_tanh(fmaf(rg, zc[2], zc[3]));
            hn[e] = fmaf(-zg + 1.0f, ht, zg * cst[e]);
            cst[e] = hn[e];
        } else {
            // lstm.c:201-238: Z = xW + b_i + hU (+ b_h); blocks i | f | g | o
            const float ig = nntk_fast_sigmoid(zc[0]);
            const float fg = nntk_fast_sigmoid(zc[1]);
            const float gg = nntk_fast_tanh(zc[2]);
            const float og = nntk_fast_sigmoid(zc[3]);
            const float cn = fmaf(fg, cst[e], ig * gg);
            cst[e] = cn;
            hn[e] = og * nntk_fast_tanh(cn);
        }
    };
    auto fin_store = [&](int t) __attribute__((always_inline)) {
        if (RR_DBG(2)) return;
        if (!RR_DBG(128)) {
            unsigned sh0, sm0, sl0, sh1, sm1, sl1;
            rr_split_pair(hn[0], hn[1], sh0, sm0, sl0);
            rr_split_pair(hn[2], hn[3], sh1, sm1, sl1);
            // no published word may equal the "not yet written" pattern (two bf16 NaNs with every payload bit set): such a pair becomes another NaN
            const fk_v2u a = {(unsigned)min(sh0, 0xfffffffeu), (unsigned)min(sh1, 0xfffffffeu)};
            const fk_v2u b = {(unsigned)min(sm0, 0xfffffffeu), (unsigned)min(sm1, 0xfffffffeu)};
            const fk_v2u c = {(unsigned)min(sl0, 0xfffffffeu), (unsigned)min(sl1, 0xfffffffeu)};
            // the whole address rides in the VECTOR offset, soffset stays immediate 0 (tools/check_store_hazard.py)
            const __amdgpu_buffer_rsrc_t rs = rs_wr(t);
            __builtin_amdgcn_raw_buffer_store_b64(a, rs, pub_vo, 0, RR_ST_AUX /* sc1 */);
            __builtin_amdgcn_raw_buffer_store_b64(b, rs, pub_vo + 1024, 0, RR_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(c, rs, pub_vo + 2048, 0, RR_ST_AUX);
            RR_BOUND(3, (size_t)t * p.hstep, pub_vo + 2048, 0, hb_bytes, 8);
        }
        olast = (rr_v4u){__float_as_uint(hn[0]), __float_as_uint(hn[1]), __float_as_uint(hn[2]), __float_as_uint(hn[3])};
        if (!RR_DBG(1024)) {      // (no branch: a call without a sequence output stores past the descriptor)
            const int vo = (row_ok && p.return_sequences && p.out) ? out_vo + t * H * 4 : 0x7fff0000;
            __builtin_amdgcn_raw_buffer_store_b128(olast, rso, vo, 0, 0);
            RR_BOUND(2, (size_t)b0 * o_row_bytes, vo, 0, o_range < 0x7fffffffL ? o_range : 0x7fffffffL, 16);
        }
    };

    // this wavefront's blocks of step t + 1 start out "pending" (past the last step: dropped).  Issued in the FIRST group of step t: every
    // later group of the step requests and consumes operand loads, whose in-order vmcnt waits retire these stores long before the step's
    // finish publishes h_t -- and a consumer asks for step t + 1's blocks only after it has seen every producer's h_t.  (Kept out of the
    // finish: with the four wavefronts in lockstep, 7 stores there and the next requests behind them queued up on the address path.)
    auto mark_next = [&](int t) __attribute__((always_inline)) {
        if (RR_DBG(128)) return;
        const __amdgpu_buffer_rsrc_t rs2 = rs_wr(t + 1 < T ? t + 1 : t);
        const int vo2 = t + 1 < T ? pub_vo : RR_OOB_F;
        const fk_v2u pend = {RR_PENDING, RR_PENDING};
        __builtin_amdgcn_raw_buffer_store_b64(pend, rs2, vo2, 0, RR_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b64(pend, rs2, vo2 + 1024, 0, RR_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b64(pend, rs2, vo2 + 2048, 0, RR_ST_AUX);
        RR_BOUND(3, (size_t)(t + 1 < T ? t + 1 : t) * p.hstep, vo2 + 2048, 0, hb_bytes, 8);
    };

    // six products per k step, smallest terms first: (A image, B image) = (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    // A fragments of consumption position `pos` of part (h: registers, low image from LDS when NMR == 2; x: LDS, prefetched into wa)
    auto lds_fetch_a = [&](bool is_h, int pos, int buf) __attribute__((always_inline)) {
        if (is_h) {
            if (NMR == 2) ulo[buf] = __builtin_bit_cast(rr_bf16x8, ULs[(ug * NKH + pos) * 64 + lane]);
        } else {
#pragma unroll
            for (int m = NMW; m < 3; ++m) wa[buf][m] = __builtin_bit_cast(rr_bf16x8, WXs[((ug * NKX + pos) * (3 - NMW) + (m - NMW)) * 64 + lane]);
        }
    };
    auto mult = [&](f32x16 &acc, bool is_h, int pos, int buf, const rr_v4u (&bsrc)[3], int j0, int j1) __attribute__((always_inline)) {
        if (RR_DBG(16)) return;
        rr_bf16x8 b[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) b[m] = __builtin_bit_cast(rr_bf16x8, bsrc[m]);
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
            if (pr < j0 || pr >= j1) continue;
            rr_bf16x8 av;
            if (is_h) av = (NMR == 2 && PA[pr] == 2) ? ulo[buf] : uh[pos][PA[pr] < NMR ? PA[pr] : 0];
            else av = PA[pr] < NMW ? wr[NMW ? pos : 0][NMW ? PA[pr] : 0] : wa[buf][PA[pr]];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[PB[pr]], acc, 0, 0, 0);
        }
    };

    // the k-th request this group-step is due to make (fragments [m0, m1)): the target groups r = gs + L whose lead is L, in order of L
    auto issue_sched = [&](int gs, int par, int t, int k, int m0, int m1) __attribute__((always_inline)) {
        int seen = 0;
#pragma unroll
        for (int L = 1; L <= ND; ++L) {
            const int r = gs + L, q = r % NG;
            if (fk_lead(q) != L) continue;
            if (seen++ != k) continue;
            const bool qh = q < NGH;
            issue_own(qh ? q : 0, qh ? 0 : q - NGH, qh, (par * NG + r) % ND, t + r / NG, m0, m1);
        }
    };
    // One step: [h part of step t on `acc`] -> finish(t) || [x part of step t + 1 on `accn`].  PAR = t & 1 (compile time: the
    // staging / ring indices of a group are its global index mod ND / mod FK_RING); HP false: the prologue (x part of step 0 only, t = -1).
    auto step = [&](auto par_tag, auto hp_tag, int t, f32x16 &acc, f32x16 &accn) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool HP = decltype(hp_tag)::value;
        fk_for<(HP ? 0 : NGH), NG>([&](auto gs_tag) __attribute__((always_inline)) {
            constexpr int gs = decltype(gs_tag)::value;
            const bool is_h = gs < NGH;
            const int pos0 = NW * (is_h ? gs : gs - NGH);                  // consumption positions of the group in its part: own first
            const int gidx = PAR * NG + gs;                               // global group index modulo 2 NG
            const int si = gidx % ND, slot = gidx % FK_RING;
            const int si1 = (gidx + 1) % ND, slot1 = (gidx + 1) % FK_RING;
            const unsigned seq = (unsigned)((t + 1) * NG + gs) + 1u;       // global group number + 1 (the prologue is step -1)
            // the group after this one, and the one ND behind this one (both may belong to the next step)
            const int q1 = (gs + 1) % NG, t1 = t + (gs + 1 >= NG ? 1 : 0);
            const bool h1 = q1 < NGH;
            f32x16 &a = is_h ? acc : accn;
            FK_STAMP(t, gs);
            if (gs == NGH) {
#pragma unroll
                for (int r = 0; r < 16; ++r) accn[r] = 0.0f;
            }
            // The non-MFMA work of a group is spread between its MFMAs in small pinned pieces (sched_barrier): a 1 KB vector-memory
            // request occupies the address path the four lock-stepped wavefronts share for ~60 cycles, an LDS burst likewise, and a
            // wavefront that waits at one cannot issue its next MFMA (one wavefront per SIMD: nobody else fills the pipe).
#define FK_PIN() __builtin_amdgcn_sched_barrier(0)
            // ---- block a: own k step, first half; the partners' k steps of this group requested from their rings; the own k step of
            //      the NEXT group looked at (its words compared in the MFMAs' shadow); ONE branch for both kinds of "not there yet"
            // LATE (FK_LEAD_CAP, the step's last group): the next group is the head of the next step -- its operand is in flight from the peers'
            // finish, and a wait for it here would hold THIS group's MFMAs back behind the hand-off chain; the look, the wait and the
            // hand-over move to the end of the group
            const bool LATE = FK_LEAD_CAP && gs == NG - 1;
            const bool chk1 = h1 && !LATE && !RR_DBG(4) && !RR_DBG(1);     // (a run-time t1 > 0 on top: step 0 reads the h_0 slot, written before the launch)
            unsigned mx0 = 0, mx1 = 0, mx2 = 0;
#ifdef NNTK_REC_STAMPS
            unsigned long long fine[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
            FK_FINE(gs, 0);
            mult(a, is_h, pos0, 0, S[si], 0, 1); FK_PIN();
#pragma unroll
            for (int j = 1; j < NW; ++j) ring_word(j, slot);
            if (chk1) { const rr_v4u v = S[si1][0]; mx0 = max(max(v.x, v.y), max(v.z, v.w)); asm volatile("" : "+v"(mx0)); }      // (pinned: hipcc would sink the look behind the branch on t1)
            FK_PIN();
            mult(a, is_h, pos0, 0, S[si], 1, 2); FK_PIN();
            ring_data(1, slot, 0, 3);
            if (BURST && NW == 4) ring_data(2, slot, 0, 3);
            if (chk1) { const rr_v4u v = S[si1][1]; mx1 = max(mx0, max(max(v.x, v.y), max(v.z, v.w))); asm volatile("" : "+v"(mx1)); }
            FK_PIN();
            mult(a, is_h, pos0, 0, S[si], 2, 3); FK_PIN();
            if (BURST && NW == 4) ring_data(3, slot, 0, 3);
            if (chk1) { const rr_v4u v = S[si1][2]; mx2 = max(mx1, max(max(v.x, v.y), max(v.z, v.w))); asm volatile("" : "+v"(mx2)); }
            lds_fetch_a(is_h, pos0 + 1, 1); FK_PIN();
            FK_FINE(gs, 1);
            {
                bool bad = false;
                if (!RR_DBG(512)) {
#pragma unroll
                    for (int j = 1; j < NW; ++j) bad |= fl[j - 1] != seq;
                }
                unsigned long long pend = chk1 ? __builtin_amdgcn_ballot_w64(mx2 == RR_PENDING) | inject : 0ull;
                if (t1 <= 0) pend = 0ull;
                if (__builtin_expect((pend | __builtin_amdgcn_ballot_w64(bad)) != 0, 0)) {
                    if (pend != 0) settle(q1, si1, t1);
                    if (!RR_DBG(512)) ring_check(slot, seq);
                }
            }
            FK_FINE(gs, 2);
#ifdef FK_COARSE_GS
            if (gs == FK_COARSE_GS) FK_STAMP(t, 40);
#endif
            // ---- block b: own k step, second half, the next group's own k step handed to the partners between its MFMAs
            if (!RR_DBG(512) && !LATE) ring_put(si1, slot1, seq + 1u, 0);
            FK_PIN();
            mult(a, is_h, pos0, 0, S[si], 3, 4); FK_PIN();
            if (!RR_DBG(512) && !LATE) ring_put(si1, slot1, seq + 1u, 1);
            FK_PIN();
            mult(a, is_h, pos0, 0, S[si], 4, 5); FK_PIN();
            if (!RR_DBG(512) && !LATE) ring_put(si1, slot1, seq + 1u, 2);
            FK_PIN();
            mult(a, is_h, pos0, 0, S[si], 5, 6); FK_PIN();
            if (HP && gs == 0) { mark_next(t); FK_PIN(); }
#ifdef FK_COARSE_GS
            if (gs == FK_COARSE_GS) FK_STAMP(t, 41);
#endif
            FK_FINE(gs, 3);
            // ---- blocks c ..: the partners' k steps multiplied; the own k step of ND groups ahead requested, one fragment per two
            //      MFMAs; the A fragments of the next k step on their way.  The first FIN_G x groups also carry the finish of the step
            //      whose h part has just ended: their non-MFMA work is left to the scheduler, a few vector-ALU instructions per MFMA
            const int gx = gs - NGH;
            const bool fin_here = FK_FIN_SLICED && HP && !is_h && gx < FIN_G;
#pragma unroll
            for (int j = 1; j < NW; ++j) {
                mult(a, is_h, pos0 + j, j & 1, P[j - 1], 0, 2); if (!fin_here) FK_PIN();
                if (j + 1 < NW && !BURST) ring_data(j + 1, slot, 0, 1);
                issue_sched(gs, PAR, t, j - 1, 0, 1); if (!fin_here) FK_PIN();
                if (j + 1 < NW) lds_fetch_a(is_h, pos0 + j + 1, (j + 1) & 1);
                else lds_fetch_a(h1, NW * (h1 ? q1 : q1 - NGH), 0);
                if (!fin_here) FK_PIN();
                mult(a, is_h, pos0 + j, j & 1, P[j - 1], 2, 4); if (!fin_here) FK_PIN();
                if (j + 1 < NW && !BURST) ring_data(j + 1, slot, 1, 2);
                issue_sched(gs, PAR, t, j - 1, 1, 2); if (!fin_here) FK_PIN();
                mult(a, is_h, pos0 + j, j & 1, P[j - 1], 4, 6); if (!fin_here) FK_PIN();
                if (j + 1 < NW && !BURST) ring_data(j + 1, slot, 2, 3);
                issue_sched(gs, PAR, t, j - 1, 2, 3); if (!fin_here) FK_PIN();
                FK_FINE(gs, 3 + j);
#ifdef FK_COARSE_GS
                if (gs == FK_COARSE_GS) FK_STAMP(t, 41 + j);
#endif
            }
            // (requests of this group-step that found no partner block to ride in: pairs have one block and may be due two or three)
#pragma unroll
            for (int k = NW - 1; k < 3; ++k) issue_sched(gs, PAR, t, k, 0, 3);
            if (fin_here) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e * FIN_G / 4 == gx) fin_unit(e, acc);
                if (gx == FIN_G - 1) fin_store(t);
#pragma unroll
                for (int k = 0; k < 6 * (NW - 1); ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, FK_FIN_VALU, 0);
                    if (k % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x030, 1, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            if (LATE) {
                if (h1 && t1 > 0 && !RR_DBG(4) && !RR_DBG(1)) {
                    const unsigned long long pend = probe(si1) | inject;
                    if (__builtin_expect(pend != 0, 0)) settle(q1, si1, t1);
                }
                if (!RR_DBG(512)) ring_put(si1, slot1, seq + 1u, -1);
            }
            __builtin_amdgcn_sched_barrier(0);
            FK_FINE_OUT(gs, t);
            if (!FK_FIN_SLICED && HP && gs == NGH - 1) {
                // the finish as one piece between the h part and the x part.  (Sliced between the x part's MFMAs -- FK_FIN_SLICED -- it was
                // measured SLOWER: one wavefront per SIMD is bound by its instruction issue, not by the pipe: profiles/r05_fk_*.log)
#pragma unroll
                for (int e = 0; e < 4; ++e) fin_unit(e, acc);
                fin_store(t);
                FK_STAMP(t, 33);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (FK_FIN_SLICED && HP && !is_h && gx == FIN_G - 1) FK_STAMP(t, 33);
        });
    };

    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[r] = 0.0f; accB[r] = 0.0f; }
    // prime the pipeline: every request whose turn would have come before the prologue's first group (x part of step 0; a short x part
    // runs on into the h part of step 0); the first group handed to the partners
    {
#pragma unroll
        for (int r = NGH; r < NGH + ND; ++r) {
            const int q = r % NG;
            if (r - fk_lead(q) >= NGH) continue;          // its turn comes inside the prologue
            const bool qh = q < NGH;
            issue_own(qh ? q : 0, qh ? 0 : q - NGH, qh, (NG + r) % ND, -1 + r / NG);
        }
        ring_put((NG + NGH) % ND, (NG + NGH) % FK_RING, (unsigned)NGH + 1u, -1);
        lds_fetch_a(false, 0, 0);
    }
    step(I1{}, Ff{}, -1, accB, accA);
    for (int t = 0; t < T; t += 2) {
        step(I0{}, Tt{}, t, accA, accB);
        if (t + 1 >= T) break;
        step(I1{}, Tt{}, t + 1, accB, accA);
    }
    // ---- last output / final state: the lanes still hold h_{T-1} and the state ----
    {
        int b0e = b0;
        asm volatile("" : "+s"(b0e));
        const int row = b0e + hf * 32 + n;
        if (row < p.B && grp_ok) {
            if (!p.return_sequences && p.out) *reinterpret_cast<rr_v4u *>(p.out + (size_t)row * H + jf) = olast;
            if (p.hT) *reinterpret_cast<rr_v4u *>(p.hT + (size_t)row * H + jf) = olast;
            if (p.cT) *reinterpret_cast<rr_v4u *>(p.cT + (size_t)row * H + jf) =
                (rr_v4u){__float_as_uint(cst[0]), __float_as_uint(cst[1]), __float_as_uint(cst[2]), __float_as_uint(cst[3])};
        }
    }
}
template <int NKH, int NKX, int NW, int ND>
__global__ __launch_bounds__(256) void lstm_fk_kernel(RRParams p) { fk_body<NKH, NKX, 0, NW, ND>(p); }
template <int NKH, int NKX, int NW, int ND>
__global__ __launch_bounds__(256) void gru_fk_kernel(RRParams p) { fk_body<NKH, NKX, 1, NW, ND>(p); }

// ---- host side --------------------------------------------------------------------------------------------------
// NW = 4 (32 rows x 32 units per workgroup, half the operand traffic) when W's images for 32 units fit LDS (in <= 128); else NW = 2
// rec_fk: 1 every shape the family takes; 0 none; auto: the shapes where it beats the split-K family -- inputs of 129 .. 256 channels (GRU
// 256 -> 256: 5.60 against 6.13 ms, LSTM 5.83 against 6.33, GRU 200 -> 192: 4.55 against 5.21; at in <= 128 it is 1-4 % behind)
static bool fk_shape(int H, int in, int *NKH, int *NKX, int *NW) {
    if ((H % 16) != 0 || H <= 128 || H > 256 || in <= 64 || in > 256) return false;
    const int on = nntk_options().rec_fk;
    if (on == 0 || (on < 0 && in <= 128)) return false;
    *NKH = 16;
    *NKX = in <= 128 ? 8 : 16;
    *NW = 4;
#ifdef FK_NW2_WIDE
    if (in > 128) *NW = 2;        // (A/B: the 256-wide layer on pairs, 64 rows x 16 units)
#endif
    return true;
}
static size_t fk_lds_bytes(int NKH, int NKX, int NW) {
    return (size_t)(NW * NKH * (3 - fk_nmr(NKH)) + NW * NKX * (3 - fk_nmw(NKX, NW)) + 4 * fk_ring(NKX, NW) * 3) * 1024 + 4 * fk_ring(NKX, NW) * 64 * 4;
}
extern "C" size_t nntk_shim_fk_image_floats(int H, int in) {
    int NKH, NKX, NW;
    if (!fk_shape(H, in, &NKH, &NKX, &NW)) return 0;
    return (size_t)((H + 8 * NW - 1) / (8 * NW)) * fk_blocks_per_ct(NKH, NKX, NW) * 256;
}
static int fk_pack(bool raw, const float *d_u, const float *d_w, float *d_img, int H, int in) {
    int NKH, NKX, NW;
    if (!fk_shape(H, in, &NKH, &NKX, &NW)) return nntk_fail_msg("fk_pack: shape not taken by the full-K register-resident kernel");
    const int Hj_p = (H + 15) & ~15, Hk_p = (H + 31) & ~31;
    int Kin_p, N_p;
    nntk_shim_conv_pack_sizes(in, 4 * H, 1, &Kin_p, &N_p);
    const int NCT = (H + 8 * NW - 1) / (8 * NW), NMR = NKH <= 16 ? 3 : 2;
    const long total = (long)NCT * fk_blocks_per_ct(NKH, NKX, NW) * 64;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (raw) hipLaunchKernelGGL(fk_pack_kernel<true>, dim3((unsigned)g), dim3(256), 0, nntk_stream(), d_u, d_w, (rr_v4u *)d_img, H, in, 0, 0, 0, NKH, NKX, NMR, NW, NCT);
    else     hipLaunchKernelGGL(fk_pack_kernel<false>, dim3((unsigned)g), dim3(256), 0, nntk_stream(), d_u, d_w, (rr_v4u *)d_img, H, in, Hj_p, Hk_p, Kin_p, NKH, NKX, NMR, NW, NCT);
    NNTK_LAUNCH_CHECK("fk_pack_kernel");
    return 0;
}
// d_ut / d_wp: the per-gate U^T and the packed W^T (host: core_upload)
extern "C" int nntk_shim_fk_pack(const float *d_ut, const float *d_wp, float *d_img, int H, int in) { return fk_pack(false, d_ut, d_wp, d_img, H, in); }
// the caller-layout matrices U [H][4H], W [in][4H] (the GRU's four-slot matrices)
extern "C" int nntk_shim_fk_pack_raw(const float *d_U, const float *d_W, float *d_img, int H, int in) { return fk_pack(true, d_U, d_W, d_img, H, in); }

// 0 = launched; 1 = not taken; -1 = error.  q: the parameters rr_launch (recurrent_rr.hip) has filled (frag3 x, T-deep hand-off with its
// first two steps preset to the pending pattern, the h_0 slot); d_imgfk: the images of fk_pack.
int nntk_fk_launch(RRParams q, const float *d_imgfk, int cell, size_t *launches) {
    int NKH, NKX, NW;
    if (!d_imgfk || !q.xf3 || q.x_tm || q.out_tm || !fk_shape(q.H, q.in, &NKH, &NKX, &NW)) return 1;
    void (*kern)(RRParams) = nullptr;
    if (cell == 1) kern = NKX == 8 ? gru_fk_kernel<16, 8, 4, FK_ND_4> : gru_fk_kernel<16, 16, 4, FK_ND_4W>;
    else kern = NKX == 8 ? lstm_fk_kernel<16, 8, 4, FK_ND_4> : lstm_fk_kernel<16, 16, 4, FK_ND_4W>;
#ifdef FK_NW2_WIDE      // A/B build only: the 256-wide layer on pairs of wavefronts (64 rows x 16 units), measured 8 % behind split-K
    if (NW == 2) kern = cell == 1 ? gru_fk_kernel<16, 16, 2, FK_ND_2> : lstm_fk_kernel<16, 16, 2, FK_ND_2>;
#endif
    const size_t lds = fk_lds_bytes(NKH, NKX, NW);
    if (lds > 160 * 1024) return 1;
    if (nntk_set_max_dynamic_lds((const void *)kern, lds)) return -1;
    const int NCT = (q.H + 8 * NW - 1) / (8 * NW);
    const int rows = 128 / NW;                         // batch rows per workgroup
    const int resident = nntk_resident_blocks((const void *)kern, 256, lds, 1);
    const int tiles_per_launch = resident / NCT;
    if (tiles_per_launch < 1) return 1;
    q.img = (const rr_v4u *)d_imgfk;
    q.NCT = NCT;
    const int nbt_total = (q.B + rows - 1) / rows;
    for (int bt0 = 0; bt0 < nbt_total; bt0 += tiles_per_launch) {
        const int nbt = nbt_total - bt0 < tiles_per_launch ? nbt_total - bt0 : tiles_per_launch;
        q.NBT = nbt; q.b_base = bt0 * rows;
        hipLaunchKernelGGL(kern, dim3((unsigned)(nbt * NCT)), dim3(256), lds, nntk_stream(), q);
    }
    if (launches) *launches = (size_t)((nbt_total + tiles_per_launch - 1) / tiles_per_launch);
    static const char *const names[2][3] = {{"lstm_fk_kernel<16,8,4>", "lstm_fk_kernel<16,16,4>", "lstm_fk_kernel<16,16,2>"}, {"gru_fk_kernel<16,8,4>", "gru_fk_kernel<16,16,4>", "gru_fk_kernel<16,16,2>"}};
    nntk_set_last_rec_kernel(names[cell == 1][NW == 2 ? 2 : NKX == 16]);
    return 0;
}
