// recurrent_rr.hip -- K4b: persistent LSTM on the bf16 MFMA with the weights REGISTER-RESIDENT and the input
// projection fused into the step.  Reference semantics: layers/lstm.c:185-239 (cell), :426-475 (batch forward,
// zero or carried state).
//
// Why another LSTM kernel.  rec_persistent_kernel (recurrent.hip) keeps a [64 x H] f32 slice of U^T in LDS and
// multiplies on the exact-f32 MFMA: 0.74 of that pipe's 157 TFLOP/s, and the pipe is the bound (9.2 us per step at
// LSTM-512, B = 512), with the x W projection as a separate 2.5 ms GEMM whose [T, B, 4H] result (4.2 GB) makes a round
// trip through HBM.  The split-bf16 x 3 contraction (conv1d.hip: every f32 operand = hi + mid + lo in bf16 exactly,
// six bf16 products accumulated in f32 -- f32 accuracy, 2.67x less MFMA time) does not fit that design: three bf16
// images of the U^T slice are 196 KB, more than the CU's 160 KB of LDS.  What does hold them is the REGISTER FILE:
// one wavefront per SIMD may own 512 VGPRs, 512 KB per CU.  So:
//
//   * workgroup = 4 wavefronts (one per SIMD), 16 hidden units x 4 gates = 64 gate columns (two 32-row MFMA tiles) for a
//     64-row batch tile, for the whole sequence; K = [h (H) | x_t (in)] is split over the 4 wavefronts, and each keeps the
//     hi / mid images of its K range of U^T in REGISTERS as ready-made MFMA A fragments (LSTM-512: 128 VGPRs), loaded
//     once per launch; the lo image (used by one product in six) and the three images of W^T sit in LDS (112 KB).
//   * h travels between workgroups ALREADY SPLIT: the producer of h_t[b][j] writes its three bf16 images in the order
//     the consumers' MFMA B fragments want them (1 KB blocks = 64 lanes x 16 B), so a consumer's operand is 24 16-byte
//     loads per lane straight into registers, no LDS, no conversion.  Splitting once at the producer costs 6 bytes per
//     element on the wire instead of 4, and 32x fewer VALU instructions than splitting at every consumer.
//   * x_t W is computed inside the step on the same accumulators (x_t is split by its consumer: it does not depend on
//     the recurrence and is fetched one half-step ahead).  No projection launch, no [T, B, 4H] tensor.
//   * the 64 batch rows are two independent 32-row halves with their own hand-off chains, worked in strict alternation
//     by ONE instruction stream per SIMD: while half A multiplies, half B's publication -> arrival -> poll -> operand
//     fetch elapses, and the other way round.  The events of the idle half (drain + arrival of what it published, poll
//     for its next operand, operand loads) are placed INSIDE the multiplying half's MFMA sequence.
//
// Hand-off protocols.  Both follow the placement-independent recipe of the CDNA4 guide (Guideline 16 R1, sc1 row): write-through
// (sc1) 16-byte stores by the producer, every load of handed-off bytes an sc1 buffer load.  They differ in how a consumer learns
// that a block has been written:
//   * FLAGS (KH = 8, the LSTM-512 class): the publishing wave waits until its stores have drained (a counted vmcnt), then stores
//     t + 1 into its column tile's flag word; a consumer wave reads the batch tile's flags with one wave-wide sc1 load before it
//     requests the operand.  Cheapest in instructions; the chain publication -> drain -> flag -> poll -> operand request -> operand
//     is ~5 k cycles, which a 10-k-step half-step covers.
//   * PENDING PATTERN (KH = 4, H <= 256): that chain is LONGER than a 6..8-k-step half-step (stamps, DESIGN K4b round 4: the
//     half-step took as long as the chain, 5.15 k cycles against 2.3 k of MFMA time), so these shapes drop the flag: a block that
//     has not been written yet holds the word 0xffffffff in every position, a published block never does (the publisher maps the
//     one bf16 pair that would -- two NaNs with all payload bits set -- to another NaN), and the consumer simply requests the
//     operand and LOOKS at it: every word of every fragment of a k step is compared (v_max3 over 12 words, in the shadow of the
//     previous k step's MFMAs), and a k step that still holds a pending word is requested again (rare; bounded like every spin).
//     No drain wait, no flag hop, no poll; per-word comparison, so no assumption about how a 16-byte store becomes visible.
//     Who writes the pattern: the T-deep hand-off is also the layer's OUTPUT, so it cannot be cleared behind the consumers.  The
//     host presets the first two timesteps; after that the workgroup that owns a block marks its block of step t + 2 while it
//     publishes step t.  Why a stale block of an earlier launch can never be taken for data: a consumer requests blocks of step
//     t + 2 only after it has multiplied step t + 1's operand, i.e. after it has SEEN every producer's data of step t + 1; a
//     producer issues that data a whole step (>= 4 workgroup barriers and > 20 consumed operand loads, whose in-order vmcnt waits
//     also retire the older mark stores) after its marks of step t + 2; both are write-through stores of the same workgroup to
//     memory.  So "data(t + 1) visible" implies "mark(t + 2) visible", and from then on the block reads as pending until its own
//     data lands on top of the mark.  The marks are issued by the OTHER half's publishing wave, which is idle in that slice: it has
//     retired them (the vmcnt waits of the operand loads it consumes in every k step) two steps of workgroup barriers before the
//     publishing wave issues the data for the same addresses.
// Spins are bounded and poison the counter (runtime.hip fault word).
//
// Numerics: the contraction is the split form of conv1d.hip (error <= the f32 chain's against float64); gates are the
// exp2 / rcp forms of nntk_common.hpp.  Not bit-identical to rec_persistent_kernel (another summation order), same
// tolerance; results do not depend on the batch tile a row lands in, so shards equal the whole batch bit for bit.
#include "recurrent_rr_common.hpp"

// ---- weight images ------------------------------------------------------------------------------------------------
// Tile row c (0..63) of column tile ct  <->  gate g = (c >> 3) & 3, hidden unit j = 16 ct + 8 ((c >> 2) & 1) + 4 (c >> 5) + (c & 3):
// with the 32x32 MFMA's D layout (lane (n, kh) holds rows 8 q + 4 kh + e of a tile, q = 0..3, e = 0..3) a lane then owns
// all four gates of hidden units 8 kh + 4 mt + e -- and lane (n, kh) of the publishing wave owns EIGHT CONSECUTIVE
// hidden units of batch row n, which is exactly one B fragment (8 consecutive k) of the consumers.
// Blocks of 1 KB = the A fragment of one (32-row tile mt, 16-deep k step): lane l holds row (l & 31), k = 8 (l >> 5) .. + 7.
// Per column tile, in this order (KH / KX = k steps per wavefront of the h / x part):
//   UH [w][i < KH][mt][m = hi, mid]       -> registers
//   UL [w][i < KH][mt]                    -> LDS
//   WX [w][ix < KX][mt][m = hi, mid, lo]  -> LDS
__host__ __device__ inline int rr_blocks_per_ct(int KH, int KX) { return 4 * KH * 2 * 2 + 4 * KH * 2 + 4 * KX * 2 * 3; }

// ut [4][Hj_p][Hk_p] f32 (U^T per gate), wp [4H padded][Kin_p] f32 (W^T, K-contiguous) -> images.
// RAW: the sources are the caller's own layout instead, U [H][4H] and W [in][4H] (recurrent_private.c:29-36): the training
// forward re-packs every mini-batch (the weights change with every optimiser step) straight from the uploaded block.
// HF (uscale > 0): the images of the HF instantiations (h.U on two f16 images): UH [..][m = hi, lo] are the two f16 images of U * uscale, UL is
// unused (zeros), WX the three bf16 images of W * wscale (wscale = 2^15 uscale: both parts of Z then carry the same power of two).
template <bool RAW>
__global__ __launch_bounds__(256) void rr_pack_kernel(const float *__restrict__ ut, const float *__restrict__ wp,
                                                      rr_v4u *__restrict__ img, int H, int in, int Hj_p, int Hk_p, int Kin_p,
                                                      int KH, int KX, int NCT, float uscale = 0.0f, float wscale = 1.0f) {
    const int bpc = rr_blocks_per_ct(KH, KX);
    const long total = (long)NCT * bpc * 64;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int l = (int)(e & 63);
        const long blk = e >> 6;
        const int ct = (int)(blk / bpc);
        int r = (int)(blk % bpc);
        int part, w, i, mt, m;                    // part 0 = UH, 1 = UL, 2 = WX
        const int n_uh = 4 * KH * 4, n_ul = 4 * KH * 2;
        if (r < n_uh) { part = 0; m = r & 1; mt = (r >> 1) & 1; i = (r >> 2) % KH; w = (r >> 2) / KH; }
        else if (r < n_uh + n_ul) { r -= n_uh; part = 1; m = 2; mt = r & 1; i = (r >> 1) % KH; w = (r >> 1) / KH; }
        else { r -= n_uh + n_ul; part = 2; m = r % 3; mt = (r / 3) & 1; i = (r / 6) % KX; w = (r / 6) / KX; }
        const int c = 32 * mt + (l & 31);
        const int g = (c >> 3) & 3;
        const int j = 16 * ct + 8 * ((c >> 2) & 1) + 4 * (c >> 5) + (c & 3);
        const int ks = part == 2 ? w * KX + i : w * KH + i;
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
        if (uscale > 0.0f) {
            if (part == 0) {
                unsigned h4[4], l4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) rr_split_pair_f16(v[2 * q] * uscale, v[2 * q + 1] * uscale, h4[q], l4[q]);
                img[e] = m == 0 ? (rr_v4u){h4[0], h4[1], h4[2], h4[3]} : (rr_v4u){l4[0], l4[1], l4[2], l4[3]};
                continue;
            }
            if (part == 1) { img[e] = (rr_v4u){0u, 0u, 0u, 0u}; continue; }
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] *= wscale;
        }
        rr_split8(v, hi, mid, lo);
        img[e] = m == 0 ? hi : m == 1 ? mid : lo;
    }
}

// h_0 [B][H] f32 -> the split hand-off layout of the h_0 slot: block (((bt * 2 + half) * NKS + ks) * 3 + m), lane (n, kh); NKS = H / 16
__global__ __launch_bounds__(256) void rr_tile_h0_kernel(const float *__restrict__ h0, rr_v4u *__restrict__ hb, int B, int H, int NKS) {
    const long total = (long)((B + 63) / 64) * 2 * (H / 16) * 64;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int l = (int)(e & 63);
        long r = e >> 6;
        const int ks = (int)(r % (H / 16)); r /= (H / 16);
        const int half = (int)(r & 1);
        const long bt = r >> 1;
        const long row = bt * 64 + half * 32 + (l & 31);
        if (row >= B) continue;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = h0[row * H + 16 * ks + 8 * (l >> 5) + q];
        rr_v4u hi, mid, lo;
        rr_split8(v, hi, mid, lo);
        rr_v4u *dst = hb + ((((size_t)bt * 2 + half) * NKS + ks) * 3) * 64 + l;
        dst[0] = hi; dst[64] = mid; dst[128] = lo;
    }
}

// The kernel is a software pipeline of HALF-STEPS.  Half-step s multiplies half Y = s & 1 at timestep t = s >> 1 (its
// operand fetched during half-step s - 1) and, sliced between the k steps of that MFMA sequence, FINISHES half X = 1 - Y,
// whose partial sums half-step s - 1 left in LDS:
//   k step S_RED   barrier; partial sums of X read and added (fixed order); gates; h written to the exchange image
//          S_PUB   barrier; the publishing wave reads 8 consecutive hidden units of a row, already split into the three bf16
//                  images by the lanes that computed them, and publishes them write-through
//          S_XSPL  x_t of X's next step split into its three images; x_{t+1} of Y requested (a whole half-step ahead)
//   FLAGS:
//          S_E1    the publication has had S_E1 - S_PUB k steps to drain: counted vmcnt, then the publishing wave raises its flag
//          S_E2-2  every wave requests the flags of X's next operand (POLL_LEAD)
//          S_E2    check them (spin only if one is missing), request the head of X's next operand
//   PENDING PATTERN:
//          S_HEAD  request the head of X's next operand (no earlier than the data can be there: RR_HEAD_LEAD)
//          every h k step: the NEXT k step's fragments are looked at (probe_h); a pending one is settled before its MFMAs
// so that the vector ALU, LDS and memory work of one half runs in the shadow of the other half's MFMAs (an MFMA holds the
// issue port 8 of its 32 cycles).  Every vector-memory operation is a branch-free buffer operation (lanes that must not
// load / store get an out-of-range offset), so the steady-state half-step is straight-line code apart from the poll.
//
// ONE wavefront publishes a half (wave 2 * half: all 64 lanes = 32 rows x 2 k halves, three full 1 KB write-through stores)
// and another one (wave 2 * half + 1) stores the f32 layer output: with the rows spread over the four wavefronts every one of
// them issued five quarter-filled store instructions per half-step, and a store occupies the shared address path as long
// whatever it carries (measured: publication + output stores cost 1.06 us of a 7.0 us step that way).
//
// FLAGS protocol: signalling is by FLAG WORDS, not by an arrival counter: the publishing wave of column tile ct stores (t + 1)
// write-through into flags[batch tile][half][ct] once its publishing stores have drained, and every consumer wave reads the batch
// tile's flags with one wave-wide sc1 load.  Against the agent-scope counter of rec_persistent_kernel this removes the
// serialisation of 32 read-modify-writes at the memory side (~12 ns each) and the cross-wave gather in front of the add,
// and makes the arrival a plain store.  Valid by the guide's sc1 hand-off table: the flag covers exactly the stores of the
// wave that raises it, after that wave's vmcnt wait.
// KH / KX: k steps (of 16) per wavefront for the h / x part: H <= 64 KH, in <= 64 KX (zero padded).
// TRAIN: the training forward pass (LSTMApplyTrainingBatch, lstm.c:426-475): additionally keeps c_t and the gates' pre-activations
// and activations of every step for back-propagation through time.
// CELL 1: GRU (gru.c:129-187) on the same frame.  The GRU's three gates ride in the four gate slots as z | r | h.U_h | x.W_h: the
// host packs U as [U_z | U_r | U_h | 0] and W as [W_z | W_r | 0 | W_h] (recurrent.c core_try_gru_rr), so slot 2 collects only the
// recurrent part of the candidate and slot 3 only its input part -- the reset gate multiplies the former (reset-after form) -- and the
// multiply phase is the LSTM's, instruction for instruction; only the gate arithmetic (fin_gates) differs.  The state register
// that is the LSTM's c_t holds the GRU's own f32 h_t.
// ULR (in > 128, H <= 256): U's low image lives in registers (32 VGPRs at KH = 4) instead of LDS, which makes room for the
// 96 KB of W images a 256-wide input needs.
// XF: x arrives as a frag3 tensor (frag3.hip: already split into its three bf16 images, in THIS kernel's B-fragment order, 32-row
// blocks of one timestep contiguous): 3 KX coalesced 1 KB requests per half-step straight into the MFMA operand registers instead
// of 2 KX requests that touch 32 rows each plus the 44-instruction split (DESIGN K4b "what the x part costs": the requests, not
// the split, were 0.5 us of a 6.7 us step).  Each half owns an operand set, requested a whole half-step ahead.
// HF (LSTM, H > 256, XF): the h part of Z on TWO f16 images (frag3.hip FRAG2H: |h| < 1, h 2^15 = hi + lo) and three products per k step
// instead of three bf16 images and six: U's two f16 images of U 2^q sit in the registers that hold hi / mid (its LDS image is unused), the
// hand-off carries two blocks per (row block, k step) -- and IS the dense layer's FRAG2H operand -- W's bf16 images are packed
// pre-multiplied by 2^(15 + q), so both parts accumulate at one scale, taken out (exactly) where the bias is added.
template <int KH, int KX, bool TRAIN, int CELL, bool XF, bool HF = false>
__device__ __forceinline__ void rr_body(const RRParams &p) {
    constexpr int NH = HF ? 2 : 3;                    // images of the h hand-off
#ifndef RR_ULR8
#define RR_ULR8 0
#endif
    constexpr bool ULR = !HF && (KX > 2 || (RR_ULR8 && KH == 8 && !TRAIN));      // (HF: U has no low image at all)
    constexpr int NST = KX + KH;                      // k steps one wavefront multiplies per half
#ifndef RR_S_RED
#define RR_S_RED 0               // k step of the reduce + gates slice; 1 (KH = 8: the half-step's first 12 MFMAs issue before its barrier) measured
                                 // 1-4 % SLOWER on LSTM-512 (6.30-6.46 vs 6.19-6.23 ms, same box, alternating processes: profiles/r04_lstm_sred_ab.log)
#endif
    constexpr int S_RED = (RR_S_RED) && KH == 8 ? 1 : 0, S_PUB = S_RED + 1;       // (the KH = 4 shapes have no k step to spare)
    constexpr int S_XSPL = KX > 2 ? KX : S_PUB + 1;
    // Hand-off protocol (see the header comment): PEND = the published fragments carry their own validity (KH = 4 shapes, whose
    // half-step is shorter than the flag chain), else flag words (KH = 8: the chain fits, and the flags cost fewer instructions)
#ifndef RR_PEND
#define RR_PEND (KH == 4)
#endif
    constexpr bool PEND = RR_PEND;
#ifndef RR_S_E1_XLATE
#define RR_S_E1_XLATE 2           // the KX = 4 shapes (x requests behind the poll) may arrive from k step 2 on: 6.82 -> 6.66 us per step; 4: 8.3 -- the chain is that tight
#endif
    // HF: a k step of the h part is 6 MFMAs, not 12 -- the same number of k steps is half the time, and the flag chain (publication -> drain ->
    // flag -> the consumers' poll -> their fetch) is what the half-step waits for.  RR_HF_* are that instantiation's own knobs, measured on
    // LSTM-512 (same box, alternating processes, profiles/r05_lstm_hf_knobs.log): arrival at k step 3 / 5 / 2: 5.27 / 5.63 / 4.73-5.10 ms -- the
    // flag goes up as early as the frame allows; operand k steps requested ahead 1 / 2 / 3 / 4 / 5 / 6 / 8 (at arrival 2): 5.27 / 5.10 / 4.93 /
    // 4.76 / 4.71 / 4.70 / 4.96; flags requested 1 / 3 / 4 k steps before they are looked at instead of 2: +0.05 / +-0 / +0.27 ms.
#ifndef RR_HF_S_E1
#define RR_HF_S_E1 2
#endif
    constexpr int S_E1 = (KX > 2 ? RR_S_E1_XLATE : HF ? RR_HF_S_E1 : RR_S_E1) + S_RED;                  // (flag protocol only)
    // X_LATE (KX = 4): the eight x requests of a half-step touch 32 rows each (2 x 16 bytes per row and request): they hold the
    // address path for ~2 k cycles and take ~3 us to return, and the poll's vmcnt(0) at S_E2 waited for them (stamps: 6 k cycles
    // in that k step).  They go out AFTER the poll instead, at the end of the half-step, and have the next half-step up to its
    // S_XSPL to arrive.
    constexpr bool X_LATE = KX > 2;
    constexpr int S_E2 = NST - 1;
#ifndef RR_HEAD_LEAD
#define RR_HEAD_LEAD 1            // PEND: the head of the other half's next operand goes out this many k steps before the half-step's last one
#endif                            // (GRU-256 pair: 0: 10.6, 1: 10.2, 2: 11.2, 3: 12.0 ms -- earlier than that the fragments are still pending and the second look costs a round trip)
    constexpr int S_HEAD = !PEND ? S_E2 : S_E2 - RR_HEAD_LEAD > S_PUB ? S_E2 - RR_HEAD_LEAD : S_PUB + 1;
    // Operand schedule: the fragments of h k step i are requested NPRE k steps ... see issue_h below: i < NPRE at S_HEAD of the
    // OTHER half's sequence (flag protocol: after the poll), i >= NPRE at k step i - NPRE of the half's own sequence (needed at KX + i).
#ifndef RR_HF_NPRE
#define RR_HF_NPRE 5
#endif
    constexpr int NPRE = (HF ? RR_HF_NPRE : RR_NPRE) < KH ? (HF ? RR_HF_NPRE : RR_NPRE) : KH;
    // flag protocol: vector-memory operations a wave issues between a publication (S_PUB) and its arrival (S_E1): the own-sequence
    // operand requests of k steps S_PUB .. S_E1 - 1 and the x request at S_XSPL -- what the arrival's counted wait leaves in flight
    constexpr int own_lo = S_PUB + NPRE < KH ? S_PUB + NPRE : KH, own_hi = S_E1 + NPRE < KH ? S_E1 + NPRE : KH;
    constexpr int NXR = XF ? 3 * KX : 2 * KX;         // vector-memory requests of one x fetch
    constexpr int N_X_AFTER_PUB = (X_LATE || S_XSPL >= S_E1) ? 0 : NXR;      // (S_XSPL == S_E1: the arrival precedes the x requests in its k step)
    constexpr int N_AFTER_PUB = NH * (own_hi - own_lo) + N_X_AFTER_PUB;
    static_assert(!HF || (!PEND && !TRAIN && XF && CELL == 0 && !ULR), "HF: the flag-protocol LSTM instantiations with a frag3 x operand");
    static_assert(NST >= 5 && S_XSPL >= KX && S_XSPL > S_PUB && KH - NPRE <= S_E2 && S_HEAD > S_PUB && S_HEAD <= S_E2, "slice schedule");
    // flag protocol: the flags are requested POLL_LEAD k steps before they are looked at.  Two k steps (~1 k cycles) cover the load's round
    // trip; with one the check waited for it: LSTM-512 5.94-5.98 -> 5.78-5.85 ms in four alternating rounds, three k steps 5.82-5.85
    // (profiles/r04_rr_poll_lead_ab.log)
#ifndef RR_POLL_LEAD_RR
#define RR_POLL_LEAD_RR (NST >= 9 ? 2 : RR_POLL_LEAD)
#endif
#ifndef RR_HF_POLL_LEAD
#define RR_HF_POLL_LEAD RR_POLL_LEAD_RR
#endif
    constexpr int POLL_LEAD = HF ? RR_HF_POLL_LEAD : RR_POLL_LEAD_RR;
    static_assert(PEND ? S_XSPL < S_E2 : (S_E1 < S_E2 && S_E1 > S_PUB && S_XSPL < S_E2 - POLL_LEAD), "slice schedule");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    rr_v4u *ULs = reinterpret_cast<rr_v4u *>(smem);                       // [4][KH][2] blocks (not with ULR)
    rr_v4u *WXs = ULs + ((ULR || HF) ? 0 : 4 * KH * 2 * 64);              // [4][KX][2][3] blocks
    rr_v4u *red = WXs + 4 * KX * 6 * 64;                                  // [dst 4][src 4][2] blocks: split-K exchange
    float *hx = reinterpret_cast<float *>(red + 32 * 64);                 // [32][RR_HX_LD] h exchange (f32: the output wave's rows)
    unsigned *hs = reinterpret_cast<unsigned *>(hx + 32 * RR_HX_LD);      // [3 images][32][RR_HS_LD] the same h, already split (the publishing wave's rows)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, kh = lane >> 5;
    const int bt = blockIdx.x % p.NBT;
    const int ct = blockIdx.x / p.NBT;
    const int bt_abs = p.b_base / 64 + bt;
    const int b0 = p.b_base + bt * 64;
    const int H = p.H, T = p.T;
    const int NKS = H >> 4;                          // k steps the hand-off stores per row block (H % 16 == 0)
    const int rows_valid = p.B - b0 < 64 ? p.B - b0 : 64;

    // ---- resident operands ----
    const rr_v4u *img = p.img + (size_t)ct * rr_blocks_per_ct(KH, KX) * 64;
    rr_bf16x8 uh[KH][2][2];
#pragma unroll
    for (int i = 0; i < KH; ++i)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                uh[i][mt][m] = __builtin_bit_cast(rr_bf16x8, img[((((size_t)w * KH + i) * 2 + mt) * 2 + m) * 64 + lane]);
                RR_PIN_A(uh[i][mt][m]);
            }
    rr_bf16x8 ulr[ULR ? KH : 1][2];
    if (ULR || HF) {
        if constexpr (ULR) {
#pragma unroll
        for (int i = 0; i < KH; ++i)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                ulr[i][mt] = __builtin_bit_cast(rr_bf16x8, img[(size_t)4 * KH * 4 * 64 + ((w * KH + i) * 2 + mt) * 64 + lane]);
        }
        const rr_v4u *src = img + (size_t)4 * KH * 4 * 64 + (size_t)4 * KH * 2 * 64;      // WX only
        constexpr int n16 = 4 * KX * 6 * 64;
        for (int e = tid; e < n16; e += 256) WXs[e] = src[e];
    } else {
        const rr_v4u *src = img + (size_t)4 * KH * 4 * 64;                // UL then WX, contiguous, same order as in LDS
        constexpr int n16 = (4 * KH * 2 + 4 * KX * 6) * 64;
        for (int e = tid; e < n16; e += 256) ULs[e] = src[e];
    }
    // this lane finishes hidden units jf, jf + 1 (all four gates) of batch row n of each half
    const int jl = 8 * kh + 4 * (w >> 1) + 2 * (w & 1);
    const int jf = 16 * ct + jl;
    float bsum[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 2; ++e)
            bsum[g][e] = jf + e < H ? p.bi[g * H + jf + e] + (p.bh ? p.bh[g * H + jf + e] : 0.0f) : 0.0f;
    float cst[2][2];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = b0 + half * 32 + n;
            cst[half][e] = (p.c0 && row < p.B && jf + e < H) ? p.c0[(size_t)row * H + jf + e] : 0.0f;
        }
    // ---- buffer descriptors and per-lane offsets (everything that must be clipped rides in the range-checked vector offset) ----
    const int hb_bytes = (int)p.hstep;
    // the hand-off of step t: read h_{t-1} (t == 0: the h_0 slot), write h_t -- one descriptor per timestep, 32-bit offsets inside it
    auto rs_rd = [&](int t) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(t ? p.hseq + (size_t)(t - 1) * p.hstep : p.h0f), 0, hb_bytes, 0x00020000);
    };
    auto rs_wr = [&](int t) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(p.hseq + (size_t)t * p.hstep), 0, hb_bytes, 0x00020000);
    };
    const int lane16 = lane * 16;
    // k steps past H / 16 (H between the compiled depths: 4 KH * 16 > H) are not stored: their fragments read as zeros through an
    // out-of-range vector offset
    int hvo[KH];
#pragma unroll
    for (int i = 0; i < KH; ++i) hvo[i] = w * KH + i < NKS ? lane16 : RR_OOB_F;
    // x: the tile's rows [b0, b0 + rows_valid) of [B][T][in] -- or of the time-major [T][B][in] a stacked layer hands over
    // (p.x_tm: a row's timesteps are then B * in apart and the tile's 32 rows of one step are contiguous: coalesced requests).
    // Rows past the batch are masked PER LANE AND HALF (xok): the half rides in the scalar offset, which the range check does
    // not see, so the descriptor's range alone would let half 1 read up to 32 rows past the end of the tensor.
    const long x_row_bytes = p.x_tm ? (long)p.in * 4 : (long)p.T * p.in * 4;
    const long x_step_bytes = p.x_tm ? (long)p.B * p.in * 4 : (long)p.in * 4;
    const long x_range = p.x_tm ? (long)(T - 1) * x_step_bytes + rows_valid * x_row_bytes : rows_valid * x_row_bytes;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + (size_t)b0 * (x_row_bytes / 4)), 0,
                                                                          (int)(x_range < 0x7fffffffL ? x_range : 0x7fffffffL), 0x00020000);
    const bool xok0 = n < rows_valid, xok1 = 32 + n < rows_valid;
    int xvo[KX][2];
#pragma unroll
    for (int ix = 0; ix < KX; ++ix)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = (w * KX + ix) * 16 + 8 * kh + 4 * q;
            xvo[ix][q] = XF ? (w * KX + ix < p.NKSx ? lane16 : RR_OOB_F)        // (XF: only [ix][0] is used: the block's lane offset)
                            : k < p.in ? (int)(n * x_row_bytes) + k * 4 : RR_OOB;
        }
    // publication: lane (n, kh) of the publishing wave owns hidden units 8 kh .. 8 kh + 7 of row n -- one B fragment of the
    // consumers; output rows past the batch fall outside the descriptor; hand-off rows past the batch are written too
    // (padding rows compute on zero inputs: finite, read only by themselves)
    // (p.out_tm: the sequence output goes out time-major, [T][B][H], for the next layer of a stack -- rows are then masked per lane)
    const long o_row_bytes = p.out_tm ? (long)H * 4 : (long)(p.return_sequences ? p.T : 1) * H * 4;
    const long o_step_bytes = p.out_tm ? (long)p.B * H * 4 : (long)H * 4;
    const long o_range = p.out_tm ? (long)(T - 1) * o_step_bytes + rows_valid * o_row_bytes : rows_valid * o_row_bytes;
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.out + (size_t)b0 * (o_row_bytes / 4)), 0, (int)(o_range < 0x7fffffffL ? o_range : 0x7fffffffL), 0x00020000);
    const int out_vo = (int)(n * o_row_bytes) + (16 * ct + 8 * kh) * 4;
    const int out_half = (int)(32 * o_row_bytes);
    const int out_step = (int)o_step_bytes;
    RR_BARRIER();

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using Tt = std::true_type;
    using Ff = std::false_type;
    rr_v4u hf[2][KH][NH];          // the h operand of each half
    constexpr bool ULPRE = !ULR && KH == 8 && !HF;
    rr_bf16x8 ulo_n[2];            // ULPRE: U's low image of the next h k step (LDS -> registers one k step ahead)
    rr_v4u xr[XF ? 1 : KX][2];     // raw x_t (f32) of the half that multiplies next (not with XF)
    rr_bf16x8 xf[XF ? 2 : 1][KX][3];   // ... and its three bf16 images; XF: one set per half, filled by the loads themselves

    // The operand's 3 KH fragment loads are SPREAD over the MFMA sequences instead of issued in one burst: a 1 KB wave-wide load
    // occupies the CU's address path for ~16 cycles and the four wavefronts (in lockstep) share that path, so a burst of 24
    // stalls the issuing wave -- and its MFMAs -- for ~1.5 k cycles.  Only the first NPRE k steps go out ahead (after the poll,
    // at the end of the other half's sequence); k step i >= NPRE is requested at k step i - NPRE of the half's own sequence,
    // KX + NPRE k steps before the MFMAs that consume it.
    auto issue_h = [&](auto half_tag, int t, int j0, int j1) __attribute__((always_inline)) {      // fragments [j0, j1): constants once unrolled
        constexpr int half = decltype(half_tag)::value;
        const int so = ((bt_abs * 2 + half) * NKS + w * KH) * NH * 1024;
        const __amdgpu_buffer_rsrc_t rs = rs_rd(t);
#pragma unroll
        for (int blk = 0; blk < NH * KH; ++blk) {
            if (blk < j0 || blk >= j1) continue;
            // four 1 KB blocks per scalar offset: the rest of the address rides in the instruction's immediate
            const int i = blk / NH, m = blk % NH;
            hf[half][i][m] = __builtin_amdgcn_raw_buffer_load_b128(rs, hvo[i] + (blk & 3) * 1024, so + (blk >> 2) * 4096, 16 /* sc1 */);
            RR_BOUND(t ? 3 : 4, t ? (size_t)(t - 1) * p.hstep : 0, hvo[i] + (blk & 3) * 1024, so + (blk >> 2) * 4096, hb_bytes, 16);
        }
    };
    auto issue_x = [&](auto half_tag, int t) __attribute__((always_inline)) {
        constexpr int half = decltype(half_tag)::value;
        if constexpr (XF) {
            // the half's own operand set: its MFMAs of this half-step are done, x_{t} of its NEXT step lands where they read from
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.xf3 + (size_t)t * p.xstep), 0, (int)p.xstep, 0x00020000);
            const int so = ((bt_abs * 2 + half) * p.NKSx + w * KX) * 3 * 1024;
#pragma unroll
            for (int ix = 0; ix < KX; ++ix)
#pragma unroll
                for (int m = 0; m < 3; ++m)
                {
                    xf[half][ix][m] = __builtin_bit_cast(rr_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, xvo[ix][0] + ((3 * ix + m) & 3) * 1024, so + ((3 * ix + m) >> 2) * 4096, 0));
                    RR_BOUND(1, (size_t)t * p.xstep, xvo[ix][0] + ((3 * ix + m) & 3) * 1024, so + ((3 * ix + m) >> 2) * 4096, p.xstep, 16);
                }
        } else {
            const int so = (int)((half * 32) * x_row_bytes) + t * (int)x_step_bytes;
            const bool ok = half ? xok1 : xok0;
#pragma unroll
            for (int ix = 0; ix < KX; ++ix)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    xr[ix][q] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? xvo[ix][q] : RR_OOB, so, 0);
                    RR_BOUND(0, (size_t)b0 * x_row_bytes, ok ? xvo[ix][q] : RR_OOB, so, x_range < 0x7fffffffL ? x_range : 0x7fffffffL, 16);
                }
        }
    };
    auto split_x = [&]() __attribute__((always_inline)) {
        if constexpr (XF) return;
#pragma unroll
        for (int ix = 0; ix < KX; ++ix) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                v[4 * q] = __uint_as_float(xr[ix][q].x); v[4 * q + 1] = __uint_as_float(xr[ix][q].y);
                v[4 * q + 2] = __uint_as_float(xr[ix][q].z); v[4 * q + 3] = __uint_as_float(xr[ix][q].w);
            }
            rr_v4u a, b, c;
            rr_split8(v, a, b, c);
            xf[0][ix][0] = __builtin_bit_cast(rr_bf16x8, a); xf[0][ix][1] = __builtin_bit_cast(rr_bf16x8, b); xf[0][ix][2] = __builtin_bit_cast(rr_bf16x8, c);
        }
    };
    // ---- the finish of a half, in slices ----
    float z[4][2];
    auto fin_reduce = [&]() __attribute__((always_inline)) {
        RR_BARRIER();                                   // every wave's partial sums are in `red`
        // (Measured and not kept: a few idle cycles per wave index behind this barrier, so that the four wavefronts do not issue every LDS /
        // vector-memory burst of the half-step at the same moment: LSTM-512 +0.1 ms, GRU pair +-0: profiles/r05_rr_skew.log)
        rr_v4u s0[4], s1[4];
#pragma unroll
        for (int src = 0; src < 4; ++src) {
            if (RR_DBG(256)) { s0[src] = (rr_v4u){0x3c000000u + lane, 0x3c100000u, 0x3c200000u, 0x3c300000u}; s1[src] = s0[src]; continue; }
            s0[src] = red[((w * 4 + src) * 2 + 0) * 64 + lane];
            s1[src] = red[((w * 4 + src) * 2 + 1) * 64 + lane];
        }
        auto sum4 = [&](unsigned a, unsigned b, unsigned c, unsigned d) {     // fixed order: ((w0 + w1) + w2) + w3
            return ((__uint_as_float(a) + __uint_as_float(b)) + __uint_as_float(c)) + __uint_as_float(d);
        };
        z[0][0] = sum4(s0[0].x, s0[1].x, s0[2].x, s0[3].x); z[0][1] = sum4(s0[0].y, s0[1].y, s0[2].y, s0[3].y);
        z[1][0] = sum4(s0[0].z, s0[1].z, s0[2].z, s0[3].z); z[1][1] = sum4(s0[0].w, s0[1].w, s0[2].w, s0[3].w);
        z[2][0] = sum4(s1[0].x, s1[1].x, s1[2].x, s1[3].x); z[2][1] = sum4(s1[0].y, s1[1].y, s1[2].y, s1[3].y);
        z[3][0] = sum4(s1[0].z, s1[1].z, s1[2].z, s1[3].z); z[3][1] = sum4(s1[0].w, s1[1].w, s1[2].w, s1[3].w);
    };
    auto fin_gates = [&](auto half_tag, int t) __attribute__((always_inline)) {      // lstm.c:201-238: Z = xW + b_i + hU (+ b_h); blocks i | f | g | o
        constexpr int half = decltype(half_tag)::value;
        float hn[2], zc[4][2], ac[4][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (RR_DBG(64)) { hn[e] = z[0][e] + z[1][e] + z[2][e] + z[3][e] + cst[half][e]; continue; }
#pragma unroll
            for (int g = 0; g < 4; ++g) zc[g][e] = HF ? fmaf(z[g][e], p.z_scale, bsum[g][e]) : z[g][e] + bsum[g][e];
            if (CELL == 1) {
                // gru.c:144-186 (same expressions as rec_persistent_kernel's): slots z | r | h.U_h + b_h | x.W_h + b_i
                const float zg = nntk_fast_sigmoid(zc[0][e]);
                const float rg = nntk_fast_sigmoid(zc[1][e]);
                const float ht = nntk_fast_tanh(fmaf(rg, zc[2][e], zc[3][e]));
                hn[e] = fmaf(-zg + 1.0f, ht, zg * cst[half][e]);
                cst[half][e] = hn[e];
                if (TRAIN) {      // the BPTT caches (train.hip gru_train_fwd_step_kernel's): Z_z | Z_r | Z_h | z | r | h~, and h.U_h + b_h
                    zc[2][e] = fmaf(rg, zc[2][e], zc[3][e]);       // Z_h; zc[3] keeps h.U_h + b_h from here on
                    zc[3][e] = z[2][e] + bsum[2][e];
                    ac[0][e] = zg; ac[1][e] = rg; ac[2][e] = ht;
                }
                continue;
            }
            const float ig = nntk_fast_sigmoid(zc[0][e]);
            const float fg = nntk_fast_sigmoid(zc[1][e]);
            const float gg = nntk_fast_tanh(zc[2][e]);
            const float og = nntk_fast_sigmoid(zc[3][e]);
            const float cn = fmaf(fg, cst[half][e], ig * gg);
            cst[half][e] = cn;
            hn[e] = og * nntk_fast_tanh(cn);
            ac[0][e] = ig; ac[1][e] = fg; ac[2][e] = gg; ac[3][e] = og;
        }
        *reinterpret_cast<float2 *>(hx + n * RR_HX_LD + jl) = make_float2(hn[0], hn[1]);
        // every lane splits its own two hidden units (11 instructions in each of the four waves, here in the reduce slice, instead of 44
        // in the publishing wave between its barrier and its stores: that wave's half-steps were ~400 cycles longer than the others').
        // (Publishing from here as well -- every lane storing its own word of the fragment slot, three write-through 4-byte stores, no
        // LDS round trip and no publishing wave -- was measured SLOWER with the pending-pattern protocol: GRU-256 pair 10.33 vs 9.90 ms,
        // profiles/r04_rr_direct_publication.log: the quarter-filled stores reach the consumers later than three full 1 KB ones.)
        if (HF && !RR_DBG(128)) {                       // the hand-off's two f16 words (images 0, 1 of `hs`)
            unsigned sh, sl;
            rr_split_pair_h2(hn[0], hn[1], sh, sl);
            unsigned *d = hs + n * RR_HS_LD + (jl >> 1);
            d[0] = sh; d[32 * RR_HS_LD] = sl;
        } else if (!RR_DBG(128)) {
            unsigned sh, sm, sl;
            rr_split_pair(hn[0], hn[1], sh, sm, sl);
            // PEND: no published word may equal the "not yet written" pattern (two bf16 NaNs with every payload bit set): such a pair
            // becomes another NaN
            if (PEND) { sh = min(sh, 0xfffffffeu); sm = min(sm, 0xfffffffeu); sl = min(sl, 0xfffffffeu); }
            unsigned *d = hs + n * RR_HS_LD + (jl >> 1);
            d[0] = sh; d[32 * RR_HS_LD] = sm; d[2 * 32 * RR_HS_LD] = sl;
        }
        if (TRAIN && CELL == 1) {
            const int row = b0 + half * 32 + n;
            if (row < p.B && jf + 1 < H + 1) {
                float *zrow = p.z_cache + ((size_t)row * T + t) * 6 * H + jf;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    *reinterpret_cast<float2 *>(zrow + g * H) = make_float2(zc[g][0], zc[g][1]);
                    *reinterpret_cast<float2 *>(zrow + (3 + g) * H) = make_float2(ac[g][0], ac[g][1]);
                }
                *reinterpret_cast<float2 *>(p.c_cache + ((size_t)row * T + t) * H + jf) = make_float2(zc[3][0], zc[3][1]);
            }
        } else if (TRAIN) {
            const int row = b0 + half * 32 + n;
            if (row < p.B && jf + 1 < H + 1) {                 // H % 2 == 0 here (H % 16 == 0): both cells or none
                float *zrow = p.z_cache + ((size_t)row * T + t) * 8 * H + jf;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *reinterpret_cast<float2 *>(zrow + g * H) = make_float2(zc[g][0], zc[g][1]);
                    *reinterpret_cast<float2 *>(zrow + (4 + g) * H) = make_float2(ac[g][0], ac[g][1]);
                }
                *reinterpret_cast<float2 *>(p.c_cache + ((size_t)row * T + t) * H + jf) = make_float2(cst[half][0], cst[half][1]);
            }
        }
    };
    auto fin_publish = [&](auto half_tag, auto last_tag, int t) __attribute__((always_inline)) {
        constexpr int half = decltype(half_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;     // the half's last step: final state / last output leave from here
        RR_BARRIER();                                   // the half's h row pieces are in `hx`
        if (w == 2 * half) {                            // the publishing wave: three full write-through stores
            rr_v4u a, b, c;
            if (!RR_DBG(128)) {
                const unsigned *src = hs + n * RR_HS_LD + 4 * kh;
                a = *reinterpret_cast<const rr_v4u *>(src);
                b = *reinterpret_cast<const rr_v4u *>(src + 32 * RR_HS_LD);
                if (!HF) c = *reinterpret_cast<const rr_v4u *>(src + 2 * 32 * RR_HS_LD);
            }
            // the block's offset rides in the VECTOR offset and soffset stays immediate 0: a wide buffer store with an SGPR
            // soffset followed by a VALU write of its data registers stores the overwritten value in some lanes on MI355X, and
            // the compiler inserts no wait state for that form (tools/check_store_hazard.py, tests/test_isa_lint.py)
            const int vo = lane16 + (((bt_abs * 2 + half) * NKS + ct) * NH) * 1024;
            if (!RR_DBG(128)) {
                const __amdgpu_buffer_rsrc_t rs = rs_wr(t);
                __builtin_amdgcn_raw_buffer_store_b128(a, rs, vo, 0, RR_ST_AUX /* sc1 */);
                __builtin_amdgcn_raw_buffer_store_b128(b, rs, vo + 1024, 0, RR_ST_AUX);
                if (!HF) __builtin_amdgcn_raw_buffer_store_b128(c, rs, vo + 2048, 0, RR_ST_AUX);
                RR_BOUND(3, (size_t)t * p.hstep, vo + (NH - 1) * 1024, 0, hb_bytes, 16);
            }
        } else if (w == 2 * half + 1) {                 // the output wave: the same row pieces in f32
            const rr_v4u o0 = *reinterpret_cast<const rr_v4u *>(hx + n * RR_HX_LD + 8 * kh);
            const rr_v4u o1 = *reinterpret_cast<const rr_v4u *>(hx + n * RR_HX_LD + 8 * kh + 4);
            if (p.return_sequences && p.out && !RR_DBG(1024)) {      // (p.out NULL: the caller takes the layer output in frag3 form -- hseq -- only)
                const int vo = (half ? xok1 : xok0) ? out_vo + half * out_half + t * out_step : 0x7fff0000;   // (past every range, room for + 16); soffset immediate, as above
                __builtin_amdgcn_raw_buffer_store_b128(o0, rso, vo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o1, rso, vo + 16, 0, 0);
                RR_BOUND(2, (size_t)b0 * o_row_bytes, vo + 16, 0, o_range < 0x7fffffffL ? o_range : 0x7fffffffL, 16);
            } else if (p.out_h2 && !RR_DBG(1024)) {
                // the same eight values as the FRAG2H block of (row block, k step ct): this lane IS lane 32 kh + n of the fragment, so the two
                // images leave as two coalesced 1 KB stores -- in the slot of the two f32 stores above (the host passes one form or the other).
                // Ordinary stores: the consumer is a later kernel
                const float v8[8] = {__uint_as_float(o0.x), __uint_as_float(o0.y), __uint_as_float(o0.z), __uint_as_float(o0.w),
                                     __uint_as_float(o1.x), __uint_as_float(o1.y), __uint_as_float(o1.z), __uint_as_float(o1.w)};
                rr_v4u h2hi, h2lo;
                rr_split8_h2(v8, h2hi, h2lo);
                const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.out_h2 + (size_t)t * p.h2step), 0, (int)p.h2step, 0x00020000);
                const int vo = lane16 + (((bt_abs * 2 + half) * NKS + ct) * 2) * 1024;
                __builtin_amdgcn_raw_buffer_store_b128(h2hi, rs2, vo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(h2lo, rs2, vo + 1024, 0, 0);
                RR_BOUND(5, (size_t)t * p.h2step, vo + 1024, 0, p.h2step, 16);
            }
            if (LAST) {
                int b0e = b0;
                asm volatile("" : "+s"(b0e));             // opaque: keeps these addresses from being computed (and spilled) ahead of the loop
                const int row = b0e + half * 32 + n;
                if (row < p.B) {
                    if (!p.return_sequences && p.out) {
                        float *o = p.out + (size_t)row * H + 16 * ct + 8 * kh;
                        *reinterpret_cast<rr_v4u *>(o) = o0; *reinterpret_cast<rr_v4u *>(o + 4) = o1;
                    }
                    if (p.hT) {
                        float *o = p.hT + (size_t)row * H + 16 * ct + 8 * kh;
                        *reinterpret_cast<rr_v4u *>(o) = o0; *reinterpret_cast<rr_v4u *>(o + 4) = o1;
                    }
                }
            }
        } else if (PEND && w == ((2 * half + 2) & 3)) {
            // the OTHER half's publishing wave (idle in this slice) marks this column tile's blocks of step t + 2 pending (see "Hand-off
            // protocol" above); past the last step the offset is out of range and the stores are dropped
            if (!RR_DBG(128)) {
                const int vo = lane16 + (((bt_abs * 2 + half) * NKS + ct) * 3) * 1024;
                const __amdgpu_buffer_rsrc_t rs2 = rs_wr(t + 2 < T ? t + 2 : t);
                const int vo2 = t + 2 < T ? vo : RR_OOB_F;
                const rr_v4u pend = {RR_PENDING, RR_PENDING, RR_PENDING, RR_PENDING};
                __builtin_amdgcn_raw_buffer_store_b128(pend, rs2, vo2, 0, RR_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(pend, rs2, vo2 + 1024, 0, RR_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(pend, rs2, vo2 + 2048, 0, RR_ST_AUX);
                RR_BOUND(3, (size_t)(t + 2 < T ? t + 2 : t) * p.hstep, vo2 + 2048, 0, hb_bytes, 16);
            }
        }
    };
    // ---- flag protocol (!PEND) ----
    unsigned *const flags0 = p.flags + (size_t)bt * 2 * RR_FLAGS, *const flags1 = flags0 + RR_FLAGS;
    // arrival (publishing wave only): its publishing stores have drained -> raise the column tile's flag (write-through store
    // of t + 1).  Counted wait: the x request and the own-sequence operand requests issued after the publication stay in flight.
    // XLIVE: the half-step carries the x request (the last half-step of each half does not: nothing would consume it, the
    // compiler would drop the loads and the count would be 2 KX too lenient -- tools/check_rr_waits.py checks every
    // instantiation's counts against the ISA)
    auto arrive = [&](int half, int t, auto xlive_tag) __attribute__((always_inline)) {
        if (w == 2 * half) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N_AFTER_PUB - (decltype(xlive_tag)::value ? 0 : N_X_AFTER_PUB)) : "memory");
            if (lane == 0)
                __hip_atomic_store((half ? flags1 : flags0) + ct, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // every wave polls for itself: one sc1 load covers the batch tile's flags; requested POLL_LEAD k steps before it is looked at
    unsigned pv0 = 0;
    const bool flag_live = lane < p.NCT;               // lanes past the last column tile read padding words (always 0)
    auto poll_a = [&](int half) __attribute__((always_inline)) {
        const unsigned *f = (half ? flags1 : flags0) + lane;
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(pv0) : "v"(f) : "memory");
    };
    auto poll_b = [&](int half, int t) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv0) :: "memory");
        const unsigned target = (unsigned)t;
        bool ok = !flag_live || pv0 >= target;
        // spin_ticks == 0 is fault injection (tests): behave as if the very first poll had found nothing and run out of budget
        if (__builtin_amdgcn_ballot_w64(!ok) != 0 || p.spin_ticks == 0) {  // something has not arrived yet: spin (bounded)
            const unsigned *f = (half ? flags1 : flags0) + lane;
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            bool expired = p.spin_ticks == 0;
            while (!expired) {
                const unsigned a = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = !flag_live || a >= target;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                // a peer that gave up has raised the fault word: every other spin ends at once
                if (__hip_atomic_load(p.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { expired = true; break; }
                __builtin_amdgcn_s_sleep(1);
                expired = __builtin_amdgcn_s_memrealtime() - t_start > p.spin_ticks;
            }
            if (expired && lane == 0) __hip_atomic_fetch_or(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // ---- pending-pattern protocol (PEND) ----
    // validity of a fetched k step: no word of its three fragments is the pending pattern.  The fast path is 7 vector-ALU
    // instructions and an untaken branch per k step; a fragment that is still pending is requested again (bounded; a peer that gave up
    // has raised the fault word and ends every other spin at once).  spin_ticks == 0 is fault injection (tests): behave as if the
    // first look had found nothing and the budget had run out.
    // probe_h looks (the mask of lanes that hold a pending word), settle_h acts on it: the look at k step i + 1 is computed at the END
    // of k step i, in the shadow of its last MFMAs, so the multiply of k step i + 1 starts behind one scalar compare
    auto probe_h = [&](auto half_tag, int i) __attribute__((always_inline)) {
        constexpr int half = decltype(half_tag)::value;
        unsigned mx = 0;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const rr_v4u v = hf[half][i][m];
            mx = max(mx, max(max(v.x, v.y), max(v.z, v.w)));
        }
        return __builtin_amdgcn_ballot_w64(mx == RR_PENDING);
    };
    const unsigned long long inject = p.spin_ticks == 0 ? ~0ull : 0ull;
    auto settle_h = [&](auto half_tag, int t, int i, unsigned long long pend) __attribute__((always_inline)) {
        if (__builtin_expect((pend | inject) != 0, 0)) {
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            bool expired = p.spin_ticks == 0;
            while (!expired) {
                issue_h(half_tag, t, 3 * i, 3 * i + 3);
                if (probe_h(half_tag, i) == 0) break;
                if (__hip_atomic_load(p.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { expired = true; break; }
                __builtin_amdgcn_s_sleep(1);
                expired = __builtin_amdgcn_s_memrealtime() - t_start > p.spin_ticks;
            }
            if (expired && lane == 0) __hip_atomic_fetch_or(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // ---- split-K exchange: wave `dst` finishes registers 4 g + 2 (dst & 1) + {0, 1} of tile dst >> 1 (read in the NEXT half-step) ----
    auto exchange = [&](const f32x16 (&acc)[2], int d0, int d1) __attribute__((always_inline)) {
        if (RR_DBG(32)) return;
#pragma unroll
        for (int dst = 0; dst < 4; ++dst) {
            if (dst < d0 || dst >= d1) continue;
            const int mt = dst >> 1, o = 2 * (dst & 1);
            const rr_v4u q0 = {__float_as_uint(acc[mt][o]), __float_as_uint(acc[mt][o + 1]), __float_as_uint(acc[mt][4 + o]), __float_as_uint(acc[mt][5 + o])};
            const rr_v4u q1 = {__float_as_uint(acc[mt][8 + o]), __float_as_uint(acc[mt][9 + o]), __float_as_uint(acc[mt][12 + o]), __float_as_uint(acc[mt][13 + o])};
            red[((dst * 4 + w) * 2 + 0) * 64 + lane] = q0;
            red[((dst * 4 + w) * 2 + 1) * 64 + lane] = q1;
        }
    };
    // one half-step: multiply half Y at step t; FIN: finish half X = 1 - Y (its step tX) on the way; NEXT: fetch X's operand of
    // step tX + 1 (POLL: it was published inside this launch)
    auto half_step = [&](auto y_tag, auto fin_tag, auto next_tag, auto poll_tag, auto last_tag, auto xlive_tag, int t, int tX) __attribute__((always_inline)) {
        constexpr bool XLIVE = decltype(xlive_tag)::value;      // x_{t+1} of this half exists and a later half-step consumes it
        constexpr int Y = decltype(y_tag)::value;
        using XT = std::integral_constant<int, 1 - Y>;
        constexpr int X = 1 - Y;
        constexpr bool FIN = decltype(fin_tag)::value, NEXT = decltype(next_tag)::value, POLL = decltype(poll_tag)::value;
        constexpr bool YCHK = PEND;      // this half's operand is checked where it is used (at run time: from its second step on)
        f32x16 acc[2];
        unsigned long long pend_next = 0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
        // six products per (k step, tile), smallest terms first: (A image, B image) = (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            RR_STAMP(Y, t, s);                            // diagnostics build: start of every k step
            // ---- this k step's MFMA operands from LDS, requested before the slice's own LDS traffic ----
            rr_bf16x8 wa[2][3], ulo[2];
            if (s < KX) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int m = 0; m < 3; ++m)
                        wa[mt][m] = __builtin_bit_cast(rr_bf16x8, WXs[((((w * KX + s) * 2 + mt) * 3) + m) * 64 + lane]);
            } else if (ULR) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) ulo[mt] = ulr[s - KX][mt];
            } else if (HF) {                              // (both images of U are in registers)
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    ulo[mt] = ULPRE ? ulo_n[mt]           // requested during the previous k step (below)
                                    : __builtin_bit_cast(rr_bf16x8, ULs[((w * KH + (s - KX)) * 2 + mt) * 64 + lane]);
            }
            // ---- slices of the other half's finish and of its next fetch ----
            // (issuing 2 / 4 / 6 of this k step's x products BEFORE the reduce slice's barrier -- they do not depend on it -- measured
            // +0.1 ms on LSTM-512 and nothing on GRU-256: profiles/r04_rr_exchange_ab.log)
            if (s == S_RED && FIN && !RR_DBG(2)) { fin_reduce(); fin_gates(XT{}, tX); }
            if (s == S_PUB && FIN && !RR_DBG(2)) { fin_publish(XT{}, last_tag, tX); }
            if (!PEND) {
                if (s == S_E1 && FIN && !RR_DBG(4)) { arrive(X, tX, xlive_tag); }
                if (s == S_E2 - POLL_LEAD && NEXT && POLL && !RR_DBG(4)) poll_a(X);
                if (s == S_E2 && NEXT && POLL && !RR_DBG(4)) poll_b(X, tX + 1);
            }
            if (s == S_HEAD && NEXT && !RR_DBG(1)) issue_h(XT{}, tX + 1, 0, NH * NPRE);        // head of X's next operand (PEND: speculative, checked where it is used)
            if (s == S_E2 && X_LATE && XLIVE && !RR_DBG(8)) issue_x(y_tag, t + 1);            // (xr is free since this half-step's S_XSPL)
            if (s + NPRE < KH && !RR_DBG(1)) issue_h(y_tag, t, NH * (s + NPRE), NH * (s + NPRE + 1));       // THIS half's operand, k step s + NPRE (published long ago)
            // ---- multiply ----
            if (RR_DBG(16)) {
            } else if (s < KX) {
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[mt][PA[pr]], xf[XF ? Y : 0][s][PB[pr]], acc[mt], 0, 0, 0);
            } else {
                const int i = s - KX;
                if (YCHK && t > 0 && !RR_DBG(4) && !RR_DBG(1)) settle_h(y_tag, t, i, pend_next);   // (step 0 reads the h_0 slot, written before the launch)
                if constexpr (HF) {
                    // three products per tile, smallest terms first: (U image, h image) = (hi, lo) (lo, hi) (hi, hi); the last k step tile by tile, as below
                    constexpr int QA[3] = {0, 1, 0}, QB[3] = {1, 0, 0};
                    rr_f16x8 bh[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) bh[m] = __builtin_bit_cast(rr_f16x8, hf[Y][i][m]);
                    if (s == NST - 1) {
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                            for (int pr = 0; pr < 3; ++pr)
                                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rr_f16x8, uh[i][mt][QA[pr]]), bh[QB[pr]], acc[mt], 0, 0, 0);
                            if (mt == 0) exchange(acc, 0, 2);
                        }
                    } else {
#pragma unroll
                        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
                                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rr_f16x8, uh[i][mt][QA[pr]]), bh[QB[pr]], acc[mt], 0, 0, 0);
                    }
                } else {
                rr_bf16x8 b[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) b[m] = __builtin_bit_cast(rr_bf16x8, hf[Y][i][m]);
                if (s == NST - 1) {
                    // the half-step's last k step runs tile by tile (the same six products per tile in the same order): tile 0's
                    // partial sums leave for the exchange while tile 1 still multiplies
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                        for (int pr = 0; pr < 6; ++pr) {
                            const rr_bf16x8 av = PA[pr] == 2 ? ulo[mt] : uh[i][mt][PA[pr]];
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[PB[pr]], acc[mt], 0, 0, 0);
                        }
                        if (mt == 0) exchange(acc, 0, 2);
                    }
                } else {
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const rr_bf16x8 av = PA[pr] == 2 ? ulo[mt] : uh[i][mt][PA[pr]];
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[PB[pr]], acc[mt], 0, 0, 0);
                    }
                }
                }
            }
            if (YCHK && s + 1 >= KX && s + 1 < NST && !RR_DBG(4) && !RR_DBG(1)) pend_next = probe_h(y_tag, s + 1 - KX);
            // U's low image of the NEXT k step: an LDS round trip (~100+ cycles with four waves reading) ahead of the product that
            // needs it (the second of the six).  KH = 8: LSTM-512 6.04 -> 5.98 ms in four alternating rounds; nothing at KH = 4
            // (profiles/r04_rr_ulo_prefetch_ab.log; the same for W's images in the x k steps: no difference on either shape)
            if (ULPRE && s + 1 >= KX && s + 1 < NST) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) ulo_n[mt] = __builtin_bit_cast(rr_bf16x8, ULs[((w * KH + (s + 1 - KX)) * 2 + mt) * 64 + lane]);
            }
            // x_t of the half that multiplies next: split (this half's own x part has been multiplied), then request this
            // half's x_{t+1} -- a whole half-step ahead of its use
            if (s == S_XSPL) {
                if (NEXT && !RR_DBG(8)) split_x();
                if (!X_LATE && XLIVE && !RR_DBG(8)) issue_x(y_tag, t + 1);  // the arrival's counted wait counts these requests
            }
#ifndef RR_NO_INTERLEAVE
            // spread this k step's vector-memory instructions between its MFMAs (2 MFMAs, then at most 1 memory operation,
            // six times): the four wavefronts run in lockstep and share one address path, so back-to-back requests queue up
            // behind each other and stall the issuing wave together with its MFMAs
#ifdef RR_VALU_IL
#pragma unroll
            for (int j = 0; j < 12; ++j) {                // ... and the slice's vector-ALU work in the MFMAs' shadow, RR_VALU_IL per MFMA
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, RR_VALU_IL, 0);
                if (j & 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
#else
            if (s == NST - 1) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {             // tile 1's MFMAs, tile 0's four LDS writes and the rest of the requests between them
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            }
#endif
#endif
            __builtin_amdgcn_sched_barrier(0);            // a slice stays with its k step
        }
        RR_STAMP(Y, t, NST);
        // (tile 1's half of the split-K exchange; tile 0's went out inside the last k step)
        exchange(acc, 2, 4);
    };

    // prologue: operands of half A, step 0 (h_0 sits in parity 0: no poll); x_0 of half B
    issue_x(I0{}, 0);
    issue_h(I0{}, 0, 0, NH * NPRE);
    split_x();
    issue_x(I1{}, 0);
    if (T > 1) {
        half_step(I0{}, Ff{}, Tt{}, Ff{}, Ff{}, Tt{}, 0, -1);                   // A(0); fetch B(0)
        half_step(I1{}, Tt{}, Tt{}, Tt{}, Ff{}, Tt{}, 0, 0);                    // B(0); finish A(0); fetch A(1)
        for (int t = 1; t < T - 1; ++t) {
            half_step(I0{}, Tt{}, Tt{}, Tt{}, Ff{}, Tt{}, t, t - 1);            // A(t); finish B(t-1); fetch B(t)
            half_step(I1{}, Tt{}, Tt{}, Tt{}, Ff{}, Tt{}, t, t);                // B(t); finish A(t); fetch A(t+1)
        }
        half_step(I0{}, Tt{}, Tt{}, Tt{}, Ff{}, Ff{}, T - 1, T - 2);            // A(T-1); finish B(T-2); fetch B(T-1)
    } else {
        half_step(I0{}, Ff{}, Tt{}, Ff{}, Ff{}, Ff{}, 0, -1);                   // T == 1: A(0); fetch B(0)
    }
    half_step(I1{}, Tt{}, Ff{}, Ff{}, Tt{}, Ff{}, T - 1, T - 1);                // B(T-1); finish A(T-1): its last step
    // drain: finish B(T-1)
    fin_reduce();
    fin_gates(I1{}, T - 1);
    fin_publish(I1{}, Tt{}, T - 1);
    // ---- final cell state ----
    int b0e = b0;
    asm volatile("" : "+s"(b0e));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int row = b0e + half * 32 + n;
        if (p.cT && row < p.B) {
            if (jf < H) p.cT[(size_t)row * H + jf] = cst[half][0];
            if (jf + 1 < H) p.cT[(size_t)row * H + jf + 1] = cst[half][1];
        }
    }
}
template <int KH, int KX, bool TRAIN = false, bool XF = false, bool HF = false>
__global__ __launch_bounds__(256) void lstm_rr_kernel(RRParams p) { rr_body<KH, KX, TRAIN, 0, XF, HF>(p); }
template <int KH, int KX, bool TRAIN = false, bool XF = false>
__global__ __launch_bounds__(256) void gru_rr_kernel(RRParams p) { rr_body<KH, KX, TRAIN, 1, XF>(p); }

// ---- host side --------------------------------------------------------------------------------------------------
#ifdef NNTK_RR_BOUNDS
static unsigned long long *g_rr_bounds = nullptr;
// copies the eight recorded extents to the host and clears them (diagnostics build only; synchronises the device)
extern "C" int nntk_shim_rr_bounds_fetch(unsigned long long out[8]) {
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (!g_rr_bounds) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(out, g_rr_bounds, 64, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    (void)hipMemset(g_rr_bounds, 0, 64);
    return 0;
}
#endif
// xf: x arrives as a frag3 tensor (any in: the pack kernel zero-pads to whole k steps); else f32 rows read 16 bytes at a time
static bool rr_shape(int H, int in, bool xf, int *KH, int *KX) {
    if (H < 64 || H > 512 || (H % 16) != 0 || in < 1) return false;
    if (!xf && (in < 8 || (in % 8) != 0)) return false;
    *KH = H <= 256 ? 4 : 8;
    // in <= 256 (KX = 4) needs 96 KB of LDS for the W images: it fits once U's low image moves to registers, which the register
    // budget allows at KH = 4 (H <= 256) only
    *KX = in <= 64 ? 1 : in <= 128 ? 2 : (in <= 256 && *KH == 4) ? 4 : 0;
    return *KX != 0;
}
// the HF instantiations (H > 256): U has no LDS image, so the 96 KB of W images of a 256-wide input fit there too
static bool rr_shape_hf(int H, int in, int *KH, int *KX) {
    if (H <= 256 || H > 512 || (H % 16) != 0 || in < 1) return false;
    *KH = 8;
    *KX = in <= 64 ? 1 : in <= 128 ? 2 : in <= 256 ? 4 : 0;
    return *KX != 0;
}
static size_t rr_lds_bytes(int KH, int KX, bool train = false, bool hf = false) {
    const bool ulr = hf || KX > 2 || (RR_ULR8 && KH == 8 && !train);
    return (size_t)((ulr ? 0 : 4 * KH * 2) + 4 * KX * 6 + 32) * 1024 + 32 * RR_HX_LD * 4 + 3 * 32 * RR_HS_LD * 4;
}
// one timestep of the frag3 hand-off / layer output: [NHT = 2 * batch tiles][H / 16 k steps][3 images] blocks of 1 KB
static size_t rr_step_bytes(int B, int H) { return (size_t)((B + 63) / 64) * 2 * (H / 16) * 3 * 1024; }

extern "C" size_t nntk_shim_lstm_rr_image_floats(int H, int in) {
    int KH, KX;
    if (!rr_shape(H, in, false, &KH, &KX)) return 0;
    return (size_t)((H + 15) / 16) * rr_blocks_per_ct(KH, KX) * 256;        // 1 KB blocks, in floats
}
// the same for a layer whose input arrives in frag3 form (any in up to the LDS limit)
extern "C" size_t nntk_shim_rr_image_floats_xf(int H, int in) {
    int KH, KX;
    if (!rr_shape(H, in, true, &KH, &KX)) return 0;
    return (size_t)((H + 15) / 16) * rr_blocks_per_ct(KH, KX) * 256;
}
// d_work: the h_0 slot (one timestep of the hand-off) followed by the flag words
extern "C" size_t nntk_shim_lstm_rr_work_floats(int B, int H) {
    const size_t nht = (size_t)(B + 127) / 128 * 4;        // flag words for whole 128-row tiles (the four-stream kernels' unit)
    return rr_step_bytes(B, H) / 4 + nht * RR_FLAGS;
}
// d_hseq: T timesteps of the hand-off = the layer output as a frag3 tensor (same size as nntk_shim_frag3_floats(B, T, H))
extern "C" size_t nntk_shim_rr_hseq_floats(int B, int T, int H) { return (size_t)T * (rr_step_bytes(B, H) / 4); }
// d_ut / d_wp: the per-gate U^T and packed W^T the other kernels use (host: core_upload); d_img: nntk_shim_lstm_rr_image_floats
extern "C" int nntk_shim_lstm_rr_pack(const float *d_ut, const float *d_wp, float *d_img, int H, int in) {
    int KH, KX;
    if (!rr_shape(H, in, true, &KH, &KX)) return nntk_fail_msg("lstm_rr_pack: shape not taken by the register-resident kernel");
    const int Hj_p = (H + 15) & ~15, Hk_p = (H + 31) & ~31;
    int Kin_p, N_p;
    nntk_shim_conv_pack_sizes(in, 4 * H, 1, &Kin_p, &N_p);
    const int NCT = (H + 15) / 16;
    const long total = (long)NCT * rr_blocks_per_ct(KH, KX) * 64;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(rr_pack_kernel<false>, dim3((unsigned)g), dim3(256), 0, nntk_stream(), d_ut, d_wp, (rr_v4u *)d_img, H, in,
                       Hj_p, Hk_p, Kin_p, KH, KX, NCT);
    NNTK_LAUNCH_CHECK("rr_pack_kernel");
    return 0;
}
// the same images from the caller-layout weights U [H][4H], W [in][4H] already on the device (training forward)
extern "C" int nntk_shim_lstm_rr_pack_raw(const float *d_U, const float *d_W, float *d_img, int H, int in) {
    int KH, KX;
    if (!rr_shape(H, in, true, &KH, &KX)) return nntk_fail_msg("lstm_rr_pack_raw: shape not taken by the register-resident kernel");
    const int NCT = (H + 15) / 16;
    const long total = (long)NCT * rr_blocks_per_ct(KH, KX) * 64;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(rr_pack_kernel<true>, dim3((unsigned)g), dim3(256), 0, nntk_stream(), d_U, d_W, (rr_v4u *)d_img, H, in,
                       0, 0, 0, KH, KX, NCT);
    NNTK_LAUNCH_CHECK("rr_pack_kernel");
    return 0;
}

// 0 = launched; 1 = shape / configuration not taken (the caller runs projection GEMM + rec_persistent_kernel); -1 = error
// d_x: f32 [B][T][in] (or time-major with x_tm), or NULL with d_xf3 = the same tensor in frag3 form (nntk_shim_frag3_pack);
// d_out: f32 layer output or NULL (the caller takes it in frag3 form: d_hseq); d_hseq: nntk_shim_rr_hseq_floats(B, T, H) floats
int nntk_fk_launch(RRParams q, const float *d_imgfk, int cell, size_t *launches);      // recurrent_fk.hip
struct RRIo {
    const float *img4;        // images of the full-K kernels (nntk_shim_fk_pack) or NULL
    const float *x; const void *xf3; float *out; float *hseq; float *work;
    const float *h0, *c0; float *hT, *cT; float *c_cache, *z_cache;
    int x_tm, out_tm;
    float *out_h2;            // the sequence output as a FRAG2H tensor (frag3.hip) instead of the f32 rows, or NULL
    int hf;                   // the HF instantiation: `hseq` is a FRAG2H tensor (two images per block), `img` packed by nntk_shim_lstm_rr_pack_hf
    float z_scale;            // ... and the scale its sums carry, inverted
};
static int rr_launch(const RRIo &io, const float *d_img, const float *d_bi, const float *d_bh,
                     int B, int T, int in, int H, int return_sequences, int cell);

extern "C" int nntk_shim_lstm_rr(const float *d_x, const void *d_xf3, const float *d_img, const float *d_img4, const float *d_bi, const float *d_bh,
                                 const float *d_h0, const float *d_c0, float *d_out, float *d_out_h2, float *d_hseq, float *d_hT, float *d_cT,
                                 float *d_work, int B, int T, int in, int H, int return_sequences) {
    RRIo io = {d_img4, d_x, d_xf3, d_out, d_hseq, d_work, d_h0, d_c0, d_hT, d_cT, nullptr, nullptr, 0, 0, d_out_h2, 0, 0.0f};
    return rr_launch(io, d_img, d_bi, d_bh, B, T, in, H, return_sequences, 0);
}
// The HF instantiations (H > 256: KH = 8): zero initial state, x as a frag3 tensor, the sequence output ONLY as the FRAG2H tensor the
// kernel's hand-off is (d_h2: nntk_shim_frag2h_floats(B, T, H) floats).  d_img: nntk_shim_lstm_rr_pack_hf(.., uscale, wscale = 2^15 uscale),
// z_scale = 1 / wscale.  1 = shape not taken.
extern "C" int nntk_shim_lstm_rr_hf_ok(int H, int in) {
    int KH, KX;
    return rr_shape_hf(H, in, &KH, &KX) && nntk_options().rec_hf != 0;
}
extern "C" size_t nntk_shim_lstm_rr_hf_image_floats(int H, int in) {
    int KH, KX;
    if (!rr_shape_hf(H, in, &KH, &KX)) return 0;
    return (size_t)((H + 15) / 16) * rr_blocks_per_ct(KH, KX) * 256;
}
extern "C" int nntk_shim_lstm_rr_pack_hf(const float *d_ut, const float *d_wp, float *d_img, int H, int in, float uscale, float wscale) {
    int KH, KX;
    if (!rr_shape_hf(H, in, &KH, &KX)) return nntk_fail_msg("lstm_rr_pack_hf: shape not taken");
    const int Hj_p = (H + 15) & ~15, Hk_p = (H + 31) & ~31;
    int Kin_p, N_p;
    nntk_shim_conv_pack_sizes(in, 4 * H, 1, &Kin_p, &N_p);
    const int NCT = (H + 15) / 16;
    const long total = (long)NCT * rr_blocks_per_ct(KH, KX) * 64;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(rr_pack_kernel<false>, dim3((unsigned)g), dim3(256), 0, nntk_stream(), d_ut, d_wp, (rr_v4u *)d_img, H, in,
                       Hj_p, Hk_p, Kin_p, KH, KX, NCT, uscale, wscale);
    NNTK_LAUNCH_CHECK("rr_pack_kernel");
    return 0;
}
extern "C" int nntk_shim_lstm_rr_hf(const void *d_xf3, const float *d_img, const float *d_bi, const float *d_bh, float *d_h2, float *d_work,
                                    int B, int T, int in, int H, float z_scale) {
    RRIo io = {nullptr, nullptr, d_xf3, nullptr, d_h2, d_work, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 1, z_scale};
    return rr_launch(io, d_img, d_bi, d_bh, B, T, in, H, 1, 0);
}
// GRU on the same kernel frame (gru_rr_kernel): d_img packed from the four-slot matrices [U_z | U_r | U_h | 0] / [W_z | W_r | 0 | W_h],
// d_b4 [4H] = b_i,z + b_h,z | b_i,r + b_h,r | b_h,h | b_i,h.  The f32 state register starts from h_0 (which is also the published operand).
// x_tm / out_tm: d_x is [T][B][in] / the sequence output is written as [T][B][H]
extern "C" int nntk_shim_gru_rr(const float *d_x, const void *d_xf3, const float *d_img, const float *d_img4, const float *d_b4, const float *d_h0, float *d_out,
                                float *d_hseq, float *d_hT, float *d_work, int B, int T, int in, int H, int return_sequences, int x_tm, int out_tm) {
    RRIo io = {d_img4, d_x, d_xf3, d_out, d_hseq, d_work, d_h0, d_h0, d_hT, nullptr, nullptr, nullptr, x_tm, out_tm, nullptr, 0, 0.0f};
    return rr_launch(io, d_img, d_b4, nullptr, B, T, in, H, return_sequences, 1);
}
// GRU training forward: zero initial state, h of every step to d_h [B][T][H], caches d_hU [B][T][H] (h.U_h + b_h) and d_Zg [B][T][6H]
extern "C" int nntk_shim_gru_rr_train_forward(const float *d_x, const float *d_img, const float *d_b4, float *d_h, float *d_hU, float *d_Zg,
                                              float *d_hseq, float *d_work, int B, int T, int in, int H) {
    RRIo io = {nullptr, d_x, nullptr, d_h, d_hseq, d_work, nullptr, nullptr, nullptr, nullptr, d_hU, d_Zg, 0, 0, nullptr, 0, 0.0f};
    return rr_launch(io, d_img, d_b4, nullptr, B, T, in, H, 1, 1);
}
// training forward: zero initial state, h of every step to d_h [B][T][H], caches d_c [B][T][H] and d_zifgo [B][T][8H]
extern "C" int nntk_shim_lstm_rr_train_forward(const float *d_x, const float *d_img, const float *d_bi, const float *d_bh,
                                               float *d_h, float *d_c, float *d_zifgo, float *d_hseq, float *d_work, int B, int T, int in, int H) {
    RRIo io = {nullptr, d_x, nullptr, d_h, d_hseq, d_work, nullptr, nullptr, nullptr, nullptr, d_c, d_zifgo, 0, 0, nullptr, 0, 0.0f};
    return rr_launch(io, d_img, d_bi, d_bh, B, T, in, H, 1, 0);
}

template <int KH, int KX>
static void (*rr_pick(int cell, bool train, bool xf))(RRParams) {
    if (cell == 1) return train ? gru_rr_kernel<KH, KX, true, false> : xf ? gru_rr_kernel<KH, KX, false, true> : gru_rr_kernel<KH, KX, false, false>;
    return train ? lstm_rr_kernel<KH, KX, true, false> : xf ? lstm_rr_kernel<KH, KX, false, true> : lstm_rr_kernel<KH, KX, false, false>;
}

static int rr_launch(const RRIo &io, const float *d_img, const float *d_bi, const float *d_bh,
                     int B, int T, int in, int H, int return_sequences, int cell) {
    if (B <= 0 || T <= 0) return 0;
    const bool xf = io.xf3 != nullptr;
    const bool train = io.c_cache != nullptr;
    if (xf && (train || io.x_tm)) return 1;
    if (!xf && !io.x) return nntk_fail_msg("lstm_rr: no input");
    if (!io.hseq || !io.work) return nntk_fail_msg("lstm_rr: no hand-off buffer");
    // time-major tensors are addressed across the whole batch with 32-bit buffer offsets
    if (io.x_tm && (double)T * B * in * 4 >= 2.0e9) return 1;
    if (io.out_tm && (!return_sequences || !io.out || (double)T * B * H * 4 >= 2.0e9)) return 1;
    // the FRAG2H output takes the f32 rows' store slot and exists in the split-K family only (the host routes full-K shapes through f32)
    if (io.out_h2 && (io.out || io.img4 || train || !return_sequences)) return nntk_fail_msg("lstm_rr: frag2h output with an f32 output / on the full-K family");
    const NntkOptions &opt = nntk_options();
    if (opt.rec_rr == 0 || opt.rec_persistent == 0 || nntk_persistent_disabled()) return 1;
    int KH, KX;
    const bool hf = io.hf != 0;
    if (!(hf ? rr_shape_hf(H, in, &KH, &KX) : rr_shape(H, in, xf, &KH, &KX))) return 1;
    if (!xf && ((((size_t)io.x) & 15) != 0 || (in % 4) != 0)) return 1;
    if (hf && (KH != 8 || cell != 0 || !xf || train || io.h0 || io.out || io.out_h2 || io.img4 || !return_sequences || opt.rec_hf == 0)) return 1;
    const int NCT = H / 16;
    // the x and out rows of one 64-row batch tile are addressed with 32-bit buffer offsets
    if ((double)64 * T * in * 4 >= 2.0e9 || (double)64 * T * H * 4 >= 2.0e9) return 1;
    void (*kern)(RRParams) = nullptr;
    if (KH == 8 && KX == 2) kern = rr_pick<8, 2>(cell, train, xf);
    else if (KH == 8 && KX == 1) kern = rr_pick<8, 1>(cell, train, xf);
    else if (KH == 4 && KX == 4) kern = rr_pick<4, 4>(cell, train, xf);
    else if (KH == 4 && KX == 2) kern = rr_pick<4, 2>(cell, train, xf);
    else if (KH == 4 && KX == 1) kern = rr_pick<4, 1>(cell, train, xf);
    if (hf) kern = KX == 4 ? lstm_rr_kernel<8, 4, false, true, true> : KX == 2 ? lstm_rr_kernel<8, 2, false, true, true> : lstm_rr_kernel<8, 1, false, true, true>;
    if (!kern) return 1;
    const size_t lds = rr_lds_bytes(KH, KX, train, hf);
    if (lds > 160 * 1024) return 1;
    if (nntk_set_max_dynamic_lds((const void *)kern, lds)) return -1;
    const int resident = nntk_resident_blocks((const void *)kern, 256, lds, 1);
    const int tiles_per_launch = resident / NCT;
    if (tiles_per_launch < 1) return 1;
    unsigned *fault = nntk_fault_word();
    if (!fault) return 1;
    const size_t step = hf ? rr_step_bytes(B, H) / 3 * 2 : rr_step_bytes(B, H);      // (HF: two images per block)
    if (step >= (size_t)RR_OOB_F) return 1;
    const int nbt_total = (B + 63) / 64;
    const size_t xstep = (size_t)nbt_total * 2 * ((in + 15) / 16) * 3 * 1024;
    if (xf && xstep >= (size_t)RR_OOB_F) return 1;
    unsigned *flags = reinterpret_cast<unsigned *>(io.work + step / 4);
    // the h_0 slot (zeros, or h_0 split below) and the flags; the T-deep part needs no clearing: every block a step reads has been
    // written by the step before it (k steps the hand-off does not store read as zeros through out-of-range offsets)
    if (nntk_shim_memset(io.work, 0, step + (size_t)((B + 127) / 128) * 4 * RR_FLAGS * sizeof(unsigned))) return -1;
    // pending-pattern protocol (KH = 4 kernels; harmless for the others, which overwrite it before their flags let anyone look): the first
    // two timesteps of the hand-off start out "pending"; from there on every publishing wave marks its own blocks two steps ahead
    if (nntk_shim_memset(io.hseq, 0xff, step * (size_t)(T < 2 ? T : 2))) return -1;
    if (io.h0) {
        long g = ((long)nbt_total * 2 * NCT * 64 + 255) / 256;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(rr_tile_h0_kernel, dim3((unsigned)g), dim3(256), 0, nntk_stream(), io.h0, (rr_v4u *)io.work, B, H, H / 16);
    }
    RRParams q;
    q.x = io.x; q.img = (const rr_v4u *)d_img; q.bi = d_bi; q.bh = d_bh;
    q.h0f = (char *)io.work; q.hseq = (char *)io.hseq; q.hstep = step;
    q.xf3 = (const char *)io.xf3; q.xstep = xstep; q.NKSx = (in + 15) / 16;
    q.c0 = io.c0; q.cT = io.cT; q.hT = io.hT; q.out = io.out;
    q.c_cache = io.c_cache; q.z_cache = io.z_cache;
    q.x_tm = io.x_tm; q.out_tm = io.out_tm;
    q.fault = fault;
    q.spin_ticks = (unsigned long long)(opt.rec_spin_us > 0 ? opt.rec_spin_us : 0) * 100ull;
    q.B = B; q.T = T; q.H = H; q.in = in; q.NCT = NCT; q.return_sequences = return_sequences;
    q.NHT = nbt_total * 2;
    q.out_h2 = (char *)io.out_h2; q.h2step = rr_step_bytes(B, H) / 3 * 2;
    q.z_scale = io.z_scale;
#ifdef NNTK_REC_STAMPS
    q.stamp = nullptr;
    const char *stamp_path = getenv("NNTK_REC_STAMP_FILE");
    if (stamp_path) {
        if (hipMalloc((void **)&q.stamp, (size_t)(T + 1) * 64 * 8) != hipSuccess) return nntk_fail_msg("stamp alloc");
        (void)hipMemset(q.stamp, 0, (size_t)(T + 1) * 64 * 8);
    }
#endif
#ifdef NNTK_RR_BOUNDS
    static unsigned long long *g_bounds = nullptr;
    if (!g_bounds) {
        if (hipMalloc((void **)&g_bounds, 64) != hipSuccess) return nntk_fail_msg("bounds alloc");
        (void)hipMemset(g_bounds, 0, 64);
    }
    q.bounds = g_bounds;
    g_rr_bounds = g_bounds;
#endif
    const int span = nntk_prof_span_begin(NNTK_SPAN_REC);
    nntk_persistent_launch_begin();
    // H <= 256 with a 128- / 256-wide input (frag3 form): the full-K family (recurrent_fk.hip: no split-K, gates in the accumulator lanes)
    size_t launches = (size_t)(nbt_total + tiles_per_launch - 1) / tiles_per_launch;
    int took4 = 1;
    if (xf && !train && io.img4 && !io.out_tm) {
        q.flags = flags;
        took4 = nntk_fk_launch(q, io.img4, cell, &launches);
        if (took4 < 0) { nntk_persistent_launch_end(); return -1; }
    }
    if (took4 == 1)
    for (int bt0 = 0; bt0 < nbt_total; bt0 += tiles_per_launch) {
        const int nbt = nbt_total - bt0 < tiles_per_launch ? nbt_total - bt0 : tiles_per_launch;
        q.NBT = nbt; q.b_base = bt0 * 64;
        q.flags = flags + (size_t)bt0 * 2 * RR_FLAGS;
        hipLaunchKernelGGL(kern, dim3((unsigned)(nbt * NCT)), dim3(256), lds, nntk_stream(), q);
    }
    const int copy_rc = nntk_fault_enqueue_copy();
    nntk_persistent_launch_end();
    if (copy_rc) return -1;
    nntk_prof_span_end(span, (long)launches, T);
#ifdef NNTK_REC_STAMPS
    if (q.stamp) {
        (void)hipStreamSynchronize(nntk_stream());
        unsigned long long *hs = (unsigned long long *)malloc((size_t)(T + 1) * 64 * 8);
        (void)hipMemcpy(hs, q.stamp, (size_t)(T + 1) * 64 * 8, hipMemcpyDeviceToHost);
        FILE *f = fopen(stamp_path, "wb");
        if (f) { fwrite(hs, 8, (size_t)(T + 1) * 64, f); fclose(f); }
        free(hs); (void)hipFree(q.stamp);
    }
#endif
    NNTK_LAUNCH_CHECK("lstm_rr_kernel");
    static const char *const names[2][2][3] = {{{"lstm_rr_kernel<4,1>", "lstm_rr_kernel<4,2>", "lstm_rr_kernel<4,4>"}, {"lstm_rr_kernel<8,1>", "lstm_rr_kernel<8,2>", ""}},
                                               {{"gru_rr_kernel<4,1>", "gru_rr_kernel<4,2>", "gru_rr_kernel<4,4>"}, {"gru_rr_kernel<8,1>", "gru_rr_kernel<8,2>", ""}}};
    if (hf) nntk_set_last_rec_kernel(KX == 4 ? "lstm_rr_kernel<8,4,hf>" : KX == 2 ? "lstm_rr_kernel<8,2,hf>" : "lstm_rr_kernel<8,1,hf>");
    else if (took4 == 1) nntk_set_last_rec_kernel(names[cell == 1][KH == 8][KX == 1 ? 0 : KX == 2 ? 1 : 2]);
    return 0;
}
