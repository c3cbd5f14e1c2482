// recurrent_rr4.hip -- the register-resident recurrent kernels with FOUR half-streams per workgroup (GRU gru.c:129-204, :246-293;
// LSTM lstm.c:185-239, :426-475; zero or carried state).
//
// Why.  recurrent_rr.hip works two independent 32-row half-streams per workgroup in alternation: while one multiplies, the other's
// publication -> write-through -> flag -> poll -> operand fetch chain (~4 k cycles between workgroups) elapses.  That hides the chain
// only if a half-step's MFMA sequence is as long as the chain.  At H <= 256 it is not (6-8 k steps x 384 cycles = 2.3-3.1 k): the two
// GRU-256 layers of BASELINE configs[3] ran at 0.30 of the split-bf16 ceiling with the MFMA pipe 43-45 % busy and 27-36 % of the wave
// cycles waiting (profiles/r03_pmc_sq_gru.json) -- chain-bound.  More streams per workgroup give every chain more half-steps to
// hide behind.  The register file has room for the state of four streams but the chip must stay full, so the workgroup changes
// shape: 128 batch rows (four half-streams) x EIGHT hidden units (one 32 x 32 MFMA tile of 4 gate slots x 8 units) instead of
// 64 rows x 16 units -- the same 256 workgroups for 1024 sequences of GRU-256, half-steps of 6 MFMAs per k step, and a stream's
// operand is polled for TWO half-steps after its peers published it instead of within the same one.
//
// What stays: the weights of the wave's K range register-resident (hi / mid images; lo image and W^T in LDS), K = [h | x_t] split over
// the four wavefronts, partial sums exchanged through LDS and added in the fixed order ((w0 + w1) + w2) + w3, h handed over ALREADY
// SPLIT in the consumers' fragment order through the T-deep frag3 buffer (recurrent_rr_common.hpp), flag words, counted drains.
// Same products in the same order per output element: BIT-IDENTICAL to gru_rr_kernel / lstm_rr_kernel (tests/test_gpu_rr4.py), so
// the host may choose between the families by size.
//
// What is simpler: no peeled half-steps.  Every half-step does everything -- multiply stream Y, finish stream Y - 1, poll / fetch h
// for stream Y + 1, request its own next x -- and the ends of the sequence are handled by data, not by code: a finish before the first step
// and the fetches past the last one are masked through their descriptors' ranges (a zero-length descriptor drops every access),
// timesteps are clamped, and one extra half-step after the loop finishes the last stream.  One loop body, one counted wait.
// Input: x as a frag3 tensor only (the host packs f32 input first: nntk_shim_frag3_pack).  Inference only.
#include "recurrent_rr_common.hpp"

// ---- weight images: per column tile of 8 hidden units, 1 KB blocks (the A fragment of the 32-row tile for one 16-deep k step:
// lane l holds tile row l & 31, k = 8 (l >> 5) .. + 7), tile row c <-> gate slot c >> 3, hidden unit 8 ct + (c & 7):
//   UH [w][i < KH][m = hi, mid]  -> registers     UL [w][i < KH] -> LDS     WX [w][ix < KX][m = hi, mid, lo] -> LDS
// With the MFMA's D layout (lane (n, kh) holds tile rows 8 q + 4 kh + e in register 4 q + e) a lane owns all four gate slots (q) of
// hidden units 4 kh + e; wave e finishes unit 4 kh + e.
__host__ __device__ inline int rr4_blocks_per_ct(int KH, int KX) { return 4 * KH * 2 + 4 * KH + 4 * KX * 3; }

template <bool RAW>
__global__ __launch_bounds__(256) void rr4_pack_kernel(const float *__restrict__ ut, const float *__restrict__ wp,
                                                       rr_v4u *__restrict__ img, int H, int in, int Hj_p, int Hk_p, int Kin_p,
                                                       int KH, int KX, int NCT) {
    const int bpc = rr4_blocks_per_ct(KH, KX);
    const long total = (long)NCT * bpc * 64;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int l = (int)(e & 63);
        const long blk = e >> 6;
        const int ct = (int)(blk / bpc);
        int r = (int)(blk % bpc);
        int part, w, i, m;                        // part 0 = UH, 1 = UL, 2 = WX
        const int n_uh = 4 * KH * 2, n_ul = 4 * KH;
        if (r < n_uh) { part = 0; m = r & 1; i = (r >> 1) % KH; w = (r >> 1) / KH; }
        else if (r < n_uh + n_ul) { r -= n_uh; part = 1; m = 2; i = r % KH; w = r / KH; }
        else { r -= n_uh + n_ul; part = 2; m = r % 3; i = (r / 3) % KX; w = (r / 3) / KX; }
        const int c = l & 31;
        const int g = c >> 3;
        const int j = 8 * ct + (c & 7);
        const int ks = part == 2 ? w * KX + i : w * KH + i;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = 16 * ks + 8 * (l >> 5) + q;
            float val = 0.0f;
            if (j < H) {
                if (RAW) {      // the caller's own layout: U [H][4H], W [in][4H]
                    if (part == 2) { if (k < in) val = wp[(size_t)k * 4 * H + (size_t)g * H + j]; }
                    else if (k < H) val = ut[(size_t)k * 4 * H + (size_t)g * H + j];
                } else {        // per-gate U^T [4][Hj_p][Hk_p], packed W^T [4H padded][Kin_p]
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

#define RR4_HX_LD 12              // floats per row of the h exchange image (8 hidden units + pad; 16-byte aligned rows)

template <int KH, int KX, int CELL>
__device__ __forceinline__ void rr4_body(const RRParams &p) {
    constexpr int NST = KX + KH;                      // k steps one wavefront multiplies per half-step
    constexpr int S_RED = 0, S_PUB = 1, S_E1 = 3, S_E2 = NST - 1;
    constexpr int NPRE = RR_NPRE < KH ? RR_NPRE : KH;
    // vector-memory operations the publishing wave issues between its publication (S_PUB) and its arrival (S_E1): the own-sequence
    // operand requests of k steps S_PUB .. S_E1 - 1 -- what the arrival's counted wait leaves in flight
    constexpr int own_lo = S_PUB + NPRE < KH ? S_PUB + NPRE : KH, own_hi = S_E1 + NPRE < KH ? S_E1 + NPRE : KH;
    constexpr int N_AFTER_PUB = 3 * (own_hi - own_lo);
    static_assert(NST >= 5 && S_E2 - RR_POLL_LEAD >= S_E1 && KH - NPRE <= S_E2, "slice schedule");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    rr_v4u *ULs = reinterpret_cast<rr_v4u *>(smem);                       // [4][KH] blocks
    rr_v4u *WXs = ULs + 4 * KH * 64;                                      // [4][KX][3] blocks
    rr_v4u *red = WXs + 4 * KX * 3 * 64;                                  // [dst 4][src 4] blocks: split-K exchange
    float *hx = reinterpret_cast<float *>(red + 16 * 64);                 // [32][RR4_HX_LD] h exchange

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, kh = lane >> 5;
    const int bt = blockIdx.x % p.NBT;
    const int ct = blockIdx.x / p.NBT;
    const int ht0 = p.b_base / 32 + bt * 4;          // the workgroup's four half-streams = half-tiles ht0 .. ht0 + 3 of the batch
    const int H = p.H, T = p.T;
    const int NKS = H >> 4;

    // ---- resident operands ----
    const rr_v4u *img = p.img + (size_t)ct * rr4_blocks_per_ct(KH, KX) * 64;
    rr_bf16x8 uh[KH][2];
#pragma unroll
    for (int i = 0; i < KH; ++i)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            uh[i][m] = __builtin_bit_cast(rr_bf16x8, img[(((size_t)w * KH + i) * 2 + m) * 64 + lane]);
            RR_PIN_A(uh[i][m]);
        }
    {
        const rr_v4u *src = img + (size_t)4 * KH * 2 * 64;               // UL then WX, contiguous, same order as in LDS
        constexpr int n16 = (4 * KH + 4 * KX * 3) * 64;
        for (int e = tid; e < n16; e += 256) ULs[e] = src[e];
    }
    // this lane finishes hidden unit jf (all four gate slots) of batch row n of every stream
    const int jl = 4 * kh + w;
    const int jf = 8 * ct + jl;
    float bsum[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bsum[g] = p.bi[g * H + jf] + (p.bh ? p.bh[g * H + jf] : 0.0f);
    float cst[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = (ht0 + s) * 32 + n;
        cst[s] = (p.c0 && row < p.B) ? p.c0[(size_t)row * H + jf] : 0.0f;
    }
    // ---- descriptors: everything that must be masked rides in a descriptor's RANGE (0 = every access dropped) or in the
    //      range-checked vector offset ----
    const int hb_bytes = (int)p.hstep, xb_bytes = (int)p.xstep;
    auto live = [&](int s) { return ht0 + s < p.NHT; };                  // streams past the batch: nothing read, nothing written
    auto rs_rd = [&](int s, int t) __attribute__((always_inline)) {     // h_{t-1}: t == 0 the h_0 slot
        return __builtin_amdgcn_make_buffer_rsrc((void *)(t ? p.hseq + (size_t)(t - 1) * p.hstep : p.h0f), 0, live(s) ? hb_bytes : 0, 0x00020000);
    };
    auto rs_wr = [&](int s, int t) __attribute__((always_inline)) {     // h_t; t outside [0, T): dropped
        const bool ok = live(s) && t >= 0 && t < T;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(p.hseq + (size_t)(ok ? t : 0) * p.hstep), 0, ok ? hb_bytes : 0, 0x00020000);
    };
    auto rs_x = [&](int s, int t) __attribute__((always_inline)) {      // x_t; past the end: x_{T-1} again (valid memory, result unused)
        return __builtin_amdgcn_make_buffer_rsrc((void *)(p.xf3 + (size_t)(t < T ? t : T - 1) * p.xstep), 0, live(s) ? xb_bytes : 0, 0x00020000);
    };
    const int lane16 = lane * 16;
    int hvo[KH], xvo[KX];
#pragma unroll
    for (int i = 0; i < KH; ++i) hvo[i] = w * KH + i < NKS ? lane16 : RR_OOB_F;       // k steps past H / 16 read as zeros
#pragma unroll
    for (int ix = 0; ix < KX; ++ix) xvo[ix] = w * KX + ix < p.NKSx ? lane16 : RR_OOB_F;
    // f32 layer output [B][T][H] (or [B][H]): one descriptor per stream over its valid rows
    const long o_row_bytes = (long)(p.return_sequences ? T : 1) * H * 4;
    auto rs_out = [&](int s, bool ok) __attribute__((always_inline)) {
        const long rows = (long)p.B - (long)(ht0 + s) * 32;
        const long rv = rows < 0 ? 0 : rows > 32 ? 32 : rows;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(p.out + (size_t)(ht0 + s) * 32 * (o_row_bytes / 4)), 0,
                                                 (ok && p.out) ? (int)(rv * o_row_bytes) : 0, 0x00020000);
    };
    const int out_vo = (int)(n * o_row_bytes) + (8 * ct + 4 * kh) * 4;
    RR_BARRIER();

    rr_v4u hf[2][KH][3];           // the h operand of the stream that multiplies (parity of the stream index) and of the next one
    // x_t: one operand set per stream.  A stream's x_{t+1} is requested right after the poll of the half-step that multiplied x_t --
    // four half-steps before its use (the tensor streams from HBM: ~2 us), and BEHIND the flag load in the wave's in-order return
    // queue: requested before it, the poll's vmcnt(0) would wait for HBM every half-step (recurrent_rr.hip X_LATE)
    rr_bf16x8 xf[4][KX][3];
    auto issue_h = [&](auto s_tag, int t, int j0, int j1) __attribute__((always_inline)) {
        constexpr int s = decltype(s_tag)::value;
        const int so = ((ht0 + s) * NKS + w * KH) * 3 * 1024;
        const __amdgpu_buffer_rsrc_t rs = rs_rd(s, t);
#pragma unroll
        for (int blk = 0; blk < 3 * KH; ++blk) {
            if (blk < j0 || blk >= j1) continue;
            const int i = blk / 3, m = blk % 3;
            hf[s & 1][i][m] = __builtin_amdgcn_raw_buffer_load_b128(rs, hvo[i] + (blk & 3) * 1024, so + (blk >> 2) * 4096, 16 /* sc1 */);
        }
    };
    auto issue_x = [&](auto s_tag, int t) __attribute__((always_inline)) {
        constexpr int s = decltype(s_tag)::value;
        const __amdgpu_buffer_rsrc_t rs = rs_x(s, t);
        const int so = ((ht0 + s) * p.NKSx + w * KX) * 3 * 1024;
#pragma unroll
        for (int ix = 0; ix < KX; ++ix)
#pragma unroll
            for (int m = 0; m < 3; ++m)
                xf[s][ix][m] = __builtin_bit_cast(rr_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, xvo[ix] + ((3 * ix + m) & 3) * 1024, so + ((3 * ix + m) >> 2) * 4096, 0));
    };
    // ---- the finish of a stream, in slices ----
    float z[4];
    auto fin_reduce = [&]() __attribute__((always_inline)) {
        RR_BARRIER();                                   // every wave's partial sums are in `red`
        rr_v4u sp[4];
#pragma unroll
        for (int src = 0; src < 4; ++src) sp[src] = red[(w * 4 + src) * 64 + lane];
        auto sum4 = [&](unsigned a, unsigned b, unsigned c, unsigned d) {     // fixed order: ((w0 + w1) + w2) + w3
            return ((__uint_as_float(a) + __uint_as_float(b)) + __uint_as_float(c)) + __uint_as_float(d);
        };
        z[0] = sum4(sp[0].x, sp[1].x, sp[2].x, sp[3].x); z[1] = sum4(sp[0].y, sp[1].y, sp[2].y, sp[3].y);
        z[2] = sum4(sp[0].z, sp[1].z, sp[2].z, sp[3].z); z[3] = sum4(sp[0].w, sp[1].w, sp[2].w, sp[3].w);
    };
    auto fin_gates = [&](auto s_tag, int t) __attribute__((always_inline)) {
        constexpr int s = decltype(s_tag)::value;
        float zc[4], hn, cn;
#pragma unroll
        for (int g = 0; g < 4; ++g) zc[g] = z[g] + bsum[g];
        if (CELL == 1) {      // gru.c:144-186, the expressions of gru_rr_kernel: slots z | r | h.U_h + b_h | x.W_h + b_i
            const float zg = nntk_fast_sigmoid(zc[0]);
            const float rg = nntk_fast_sigmoid(zc[1]);
            const float ht = nntk_fast_tanh(fmaf(rg, zc[2], zc[3]));
            hn = fmaf(-zg + 1.0f, ht, zg * cst[s]);
            cn = hn;
        } else {              // lstm.c:201-238: blocks i | f | g | o
            const float ig = nntk_fast_sigmoid(zc[0]);
            const float fg = nntk_fast_sigmoid(zc[1]);
            const float gg = nntk_fast_tanh(zc[2]);
            const float og = nntk_fast_sigmoid(zc[3]);
            cn = fmaf(fg, cst[s], ig * gg);
            hn = og * nntk_fast_tanh(cn);
        }
        cst[s] = t >= 0 ? cn : cst[s];                  // (the finish before the first step works on nothing: keep the initial state)
        hx[n * RR4_HX_LD + jl] = hn;
    };
    unsigned *const flags = p.flags + (size_t)bt * 4 * RR_FLAGS;
    auto fin_publish = [&](auto s_tag, int t) __attribute__((always_inline)) {
        constexpr int s = decltype(s_tag)::value;
        RR_BARRIER();                                   // the stream's h row pieces are in `hx`
        if (w == s) {
            // the publishing wave: lanes 0-31 hold the 8 hidden units of row n = half of the consumers' 16-wide k step 16 (ct >> 1);
            // split, three write-through stores of 512 contiguous bytes.  Offsets ride in the VECTOR offset with an immediate
            // soffset (wide-store hazard: tools/check_store_hazard.py).
            const float4 h_lo = *reinterpret_cast<const float4 *>(hx + n * RR4_HX_LD);
            const float4 h_hi = *reinterpret_cast<const float4 *>(hx + n * RR4_HX_LD + 4);
            const float v[8] = {h_lo.x, h_lo.y, h_lo.z, h_lo.w, h_hi.x, h_hi.y, h_hi.z, h_hi.w};
            rr_v4u a, b, c;
            rr_split8(v, a, b, c);
            const int vo = lane < 32 ? ((ct & 1) * 32 + n) * 16 + (((ht0 + s) * NKS + (ct >> 1)) * 3) * 1024 : RR_OOB_F;
            const __amdgpu_buffer_rsrc_t rs = rs_wr(s, t);
            __builtin_amdgcn_raw_buffer_store_b128(a, rs, vo, 0, 16 /* sc1 */);
            __builtin_amdgcn_raw_buffer_store_b128(b, rs, vo + 1024, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(c, rs, vo + 2048, 0, 16);
        } else if (w == ((s + 2) & 3)) {                // the output wave: the same row pieces in f32, 16 bytes per lane
            const rr_v4u o = *reinterpret_cast<const rr_v4u *>(hx + n * RR4_HX_LD + 4 * kh);
            if (p.return_sequences) {
                const __amdgpu_buffer_rsrc_t rs = rs_out(s, t >= 0 && t < T);
                __builtin_amdgcn_raw_buffer_store_b128(o, rs, out_vo + t * H * 4, 0, 0);
            }
            if (t == T - 1) {                           // the stream's last step: last output / final state leave from here
                const int row = (ht0 + s) * 32 + n;
                if (row < p.B) {
                    if (!p.return_sequences && p.out) *reinterpret_cast<rr_v4u *>(p.out + (size_t)row * H + 8 * ct + 4 * kh) = o;
                    if (p.hT) *reinterpret_cast<rr_v4u *>(p.hT + (size_t)row * H + 8 * ct + 4 * kh) = o;
                }
            }
        }
    };
    // arrival (publishing wave only): its publishing stores have drained -> raise the column tile's flag for this stream
    auto arrive = [&](auto s_tag, int t) __attribute__((always_inline)) {
        constexpr int s = decltype(s_tag)::value;
        if (w == s) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N_AFTER_PUB) : "memory");
            if (lane == 0)
                __hip_atomic_store(flags + s * RR_FLAGS + ct, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    unsigned pv0 = 0;
    const bool flag_live = lane < p.NCT;
    auto poll_a = [&](int s) __attribute__((always_inline)) {
        const unsigned *f = flags + s * RR_FLAGS + lane;
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(pv0) : "v"(f) : "memory");
    };
    auto poll_b = [&](int s, int t) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv0) :: "memory");
        const unsigned target = (unsigned)t;
        bool ok = !flag_live || pv0 >= target;
        if (__builtin_amdgcn_ballot_w64(!ok) != 0 || p.spin_ticks == 0) {  // something has not arrived yet: spin (bounded)
            const unsigned *f = flags + s * RR_FLAGS + lane;
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            bool expired = p.spin_ticks == 0;
            while (!expired) {
                const unsigned a = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = !flag_live || a >= target;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                if (__hip_atomic_load(p.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { expired = true; break; }
                __builtin_amdgcn_s_sleep(1);
                expired = __builtin_amdgcn_s_memrealtime() - t_start > p.spin_ticks;
            }
            if (expired && lane == 0) __hip_atomic_fetch_or(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // one half-step: multiply stream Y at step t; finish stream X = Y - 1 (its step tX) on the way; poll / fetch h
    // for stream P = Y + 1 (its step tP), then request Y's own x_{t+1}.  Identical for every (Y, t): the sequence's ends are masked by data (see the header).
    auto half_step = [&](auto y_tag, int t) __attribute__((always_inline)) {
        constexpr int Y = decltype(y_tag)::value;
        using XT = std::integral_constant<int, (Y + 3) & 3>;
        using PT = std::integral_constant<int, (Y + 1) & 3>;
        const int tX = Y == 0 ? t - 1 : t, tP = Y == 3 ? t + 1 : t;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // six products per k step, smallest terms first: (A image, B image) = (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            RR_STAMP(Y, t, s);
            rr_bf16x8 wa[3], ulo;
            if (s < KX) {
#pragma unroll
                for (int m = 0; m < 3; ++m) wa[m] = __builtin_bit_cast(rr_bf16x8, WXs[((w * KX + s) * 3 + m) * 64 + lane]);
            } else {
                ulo = __builtin_bit_cast(rr_bf16x8, ULs[(w * KH + (s - KX)) * 64 + lane]);
            }
            if (s == S_RED) { fin_reduce(); fin_gates(XT{}, tX); }
            if (s == S_PUB) fin_publish(XT{}, tX);
            if (s == S_E1) arrive(XT{}, tX);
            if (s == S_E2 - RR_POLL_LEAD) poll_a(PT::value);
            if (s == S_E2) { poll_b(PT::value, tP); issue_h(PT{}, tP, 0, 3 * NPRE); issue_x(y_tag, t + 1); }
            if (s + NPRE < KH) issue_h(y_tag, t, 3 * (s + NPRE), 3 * (s + NPRE + 1));
            if (s < KX) {
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[PA[pr]], xf[Y][s][PB[pr]], acc, 0, 0, 0);
            } else {
                const int i = s - KX;
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    const rr_bf16x8 av = PA[pr] == 2 ? ulo : uh[i][PA[pr]];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(rr_bf16x8, hf[Y & 1][i][PB[pr]]), acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {                 // at most one vector-memory operation between two MFMAs
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);            // a slice stays with its k step
        }
        RR_STAMP(Y, t, NST);
        // split-K exchange: wave `dst` finishes registers 4 g + dst (read in the NEXT half-step)
#pragma unroll
        for (int dst = 0; dst < 4; ++dst)
            red[(dst * 4 + w) * 64 + lane] = (rr_v4u){__float_as_uint(acc[dst]), __float_as_uint(acc[4 + dst]),
                                                       __float_as_uint(acc[8 + dst]), __float_as_uint(acc[12 + dst])};
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    issue_x(I0{}, 0); issue_x(I1{}, 0); issue_x(I2{}, 0); issue_x(I3{}, 0);
    issue_h(I0{}, 0, 0, 3 * NPRE);
    for (int t = 0; t < T; ++t) {
        half_step(I0{}, t);
        half_step(I1{}, t);
        half_step(I2{}, t);
        half_step(I3{}, t);
    }
    half_step(I0{}, T);                                   // finishes stream 3's last step (its own multiply works on nothing)
    // ---- final cell state (LSTM) ----
    if (p.cT) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = (ht0 + s) * 32 + n;
            if (row < p.B) p.cT[(size_t)row * H + jf] = cst[s];
        }
    }
}
template <int KH, int KX>
__global__ __launch_bounds__(256) void lstm_rr4_kernel(RRParams p) { rr4_body<KH, KX, 0>(p); }
template <int KH, int KX>
__global__ __launch_bounds__(256) void gru_rr4_kernel(RRParams p) { rr4_body<KH, KX, 1>(p); }

// ---- host side --------------------------------------------------------------------------------------------------
static bool rr4_shape(int H, int in, int *KH, int *KX) {
    if (H < 64 || H > 512 || (H % 16) != 0 || in < 1) return false;
    *KH = H <= 256 ? 4 : 8;
    *KX = in <= 64 ? 1 : in <= 128 ? 2 : (in <= 256 && *KH == 4) ? 4 : 0;
    return *KX != 0;
}
static size_t rr4_lds_bytes(int KH, int KX) { return (size_t)(4 * KH + 4 * KX * 3 + 16) * 1024 + 32 * RR4_HX_LD * 4; }

extern "C" size_t nntk_shim_rr4_image_floats(int H, int in) {
    int KH, KX;
    if (!rr4_shape(H, in, &KH, &KX)) return 0;
    return (size_t)(H / 8) * rr4_blocks_per_ct(KH, KX) * 256;
}
static int rr4_pack(bool raw, const float *d_u, const float *d_w, float *d_img, int H, int in) {
    int KH, KX;
    if (!rr4_shape(H, in, &KH, &KX)) return nntk_fail_msg("rr4_pack: shape not taken by the four-stream register-resident kernel");
    const int Hj_p = (H + 15) & ~15, Hk_p = (H + 31) & ~31;
    int Kin_p, N_p;
    nntk_shim_conv_pack_sizes(in, 4 * H, 1, &Kin_p, &N_p);
    const int NCT = H / 8;
    const long total = (long)NCT * rr4_blocks_per_ct(KH, KX) * 64;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (raw) hipLaunchKernelGGL(rr4_pack_kernel<true>, dim3((unsigned)g), dim3(256), 0, nntk_stream(), d_u, d_w, (rr_v4u *)d_img, H, in, 0, 0, 0, KH, KX, NCT);
    else     hipLaunchKernelGGL(rr4_pack_kernel<false>, dim3((unsigned)g), dim3(256), 0, nntk_stream(), d_u, d_w, (rr_v4u *)d_img, H, in, Hj_p, Hk_p, Kin_p, KH, KX, NCT);
    NNTK_LAUNCH_CHECK("rr4_pack_kernel");
    return 0;
}
// d_ut / d_wp: the per-gate U^T and the packed W^T (host: core_upload)
extern "C" int nntk_shim_rr4_pack(const float *d_ut, const float *d_wp, float *d_img, int H, int in) { return rr4_pack(false, d_ut, d_wp, d_img, H, in); }
// the caller-layout matrices U [H][4H], W [in][4H] (the GRU's four-slot matrices)
extern "C" int nntk_shim_rr4_pack_raw(const float *d_U, const float *d_W, float *d_img, int H, int in) { return rr4_pack(true, d_U, d_W, d_img, H, in); }

// 0 = launched; 1 = not taken; -1 = error.  q: the parameters rr_launch (recurrent_rr.hip) has filled (frag3 x, hand-off, flags
// for 4 ceil(B / 128) half-tiles zeroed, h_0 slot); d_img4: the images of rr4_pack.
int nntk_rr4_launch(RRParams q, const float *d_img4, int cell, size_t *lds_out) {
    const NntkOptions &opt = nntk_options();
    int KH, KX;
    if (!d_img4 || !q.xf3 || !rr4_shape(q.H, q.in, &KH, &KX)) return 1;
    void (*kern)(RRParams) = nullptr;
    if (cell == 1) {
        if (KH == 4 && KX == 2) kern = gru_rr4_kernel<4, 2>;
        else if (KH == 4 && KX == 4) kern = gru_rr4_kernel<4, 4>;
        else if (KH == 4 && KX == 1) kern = gru_rr4_kernel<4, 1>;
        else if (KH == 8 && KX == 2) kern = gru_rr4_kernel<8, 2>;
    } else {
        if (KH == 4 && KX == 2) kern = lstm_rr4_kernel<4, 2>;
        else if (KH == 4 && KX == 4) kern = lstm_rr4_kernel<4, 4>;
        else if (KH == 4 && KX == 1) kern = lstm_rr4_kernel<4, 1>;
        else if (KH == 8 && KX == 2) kern = lstm_rr4_kernel<8, 2>;
    }
    if (!kern) return 1;
    const size_t lds = rr4_lds_bytes(KH, KX);
    if (nntk_set_max_dynamic_lds((const void *)kern, lds)) return -1;
    const int NCT = q.H / 8;
    if (NCT > RR_FLAGS) return 1;
    const int resident = nntk_resident_blocks((const void *)kern, 256, lds, 1);
    const int tiles_per_launch = resident / NCT;
    if (tiles_per_launch < 1) return 1;
    (void)opt;
    q.img = (const rr_v4u *)d_img4;
    q.NCT = NCT;
    q.NHT = (q.B + 63) / 64 * 2;
    unsigned *flags0 = q.flags;
    const int nbt_total = (q.B + 127) / 128;
    for (int bt0 = 0; bt0 < nbt_total; bt0 += tiles_per_launch) {
        const int nbt = nbt_total - bt0 < tiles_per_launch ? nbt_total - bt0 : tiles_per_launch;
        q.NBT = nbt; q.b_base = bt0 * 128;
        q.flags = flags0 + (size_t)bt0 * 4 * RR_FLAGS;
        hipLaunchKernelGGL(kern, dim3((unsigned)(nbt * NCT)), dim3(256), lds, nntk_stream(), q);
    }
    if (lds_out) *lds_out = (size_t)((nbt_total + tiles_per_launch - 1) / tiles_per_launch);
    static const char *const names[2][4] = {{"lstm_rr4_kernel<4,1>", "lstm_rr4_kernel<4,2>", "lstm_rr4_kernel<4,4>", "lstm_rr4_kernel<8,2>"},
                                            {"gru_rr4_kernel<4,1>", "gru_rr4_kernel<4,2>", "gru_rr4_kernel<4,4>", "gru_rr4_kernel<8,2>"}};
    nntk_set_last_rec_kernel(names[cell == 1][KH == 8 ? 3 : KX == 1 ? 0 : KX == 2 ? 1 : 2]);
    return 0;
}
