// conv1d_flatk.hip -- the split-bf16 x 3 implicit-GEMM convolution with a FLAT K axis, for channel counts that are multiples of 8
// but not of 16 (BASELINE configs[2]: Conv1d(40 -> 128, k = 5)).  Reference semantics: layers/conv_1d.c:122-147.
//
// conv1d_mfma_bf16x3_kernel walks K as (16-channel chunk) x (tap) and pads every tap's channels to a multiple of 16: 40 channels
// become 48, K = 5 x 48 = 240 for 200 real products -- 17 % of the MFMAs multiply padding.  Here K is the flat index
// f = tap * Cin + channel, padded once at the end (200 -> 208: 13 k steps of 16 instead of 15).  A k step's 16 values may then
// straddle two taps, which the MFMA's operand layout absorbs for free when Cin % 8 == 0: the B fragment of lane (position, k half)
// is 8 consecutive K values, and each k half lies inside one tap -- the two halves simply read different window rows.
// The whole window (all channels, three bf16 images) is staged in LDS ONCE per tile -- 132 rows x 240 bytes = 31 KB for configs[2] --
// instead of once per channel chunk; only the weight fragments stream through a double buffer (one barrier per k step, as before).
// Same contraction (six products per f32 product, smallest first), but K runs tap-major here and chunk-major there: results differ
// from conv1d_mfma_bf16x3_kernel in the last bits (same tolerance), so the choice is made from the layer's SHAPE only.
#include "conv1d_kernels.hpp"

// WDIRECT: the weight fragments go global (L2) -> registers, every wave fetching the TN column tiles it multiplies, one k step ahead:
// no weight LDS, NO barrier in the K loop (the window is read-only after its staging), 32 KB of LDS per workgroup instead of 57.
// TM: 32-position tiles per wave (2: 128 output positions per workgroup; 4: 256 -- the weight stream per output halves)
template <int TN, bool WDIRECT, int TM = 2>
__global__ __launch_bounds__(256, 2) void conv1d_flatk_bf16x3_kernel(ConvParams p, int KS /* k steps of 16 */) {
    constexpr int WN = 2;
    constexpr int BM = 2 * TM * 32;
    constexpr int BN = WN * TN * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char *lds = reinterpret_cast<char *>(smem);
    const int Cin = p.Cin;
    const int RB = 6 * Cin;                                   // window row: 3 images x Cin bf16
    const int a_bytes = (p.rows_a * RB + 15) & ~15;
    constexpr int w_bytes = WDIRECT ? 0 : (BN / 32) * 3 * 1024;
    // LDS: window | 16 zero bytes (the K padding's operand) | weights [2][BN / 32][3][1 KB] | epilogue constants [6][BN]
    const int z_off = a_bytes, w_base = a_bytes + 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;          // XCD-aware tile order (conv1d_kernels.hpp)
    const int tile = (local / p.n_tiles) * 8 + xcd;
    if (tile >= p.m_tiles) return;
    const int b = tile / p.tiles_per_seq;
    const int x0 = (tile % p.tiles_per_seq) * BM;
    const int n0 = (local % p.n_tiles) * BN;

    // ---- weights: wave w stages column tile w (+ 4 c) of the workgroup's BN / 32: one 1 KB block per image and k step ----
    const size_t w_elems_total = (size_t)p.Cout_p * KS * 16;
    const int img_bytes = (int)(w_elems_total * 2);
    const __amdgpu_buffer_rsrc_t rs_w = conv_rsrc(p.wp + w_elems_total, (size_t)3 * img_bytes);
    const bool w_thread = wave < BN / 32;
    const int w_voff = w_thread ? (((n0 >> 5) + wave) * KS) * 1024 + lane * 16 : CONV_OOB;
    v4u32_t wreg[WDIRECT ? TN * 3 : 3];
    auto load_w = [&](int ks) {
        if constexpr (WDIRECT) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    wreg[j * 3 + m] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (((n0 >> 5) + wn * TN + j) * KS) * 1024 + lane * 16, ks * 1024 + m * img_bytes, 0);
        } else {
#pragma unroll
            for (int m = 0; m < 3; ++m) wreg[m] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_voff, ks * 1024 + m * img_bytes, 0);
        }
    };
    load_w(0);

    // ---- the window: every row, every channel, split into its three images, once ----
    {
        const size_t in_total = (size_t)p.B * p.T * p.Cin;
        const size_t in_off = (size_t)b * p.in_seq + (size_t)x0 * p.in_row;
        const __amdgpu_buffer_rsrc_t rs_in = conv_rsrc(p.in + in_off, (in_total - in_off) * 4);
        const int c4n = Cin >> 2;                             // 16-byte pieces per row
        const int pieces = p.rows_a * c4n;
        for (int e = tid; e < pieces; e += 256) {
            const int r = e / c4n, c4 = e - r * c4n;
            const v4u32_t x = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(((long)r * p.in_row + c4 * 4) * 4), 0, 0);
            unsigned h0, m0, l0, h1, m1, l1;
            split3_pair(__uint_as_float(x.x), __uint_as_float(x.y), h0, m0, l0);
            split3_pair(__uint_as_float(x.z), __uint_as_float(x.w), h1, m1, l1);
            char *dst = lds + r * RB + c4 * 8;
            *reinterpret_cast<uint2 *>(dst) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(dst + 2 * Cin) = make_uint2(m0, m1);
            *reinterpret_cast<uint2 *>(dst + 4 * Cin) = make_uint2(l0, l1);
        }
        if (tid < 4) reinterpret_cast<unsigned *>(lds + z_off)[tid] = 0u;
    }
    float *cst = reinterpret_cast<float *>(lds + w_base + 2 * w_bytes);
    {
        float c[6];
        conv_load_constants<BN>(p, n0, tid, c);
        conv_stage_constants<BN>(cst, tid, c);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int a_row0 = wm * TM * 32 + l31;                    // window row of this lane's position in tile i at tap kk: a_row0 + 32 i + kk
    const int w_rd = w_base + (wn * TN * 3) * 1024 + lane * 16;
    const int w_wr = w_base + (wave * 3) * 1024 + lane * 16;
    const int Kreal = p.k * Cin;
    // (tap, channel) of the k step's two halves, tracked on the scalar unit: f = 16 ks + 8 kh
    int kk0 = 0, c0 = 0;
    if (WDIRECT) __syncthreads();                             // the window is published; nothing else is shared
    for (int ks = 0; ks < KS; ++ks) {
        const int wbuf = ks & 1;
        bf16x8_t a[3][TM], w[3][TN];
        if constexpr (WDIRECT) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m) w[m][j] = __builtin_bit_cast(bf16x8_t, wreg[j * 3 + m]);
            load_w(ks + 1 < KS ? ks + 1 : ks);                // (the last one re-requests a block nobody reads: no branch)
        } else {
            if (w_thread) {
                char *dst = lds + wbuf * w_bytes + w_wr;
#pragma unroll
                for (int m = 0; m < 3; ++m) *reinterpret_cast<v4u32_t *>(dst + 1024 * m) = wreg[m];
            }
            __syncthreads();                                  // (the first one also publishes the window)
            if (ks + 1 < KS) load_w(ks + 1);
        }
        int kk1 = kk0, c1 = c0 + 8;
        if (c1 >= Cin) { c1 -= Cin; ++kk1; }
        const int kkl = kh ? kk1 : kk0, cl = kh ? c1 : c0;
        const bool pad = 16 * ks + 8 * kh >= Kreal;           // the K padding: weights are zeros, the operand must be finite zeros too
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int off = pad ? z_off : (a_row0 + 32 * i + kkl) * RB + cl * 2;
#pragma unroll
            for (int m = 0; m < 3; ++m) a[m][i] = *reinterpret_cast<const bf16x8_t *>(lds + off + (pad ? 0 : m * 2 * Cin));
        }
        if constexpr (!WDIRECT) {
            const char *Wb = lds + wbuf * w_bytes + w_rd;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m) w[m][j] = *reinterpret_cast<const bf16x8_t *>(Wb + (j * 3 + m) * 1024);
        }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};      // window image (0 hi, 1 mid, 2 lo): smallest terms first
        constexpr int PW[6] = {0, 2, 1, 0, 1, 0};      // weight image
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PW[t]][j], a[PA[t]][i], acc[i][j], 0, 0, 0);
        c0 += 16;
        while (c0 >= Cin) { c0 -= Cin; ++kk0; }
    }
    // whole-line stores through the window buffer (dead once every wavefront has left the K loop: barrier); the constants sit behind it
    v4u32_t *epi = nullptr;
#ifndef CONV_EPI_DIRECT
    if (w_base + 2 * w_bytes >= CONV_EPI_LDS_BYTES) {
        __syncthreads();
        epi = reinterpret_cast<v4u32_t *>(lds);
    }
#endif
    conv_epilogue<TM, TN, WN>(p, acc, b, x0, n0, wm, wn, l31, kh, cst, epi);
}

// 0 = launched; 1 = shape / configuration not taken (the caller runs nntk_shim_conv1d); -1 = error.
// d_wpf: [Cout_p][Kf_p] f32 with K = tap * Cin + channel, Kf_p = k * Cin rounded up to 16, followed by its three split images
// (nntk_upload_packed_weights(rows = Cout_p, ktot = Kf_p)).
extern "C" int nntk_shim_conv1d_flatk(const float *d_in, const float *d_wpf, const float *d_bias, const float *d_bn,
                                      float bn_eps, int act_kind, float relu_a, float *d_out,
                                      int B, int T, int Cin, int Cout, int k, int Tout) {
    if (B <= 0 || Tout <= 0) return 0;
    const NntkOptions &opt = nntk_options();
    if (opt.conv_flatk == 0 || opt.gemm_split_bf16 == 0) return 1;      // (conv_store: this kernel has the 16-byte quad epilogue only)
    if (act_kind == NNTK_ACT_SOFTMAX || act_kind == NNTK_ACT_CUSTOM) return 1;
    if (act_kind == NNTK_ACT_NONE) act_kind = NNTK_ACT_IDENTITY;
    if ((Cin % 8) != 0 || (Cin % 16) == 0 || Cin > 120 || k < 2 || (((size_t)d_in) & 15) != 0) return 1;
    ConvParams p;
    p.in = d_in; p.wp = d_wpf; p.bias = d_bias; p.bn = d_bn; p.out = d_out;
    p.bn_eps = bn_eps; p.relu_a = relu_a; p.act_kind = act_kind;
    p.B = B; p.T = T; p.Cin = Cin; p.Cout = Cout; p.k = k; p.stride = 1; p.Tout = Tout;
    p.in_seq = (long)T * Cin; p.in_row = Cin;
    int Cin_p;
    nntk_shim_conv_pack_sizes(Cin, Cout, k, &Cin_p, &p.Cout_p);
    p.Cin_p = Cin_p;
    // (256-position tiles -- TM = 4: half the weight stream per output, 229 VGPRs, two workgroups per CU -- measured SLOWER at configs[2]:
    // 0.438 vs 0.388 ms, same box; the kernel lives on waves in flight, not on L2 bytes.  conv_flatk = 4 selects them for A/B.)
    const int KS = (k * Cin + 15) / 16;
#ifdef NNTK_VARIANT_FLATK      // A/B variant build only (tools/build_variant.py ... conv1d_flatk.hip -DNNTK_VARIANT_FLATK)
    const bool big = opt.conv_flatk == 4 && p.Cout_p % 128 == 0 && (size_t)(255 + k) * 6 * Cin + 16 + 6 * 128 * sizeof(float) <= 72 * 1024;
#else
    const bool big = false;
#endif
    const int BM = big ? 256 : CONV_BM;
    p.tiles_per_seq = (Tout + BM - 1) / BM;
    p.out_mode = 0;
    p.rows_a = BM - 1 + k;
    p.bn_fast = opt.bn_fast == 1 ? 1 : 0;
    p.store16 = Cout % 4 == 0 && ((size_t)d_out & 15) == 0;
    p.quad = 1;
#ifdef NNTK_CONV_DBG
    p.dbg = 0;
#endif
    if (p.Cout_p % 64 != 0 || (long)p.Cout_p * KS * 16 * 6 >= (long)CONV_OOB || (long)(p.rows_a + 1) * Cin * 4 >= (long)CONV_OOB) return 1;
    const bool bn128 = p.Cout_p % 128 == 0;
    const int BN = bn128 ? 128 : 64;
#ifdef NNTK_VARIANT_FLATK
    const bool wdirect = opt.conv_flatk != 2;          // (2: weights through an LDS double buffer with a barrier per k step: 0.416 vs 0.390 ms)
#else
    const bool wdirect = true;
#endif
    const size_t lds = (size_t)((p.rows_a * 6 * Cin + 15) & ~15) + 16 + (wdirect ? 0 : 2 * (size_t)(BN / 32) * 3 * 1024) + 6 * BN * sizeof(float);
    if (lds > 100 * 1024) return 1;
    p.m_tiles = B * p.tiles_per_seq;
    p.n_tiles = p.Cout_p / BN;
    const long blocks = (long)((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
    if ((long)B * p.tiles_per_seq > 0x7fffffffL / 8 || blocks > 0x7fffffffL) return 1;
#ifdef NNTK_VARIANT_FLATK
    void (*kern)(ConvParams, int) = big ? conv1d_flatk_bf16x3_kernel<2, true, 4>
                                  : bn128 ? (wdirect ? conv1d_flatk_bf16x3_kernel<2, true> : conv1d_flatk_bf16x3_kernel<2, false>)
                                          : (wdirect ? conv1d_flatk_bf16x3_kernel<1, true> : conv1d_flatk_bf16x3_kernel<1, false>);
#else
    void (*kern)(ConvParams, int) = bn128 ? conv1d_flatk_bf16x3_kernel<2, true> : conv1d_flatk_bf16x3_kernel<1, true>;
#endif
    if (lds > 64 * 1024 && nntk_set_max_dynamic_lds((const void *)kern, lds)) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, nntk_stream(), p, KS);
    NNTK_LAUNCH_CHECK("conv1d_flatk_bf16x3_kernel");
    nntk_set_last_conv_kernel("conv1d_flatk_bf16x3_kernel");
    return 0;
}
