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
#include "conv1d_kernels.hpp"

extern "C" void nntk_shim_conv_pack_sizes(int Cin, int Cout, int k, int *Cin_p, int *Cout_p) {
    (void)k;
    *Cin_p = (Cin + CONV_KC - 1) & ~(CONV_KC - 1);   // whole K chunks: no guard in the weight staging
    *Cout_p = (Cout + 31) & ~31;                     // whole 32-wide MFMA column tiles
}


// src [rows][ktot] f32 (rows % 32 == 0, ktot % 16 == 0) -> dst [3 images][rows / 32][ktot / 16][2][32][8] bf16
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float *__restrict__ src, unsigned *__restrict__ dst,
                                                           int rows, int ktot) {
    const size_t n_pairs = (size_t)rows * ktot / 2;
    const int ksteps = ktot >> 4;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n_pairs; e += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(e / (ktot / 2)), kt = (int)(e % (ktot / 2)) * 2;
        unsigned h, m, l;
        split3_pair(src[2 * e], src[2 * e + 1], h, m, l);
        const size_t d = ((((size_t)(row >> 5) * ksteps + (kt >> 4)) * 2 + ((kt >> 3) & 1)) * 32 + (row & 31)) * 4 + ((kt & 7) >> 1);
        dst[d] = h; dst[n_pairs + d] = m; dst[2 * n_pairs + d] = l;
    }
}
extern "C" int nntk_shim_split_bf16x3(const float *d_src, void *d_dst, int rows, int ktot) {
    if (rows <= 0 || ktot <= 0) return 0;
    if ((rows & 31) || (ktot & 15)) return nntk_fail_msg("split_bf16x3: rows must be a multiple of 32 and ktot of 16");
    size_t g = ((size_t)rows * ktot / 2 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)g), dim3(256), 0, nntk_stream(), d_src, (unsigned *)d_dst, rows, ktot);
    NNTK_LAUNCH_CHECK("split_bf16x3_kernel");
    return 0;
}

// ---- weight sets the split contraction must not take ------------------------------------------------------------
// bf16 has f32's exponent range but the bf16 MFMA flushes denormals, bf16(|w| > 3.39e38) is infinite and inf - inf is
// NaN: a weight matrix holding a non-finite, huge or denormal value cannot be split exactly.  The host scans a weight
// block when it packs it (runtime.c nntk_upload_packed_weights, free: it touches every value anyway) and registers the
// device pointer here; "auto" then takes the exact-f32 kernel for that block, whose chain treats such values exactly as
// the reference's (core/default_ops.cc:224-231).  The list is normally empty: one relaxed atomic load per launch.
#include <atomic>
#include <mutex>
#include <vector>
static std::mutex g_exact_mu;
static std::vector<const void *> g_exact_only;
static std::atomic<int> g_exact_n{0};
extern "C" void nntk_shim_weights_exact_only(const void *d_wp, int on) {
    if (!d_wp) return;
    if (!on && g_exact_n.load(std::memory_order_relaxed) == 0) return;
    std::lock_guard<std::mutex> lk(g_exact_mu);
    for (size_t i = 0; i < g_exact_only.size(); ++i)
        if (g_exact_only[i] == d_wp) {
            if (on) return;
            g_exact_only[i] = g_exact_only.back();
            g_exact_only.pop_back();
            g_exact_n.store((int)g_exact_only.size(), std::memory_order_relaxed);
            return;
        }
    if (on) { g_exact_only.push_back(d_wp); g_exact_n.store((int)g_exact_only.size(), std::memory_order_relaxed); }
}
bool nntk_weights_exact_only(const void *d_wp) {
    if (g_exact_n.load(std::memory_order_relaxed) == 0) return false;
    std::lock_guard<std::mutex> lk(g_exact_mu);
    for (const void *q : g_exact_only) if (q == d_wp) return true;
    return false;
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

// sd = sqrtf(var + eps) and 1/sd behind the four BatchNorm vectors: block = gamma|beta|mean|var|sd|rsd (6 * C floats)
__global__ void bn_derive_kernel(float *blk, float eps, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sd = sqrtf(blk[3 * C + c] + eps);
        blk[4 * C + c] = sd;
        blk[5 * C + c] = 1.0f / sd;
    }
}
extern "C" int nntk_shim_bn_derive(float *d_block, float eps, int C) {
    if (C <= 0) return 0;
    hipLaunchKernelGGL(bn_derive_kernel, dim3((C + 255) / 256), dim3(256), 0, nntk_stream(), d_block, eps, C);
    NNTK_LAUNCH_CHECK("bn_derive_kernel");
    return 0;
}

template <int WM, int WN, int TM, int TN, bool A4, bool SPLIT = false>
static int launch_mfma(const ConvParams &p) {
    return p.quad ? launch_mfma_o<WM, WN, TM, TN, A4, SPLIT, true>(p) : launch_mfma_o<WM, WN, TM, TN, A4, SPLIT, false>(p);
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
    p.store16 = Cout % 4 == 0 && ((size_t)d_out & 15) == 0;
    p.quad = opt.conv_store >= 0 ? opt.conv_store != 0 : out_mode == 0;      // the time-major projection keeps the row form
#ifdef NNTK_CONV_DBG
    p.dbg = opt.conv_dbg;
#endif

    if ((long)p.Cout_p * k * p.Cin_p * 4 >= (long)CONV_OOB)
        return nntk_fail_msg("conv1d: packed weights must stay below 2 GiB (32-bit buffer offsets)");
    if ((long)(320 + 1) * Cin * 4 >= (long)CONV_OOB)
        return nntk_fail_msg("conv1d: a window of 320 input rows must stay below 2 GiB (32-bit buffer offsets)");
    const bool wide_window = p.rows_a > 192 && p.rows_a <= 320 && out_mode == 0;      // stride 2: its own instantiations
    const bool window_fits = p.rows_a <= 192 || wide_window;           // register staging budget (A_PT)
    const long Kdim = (long)Cin * k;
    if (!window_fits || Kdim < 16 || Cout < 32) {
        const long total = (long)B * Tout * Cout;
        long g = (total + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(conv1d_valu_kernel, dim3((unsigned)g), dim3(256), 0, nntk_stream(), p);
        NNTK_LAUNCH_CHECK("conv1d_valu_kernel");
        nntk_set_last_conv_kernel("conv1d_valu_kernel");
        return 0;
    }
    // 16-byte window loads need only 4-byte alignment (MUBUF dwordx4), so odd channel counts take them too; the pieces
    // that run past a row's end are masked at staging time (conv_a4 = 0: 4-byte loads for those shapes, as before)
    const bool a4 = ((Cin % 4 == 0) && ((size_t)d_in % 16 == 0)) || opt.conv_a4 != 0;
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
    // auto: every contraction except the recurrent input projection (out_mode 1), whose exact k-ordered chain the
    // streaming kernel reproduces bit for bit (recurrent.hip rec_stream_step_kernel) -- and a batch-size threshold here would
    // make a row's bits depend on the size of the batch it arrives in (the exact kernels' shards are bit-identical)
    bool split = opt.gemm_split_bf16 == 1 || (opt.gemm_split_bf16 < 0 && out_mode == 0) ||
                       (opt.gemm_split_bf16 == 2 && out_mode == 0 && k > 1) ||        // 2 / 3: convolutions only / dense only (A/B)
                       (opt.gemm_split_bf16 == 3 && out_mode == 0 && k == 1);
    if (split && opt.gemm_split_bf16 < 0 && nntk_weights_exact_only(d_wp)) split = false;      // "auto" only: 1 / 2 / 3 force the split (A/B runs)
    if (wide_window) {
        const bool sp = split && (long)p.Cout_p * k * p.Cin_p * 6 < (long)CONV_OOB;
        return nntk_conv1d_launch_s2(p, a4, sp);
    }
    if (split && (long)p.Cout_p * k * p.Cin_p * 6 < (long)CONV_OOB) {
        // 128 x 256 tile for wide dense GEMMs (TimeDistributedDense): the window is loaded, split and staged once per 256
        // columns instead of once per 128 -- on k = 1 shapes that staging, not the MFMA pipe, is the larger cost
        if (k == 1 && p.Cout_p % 256 == 0 && a4 && p.quad && opt.gemm_wide != 0) return launch_mfma<2, 2, 2, 4, true, true>(p);
        if (p.Cout_p % 128 == 0) return a4 ? launch_mfma<2, 2, 2, 2, true, true>(p) : launch_mfma<2, 2, 2, 2, false, true>(p);
        if (p.Cout_p % 64 == 0)  return a4 ? launch_mfma<4, 1, 1, 2, true, true>(p) : launch_mfma<4, 1, 1, 2, false, true>(p);
        return a4 ? launch_mfma<4, 1, 1, 1, true, true>(p) : launch_mfma<4, 1, 1, 1, false, true>(p);
    }
    if (p.Cout_p % 128 == 0) return a4 ? launch_mfma<2, 2, 2, 2, true>(p) : launch_mfma<2, 2, 2, 2, false>(p);
    if (p.Cout_p % 64 == 0)  return a4 ? launch_mfma<4, 1, 1, 2, true>(p) : launch_mfma<4, 1, 1, 2, false>(p);
    return a4 ? launch_mfma<4, 1, 1, 1, true>(p) : launch_mfma<4, 1, 1, 1, false>(p);
}

// Conv1d (+ BatchNorm + activation) whose output leaves as a frag3 tensor [Tout][2 ceil(B / 64)][ceil(Cout / 16)][3][1 KB] -- the operand
// form of the register-resident recurrent kernels and of dense_frag3_kernel -- instead of f32 [B][Tout][Cout] (conv_epilogue_frag3).
// Returns 1 when this form does not take the call (the caller then runs conv -> f32 scratch -> nntk_shim_frag3_pack: same bits):
// stride != 1, a window of 8 x (15 + k) rows past the staging budget (k > 9), the exact-f32 contraction (option or weights the split
// cannot hold), fewer than 32 output channels.
extern "C" int nntk_shim_conv1d_frag3(const float *d_in, const float *d_wp, const float *d_bias, const float *d_bn,
                                      float bn_eps, int act_kind, float relu_a, float *d_out_f3,
                                      int B, int T, int Cin, int Cout, int k, int stride, int Tout) {
    if (B <= 0 || Tout <= 0) return 0;
    if (act_kind == NNTK_ACT_SOFTMAX || act_kind == NNTK_ACT_CUSTOM)
        return nntk_fail_msg("conv1d epilogue: softmax/custom activations are not fusable");
    if (act_kind == NNTK_ACT_NONE) act_kind = NNTK_ACT_IDENTITY;
    const NntkOptions &opt = nntk_options();
    if (opt.conv_frag3_out == 0) return 1;
    ConvParams p;
    p.in = d_in; p.wp = d_wp; p.bias = d_bias; p.bn = d_bn; p.out = d_out_f3;
    p.bn_eps = bn_eps; p.relu_a = relu_a; p.act_kind = act_kind;
    p.B = B; p.T = T; p.Cin = Cin; p.Cout = Cout; p.k = k; p.stride = stride; p.Tout = Tout;
    p.in_seq = (long)T * Cin; p.in_row = Cin;
    nntk_shim_conv_pack_sizes(Cin, Cout, k, &p.Cin_p, &p.Cout_p);
    p.tiles_per_seq = (Tout + CONV_F3_TT - 1) / CONV_F3_TT;
    p.out_mode = 0;
    p.rows_a = CONV_F3_BB * (CONV_F3_TT + k - 1);
    p.bn_fast = opt.bn_fast == 1 ? 1 : 0;
    p.store16 = 1; p.quad = 1;
    p.f3_nht = (B + 63) / 64 * 2; p.f3_nks = (Cout + 15) / 16;
#ifdef NNTK_CONV_DBG
    p.dbg = opt.conv_dbg;
#endif
    const long Kdim = (long)Cin * k;
    const bool split = (opt.gemm_split_bf16 < 0 && !nntk_weights_exact_only(d_wp)) || opt.gemm_split_bf16 == 1 ||
                       (opt.gemm_split_bf16 == 2 && k > 1) || (opt.gemm_split_bf16 == 3 && k == 1);
    if (!split || stride != 1 || p.rows_a > 192 || Kdim < 16 || Cout < 32) return 1;
    if ((long)p.Cout_p * k * p.Cin_p * 6 >= (long)CONV_OOB) return 1;
    if ((double)Tout * p.f3_nht * p.f3_nks * 3 * 1024 >= (double)CONV_OOB) return 1;            // 32-bit offsets into the frag3 tensor
    if (((long)(CONV_F3_BB - 1) * T + CONV_F3_TT + k) * Cin * 4 >= (long)CONV_OOB) return 1;    // ... and into the tile's eight windows
    const bool a4 = ((Cin % 4 == 0) && ((size_t)d_in % 16 == 0)) || opt.conv_a4 != 0;
    // (same instantiation per Cout as the f32 form: same products in the same order)
    if (p.Cout_p % 128 == 0) return a4 ? launch_mfma_f3<2, 2, 2, 2, true>(p) : launch_mfma_f3<2, 2, 2, 2, false>(p);
    if (p.Cout_p % 64 == 0)  return a4 ? launch_mfma_f3<4, 1, 1, 2, true>(p) : launch_mfma_f3<4, 1, 1, 2, false>(p);
    return a4 ? launch_mfma_f3<4, 1, 1, 1, true>(p) : launch_mfma_f3<4, 1, 1, 1, false>(p);
}
