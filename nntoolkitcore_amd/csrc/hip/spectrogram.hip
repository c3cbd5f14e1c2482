// spectrogram.hip -- K1: framed STFT magnitude / PSD (signal/spectrogram.c:113-135).
// Per frame the reference does: window * x -> zero-pad to nfft -> complex DFT
// (dft.c:34-47 through kissfft) -> optional * fft_norm -> |X| / sum(w) (or PSD)
// on bins 0 .. nfft/2.
//
// Fast path (nfft = 512, the configuration every BASELINE config uses): one
// 64-lane wavefront transforms TWO frames at once as one 512-point complex FFT
// (frame A in the real part, frame B in the imaginary part), 8 points per lane,
// three radix-8 Stockham passes.  Pass 1 reads the windowed samples straight from
// global memory (lane j reads samples j, j+64, ...: fully coalesced; the 2.5x hop
// overlap between neighbouring frames is absorbed by L1/L2), passes 2 and 3
// exchange through a padded LDS image, and the two real spectra are separated in
// registers with 16 cross-lane shuffles (X_a[k] = (Z[k] + conj Z[N-k]) / 2, ...).
// HBM traffic is the algorithmic minimum: each sample read once, 257 floats
// written per frame.
//
// Other nfft (any size, power of two or not): direct O(window * nfreq) DFT per
// frame with a precomputed twiddle table -- correct for every configuration the
// reference accepts, not tuned.
#include "nntk_common.hpp"
#include <stdlib.h>

struct SpecParams {
    const float *in;      // [B, input_size]
    const float *window;  // [window_size]
    const float *tw;      // [nfft] interleaved (cos, -sin) = exp(-2*pi*i*m/nfft)
    float *out;           // [B, nts, nfreq]
    long total_frames;    // B * nts
    int B;
    int input_size, nfft, window_size, step, nfreq, nts;
    float fft_norm, scale, inv_scale;
    int mode;
};

typedef float f2 __attribute__((ext_vector_type(2)));     // (re, im): adds/subs/muls lower to v_pk_*_f32
__device__ __forceinline__ f2 cmul2(f2 a, f2 b) { return (f2){a.x, a.x} * b + (f2){-a.y, a.y} * (f2){b.y, b.x}; }
__device__ __forceinline__ f2 mul_mi2(f2 a) { return (f2){a.y, -a.x}; }     // a * (-i)
#define BF2(a, b) do { f2 _t = a; a = _t + b; b = _t - b; } while (0)

// forward 8-point DFT, natural-order output, in registers
__device__ __forceinline__ void fft8(f2 v[8]) {
    BF2(v[0], v[4]); BF2(v[1], v[5]); BF2(v[2], v[6]); BF2(v[3], v[7]);
    const float s = 0.70710678118654752440f;
    v[5] = cmul2(v[5], (f2){s, -s});
    v[6] = mul_mi2(v[6]);
    v[7] = cmul2(v[7], (f2){-s, -s});
    // two 4-point DFTs: (v0..v3) -> even outputs, (v4..v7) -> odd outputs
    BF2(v[0], v[2]); BF2(v[1], v[3]); v[3] = mul_mi2(v[3]); BF2(v[0], v[1]); BF2(v[2], v[3]);
    BF2(v[4], v[6]); BF2(v[5], v[7]); v[7] = mul_mi2(v[7]); BF2(v[4], v[5]); BF2(v[6], v[7]);
    // registers now hold X0,X4,X2,X6 | X1,X5,X3,X7
    f2 x1 = v[4], x2 = v[2], x3 = v[6], x4 = v[1], x5 = v[5], x6 = v[3];
    v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5; v[6] = x6;
}

// Every wavefront owns its private LDS image and a wave's DS instructions execute in
// order, so no s_barrier is needed between passes -- only a fence that stops the
// compiler from reordering the (aliasing) LDS accesses.  Waves of a workgroup drift freely.
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
                             __builtin_amdgcn_wave_barrier();                        \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define SPEC_LDS_PER_WAVE 592      // 512 + 8 * 8 padding, rounded to a multiple of 16
__device__ __forceinline__ int pidx(int i) { return i + ((i >> 6) << 3); }

__device__ __forceinline__ float finish_bin(const SpecParams &p, float re, float im, int k) {
    if (p.fft_norm != 1.0f) { re *= p.fft_norm; im *= p.fft_norm; }
    float m = re * re + im * im;
    // p.inv_scale = 1/scale rounded once on the host: one multiply instead of the reference's
    // divide (spectrogram.c:33) -- differs from it by at most 1 ulp, far inside the stated tolerance
    if (p.mode == 0) return sqrtf(m) * p.inv_scale;
    if (k == 0 || k == p.nfreq - 1) return m * p.inv_scale;
    return m * (2.0f * p.inv_scale);
}

// The kernel was VALU-bound in its first form (~900 vector instructions per frame pair,
// a third of them 64-bit address arithmetic, range guards and an integer division), so:
// grid.y walks utterances (no division), each utterance gets its own buffer descriptor
// (32-bit offsets, hardware range check instead of per-load guards), complex math is
// written on 2-vectors so it issues as packed f32, and everything that depends only on
// the lane (window taps, both twiddle sets) is hoisted out of the frame loop.
template <int MODE, bool NORM>
__global__ __launch_bounds__(256) void spectrogram512_kernel(SpecParams p) {
    __shared__ __attribute__((aligned(16))) f2 lds_z[4][SPEC_LDS_PER_WAVE];      // interleaved (re, im)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // scalar: frame offsets stay in SGPRs
    f2 *z = lds_z[wave];
    const int ppu = (p.nts + 1) >> 1;                 // frame pairs per utterance

    // lane-only constants
    // window taps, and the per-lane sample offset: taps beyond the window get an out-of-range offset,
    // so the load itself returns the exact zero padding (no select, even next to inf/nan samples)
    float wtap[8];
    int loff[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        const bool inwin = n < p.window_size;
        wtap[r] = inwin ? p.window[n] : 0.0f;
        loff[r] = inwin ? n * 4 : 0x7ffffff0;
    }
    f2 tw2[8], tw3[8];
    {
        const f2 *tw = reinterpret_cast<const f2 *>(p.tw);
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            tw2[r] = tw[(lane & 7) * r * 8];          // exp(-2 pi i (j%8) r / 64)
            tw3[r] = tw[lane * r];                    // exp(-2 pi i j r / 512)
        }
    }
    // output scale per bin (magnitude: 1/sum(w); PSD: 2/(fs sum w^2), edges 1/(..): spectrogram.c:41-47),
    // with fft_normalization_factor folded in AFTER the reference's own rounding points would not be
    // exact, so the factor still multiplies re/im first (NORM) exactly like spectrogram.c:130-131
    float osc[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int k = lane + 64 * r;
        osc[r] = (MODE == 0 || k == 0 || k == p.nfreq - 1) ? p.inv_scale : 2.0f * p.inv_scale;
    }
    const int i1 = pidx(lane * 8);                    // pass-1 write base (8 contiguous floats)
    // pass-2 write base, already padded: pidx(i2w + 8 r) = i2w + 8 r + 8 (lane >> 3) for r < 8
    const int i2w = (lane >> 3) * 64 + (lane & 7) + 8 * (lane >> 3);
    const int src = (64 - lane) & 63;

    for (int b = blockIdx.y; b < p.B; b += gridDim.y) {
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(p.in + (size_t)b * p.input_size), 0, p.input_size * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(p.out + (size_t)b * p.nts * p.nfreq), 0, p.nts * p.nfreq * 4, 0x00020000);
        // software prefetch: the 16 sample loads of this wave's NEXT frame pair are issued
        // before the current pair is transformed, so HBM latency overlaps the FFT math
        int pr = blockIdx.x * 4 + wave;
        unsigned nxa[8], nxb[8];
        bool nhas_b = false;
        // the frame's start is wave-uniform: it rides in the scalar offset, the vector offset is the
        // lane-only loff[] (a missing frame B re-reads frame A; its outputs are never stored)
        if (pr < ppu) {
            const int fa = 2 * pr;
            nhas_b = fa + 1 < p.nts;
            const int soa = fa * p.step * 4;
            const int sob = nhas_b ? soa + p.step * 4 : soa;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                nxa[r] = __builtin_amdgcn_raw_buffer_load_b32(rin, loff[r], soa, 0);
                nxb[r] = __builtin_amdgcn_raw_buffer_load_b32(rin, loff[r], sob, 0);
            }
        }
        for (; pr < ppu; pr += gridDim.x * 4) {
            const int fa = 2 * pr;
            const bool has_b = nhas_b;                // wave-uniform
            f2 v[8];
            // ---- pass 1 (Ns = 1): windowed samples, no twiddles ----
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const f2 x = (f2){__uint_as_float(nxa[r]), __uint_as_float(nxb[r])};
                v[r] = x * wtap[r];                                   // taps beyond the window were loaded as 0
            }
            {
                const int npr = pr + gridDim.x * 4;
                if (npr < ppu) {
                    const int nfa = 2 * npr;
                    nhas_b = nfa + 1 < p.nts;
                    const int soa = nfa * p.step * 4;
                    const int sob = nhas_b ? soa + p.step * 4 : soa;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        nxa[r] = __builtin_amdgcn_raw_buffer_load_b32(rin, loff[r], soa, 0);
                        nxb[r] = __builtin_amdgcn_raw_buffer_load_b32(rin, loff[r], sob, 0);
                    }
                }
            }
            fft8(v);
#pragma unroll
            for (int r = 0; r < 8; r += 2)               // 8 contiguous complex per lane: 4 x ds_write_b128
                *reinterpret_cast<float4 *>(z + i1 + r) = make_float4(v[r].x, v[r].y, v[r + 1].x, v[r + 1].y);
            WAVE_LDS_SYNC();
            // ---- pass 2 (Ns = 8) ----
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = z[pidx(lane + 64 * r)];
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmul2(v[r], tw2[r]);
            fft8(v);
            WAVE_LDS_SYNC();
#pragma unroll
            for (int r = 0; r < 8; ++r) z[i2w + 8 * r] = v[r];
            WAVE_LDS_SYNC();
            // ---- pass 3 (Ns = 64) ----
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = z[pidx(lane + 64 * r)];
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmul2(v[r], tw3[r]);
            fft8(v);
            WAVE_LDS_SYNC();   // LDS image is free for the next pair
            // lane j now holds Z[j + 64 r], r = 0..7.
            // ---- split the two real spectra; bins k = j + 64 r for r = 0..3 (+ k = 256 on lane 0) ----
            // output rows: frame offset in the scalar offset, lane offset in the vector offset; a missing
            // frame B (wave-uniform) is simply not stored
            const int soa = fa * p.nfreq * 4, sob = soa + p.nfreq * 4;
            const int vo = lane * 4, vo4 = lane == 0 ? 0 : 0x7ffffff0;     // only lane 0 owns bin 256
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                f2 c;
                if (r < 4) {
                    // lane != 0: partner register 7 - r on lane 64 - j; lane 0: register (8 - r) & 7 of itself
                    const f2 pz = (f2){__shfl(v[7 - r].x, src), __shfl(v[7 - r].y, src)};
                    c = lane == 0 ? v[(8 - r) & 7] : pz;
                } else {
                    c = v[4];                        // k = 256 pairs with itself (lane 0 only)
                }
                const f2 z = v[r];
                // X_a = (Z + conj(Zc)) / 2 ; X_b = (Z - conj(Zc)) / (2i)
                f2 xa = (f2){0.5f * (z.x + c.x), 0.5f * (z.y - c.y)};
                f2 xb = (f2){0.5f * (z.y + c.y), -0.5f * (z.x - c.x)};
                if (NORM) { xa = xa * p.fft_norm; xb = xb * p.fft_norm; }
                float ma = xa.x * xa.x + xa.y * xa.y, mb = xb.x * xb.x + xb.y * xb.y;
                // hardware square root (<= 1 ulp; libm's correctly rounded sqrtf costs 12 instructions a bin)
                if (MODE == 0) { ma = __builtin_amdgcn_sqrtf(ma); mb = __builtin_amdgcn_sqrtf(mb); }
                const int vor = r == 4 ? vo4 : vo;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ma * osc[r]), rout, vor, soa + 256 * r, 0);
                if (has_b) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mb * osc[r]), rout, vor, sob + 256 * r, 0);
            }
        }
    }
}

// Generic path: one workgroup per frame, direct DFT over the window support.
__global__ __launch_bounds__(256) void spectrogram_dft_kernel(SpecParams p) {
    extern __shared__ __attribute__((aligned(16))) float frame[];   // [window_size]
    for (long f = blockIdx.x; f < p.total_frames; f += gridDim.x) {
        const float *x = p.in + (f / p.nts) * (long)p.input_size + (f % p.nts) * (long)p.step;
        for (int n = threadIdx.x; n < p.window_size; n += blockDim.x) frame[n] = p.window[n] * x[n];
        __syncthreads();
        const float2 *tw = reinterpret_cast<const float2 *>(p.tw);
        for (int k = threadIdx.x; k < p.nfreq; k += blockDim.x) {
            float re = 0.f, im = 0.f;
            int m = 0;                       // (k * n) mod nfft, updated incrementally
            for (int n = 0; n < p.window_size; ++n) {
                const float2 t = tw[m];
                re += frame[n] * t.x;
                im += frame[n] * t.y;
                m += k; if (m >= p.nfft) m -= p.nfft;
            }
            p.out[f * (long)p.nfreq + k] = finish_bin(p, re, im, k);
        }
        __syncthreads();
    }
}

extern "C" int nntk_shim_spectrogram(const float *d_in, const float *d_window, const float *d_twiddle, float *d_out,
                                     int B, int input_size, int nfft, int window_size, int step,
                                     int nfreq, int nts, float fft_norm, int mode, float scale) {
    if (B <= 0 || nts <= 0) return 0;
    if (window_size > nfft) return nntk_fail_msg("spectrogram: window_size must be <= nfft");
    SpecParams p;
    p.in = d_in; p.window = d_window; p.tw = d_twiddle; p.out = d_out;
    p.total_frames = (long)B * nts;
    p.B = B;
    p.input_size = input_size; p.nfft = nfft; p.window_size = window_size; p.step = step;
    p.nfreq = nfreq; p.nts = nts; p.fft_norm = fft_norm; p.scale = scale; p.inv_scale = (float)(1.0 / (double)scale); p.mode = mode;
    if (nfft == 512) {
        if ((long)input_size * 4 >= 0x7ffffff0L || (long)nts * nfreq * 4 >= 0x7ffffff0L)
            return nntk_fail_msg("spectrogram: one utterance must stay below 2 GiB");
        const int ppu = (nts + 1) / 2;
        // frame pairs per wavefront per utterance: long runs amortise the per-utterance prologue (descriptor +
        // un-overlapped first prefetch) as long as the chip stays full.  Measured at the stack's size (256 k pairs):
        // 1 / 2 / 4 / 8 / 16 pairs = 0.245 / 0.225 / 0.220 / 0.207 / 0.199 ms; config 2 (12.5 k pairs): 4 is best.
        long ppw_l = (long)B * ppu / 4096;
        int ppw = ppw_l < 4 ? 4 : ppw_l > 16 ? 16 : (int)ppw_l;
        { const char *e = getenv("NNTK_SPEC_PPW"); if (e && atoi(e) > 0) ppw = atoi(e); }
        unsigned gx = (unsigned)((ppu + 4 * ppw - 1) / (4 * ppw));
        unsigned gy = (unsigned)(B < 65535 ? B : 65535);
        // keep the grid near 8 workgroups per CU; the kernel strides over the rest
        while ((long)gx * gy > 256L * 8 * 8 && gy > 1) gy = (gy + 1) / 2;
        const bool norm = fft_norm != 1.0f;
        auto kern = mode == 0 ? (norm ? spectrogram512_kernel<0, true> : spectrogram512_kernel<0, false>)
                              : (norm ? spectrogram512_kernel<1, true> : spectrogram512_kernel<1, false>);
        hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), 0, nntk_stream(), p);
        NNTK_LAUNCH_CHECK("spectrogram512_kernel");
    } else {
        long g = p.total_frames < 4096 ? p.total_frames : 4096;
        size_t lds = (size_t)window_size * sizeof(float);
        if (lds > 64 * 1024) return nntk_fail_msg("spectrogram: window too large for the generic DFT kernel");
        hipLaunchKernelGGL(spectrogram_dft_kernel, dim3((unsigned)g), dim3(256), lds, nntk_stream(), p);
        NNTK_LAUNCH_CHECK("spectrogram_dft_kernel");
    }
    return 0;
}
