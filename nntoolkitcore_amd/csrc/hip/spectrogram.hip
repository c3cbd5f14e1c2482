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

struct SpecParams {
    const float *in;      // [B, input_size]
    const float *window;  // [window_size]
    const float *tw;      // [nfft] interleaved (cos, -sin) = exp(-2*pi*i*m/nfft)
    float *out;           // [B, nts, nfreq]
    long total_frames;    // B * nts
    int input_size, nfft, window_size, step, nfreq, nts;
    float fft_norm, scale;
    int mode;
};

struct cf { float x, y; };
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf mul_mi(cf a) { return {a.y, -a.x}; }     // a * (-i)

#define BFLY2(a, b) do { cf _t = a; a = cadd(_t, b); b = csub(_t, b); } while (0)

// forward 8-point DFT, natural-order output, in registers
__device__ __forceinline__ void fft8(cf v[8]) {
    BFLY2(v[0], v[4]); BFLY2(v[1], v[5]); BFLY2(v[2], v[6]); BFLY2(v[3], v[7]);
    const float s = 0.70710678118654752440f;
    v[5] = cmul(v[5], cf{s, -s});
    v[6] = mul_mi(v[6]);
    v[7] = cmul(v[7], cf{-s, -s});
    // two 4-point DFTs: (v0..v3) -> even outputs, (v4..v7) -> odd outputs
    BFLY2(v[0], v[2]); BFLY2(v[1], v[3]); v[3] = mul_mi(v[3]); BFLY2(v[0], v[1]); BFLY2(v[2], v[3]);
    BFLY2(v[4], v[6]); BFLY2(v[5], v[7]); v[7] = mul_mi(v[7]); BFLY2(v[4], v[5]); BFLY2(v[6], v[7]);
    // registers now hold X0,X4,X2,X6 | X1,X5,X3,X7  (v0=X0, v1=X4, v2=X2, v3=X6, v4=X1, v5=X5, v6=X3, v7=X7)
    cf x1 = v[4], x2 = v[2], x3 = v[6], x4 = v[1], x5 = v[5], x6 = v[3];
    v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5; v[6] = x6;
}

#define SPEC_LDS_PER_WAVE 592      // 512 + 8 * 8 padding, rounded to a multiple of 16
__device__ __forceinline__ int pidx(int i) { return i + ((i >> 6) << 3); }

__device__ __forceinline__ float finish_bin(const SpecParams &p, float re, float im, int k) {
    if (p.fft_norm != 1.0f) { re *= p.fft_norm; im *= p.fft_norm; }
    float m = re * re + im * im;
    if (p.mode == 0) return sqrtf(m) / p.scale;
    if (k == 0 || k == p.nfreq - 1) return m / p.scale;
    return m * (2.0f / p.scale);
}

__global__ __launch_bounds__(256) void spectrogram512_kernel(SpecParams p) {
    __shared__ __attribute__((aligned(16))) float lds_re[4][SPEC_LDS_PER_WAVE];
    __shared__ __attribute__((aligned(16))) float lds_im[4][SPEC_LDS_PER_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float *re = lds_re[wave], *im = lds_im[wave];
    const long npairs = (p.total_frames + 1) / 2;
    const long wave_stride = (long)gridDim.x * 4;

    for (long pair = (long)blockIdx.x * 4 + wave; pair < ((npairs + 3) / 4) * 4; pair += wave_stride) {
        // all four waves of a workgroup run the same number of iterations (barriers below)
        const bool active = pair < npairs;
        const long fa = pair * 2, fb = pair * 2 + 1;
        const bool has_b = active && fb < p.total_frames;
        const float *xa = nullptr, *xb = nullptr;
        if (active) xa = p.in + (fa / p.nts) * (long)p.input_size + (fa % p.nts) * (long)p.step;
        if (has_b) xb = p.in + (fb / p.nts) * (long)p.input_size + (fb % p.nts) * (long)p.step;

        cf v[8];
        // ---- pass 1 (Ns = 1): windowed samples from global, no twiddles ----
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n = lane + 64 * r;
            float w = 0.f, a = 0.f, b = 0.f;
            if (n < p.window_size) {
                w = p.window[n];
                if (active) a = xa[n];
                if (has_b) b = xb[n];
            }
            v[r] = cf{w * a, w * b};
        }
        fft8(v);
        {
            const int base = pidx(lane * 8);     // 8 contiguous, 32-B aligned floats per lane
            *reinterpret_cast<float4 *>(re + base) = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
            *reinterpret_cast<float4 *>(re + base + 4) = make_float4(v[4].x, v[5].x, v[6].x, v[7].x);
            *reinterpret_cast<float4 *>(im + base) = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
            *reinterpret_cast<float4 *>(im + base + 4) = make_float4(v[4].y, v[5].y, v[6].y, v[7].y);
        }
        __syncthreads();
        // ---- pass 2 (Ns = 8): twiddle exp(-2 pi i (j%8) r / 64) ----
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = pidx(lane + 64 * r);
            v[r] = cf{re[i], im[i]};
        }
        {
            const int jm = lane & 7;
#pragma unroll
            for (int r = 1; r < 8; ++r) {
                const float2 t = reinterpret_cast<const float2 *>(p.tw)[jm * r * 8];
                v[r] = cmul(v[r], cf{t.x, t.y});
            }
        }
        fft8(v);
        __syncthreads();
        {
            const int base = (lane >> 3) * 64 + (lane & 7);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = pidx(base + 8 * r);
                re[i] = v[r].x; im[i] = v[r].y;
            }
        }
        __syncthreads();
        // ---- pass 3 (Ns = 64): twiddle exp(-2 pi i j r / 512) ----
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = pidx(lane + 64 * r);
            v[r] = cf{re[i], im[i]};
        }
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            const float2 t = reinterpret_cast<const float2 *>(p.tw)[lane * r];
            v[r] = cmul(v[r], cf{t.x, t.y});
        }
        fft8(v);
        __syncthreads();   // LDS image is free for the next pair
        // lane j now holds Z[j + 64 r], r = 0..7.
        // ---- split the two real spectra; bins k = j + 64 r for r = 0..3 (+ k = 256 on lane 0) ----
        const int src = (64 - lane) & 63;
        cf zc[5];          // Z[N - k] for r = 0..3, and for k = 256
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // lane != 0: partner register 7 - r on lane 64 - j; lane 0: register (8 - r) & 7 of itself
            const float pr = __shfl(v[7 - r].x, src), pi = __shfl(v[7 - r].y, src);
            const cf self = v[(8 - r) & 7];
            zc[r] = lane == 0 ? self : cf{pr, pi};
        }
        zc[4] = v[4];      // k = 256 pairs with itself (lane 0 only)
        if (active) {
            float *oa = p.out + fa * (long)p.nfreq;
            float *ob = p.out + fb * (long)p.nfreq;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                if (r == 4 && lane != 0) break;
                const int k = lane + 64 * r;
                const cf z = v[r], c = zc[r];
                // X_a = (Z + conj(Zc)) / 2 ; X_b = (Z - conj(Zc)) / (2i)
                const float ar = 0.5f * (z.x + c.x), ai = 0.5f * (z.y - c.y);
                const float br = 0.5f * (z.y + c.y), bi = -0.5f * (z.x - c.x);
                oa[k] = finish_bin(p, ar, ai, k);
                if (has_b) ob[k] = finish_bin(p, br, bi, k);
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
    p.input_size = input_size; p.nfft = nfft; p.window_size = window_size; p.step = step;
    p.nfreq = nfreq; p.nts = nts; p.fft_norm = fft_norm; p.scale = scale; p.mode = mode;
    if (nfft == 512) {
        long npairs = (p.total_frames + 1) / 2;
        long g = (npairs + 3) / 4;
        if (g > 256 * 8 * 4) g = 256 * 8 * 4;
        hipLaunchKernelGGL(spectrogram512_kernel, dim3((unsigned)g), dim3(256), 0, nntk_stream(), p);
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
