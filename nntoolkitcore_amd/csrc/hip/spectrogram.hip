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
    int ppw;              // frame pairs per wavefront
    // fused mel projection: out becomes [B, nts, n_mels] = (log of) spec x W, W's nonzero run per filter
    const int *mel_tab;   // [3][n_mels]: first bin, run length, offset into mel_w
    const float *mel_w;   // the runs, packed
    int n_mels, mel_log;
    float mel_eps;
#ifdef NNTK_SPEC_DBG
    int dbg;              // timing experiments only: 1 no stores, 2 no sample loads, 4 no LDS passes (wrong results), 8 no split / shuffles
#endif
};
#ifdef NNTK_SPEC_DBG
#define SPEC_DBG(bit) (p.dbg & (bit))
#else
#define SPEC_DBG(bit) 0
#endif

typedef float f2 __attribute__((ext_vector_type(2)));     // (re, im)
typedef unsigned v4u32x __attribute__((ext_vector_type(4)));
typedef unsigned v2u32x __attribute__((ext_vector_type(2)));

// ---- packed-f32 complex arithmetic -------------------------------------------------------------------------
// K1 is VALU-bound (measured, DESIGN.md: with loads, stores and LDS passes all removed the loop still takes 3/4 of
// its time), and on gfx950 EVERY vector instruction of a wave costs one 4-cycle issue slot, packed or not -- the
// f32 peak is reached only by v_pk_* instructions, which do two lanes' worth of work in that slot.  So the count of
// vector instructions is the cost model, and the complex arithmetic is written directly in v_pk_*_f32 with their
// operand modifiers: op_sel / op_sel_hi pick which half of each 64-bit source feeds the low / high result, neg_lo /
// neg_hi negate it.  A multiplication by -i, a conjugation or a (re, re) broadcast then costs nothing -- hipcc
// (ROCm 7.2) folds whole-vector negations and some swaps, but emitted v_xor / v_mov pairs for per-half signs:
// 277 vector instructions per frame pair before, 17x after.  The statements are plain (non-volatile) asm on register
// operands only: the compiler still schedules and interleaves them freely.
#define SPEC_PK2(name, text) \
    __device__ __forceinline__ f2 name(f2 a, f2 b) { f2 d; asm(text : "=v"(d) : "v"(a), "v"(b)); return d; }
#define SPEC_PK1(name, text) \
    __device__ __forceinline__ f2 name(f2 a) { f2 d; asm(text : "=v"(d) : "v"(a)); return d; }
#define SPEC_PK3(name, text) \
    __device__ __forceinline__ f2 name(f2 a, f2 b, f2 c) { f2 d; asm(text : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
SPEC_PK2(add_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")        // a + (-i) b
SPEC_PK2(sub_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")        // a - (-i) b
SPEC_PK1(rot_w1, "v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")        // a (1 - i)  = sqrt2 a w8^1
SPEC_PK1(rot_w3, "v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]")   // a (-1 - i) = sqrt2 a w8^3
SPEC_PK3(fma_mi, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]")   // c + b (-i) a   (b real pair)
SPEC_PK3(fms_mi, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]")   // c - b (-i) a
SPEC_PK2(cmul_lo, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]")                          // (a.x b.x, a.x b.y)
SPEC_PK3(cmul_hi, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]")   // (c.x - a.y b.y, c.y + a.y b.x)
SPEC_PK2(split_p, "v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]")       // (a.x + b.x, a.x - b.x)
SPEC_PK2(split_q, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_lo:[0,1]")       // (a.y - b.y, a.y + b.y)
SPEC_PK2(mul_blo, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]")                                 // a * (b.x, b.x)
SPEC_PK2(mul_bhi, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]")                                    // a * (b.y, b.y)
SPEC_PK3(fma_bhi, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]")                              // a * (b.y, b.y) + c
__device__ __forceinline__ f2 cmul2(f2 a, f2 b) { return cmul_hi(a, b, cmul_lo(a, b)); }   // complex a * b: 2 instructions
// S = (1/sqrt2, 1/sqrt2) is wave-uniform: it rides in an SGPR pair (one scalar source per VOP3P instruction is allowed)
__device__ __forceinline__ f2 fma_mi_s(f2 a, f2 S, f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "s"(S), "v"(c)); return d; }
__device__ __forceinline__ f2 fms_mi_s(f2 a, f2 S, f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "s"(S), "v"(c)); return d; }

// Eight 8-byte LDS reads at base + STEP * r bytes, as eight ds_read_b64.  Written in asm because hipcc fuses neighbouring
// reads into ds_read2_b64 / ds_read2st64_b64, which move 16 bytes per lane in 8 LDS cycles where two ds_read_b64 take 2 + 2
// (MI355X_MICROARCH.md, LDS table) -- and the LDS pipe is this kernel's most loaded unit.  The wait is inside the
// statement: hipcc does not count an asm load.
template <int STEP>
__device__ __forceinline__ void lds_read8_b64(f2 v[8], const float *base) {
    const unsigned addr = (unsigned)(size_t)base;     // LDS pointers are 32-bit offsets in the low half
    asm volatile("ds_read_b64 %0, %8\n\t"
                 "ds_read_b64 %1, %8 offset:%c9\n\t"
                 "ds_read_b64 %2, %8 offset:%c10\n\t"
                 "ds_read_b64 %3, %8 offset:%c11\n\t"
                 "ds_read_b64 %4, %8 offset:%c12\n\t"
                 "ds_read_b64 %5, %8 offset:%c13\n\t"
                 "ds_read_b64 %6, %8 offset:%c14\n\t"
                 "ds_read_b64 %7, %8 offset:%c15\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(addr), "i"(STEP), "i"(2 * STEP), "i"(3 * STEP), "i"(4 * STEP), "i"(5 * STEP), "i"(6 * STEP), "i"(7 * STEP)
                 : "memory");
}

// forward 8-point DFT in registers, natural-order output: 26 packed instructions (24 when v[7] is known to be zero).
// S = (1/sqrt2, 1/sqrt2).  Even outputs: 4-point DFT of the sums; odd outputs: 4-point DFT of (a4, w1 a5, -i a6, w3 a7)
// with the two 1/sqrt2 factors riding in the last butterflies' FMAs.
template <bool V7ZERO>
__device__ __forceinline__ void fft8(f2 v[8], const f2 S) {
    const f2 a0 = v[0] + v[4], a4 = v[0] - v[4];
    const f2 a1 = v[1] + v[5], a5 = v[1] - v[5];
    const f2 a2 = v[2] + v[6], a6 = v[2] - v[6];
    f2 a3, a7;
    if (V7ZERO) { a3 = v[3]; a7 = v[3]; } else { a3 = v[3] + v[7]; a7 = v[3] - v[7]; }
    const f2 b0 = a0 + a2, b2 = a0 - a2, b1 = a1 + a3, b3 = a1 - a3;
    v[0] = b0 + b1;
    v[4] = b0 - b1;
    v[2] = add_mi(b2, b3);
    v[6] = sub_mi(b2, b3);
    const f2 c4 = add_mi(a4, a6), c6 = sub_mi(a4, a6);
    const f2 t5 = rot_w1(a5), t7 = rot_w3(a7);
    const f2 u = t5 + t7, w = t5 - t7;
    v[1] = __builtin_elementwise_fma(u, S, c4);
    v[5] = __builtin_elementwise_fma(u, -S, c4);
    v[3] = fma_mi_s(w, S, c6);
    v[7] = fms_mi_s(w, S, c6);
}

// Every wavefront owns its private LDS image and a wave's DS instructions execute in
// order, so no s_barrier is needed between passes -- only a fence that stops the
// compiler from reordering the (aliasing) LDS accesses.  Waves of a workgroup drift freely.
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
                             __builtin_amdgcn_wave_barrier();                        \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define SPEC_DMA_DEFAULT false     // measured choice
#define SPEC_LDS_FLOATS 1184      // per wave: exchange 1 uses 1024 floats, exchange 2 uses 8 * 144 = 1152; multiple of 16 B

__device__ __forceinline__ float finish_bin(const SpecParams &p, float re, float im, int k) {
    if (p.fft_norm != 1.0f) { re *= p.fft_norm; im *= p.fft_norm; }
    float m = re * re + im * im;
    // p.inv_scale = 1/scale rounded once on the host: one multiply instead of the reference's
    // divide (spectrogram.c:33) -- differs from it by at most 1 ulp, far inside the stated tolerance
    if (p.mode == 0) return sqrtf(m) * p.inv_scale;
    if (k == 0 || k == p.nfreq - 1) return m * p.inv_scale;
    return m * (2.0f * p.inv_scale);
}

// One 64-lane wavefront transforms TWO consecutive frames as one 512-point complex FFT (frame A real, frame B
// imaginary), 8 points per lane, three radix-8 Stockham passes, everything through a wave-private LDS image
// (no s_barrier anywhere: a wave's DS instructions execute in order).
//
// Memory side.  Measured with the transform removed, the round-1 access pattern (28 four-byte sample loads and 10
// four-byte stores per pair, every one fully coalesced) ALONE took as long as the whole kernel: a vector-memory
// instruction occupies the CU's address path for ~15 (load) / ~25 (store) cycles whatever its width, so 4-byte
// accesses cap a CU at ~17 B/clk.  Now a pair's step + window contiguous samples come in as 3 (at most 4) 16-byte
// loads per lane, are laid down in LDS, and pass 1 picks its samples (lane + 64 r for frame A, step + lane + 64 r for
// frame B) from there -- which also serves the 2.5x overlap between neighbouring frames from LDS instead of L1.  The
// two finished rows (2 x 257 floats, contiguous in the output) are assembled in LDS and leave as two 16-byte stores
// per lane plus an 8-byte tail.  38 -> 6..7 vector-memory instructions per pair.
//
// LDS images (all in the same 4.7 KB region, separated by wave-level fences):
//   samples:    word n = sample n of the pair.
//   exchange 1 (after pass 1): lane j writes its 8 outputs as one 64-byte row; pass 2 reads element j + 64 r, i.e.
//     row (j >> 3) + 8 r, column j & 7.  Rows are 16 words apart, so plain rows would put the 8 lanes of a
//     ds_write_b128 group on two 4-bank columns (4-way conflict: 32 LDS cycles instead of 8 per instruction, the
//     6 % SQ_LDS_BANK_CONFLICT of the round-1 profile; now 0).  The 16-byte pairs of row j are therefore rotated by
//     (j >> 1) & 3 slots: a group's 8 rows land on 8 different 4-bank columns, and the reads (4 rows x 8 columns per
//     32-lane group) still cover 64 different banks.  The rotation of row (j >> 3) + 8 r does not depend on r, so the
//     8 reads are ONE per-lane address plus immediates.
//   exchange 2 (after pass 2): element 64 a + 8 b + c at word 144 a + 16 b + 2 c (writes: a, c from the lane, b =
//     register; reads: a = register, 8 b + c = lane): conflict-free both ways.
//   output:     word k = bin k of frame A, word 257 + k = bin k of frame B.
// NZ7 = the window is 385..448 samples (the BASELINE 400 is): register row 7 of pass 1 is zero padding for every lane
// (not read, its butterfly disappears) and only row 6 straddles the window's end.  Otherwise all 8 rows are masked.
// Zero padding is exact even next to inf / nan samples: out-of-window samples are ANDed away, not multiplied by 0.
// NLD = 16-byte loads per lane that cover step + window samples (3 for 160 + 400).
// MEL = the mel projection (signal/mel_filterbank.c:116-118) and log(x + 1.5849e-13) (log_mel_spectrogram.c:31-36)
// are applied to the two finished rows while they are still in LDS: each filter is a triangle, i.e. ONE contiguous run
// of nonzero weights (2 .. ~40 bins), so a lane takes one (frame, filter) pair and sums its run in ascending bin order
// -- the order of the reference's dense product with the exact zeros left out.  The kernel then writes n_mels values
// per frame instead of 257 (6.4x less write traffic for n_mels = 40) and the [B, nts, 257] tensor never exists.
// DMA = the sample image is filled by LDS-DMA (buffer_load ... lds): the pair's step + window samples go global -> LDS without
// passing through registers (no 12-register staging set per prefetched pair, no ds_write_b128 -- at 13 LDS-issue cycles the
// most expensive LDS instruction of the kernel).  Two image slots per wave (the pair being picked from, the pair in flight);
// the wave waits for its DMA with a counted vmcnt (an LDS-DMA is ordered for a ds_read only by the issuing wave's vmcnt).
#define SPEC_IMG_FLOATS 576       // one sample image: step + window <= 576 floats (2304 bytes)
template <int MODE, bool NORM, bool NZ7, int NLD, bool MEL = false, bool DMA = false>
__global__ __launch_bounds__(256) void spectrogram512_kernel(SpecParams p) {
    __shared__ __attribute__((aligned(16))) float lds_z[4][SPEC_LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) float lds_img[DMA ? 4 : 1][DMA ? 2 * SPEC_IMG_FLOATS : 4];
    __shared__ int mel_tab_s[MEL ? 3 * 257 : 1];      // MEL: the run table and the runs, shared by the workgroup
    __shared__ float mel_w_s[MEL ? 2 * 257 + 8 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // scalar: frame offsets stay in SGPRs
    float *z = lds_z[wave];
    const int ppu = (p.nts + 1) >> 1;                 // frame pairs per utterance
    constexpr int NR = NZ7 ? 7 : 8;                   // register rows of pass 1 that can hold samples
    if (MEL) {
        for (int i = threadIdx.x; i < 3 * p.n_mels; i += 256) mel_tab_s[i] = p.mel_tab[i];
        const int nw = p.mel_tab[3 * p.n_mels - 1] + p.mel_tab[2 * p.n_mels - 1];      // offset + length of the last run
        for (int i = threadIdx.x; i < nw; i += 256) mel_w_s[i] = p.mel_w[i];
        __syncthreads();
    }

    // lane-only constants
    // window taps, and the AND mask that clears samples beyond the window (rows that lie fully inside the window need
    // none: NZ7 knows rows 0..5 do).  The window multiply is two plain v_mul_f32 per row: frame A's and frame B's samples
    // arrive in unrelated registers, and a packed multiply would first need them moved into a register pair.
    float wtap[NR];
    unsigned wmask[NZ7 ? 1 : 8];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int n = lane + 64 * r;
        const bool inwin = n < p.window_size;
        wtap[r] = inwin ? p.window[n] : 0.0f;
        if (!NZ7) wmask[r] = inwin ? 0xffffffffu : 0u;
        else if (r == 6) wmask[0] = inwin ? 0xffffffffu : 0u;
    }
#pragma unroll
    for (int r = 0; r < (NZ7 ? 1 : 8); ++r) asm volatile("" : "+v"(wmask[r]));     // keep them masks: hipcc turned x & mask back into a v_cndmask
    f2 tw2[8], tw3[8];
    {
        const f2 *tw = reinterpret_cast<const f2 *>(p.tw);
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            tw2[r] = tw[(lane & 7) * r * 8];          // exp(-2 pi i (j%8) r / 64)
            tw3[r] = tw[lane * r];                    // exp(-2 pi i j r / 512)
        }
    }
    // The two real spectra come out of the split below as 2 X_a, 2 X_b, so the factor 1/2 (1/4 for the PSD's squares)
    // rides in the output scale -- exact, a power of two.  Output scale per bin (magnitude: 1/sum(w); PSD:
    // 2/(fs sum w^2), edges 1/(..): spectrogram.c:41-47): wave-uniform except the PSD's two edge bins, which lane 0
    // owns in rows 0 and 4.  fft_normalization_factor still multiplies re/im first (NORM), like spectrogram.c:130-131.
    const float o_in = 0.5f * p.inv_scale;            // interior bins (PSD: 2 / 4)
    const float o_edge = MODE == 0 ? 0.5f * p.inv_scale : 0.25f * p.inv_scale;
    const f2 osc_in = (f2){o_in, o_in};
    const float o0 = lane == 0 ? o_edge : o_in;
    const f2 osc_r0 = (f2){o0, o0}, osc_r4 = (f2){o_edge, o_edge};
    const f2 S = (f2){0.70710678118654752440f, 0.70710678118654752440f};
    const f2 norm2 = (f2){p.fft_norm, p.fft_norm};
    // lane 0 pairs bin 64 r with bin 64 (8 - r) of ITSELF, every other lane with a register of lane 64 - j: the
    // partner is blended as own * m.x + shuffled * m.y with m = (1, 0) on lane 0 and (0, 1) elsewhere -- exact, and
    // two packed instructions instead of two v_cndmask (measured 4.4x the issue time of a v_pk_* each)
    const f2 msel = lane == 0 ? (f2){1.0f, 0.0f} : (f2){0.0f, 1.0f};
    // sample image: 16-byte pieces in, 4-byte picks out
    const int need_bytes = (p.step + p.window_size) * 4;
    int ld_off[NLD];                                         // lanes past the pair's last sample load nothing
#pragma unroll
    for (int i = 0; i < NLD; ++i) ld_off[i] = (lane * 16 + 1024 * i < need_bytes) ? lane * 16 + 1024 * i : 0x7ffffff0;
    const int sA = lane, sB = p.step + lane;                 // + 64 r (floats)
    // exchange 1: write address of 16-byte pair q of this lane's row; read address (floats) of element lane + 64 r
    int w1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w1[q] = 16 * lane + 4 * ((q + (lane >> 1)) & 3);
    const int r1 = 16 * (lane >> 3) + 4 * ((((lane & 7) >> 1) + (lane >> 4)) & 3) + 2 * (lane & 1);      // + 128 r
    // exchange 2
    const int w2 = 144 * (lane >> 3) + 2 * (lane & 7);      // + 16 k
    const int r2 = 2 * lane;                                 // + 144 r
    const int src = ((64 - lane) & 63) * 4;                  // ds_bpermute address of the conjugate-partner lane
    const int stride = gridDim.x * 4;                        // pairs between two pairs of one wave
    const int st_tail = lane == 0 ? 2048 : 0x7ffffff0;       // the 8-byte tail of a pair's two rows (bins 255, 256 of B)
    const int vo = lane * 4, vo4 = lane == 0 ? 0 : 0x7ffffff0;     // odd last frame: only lane 0 owns bin 256

    for (int b = blockIdx.y; b < p.B; b += gridDim.y) {
        // descriptors of this utterance: every offset that must be clipped at its end rides in the (range-checked)
        // vector offset -- a pair at the very end reads zeros past the signal, never the next utterance
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(p.in + (size_t)b * p.input_size), 0, p.input_size * 4, 0x00020000);
        const int orow = MEL ? p.n_mels : p.nfreq;       // floats per output row
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(p.out + (size_t)b * p.nts * orow), 0, p.nts * orow * 4, 0x00020000);
        // Software prefetch, TWO pairs deep: the sample loads of pair n + 2 are issued while pair n is transformed
        // (two register sets, loop unrolled by two).
        int pr = blockIdx.x * 4 + wave;
        if (pr >= ppu) continue;
        v4u32x ld0[DMA ? 1 : NLD], ld1[DMA ? 1 : NLD];
        // LDS-DMA: slot `slot` of this wave's two sample images <- the pair's samples (lanes past the image are masked off:
        // a DMA writes base + 16 * lane whatever the source offset says)
#define SPEC_DMA(slot, pair) do {                                                                       \
            const int soa_ = 2 * ((pair) < ppu ? (pair) : pr) * p.step * 4;                              \
            float *img_ = lds_img[wave] + (slot) * SPEC_IMG_FLOATS;                                      \
            _Pragma("unroll") for (int i = 0; i < NLD; ++i)                                              \
                if (lane * 16 + 1024 * i < need_bytes)                                                   \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void *)(img_ + 256 * i), 16, lane * 16 + 1024 * i + soa_, 0, 0, 0); \
            } while (0)

        // a pair beyond the wave's last one re-reads the last valid one (cache hit, result unused): no branches around loads
#define SPEC_ISSUE(ld, pair) do {                                                                       \
            const int soa_ = 2 * ((pair) < ppu ? (pair) : pr) * p.step * 4;                              \
            _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                            \
                if (SPEC_DBG(2)) { ld[i] = (v4u32x){0x3f000000u + lane, 0x3e000000u + i, 0x3d800000u, 0x3e800000u}; continue; } \
                ld[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, ld_off[i] + soa_, 0, 0);              \
            } } while (0)

        auto transform = [&](v4u32x (&ld)[DMA ? 1 : NLD], int slot) __attribute__((always_inline)) {
            const int fa = 2 * pr;
            const bool has_b = fa + 1 < p.nts;        // wave-uniform
            f2 v[8];
            unsigned xa[NR], xb[NR];
            if (DMA) {
                // this pair's image has landed once at most the NEXT pair's NLD requests and the previous pair's 3 output
                // stores (younger, in issue order) are outstanding
                asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NLD + 3) : "memory");
                const float *img = lds_img[wave] + slot * SPEC_IMG_FLOATS;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    xa[r] = __float_as_uint(img[sA + 64 * r]);
                    xb[r] = __float_as_uint(img[sB + 64 * r]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the picks are in registers: the slot is free
                SPEC_DMA(slot, pr + 2 * stride);      // ... for the pair after next
            } else {
            // ---- samples: registers -> LDS image -> the 2 x NR picks of pass 1 ----
#pragma unroll
            for (int i = 0; i < NLD; ++i) *reinterpret_cast<v4u32x *>(z + lane * 4 + 256 * i) = ld[i];
            SPEC_ISSUE(ld, pr + 2 * stride);          // this register set is free again: fetch the pair after next
            WAVE_LDS_SYNC();
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                xa[r] = __float_as_uint(z[sA + 64 * r]);
                xb[r] = __float_as_uint(z[sB + 64 * r]);
            }
            WAVE_LDS_SYNC();
            }
            // ---- pass 1 (Ns = 1): windowed samples, no twiddles ----
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (!NZ7) { xa[r] &= wmask[r]; xb[r] &= wmask[r]; }
                else if (r == 6) { xa[r] &= wmask[0]; xb[r] &= wmask[0]; }
                float va, vb;
                asm("v_mul_f32 %0, %1, %2" : "=v"(va) : "v"(__uint_as_float(xa[r])), "v"(wtap[r]));
                asm("v_mul_f32 %0, %1, %2" : "=v"(vb) : "v"(__uint_as_float(xb[r])), "v"(wtap[r]));
                v[r] = (f2){va, vb};
            }
            if (NZ7) v[7] = (f2){0.f, 0.f};
            if (SPEC_DBG(16)) {      // memory pattern only: no transform at all
                const int soa = fa * p.nfreq * 4;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    __builtin_amdgcn_raw_buffer_store_b128((v4u32x){__float_as_uint(v[2 * i].x), __float_as_uint(v[2 * i].y),
                                                            __float_as_uint(v[2 * i + 1].x), __float_as_uint(v[2 * i + 1].y)},
                                                           rout, lane * 16 + 1024 * i, soa, 0);
                return;
            }
            fft8<NZ7>(v, S);
            if (!SPEC_DBG(4)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {                // the lane's row, 16-byte pairs rotated (see above)
                *reinterpret_cast<f2 *>(z + w1[q]) = v[2 * q];
                *reinterpret_cast<f2 *>(z + w1[q] + 2) = v[2 * q + 1];
            }
            WAVE_LDS_SYNC();
            // ---- pass 2 (Ns = 8) ----
            lds_read8_b64<512>(v, z + r1);
            }
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmul2(v[r], tw2[r]);
            fft8<false>(v, S);
            if (!SPEC_DBG(4)) {
            WAVE_LDS_SYNC();
#pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<f2 *>(z + w2 + 16 * k) = v[k];
            WAVE_LDS_SYNC();
            // ---- pass 3 (Ns = 64) ----
            lds_read8_b64<576>(v, z + r2);
            }
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmul2(v[r], tw3[r]);
            fft8<false>(v, S);
            WAVE_LDS_SYNC();   // the exchange image is dead: the output rows are assembled in its place
            // lane j now holds Z[j + 64 r], r = 0..7.
            // ---- split the two real spectra; bins k = j + 64 r for r = 0..3 (+ k = 256 on lane 0) ----
            //   2 X_a[k] = Z[k] + conj Z[512 - k],  2i X_b[k] = Z[k] - conj Z[512 - k]; with c = Z[512 - k]:
            //   P = (z.x + c.x, z.x - c.x), Q = (z.y - c.y, z.y + c.y)  ->  (|2 X_a|^2, |2 X_b|^2) = P P + Q Q
            f2 m[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                f2 c;
                if (r < 4) {
                    f2 pz = v[7 - r];
                    if (!SPEC_DBG(8))
                        pz = (f2){__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(pz.x))),
                                  __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(pz.y)))};
                    c = fma_bhi(pz, msel, mul_blo(v[(8 - r) & 7], msel));
                } else {
                    c = v[4];                        // k = 256 pairs with itself (lane 0 only)
                }
                f2 P = split_p(v[r], c), Q = split_q(v[r], c);
                if (NORM) { P = P * norm2; Q = Q * norm2; }
                f2 mm = __builtin_elementwise_fma(Q, Q, P * P);
                // hardware square root (<= 1 ulp; libm's correctly rounded sqrtf costs 12 instructions a bin)
                if (MODE == 0) mm = (f2){__builtin_amdgcn_sqrtf(mm.x), __builtin_amdgcn_sqrtf(mm.y)};
                m[r] = mm * (r == 0 ? osc_r0 : r == 4 ? osc_r4 : osc_in);
            }
            const int soa = fa * p.nfreq * 4;         // the pair's first output row (bytes): wave-uniform scalar offset
            if (MEL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z[lane + 64 * r] = m[r].x;
                    z[257 + lane + 64 * r] = m[r].y;
                }
                if (lane == 0) { z[256] = m[4].x; z[513] = m[4].y; }
                WAVE_LDS_SYNC();
                // one lane per filter, both frames in the same pass over the filter's run (table and runs are in LDS)
                for (int mm = lane; mm < p.n_mels; mm += 64) {
                    const int k0 = mel_tab_s[mm], len = mel_tab_s[p.n_mels + mm];
                    const float *w = mel_w_s + mel_tab_s[2 * p.n_mels + mm];
                    const float *row = z + k0;
                    float acc_a = 0.0f, acc_b = 0.0f;
                    for (int i = 0; i < len; ++i) {
                        const float wi = w[i];
                        acc_a = fmaf(row[i], wi, acc_a);
                        acc_b = fmaf(row[257 + i], wi, acc_b);
                    }
                    if (p.mel_log) { acc_a = logf(acc_a + p.mel_eps); acc_b = logf(acc_b + p.mel_eps); }
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc_a), rout, mm * 4, fa * p.n_mels * 4, 0);
                    if (has_b) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc_b), rout, (p.n_mels + mm) * 4, fa * p.n_mels * 4, 0);
                }
                WAVE_LDS_SYNC();
            } else if (has_b) {
                // ---- both rows are contiguous in the output (2 x 257 floats from row fa): LDS, then 16-byte stores ----
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z[lane + 64 * r] = m[r].x;
                    z[257 + lane + 64 * r] = m[r].y;
                }
                if (lane == 0) { z[256] = m[4].x; z[513] = m[4].y; }
                WAVE_LDS_SYNC();
                const v4u32x q0 = *reinterpret_cast<const v4u32x *>(z + lane * 4);
                const v4u32x q1 = *reinterpret_cast<const v4u32x *>(z + 256 + lane * 4);
                const v2u32x qt = *reinterpret_cast<const v2u32x *>(z + 512);
                if (!SPEC_DBG(1)) {
                    __builtin_amdgcn_raw_buffer_store_b128(q0, rout, lane * 16, soa, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(q1, rout, lane * 16 + 1024, soa, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(qt, rout, st_tail, soa, 0);
                }
                WAVE_LDS_SYNC();   // the image is free for the next pair's samples
            } else {
                // odd frame count: the utterance's last pair has only frame A (once per utterance at most)
#pragma unroll
                for (int r = 0; r < 5; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[r].x), rout, r == 4 ? vo4 : vo, soa + 256 * r, 0);
                if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // not the 3 stores the counted wait assumes: drain
            }
        };

        if (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing of the previous utterance is in flight: the counts below start clean
            SPEC_DMA(0, pr);
            SPEC_DMA(1, pr + stride);
            // the steady-state wait allows NLD + 3 younger operations; before the first stores exist that is 3 too many for the
            // first pair: wait for it outright
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NLD) : "memory");
        } else {
            SPEC_ISSUE(ld0, pr);
            SPEC_ISSUE(ld1, pr + stride);
        }
        for (;;) {
            transform(ld0, 0);
            pr += stride;
            if (pr >= ppu) break;
            transform(ld1, 1);
            pr += stride;
            if (pr >= ppu) break;
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two requests past the wave's last pair must land before the slots are reused
#undef SPEC_ISSUE
#undef SPEC_DMA
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Mixed-radix path: any nfft = 2^a 3^b 5^c up to 4096 that is not 512 (256 / 1024 / 2048 are ordinary speech settings;
// the reference takes every size through kissfft's mixed-radix plan, signal/dft.c:23-47).  One workgroup transforms
// two consecutive frames as ONE complex FFT of nfft points (frame A real, frame B imaginary), Stockham autosort
// passes of radix 4 / 2 / 3 / 5 ping-ponging between two LDS images, twiddles from the host's table
// exp(-2 pi i m / nfft) (evaluated in double).  O(N log N) instead of the direct kernel's O(window x nfreq).
// ---------------------------------------------------------------------------------------------------------------
struct MixedPlan { int nfac; int fac[12]; };

__device__ __forceinline__ f2 cmulf(f2 a, f2 b) { return (f2){a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

template <int R>
__device__ __forceinline__ void dft_small(f2 (&v)[5]) {
    if (R == 2) {
        const f2 a = v[0], b = v[1];
        v[0] = a + b; v[1] = a - b;
    } else if (R == 4) {
        const f2 a = v[0] + v[2], b = v[0] - v[2], c = v[1] + v[3], d = v[1] - v[3];
        const f2 dmi = (f2){d.y, -d.x};                      // -i d
        v[0] = a + c; v[1] = b + dmi; v[2] = a - c; v[3] = b - dmi;
    } else if (R == 3) {
        const float s3 = 0.86602540378443864676f;            // sin(2 pi / 3)
        const f2 t = v[1] + v[2], d = v[1] - v[2];
        const f2 m = v[0] - 0.5f * t;
        const f2 e = (f2){s3 * d.y, -s3 * d.x};              // -i sin * d
        v[0] = v[0] + t; v[1] = m + e; v[2] = m - e;
    } else {                                                 // 5
        const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;      // cos(2 pi / 5), cos(4 pi / 5)
        const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;       // sin(2 pi / 5), sin(4 pi / 5)
        const f2 t1 = v[1] + v[4], t2 = v[2] + v[3], d1 = v[1] - v[4], d2 = v[2] - v[3];
        const f2 m1 = v[0] + c1 * t1 + c2 * t2, m2 = v[0] + c2 * t1 + c1 * t2;
        const f2 u1 = s1 * d1 + s2 * d2, u2 = s2 * d1 - s1 * d2;
        const f2 e1 = (f2){u1.y, -u1.x}, e2 = (f2){u2.y, -u2.x};                        // -i u
        v[0] = v[0] + t1 + t2; v[1] = m1 + e1; v[4] = m1 - e1; v[2] = m2 + e2; v[3] = m2 - e2;
    }
}

template <int R>
__device__ __forceinline__ void stockham_pass(const f2 *in, f2 *out, const f2 *tw, int N, int Ns) {
    const int nb = N / R;                                    // butterflies of this pass
    const int tstep = N / (Ns * R);                          // table stride of exp(-2 pi i / (Ns R))
    for (int j = threadIdx.x; j < nb; j += blockDim.x) {
        const int k = j % Ns;
        f2 v[5];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            v[r] = in[j + r * nb];
            if (r > 0) v[r] = cmulf(v[r], tw[(r * k * tstep) % N]);
        }
        dft_small<R>(v);
        const int j0 = (j / Ns) * Ns * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) out[j0 + r * Ns] = v[r];
    }
}

__global__ __launch_bounds__(256) void spectrogram_mixed_kernel(SpecParams p, MixedPlan plan) {
    extern __shared__ __attribute__((aligned(16))) float smem_mixed[];
    const int N = p.nfft;
    f2 *za = reinterpret_cast<f2 *>(smem_mixed), *zb = za + N;
    const f2 *tw = reinterpret_cast<const f2 *>(p.tw);
    const int ppu = (p.nts + 1) >> 1;
    const long total_pairs = (long)p.B * ppu;
    for (long pair = blockIdx.x; pair < total_pairs; pair += gridDim.x) {
        const int b = (int)(pair / ppu), fa = 2 * (int)(pair % ppu);
        const bool has_b = fa + 1 < p.nts;
        const float *xa = p.in + (size_t)b * p.input_size + (size_t)fa * p.step;
        const float *xb = has_b ? xa + p.step : xa;
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
            f2 v = (f2){0.f, 0.f};
            if (n < p.window_size) { const float w = p.window[n]; v = (f2){w * xa[n], w * xb[n]}; }     // zero-padded to nfft (spectrogram.c:120-124)
            za[n] = v;
        }
        __syncthreads();
        f2 *src = za, *dst = zb;
        int Ns = 1;
        for (int i = 0; i < plan.nfac; ++i) {
            const int R = plan.fac[i];
            if (R == 4) stockham_pass<4>(src, dst, tw, N, Ns);
            else if (R == 2) stockham_pass<2>(src, dst, tw, N, Ns);
            else if (R == 3) stockham_pass<3>(src, dst, tw, N, Ns);
            else stockham_pass<5>(src, dst, tw, N, Ns);
            Ns *= R;
            __syncthreads();
            f2 *t = src; src = dst; dst = t;
        }
        // split the two real spectra: X_a[k] = (Z[k] + conj Z[N-k]) / 2, X_b[k] = (Z[k] - conj Z[N-k]) / (2i)
        float *oa = p.out + ((size_t)b * p.nts + fa) * p.nfreq;
        for (int k = threadIdx.x; k < p.nfreq; k += blockDim.x) {
            const f2 z = src[k], c = src[k == 0 ? 0 : N - k];
            oa[k] = finish_bin(p, 0.5f * (z.x + c.x), 0.5f * (z.y - c.y), k);
            if (has_b) oa[p.nfreq + k] = finish_bin(p, 0.5f * (z.y + c.y), -0.5f * (z.x - c.x), k);
        }
        __syncthreads();
    }
}

// nfft -> radix list (4s first, then 2, 3, 5); false if another prime divides it
static bool mixed_plan(int n, MixedPlan *plan) {
    plan->nfac = 0;
    if (n < 2) return false;
    for (int r : {4, 2, 3, 5})
        while (n % r == 0 && plan->nfac < 12) { plan->fac[plan->nfac++] = r; n /= r; }
    return n == 1;
}

// Last resort (nfft with a prime factor above 5, or above 4096): one workgroup per frame, direct DFT over the window support.
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

static int spectrogram_launch(const float *d_in, const float *d_window, const float *d_twiddle, float *d_out,
                              int B, int input_size, int nfft, int window_size, int step,
                              int nfreq, int nts, float fft_norm, int mode, float scale,
                              const int *d_mel_tab, const float *d_mel_w, int n_mels, float mel_eps, int mel_log) {
    if (B <= 0 || nts <= 0) return 0;
    if (window_size > nfft) return nntk_fail_msg("spectrogram: window_size must be <= nfft");
    SpecParams p;
    p.in = d_in; p.window = d_window; p.tw = d_twiddle; p.out = d_out;
    p.total_frames = (long)B * nts;
    p.B = B;
    p.input_size = input_size; p.nfft = nfft; p.window_size = window_size; p.step = step;
    p.nfreq = nfreq; p.nts = nts; p.fft_norm = fft_norm; p.scale = scale; p.inv_scale = (float)(1.0 / (double)scale); p.mode = mode;
    p.mel_tab = d_mel_tab; p.mel_w = d_mel_w; p.n_mels = n_mels; p.mel_eps = mel_eps; p.mel_log = mel_log;
    const bool mel = d_mel_tab != nullptr;
    if (mel && nfft != 512) return 1;       // the caller runs the two-kernel form (K1, then the k = 1 GEMM)
#ifdef NNTK_SPEC_DBG
    p.dbg = nntk_options().conv_dbg;
#endif
    if (nfft == 512) {
        if ((long)input_size * 4 >= 0x7ffffff0L || (long)nts * nfreq * 4 >= 0x7ffffff0L)
            return nntk_fail_msg("spectrogram: one utterance must stay below 2 GiB");
        if (mel && n_mels > 257) return 1;
        const int ppu = (nts + 1) / 2;
        // frame pairs per wavefront per utterance: long runs amortise the per-utterance prologue (descriptor +
        // un-overlapped first prefetch) as long as the chip stays full.  Measured at the stack's size (256 k pairs):
        // 1 / 2 / 4 / 8 / 16 pairs = 0.245 / 0.225 / 0.220 / 0.207 / 0.199 ms; config 2 (12.5 k pairs): 4 is best.
        long ppw_l = (long)B * ppu / 4096;
        int ppw = ppw_l < 4 ? 4 : ppw_l > 16 ? 16 : (int)ppw_l;
        if (nntk_options().spec_ppw > 0) ppw = nntk_options().spec_ppw;
        unsigned gx = (unsigned)((ppu + 4 * ppw - 1) / (4 * ppw));
        unsigned gy = (unsigned)(B < 65535 ? B : 65535);
        // keep the grid near 8 workgroups per CU; the kernel strides over the rest
        while ((long)gx * gy > 256L * 8 * 8 && gy > 1) gy = (gy + 1) / 2;
        const bool norm = fft_norm != 1.0f;
        const bool nz7 = window_size > 384 && window_size <= 448;
        const bool ld3 = (step + window_size) * 4 <= 3072;          // 16-byte loads per lane for one pair's samples: 3 or 4
        // LDS-DMA sample images: the BASELINE geometry (3 requests per pair, image <= 576 floats), no fused mel; option spec_dma = 0 / 1 forces
        // (measured 4-6 % slower, profiles/r03_k1_dma_ab.log: instantiated only in the A/B variant build -DNNTK_VARIANT_SPEC_DMA, tools/build_variant.py)
#ifdef NNTK_VARIANT_SPEC_DMA
        const bool dma = ld3 && !mel && (step + window_size) <= SPEC_IMG_FLOATS && (nntk_options().spec_dma < 0 ? SPEC_DMA_DEFAULT : nntk_options().spec_dma == 1);
#define SPEC_KERN2(M, N, Z) (mel ? (ld3 ? spectrogram512_kernel<M, N, Z, 3, true> : spectrogram512_kernel<M, N, Z, 4, true>) \
                                 : dma ? spectrogram512_kernel<M, N, Z, 3, false, true> \
                                 : (ld3 ? spectrogram512_kernel<M, N, Z, 3> : spectrogram512_kernel<M, N, Z, 4>))
#else
#define SPEC_KERN2(M, N, Z) (mel ? (ld3 ? spectrogram512_kernel<M, N, Z, 3, true> : spectrogram512_kernel<M, N, Z, 4, true>) \
                                 : (ld3 ? spectrogram512_kernel<M, N, Z, 3> : spectrogram512_kernel<M, N, Z, 4>))
#endif
#define SPEC_KERN(M, N) (nz7 ? SPEC_KERN2(M, N, true) : SPEC_KERN2(M, N, false))
        auto kern = mode == 0 ? (norm ? SPEC_KERN(0, true) : SPEC_KERN(0, false))
                              : (norm ? SPEC_KERN(1, true) : SPEC_KERN(1, false));
        p.ppw = ppw;
        hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), 0, nntk_stream(), p);
        NNTK_LAUNCH_CHECK("spectrogram512_kernel");
    } else if (MixedPlan plan; nfft <= 4096 && mixed_plan(nfft, &plan)) {
        const long pairs = (long)B * ((nts + 1) / 2);
        const long g = pairs < 8192 ? pairs : 8192;
        int bs = (nfft / 4 + 63) & ~63;                     // one radix-4 butterfly per thread where the frame is long enough
        bs = bs < 64 ? 64 : bs > 256 ? 256 : bs;
        hipLaunchKernelGGL(spectrogram_mixed_kernel, dim3((unsigned)g), dim3(bs), (size_t)nfft * 2 * sizeof(f2), nntk_stream(), p, plan);
        NNTK_LAUNCH_CHECK("spectrogram_mixed_kernel");
    } else {
        long g = p.total_frames < 4096 ? p.total_frames : 4096;
        size_t lds = (size_t)window_size * sizeof(float);
        if (lds > 64 * 1024) return nntk_fail_msg("spectrogram: window too large for the generic DFT kernel");
        hipLaunchKernelGGL(spectrogram_dft_kernel, dim3((unsigned)g), dim3(256), lds, nntk_stream(), p);
        NNTK_LAUNCH_CHECK("spectrogram_dft_kernel");
    }
    return 0;
}

extern "C" int nntk_shim_spectrogram(const float *d_in, const float *d_window, const float *d_twiddle, float *d_out,
                                     int B, int input_size, int nfft, int window_size, int step,
                                     int nfreq, int nts, float fft_norm, int mode, float scale) {
    return spectrogram_launch(d_in, d_window, d_twiddle, d_out, B, input_size, nfft, window_size, step, nfreq, nts, fft_norm,
                              mode, scale, nullptr, nullptr, 0, 0.f, 0);
}

// K1 with the mel projection (and optionally log(x + eps)) fused into its output stage: d_out is [B, nts, n_mels].
// d_mel_tab = [3][n_mels] ints (first bin, run length, offset of the run in d_mel_w).  Returns 1 (nothing launched) for
// configurations the fused kernel does not take (nfft != 512): the caller then runs K1 and the k = 1 GEMM.
extern "C" int nntk_shim_spectrogram_mel(const float *d_in, const float *d_window, const float *d_twiddle, float *d_out,
                                         int B, int input_size, int nfft, int window_size, int step,
                                         int nfreq, int nts, float fft_norm, int mode, float scale,
                                         const int *d_mel_tab, const float *d_mel_w, int n_mels, float eps, int do_log) {
    if (!d_mel_tab || !d_mel_w || n_mels <= 0) return nntk_fail_msg("spectrogram_mel: missing filter table");
    return spectrogram_launch(d_in, d_window, d_twiddle, d_out, B, input_size, nfft, window_size, step, nfreq, nts, fft_norm,
                              mode, scale, d_mel_tab, d_mel_w, n_mels, eps, do_log);
}
