// nntk_common.hpp -- shared device helpers and launch plumbing for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "nntk_shim.h"

hipStream_t nntk_stream();                        // the calling THREAD's current stream
int nntk_fail(const char *what, hipError_t err);
int nntk_fail_msg(const char *what);
enum { NNTK_SPAN_REC = 0, NNTK_SPAN_SPEC = 1 };
int nntk_prof_span_begin(int kind);               // -1 when profiling is off
void nntk_prof_span_end(int idx, long launches, long units);

// Tuning / diagnostics knobs (runtime.hip): environment defaults read once, nntk_hip_set_option() at run time.
// -1 = "auto" (the launcher's own measured choice).
struct NntkOptions {
    int rec_persistent = -1;     // 0: always the per-timestep recurrent kernels
    int rec_xw = -1;             // where the persistent kernel issues its xW loads (see recurrent.hip)
    int rec_pingpong = -1;       // ping-pong halves in the persistent kernel
    int rec_groups = 2;          // split-K groups of the per-timestep kernel
    int rec_spin_us = 1000000;   // budget of every in-kernel spin before it gives up and raises the fault word
    int rec_stream = -1;         // small-batch streaming kernel (0 off)
    int rec_fused2 = -1;         // fused two-layer GRU kernel (0 off)
    int rec_rr = -1;             // register-resident split-bf16 LSTM kernel with the fused input projection (0 off, 1 also for small batches)
    int rec_xf = -1;             // register-resident kernels: f32 input packed into frag3 form first (1 always, 0 only when the f32 path cannot take the shape; auto: see recurrent.c)
    int rec_fk = -1;             // full-K register-resident kernels (recurrent_fk.hip): 1 every shape they take, 0 none; auto: 256-wide inputs (where they beat the split-K family)
    int rec_hf = -1;             // LSTM (H > 256) -> FRAG2H output: the HF instantiation of lstm_rr_kernel (h.U on two f16 images, three products); 0 off
    int dense_frag3 = -1;        // dense GEMM with a frag3 A operand (0: consumers unpack to f32 and run the LDS-staged GEMM)
    int dense_f16x2 = -1;        // LSTM -> TimeDistributedDense: the tensor in between as two f16 images, three products (frag3.hip FRAG2H); 0: the frag3 route
    int train_outer_plain = -1;  // weight-gradient products of plain matrices on the VALU-free MFMA kernel (0: the general one; A/B)
    int train_bptt = -1;         // GRU / LSTM gradient: the whole BPTT loop in one persistent kernel (0: two launches per timestep)
    int spec_ppw = 0;            // frame pairs per wavefront in K1 (0 = auto)
    int spec_variant = -1;       // 1: log-mel as two kernels (K1, then the GEMM) instead of the fused output stage (A/B, tests)
#ifdef NNTK_VARIANT_SPEC_DMA
    int spec_dma = -1;           // K1 sample images by LDS-DMA (1) or through registers (0); auto = registers (measured faster)
#endif
    int bn_fast = 0;             // reciprocal-multiply BatchNorm (not the reference's divide)
    int gemm_tm_batch = -1;      // tile time-major GEMM outputs over the batch
    int conv_a4 = 1;             // 16-byte window loads also for channel counts that are not multiples of 4 (0: 4-byte loads there)
    int conv_flatk = -1;         // flat-K split convolution for Cin % 8 == 0, Cin % 16 != 0 (conv1d_flatk.hip); 0 off (variant builds: 2, 4)
    int conv_frag3_out = -1;     // Conv1d...ApplyDeviceFrag3: -1 / 1 the frag3 epilogue where it applies, 0 always conv -> f32 scratch -> pack (same bits)
    int conv_store = -1;         // GEMM epilogue: -1 auto, 0 row form (4-byte stores), 1 quad form (16-byte stores); bit-identical
    int gemm_split_bf16 = -1;    // 3-way split-bf16 contraction: -1 auto (conv / dense / TDD / mel, not the recurrent xW), 0 never, 1 all
    int gemm_wide = -1;          // 128 x 256 tile for wide dense GEMMs on the split path (0 off)
    int conv_dbg = 0;            // diagnostics build only
    int weights_check = -1;      // host-pointer Apply: 1 (default) whole-block compare with the shadow each call, sampled only on
                                 // the streaming recurrent call; 2 whole block everywhere; 0 trust *SyncWeights
};
const NntkOptions &nntk_options();

unsigned *nntk_fault_word();                      // device word the persistent kernel ORs into on a spin timeout
int nntk_fault_enqueue_copy();                    // async copy of the word to its pinned mirror on the current stream
int nntk_persistent_disabled();                   // set after a fault: take the per-timestep kernels
void nntk_persistent_launch_begin();              // orders persistent launches across the process's streams (holds a mutex)
void nntk_persistent_launch_end();
bool nntk_weights_exact_only(const void *d_wp);     // conv1d.hip: the packed weight block holds a value the bf16 split cannot represent
int nntk_cu_count();                              // cached per device
void nntk_set_last_conv_kernel(const char *name); // static string; read back with nntk_hip_last_conv_kernel()
void nntk_set_last_rec_kernel(const char *name);  // static string; read back with nntk_hip_last_recurrent_kernel()
int nntk_set_max_dynamic_lds(const void *kernel, size_t bytes);   // hipFuncSetAttribute once per (kernel, device)
// train.hip: C^T-sliced weight-gradient product on the f32 MFMA (outer_mfma_kernel); 0 = shape not taken, else slices written
int nntk_outer_mfma_launch(const float *d_A, const float *d_B, float *d_partial, long rows, int I, int K, int a_shift_T,
                           long rps, long seq_pitch, long row_pitch);
int nntk_resident_blocks(const void *kernel, int threads, size_t lds, int max_per_cu);   // occupancy x CUs (0: does not fit)

#define NNTK_HIP_TRY(expr)                                             \
    do {                                                               \
        hipError_t _e = (expr);                                        \
        if (_e != hipSuccess) return nntk_fail(#expr, _e);             \
    } while (0)

#define NNTK_LAUNCH_CHECK(name)                                        \
    do {                                                               \
        hipError_t _e = hipGetLastError();                             \
        if (_e != hipSuccess) return nntk_fail(name, _e);              \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Activation semantics follow layers/activation_default.c:
//   sigmoid :28-33  = 1 / (1 + exp(-x)) with a true divide
//   tanh    :65-67  = tanhf
//   relu    :123-129 = max(x, 0) then * a (a is an OUTPUT scale, skipped when a == 1)
//   identity:98-103
__device__ __forceinline__ float nntk_act(int kind, float x, float relu_a) {
    switch (kind) {
    case NNTK_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
    case NNTK_ACT_TANH:    return tanhf(x);
    case NNTK_ACT_RELU: {
        float y = fmaxf(x, 0.0f);
        return relu_a != 1.0f ? y * relu_a : y;
    }
    case NNTK_ACT_LOG_EPS: return logf(x + relu_a);
    default: return x;
    }
}

// Gate math for the recurrent step epilogue, where it sits on the sequential critical path
// (measured: libm-accurate expf/tanhf + IEEE divides cost ~2.9k cycles per step, 8 % of an
// LSTM-512 step).  Hardware exp2 / rcp forms: sigmoid(x) = rcp(1 + exp2(-x*log2 e)),
// tanh(x) = sign(x) * (1 - e) * rcp(1 + e), e = exp2(-2|x|*log2 e).  Max abs error vs the
// libm forms ~1.5e-7 (both functions are bounded by 1), covered by the stated 1e-5 / 1e-4
// tolerances and checked by the T = 1000 parity tests.
__device__ __forceinline__ float nntk_fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float nntk_fast_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(-2.88539008177792681f * fabsf(x));
    const float r = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(r, x);
}
// `a` = ReLU's output scale (activation_default.c:123-129: max(x, 0), then * a when a != 1); ignored by the other kinds
__device__ __forceinline__ float nntk_gate_act(int kind, float x, float a = 1.0f) {
    switch (kind) {
    case NNTK_ACT_SIGMOID: return nntk_fast_sigmoid(x);
    case NNTK_ACT_TANH:    return nntk_fast_tanh(x);
    case NNTK_ACT_RELU: {
        const float y = fmaxf(x, 0.0f);
        return a != 1.0f ? y * a : y;
    }
    default: return x;
    }
}

static inline int nntk_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
