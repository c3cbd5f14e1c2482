// nntk_common.hpp -- shared device helpers and launch plumbing for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "nntk_shim.h"

hipStream_t nntk_stream();
int nntk_fail(const char *what, hipError_t err);
int nntk_fail_msg(const char *what);
int nntk_prof_span_begin();                       // -1 when profiling is off
void nntk_prof_span_end(int idx, long launches, long units);
void nntk_set_post_sync_hook(int (*hook)());      // runs after every host-visible stream sync

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
    default: return x;
    }
}

__device__ __forceinline__ float nntk_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

static inline int nntk_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
