// Microbenchmark: does VALU work of one wavefront make progress while the other wavefront of the
// same SIMD issues back-to-back MFMAs?  One workgroup of 8 waves per CU (2 per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PRIO>   // MODE 0: waves 0-3 MFMA, 4-7 VALU; 1: only VALU waves work; 2: only MFMA; 3: swap roles
__global__ __launch_bounds__(512) void coissue(float *sink, unsigned long long *cyc, int iters) {
    const int w8 = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool first = w8 < 4;
    const bool do_mfma = (MODE == 0 && first) || (MODE == 2 && first) || (MODE == 3 && !first);
    const bool do_valu = (MODE == 0 && !first) || (MODE == 1 && !first) || (MODE == 3 && first);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    if (do_mfma) {
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    }
    if (do_valu) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        float v0 = lane * 0.01f, v1 = 0.5f, v2 = 0.25f, v3 = 0.125f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {      // 64 dependent-ish VALU ops + 4 transcendentals per trip
                v0 = __builtin_fmaf(v0, 1.0001f, v1);
                v1 = __builtin_fmaf(v1, 0.9999f, v2);
                v2 = __builtin_fmaf(v2, 1.0002f, v3);
                v3 = __builtin_fmaf(v3, 0.9998f, v0);
            }
            v0 = __builtin_amdgcn_exp2f(v0 * 1e-30f);
            v1 = __builtin_amdgcn_rcpf(v1 + 2.0f);
        }
        r = v0 + v1 + v2 + v3;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[w8] = t1 - t0;
    if (r == 1234.5678f) sink[threadIdx.x] = r;
}

template <int MODE, int PRIO>
static void run(const char *name, float *sink, unsigned long long *d_cyc) {
    unsigned long long h[8];
    (void)hipMemset(d_cyc, 0, 64);
    coissue<MODE, PRIO><<<256, 512>>>(sink, d_cyc, 200);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d_cyc, 64, hipMemcpyDeviceToHost);
    printf("%-46s waves0-3: %7llu %7llu %7llu %7llu   waves4-7: %7llu %7llu %7llu %7llu cycles\n", name,
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
}

int main() {
    float *sink; unsigned long long *cyc;
    (void)hipMalloc(&sink, 4096); (void)hipMalloc(&cyc, 64);
    run<2, 0>("MFMA alone (waves 0-3)", sink, cyc);
    run<1, 0>("VALU alone (waves 4-7)", sink, cyc);
    run<0, 0>("MFMA (0-3) + VALU (4-7)", sink, cyc);
    run<0, 1>("MFMA (0-3) prio0 + VALU (4-7) prio3", sink, cyc);
    run<3, 0>("VALU (0-3) + MFMA (4-7)", sink, cyc);
    run<3, 1>("VALU (0-3) prio3 + MFMA (4-7) prio0", sink, cyc);
    return 0;
}
