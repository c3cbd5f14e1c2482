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

// one wave per SIMD pair... every wave: K MFMAs (32x32x2, 64 cycles each), each followed by NV independent VALU ops
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, bool LDS>
__global__ __launch_bounds__(256) void intrawave(float *sink, unsigned long long *cyc, int iters) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0; a1[r] = 0; }
    float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = lane * 0.1f + k;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) v[(u + k) & 7] = __builtin_fmaf(v[(u + k) & 7], 1.0001f, 0.5f);
            if (LDS) x += lds[(lane + u * 64 + i) & 4095];
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) v[(u + k + 4) & 7] = __builtin_fmaf(v[(u + k + 4) & 7], 0.9999f, 0.25f);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
    float r = a0[0] + a1[5];
    for (int k = 0; k < 8; ++k) r += v[k];
    if (r == 1234.5678f) sink[threadIdx.x] = r;
}
// each 32x32x2 MFMA followed by ND independent ds_read_b128 (results never consumed) and NG global loads
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int ND, int NG>
__global__ __launch_bounds__(256) void intrawave_mem(float *sink, unsigned long long *cyc, const float *gsrc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
    __syncthreads();
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0; a1[r] = 0; }
    float x = lane * 0.001f, y = 1.0f + lane * 0.002f;
    const unsigned laddr = (unsigned)(size_t)(lds) + (threadIdx.x & 255) * 16;
    const float *gp = gsrc + (blockIdx.x * 256 + threadIdx.x) * 4;
    f32x4v t0v, t1v;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < ND; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t0v) : "v"(laddr), "n"(k * 4096));
#pragma unroll
            for (int k = 0; k < NG; ++k) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t1v) : "v"(gp));
            if (ND + NG > 0 && (u & 3) == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
    float r = a0[0] + a1[5];
    if (r == 1234.5678f) sink[threadIdx.x] = r + t0v[0] + t1v[0];
}
// The recurrent K loop in isolation: 16 chunks x (4 ds_read_b128 of the NEXT chunk's fragments, 16 dependent-free
// v_mfma_f32_16x16x4 on the current ones), one wave per SIMD, B operand in registers.
typedef float f32x4m __attribute__((ext_vector_type(4)));
template <int AHEAD>
__global__ __launch_bounds__(256) void kloop(float *sink, unsigned long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float U[];          // [64][520]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 520; i += 256) U[i] = (float)(i & 255) * 0.001f;
    __syncthreads();
    const int l15 = lane & 15, q = lane >> 4;
    const float *ub = &U[l15 * 520 + q * 4];
    float hv[16][4];
    for (int c = 0; c < 16; ++c) for (int k = 0; k < 4; ++k) hv[c][k] = lane * 0.01f + c + k;
    f32x4m acc[4];
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4m){0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float4 un[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) un[g] = *reinterpret_cast<const float4 *>(ub + g * 16 * 520);
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) {
            float4 uc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) uc[g] = un[g];
            if (ch + 1 < 16) {
#pragma unroll
                for (int g = 0; g < 4; ++g) un[g] = *reinterpret_cast<const float4 *>(ub + g * 16 * 520 + (ch + 1) * 32);
            }
            if (AHEAD) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float u = s2 == 0 ? uc[g].x : s2 == 1 ? uc[g].y : s2 == 2 ? uc[g].z : uc[g].w;
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(u, hv[ch][s2], acc[g], 0, 0, 0);
                }
            if (AHEAD) __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
    float r = 0;
    for (int g = 0; g < 4; ++g) r += acc[g][0] + acc[g][3];
    if (r == 1234.5678f) sink[threadIdx.x] = r;
}
template <int AHEAD>
static void run_kloop(const char *name, float *sink, unsigned long long *d_cyc) {
    unsigned long long h[8];
    (void)hipFuncSetAttribute((const void *)kloop<AHEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 520 * 4);
    kloop<AHEAD><<<256, 256, 64 * 520 * 4>>>(sink, d_cyc, 100);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d_cyc, 64, hipMemcpyDeviceToHost);
    printf("%-60s %7.0f cycles per 256-MFMA K loop (pure MFMA = 8192)\n", name, h[0] / 100.0);
}

template <int ND, int NG>
static void run_mem(const char *name, float *sink, unsigned long long *d_cyc, const float *gsrc) {
    unsigned long long h[8];
    intrawave_mem<ND, NG><<<256, 256>>>(sink, d_cyc, gsrc, 200);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d_cyc, 64, hipMemcpyDeviceToHost);
    printf("%-60s %7llu cycles  (MFMA alone = %d)\n", name, h[0], 200 * 16 * 64);
}

template <int NV, bool LDS>
static void run_intra(const char *name, float *sink, unsigned long long *d_cyc) {
    unsigned long long h[8];
    intrawave<NV, LDS><<<256, 256>>>(sink, d_cyc, 200);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d_cyc, 64, hipMemcpyDeviceToHost);
    printf("%-60s %7llu cycles  (MFMA alone = %d)\n", name, h[0], 200 * 16 * 64);
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
    run_intra<0, false>("1 wave/SIMD: 32x32x2 MFMAs only", sink, cyc);
    run_intra<2, false>("1 wave/SIMD: each MFMA + 2 independent v_fma", sink, cyc);
    run_intra<4, false>("1 wave/SIMD: each MFMA + 4 independent v_fma", sink, cyc);
    run_intra<8, false>("1 wave/SIMD: each MFMA + 8 independent v_fma", sink, cyc);
    float *gsrc; (void)hipMalloc(&gsrc, 256 * 256 * 16); (void)hipMemset(gsrc, 0, 256 * 256 * 16);
    run_mem<0, 0>("1 wave/SIMD: 3200 MFMAs (asm harness)", sink, cyc, gsrc);
    run_mem<1, 0>("1 wave/SIMD: each MFMA + 1 ds_read_b128", sink, cyc, gsrc);
    run_mem<2, 0>("1 wave/SIMD: each MFMA + 2 ds_read_b128", sink, cyc, gsrc);
    run_mem<0, 1>("1 wave/SIMD: each MFMA + 1 global_load_dwordx4", sink, cyc, gsrc);
    run_mem<1, 1>("1 wave/SIMD: each MFMA + 1 ds_read_b128 + 1 global load", sink, cyc, gsrc);
    run_kloop<0>("K loop, 1 wave/SIMD, scheduler's placement", sink, cyc);
    run_kloop<1>("K loop, 1 wave/SIMD, LDS reads pinned one chunk ahead", sink, cyc);
    return 0;
}
