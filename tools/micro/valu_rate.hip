// valu_rate.hip -- what does one vector instruction cost on gfx950?  Cycles per wave-instruction of v_add_f32 / v_fma_f32 /
// v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32 (with and without op_sel modifiers) / v_sqrt_f32 / v_cndmask / v_mov, at
// 1, 2 and 4 wavefronts per SIMD, 8 independent dependency chains per wave.  One workgroup per CU, s_memtime around the loop.
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o bin/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 2000

template <int KIND>
__global__ void k(float *out, unsigned long long *cyc, float seed) {
    f2 a[8];
    float s[8];
    for (int i = 0; i < 8; ++i) { a[i] = (f2){seed + i + threadIdx.x, seed * i}; s[i] = seed + i * 0.5f + threadIdx.x; }
    const f2 c = (f2){seed * 0.999f, seed * 1.001f};
    const float cs = seed * 0.999f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(cs));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[i]) : "v"(cs));
            if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
            if (KIND == 6) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[i]));
            if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(cs));
            if (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "+v"(s[i]) : "v"(cs));
            if (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(cs));
            if (KIND == 10) asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(a[i]), "+v"(s[i]) : "v"(c), "v"(cs));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int per_wave_insts) {
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps, blocks = 256;
        float *out; unsigned long long *cyc;
        hipMalloc(&out, sizeof(float) * threads * blocks);
        hipMalloc(&cyc, 8 * blocks * threads / 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, threads>>>(out, cyc, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<blocks, threads>>>(out, cyc, 1.0f);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * threads / 64);
        hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += (double)v;
        const double cyc_wave = sum / h.size();
        const double n = (double)ITER * 8 * per_wave_insts;
        printf("%-28s waves/SIMD %d: %6.2f cycles per instruction per wave, %6.2f SIMD cycles per instruction (%.1f us)\n", name, wps,
               cyc_wave / n, cyc_wave / n / wps, ms * 1e3);
        hipFree(out); hipFree(cyc);
    }
}

int main() {
    run<0>("v_add_f32", 1); run<1>("v_fma_f32", 1); run<9>("v_mul_f32", 1); run<2>("v_pk_add_f32", 1); run<3>("v_pk_fma_f32", 1);
    run<4>("v_pk_mul_f32", 1); run<5>("v_pk_add_f32 op_sel+neg", 1); run<6>("v_sqrt_f32", 1); run<7>("v_cndmask_b32", 1);
    run<8>("v_mov_b32", 1); run<10>("v_pk_add + v_add pair", 2);
    return 0;
}
