// Microbenchmark: how fast can every CU pull its 128 KB h tile (64 rows x 512 f32) per step?
// 256 workgroups x 512 threads, workgroup i reads tile (i % 8) -- 32 workgroups share a tile,
// as in rec_persistent_kernel.  One "step" = all loads, s_waitcnt vmcnt(0), s_barrier.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));

#define H 512
template <int PATTERN, int AUX>
__global__ __launch_bounds__(512) void hload(const float *h, float *sink, int steps, int ntiles) {
    const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6;
    const int bt = blockIdx.x % ntiles;
    const char *base = (const char *)(h + (size_t)bt * 64 * H);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 64 * H * 4, 0x00020000);
    v4u32 acc = {0, 0, 0, 0};
    for (int t = 0; t < steps; ++t) {
        v4u32 r[16];
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) {
            int off;
            if (PATTERN == 0) {          // kernel's: 16 rows x 64 B per instruction
                const int grp = w8 >> 2, slab = w8 & 3, l15 = lane & 15, q = lane >> 4;
                off = (slab * 16 + l15) * H * 4 + (ch * 32 + grp * 16 + q * 4) * 4;
            } else if (PATTERN == 1) {   // 8 rows x 128 B per instruction
                const int row = w8 * 8 + (lane >> 3);
                off = row * H * 4 + ch * 128 + (lane & 7) * 16;
            } else if (PATTERN == 2) {   // 1 KB contiguous per instruction (half a row)
                const int row = w8 * 8 + (ch >> 1);
                off = row * H * 4 + (ch & 1) * 1024 + lane * 16;
            } else {                     // 4 rows x 256 B per instruction
                const int row = w8 * 8 + (ch >> 3) * 4 + (lane >> 4);
                off = row * H * 4 + (ch & 7) * 256 + (lane & 15) * 16;
            }
            r[ch] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX);
        }
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) acc += r[ch];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (acc.x == 0x12345678u) sink[tid] = (float)acc.y;
}

template <int PATTERN, int AUX>
static void run(const char *name, const float *d_h, float *d_sink, int steps, int ntiles) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hload<PATTERN, AUX><<<256, 512>>>(d_h, d_sink, 50, ntiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hload<PATTERN, AUX><<<256, 512>>>(d_h, d_sink, steps, ntiles);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s tiles %d: %.2f us/step  (%.1f GB/s per CU, %.2f TB/s chip)\n", name, ntiles, ms * 1000.0 / steps,
           131072.0 / (ms * 1e-3 / steps) / 1e9, 256 * 131072.0 / (ms * 1e-3 / steps) / 1e12);
}

int main() {
    float *d_h, *d_sink;
    hipMalloc(&d_h, (size_t)256 * 64 * H * 4);
    hipMemset(d_h, 0, (size_t)256 * 64 * H * 4);
    hipMalloc(&d_sink, 4096);
    const int steps = 1000;
    for (int ntiles : {8, 256}) {
        run<0, 16>("kernel pattern (16 rows x 64 B), sc1", d_h, d_sink, steps, ntiles);
        run<0, 0>("kernel pattern, plain", d_h, d_sink, steps, ntiles);
        run<0, 1>("kernel pattern, sc0", d_h, d_sink, steps, ntiles);
        run<0, 17>("kernel pattern, sc0 sc1", d_h, d_sink, steps, ntiles);
        run<0, 2>("kernel pattern, nt", d_h, d_sink, steps, ntiles);
        run<0, 18>("kernel pattern, sc1 nt", d_h, d_sink, steps, ntiles);
        run<1, 16>("8 rows x 128 B, sc1", d_h, d_sink, steps, ntiles);
        run<1, 0>("8 rows x 128 B, plain", d_h, d_sink, steps, ntiles);
        run<3, 16>("4 rows x 256 B, sc1", d_h, d_sink, steps, ntiles);
        run<2, 16>("1 KB contiguous, sc1", d_h, d_sink, steps, ntiles);
        run<2, 0>("1 KB contiguous, plain", d_h, d_sink, steps, ntiles);
    }
    return 0;
}
