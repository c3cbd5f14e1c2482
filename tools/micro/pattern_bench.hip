// Microbenchmark: cost of vector-memory access patterns (all 256 CUs busy, 512 threads each).
// Each "step" moves 128 KB per workgroup; patterns differ only in which lane touches which address.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));

// ROWB = row stride in bytes of the [rows][K] matrix the tile is cut from
template <int PATTERN, int ROWB>
__global__ __launch_bounds__(512) void loads(const char *src, float *sink, int steps) {
    const int tid = threadIdx.x;
    const char *base = src + (size_t)blockIdx.x * (1 << 20);      // 1 MB region per workgroup
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 1 << 20, 0x00020000);
    v4u32 acc = {0, 0, 0, 0};
    for (int t = 0; t < steps; ++t) {
        v4u32 r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int off;
            if (PATTERN == 0)        // 4 adjacent lanes = 64 B of a row, 128 rows per pass (GEMM window, KC = 16)
                off = ((tid >> 2)) * ROWB + (tid & 3) * 16 + i * 64;
            else if (PATTERN == 1)   // 8 adjacent lanes = 128 B of a row, 64 rows per pass
                off = ((tid >> 3)) * ROWB + (tid & 7) * 16 + i * 128;
            else if (PATTERN == 2)   // 32 adjacent lanes = 512 B of a row (weight chunk rows)
                off = ((tid >> 5) + 16 * (i & 7)) * ROWB + (tid & 31) * 16 + (i >> 3) * 512;
            else                     // lanes 16 apart share a row: 16 rows x 64 B per wave, neighbours in different rows
                off = ((tid >> 6) * 16 + (tid & 15)) * ROWB + ((tid >> 4) & 3) * 16 + i * 64;
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += r[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (acc.x == 0x12345678u) sink[tid] = (float)acc.y;
}

template <int PATTERN, int ROWB>
__global__ __launch_bounds__(512) void stores(char *dst, int steps) {
    const int tid = threadIdx.x;
    char *base = dst + (size_t)blockIdx.x * (1 << 20);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 1 << 20, 0x00020000);
    for (int t = 0; t < steps; ++t) {
        if (PATTERN == 0) {          // GEMM epilogue: b32, 32 lanes = 128 B of a row, 2 rows per wave-instruction; 64 instr
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const int row = (tid >> 6) * 16 + ((tid >> 5) & 1) * 8 + (i & 7) + 128 * 0;
                const int off = (row + 0) * ROWB + (tid & 31) * 4 + (i >> 3) * 128;
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)t, rs, off, 0, 0);
            }
        } else if (PATTERN == 1) {   // b128, 8 adjacent lanes = 128 B of a row; 16 instr
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int off = (tid >> 3) * ROWB + (tid & 7) * 16 + i * 128;
                v4u32 v = {(unsigned)t, 1u, 2u, 3u};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
            }
        } else if (PATTERN == 2) {   // b128, 1 KB contiguous per wave-instruction
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int off = tid * 16 + i * 8192;
                v4u32 v = {(unsigned)t, 1u, 2u, 3u};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
            }
        } else {                     // b128, one lane = 16 B of its own row (32x32 MFMA transposed output): 64 rows per wave-instr
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int off = (tid & 127) * ROWB + (tid >> 7) * 16 + i * 64;
                v4u32 v = {(unsigned)t, 1u, 2u, 3u};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

template <typename F> static void timeit(const char *name, F launch, int steps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(20);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch(steps);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-78s %.2f us per 128 KB/CU  (%.2f TB/s chip)\n", name, ms * 1000.0 / steps, 256 * 131072.0 / (ms * 1e-3 / steps) / 1e12);
}

int main() {
    char *buf; float *sink;
    (void)hipMalloc(&buf, (size_t)256 << 20);
    (void)hipMemset(buf, 0, (size_t)256 << 20);
    (void)hipMalloc(&sink, 4096);
    const int steps = 500;
#define L(P, R, name) timeit(name, [&](int s) { loads<P, R><<<256, 512>>>(buf, sink, s); }, steps)
#define S(P, R, name) timeit(name, [&](int s) { stores<P, R><<<256, 512>>>(buf, s); }, steps)
    L(0, 2048, "load b128: 4 lanes = 64 B/row, row stride 2 KB   (GEMM window KC=16, K=512)");
    L(0, 512,  "load b128: 4 lanes = 64 B/row, row stride 512 B  (GEMM window KC=16, K=128)");
    L(1, 2048, "load b128: 8 lanes = 128 B/row, row stride 2 KB  (GEMM window KC=32, K=512)");
    L(1, 512,  "load b128: 8 lanes = 128 B/row, row stride 512 B");
    L(2, 4096, "load b128: 32 lanes = 512 B/row, row stride 4 KB (weight chunk)");
    L(3, 2048, "load b128: lanes 16 apart share a row (old h pattern), row stride 2 KB");
    S(0, 4096, "store b32: 32 lanes = 128 B/row, 2 rows/instr, stride 4 KB (GEMM epilogue, N=1024)");
    S(1, 4096, "store b128: 8 lanes = 128 B/row, 8 rows/instr, stride 4 KB");
    S(2, 4096, "store b128: 1 KB contiguous per instr");
    S(3, 4096, "store b128: 1 lane = 16 B of its own row, 64 rows/instr, stride 4 KB");
    return 0;
}
