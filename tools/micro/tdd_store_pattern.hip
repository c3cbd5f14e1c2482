// Microbenchmark: what does the store pattern of dense_frag3_kernel's epilogue cost, alone?
//   The TimeDistributedDense of the stack writes out[b][t][0..999] (f32) from 256 x 256 workgroup tiles whose rows are 32 UTTERANCES of one
//   timestep (the frag3 row block): a store instruction's 64 lanes write 16 bytes each to 32 rows that lie T * N * 4 = 4 MB apart (two lanes
//   per row), four instructions complete a row's 128-byte line.  Compile-time ablations of the real kernel say the epilogue costs 0.5 ms of
//   2.5 (profiles/r05_tdd_store_ablation.log).  Question: would an LDS-transposed epilogue, whose instructions write 8 WHOLE lines, be
//   cheaper?  Modes (same bytes, same lines, every element written once):
//     0: the product's pattern (row l31, quad 2 g + kh)          1: whole lines (row 8 g + lane / 8, quad lane % 8)
//     2: two rows x 512 bytes per instruction (the wave's 128 columns of two rows)
//   All workgroups start together and store their tile back to back -- the burst the lock-stepped GEMM produces -- `tiles` tiles each.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tdd_store_pattern.hip -o tools/micro/bin/tdd_store_pattern && tools/micro/bin/tdd_store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void store_tiles(float *out, int B, int T, int N, int NHT, int m_tiles, int n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    for (int w = blockIdx.x; w < m_tiles * n_tiles; w += gridDim.x) {          // persistent: gridDim = CUs, tiles back to back
        const int tile = w / n_tiles, n0 = (w % n_tiles) * 256;
        const f32x4 v = {(float)w, (float)lane, 1.0f, 2.0f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long rb = (long)tile * 8 + wm * 4 + i;
            const int t = (int)(rb / NHT), ht = (int)(rb % NHT);
            if (t >= T) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int b, c;
                    if (MODE == 0) { b = ht * 32 + l31; c = n0 + (wn * 4 + j) * 32 + 8 * g + 4 * kh; }
                    else if (MODE == 1) { b = ht * 32 + 8 * g + (lane >> 3); c = n0 + (wn * 4 + j) * 32 + 4 * (lane & 7); }
                    else { b = ht * 32 + 8 * j + 2 * g + (lane >> 5); c = n0 + wn * 128 + 4 * (lane & 31); }
                    if (b < B && c < N) *reinterpret_cast<f32x4 *>(out + ((size_t)b * T + t) * N + c) = v;
                }
        }
    }
}

int main() {
    const int B = 512, T = 996, N = 1000, NHT = 16;
    const long rows = (long)T * NHT;                    // row blocks
    const int m_tiles = (int)((rows + 7) / 8), n_tiles = 4;
    float *out;
    const size_t bytes = (size_t)B * T * N * 4;
    if (hipMalloc(&out, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {8, 16, 32, 64, 128, 256, 512, 1024}) {       // (8 .. 128: how much of the write rate a fraction of the CUs reaches)
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(store_tiles<0>, dim3(grid), dim3(256), 0, 0, out, B, T, N, NHT, m_tiles, n_tiles);
                if (mode == 1) hipLaunchKernelGGL(store_tiles<1>, dim3(grid), dim3(256), 0, 0, out, B, T, N, NHT, m_tiles, n_tiles);
                if (mode == 2) hipLaunchKernelGGL(store_tiles<2>, dim3(grid), dim3(256), 0, 0, out, B, T, N, NHT, m_tiles, n_tiles);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("grid %4d mode %d: %.3f ms = %.2f TB/s\n", grid, mode, best, bytes / (best * 1e-3) / 1e12);
        }
    }
    hipFree(out);
    return 0;
}
