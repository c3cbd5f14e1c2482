// mempat_bench.hip -- what can K1's memory pattern reach with NO compute at all?
//  (a) float4 grid-stride copy (the guide's 6.29 TB/s reference), 427 MB in + 427 MB out
//  (b) K1's traffic shape: per frame pair one wave reads 3 x 1 KB contiguous (560 samples used, pairs overlap 2.5x)
//      and writes 2 x 1 KB + 8 B at a 2056-byte stride (only 8-byte aligned) -- stack size 512 x 500 pairs
//  (c) same reads, stores 16-byte ALIGNED (rows padded to 2064 B)        (d) same reads, b32 stores (round 1's form)
//  (e) reads only   (f) stores only
// build: hipcc -O3 --offload-arch=gfx950 mempat_bench.hip -o bin/mempat_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v4 __attribute__((ext_vector_type(4)));
typedef unsigned v2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void copy4(const float4 *in, float4 *out, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) out[i] = in[i];
}

// MODE bit0: do loads, bit1: do stores; ST: 0 = x4 at 2056 stride, 1 = x4 aligned (2064 stride), 2 = b32 x 10
template <int MODE, int ST, int CONSEC = 0>
__global__ __launch_bounds__(256) void k1pat(const float *in, float *out, int B, int N, int ppu, int ppw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stride = gridDim.x * 4;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)b * N), 0, N * 4, 0x00020000);
        const size_t orow = ST == 1 ? 2064 : ST == 3 ? 2048 : 2056;
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)((char *)out + (size_t)b * ppu * orow), 0, (int)(ppu * orow), 0x00020000);
        v4 acc = {1u, 2u, 3u, 4u};
        const int p0 = CONSEC ? (blockIdx.x * 4 + wave) * ppw : blockIdx.x * 4 + wave;
        const int p1 = CONSEC ? (p0 + ppw < ppu ? p0 + ppw : ppu) : ppu;
        for (int pr = p0; pr < p1; pr += CONSEC ? 1 : stride) {
            if (MODE & 1) {
                const int so = pr * 1280;
                v4 a = __builtin_amdgcn_raw_buffer_load_b128(rin, lane * 16 + so, 0, 0);
                v4 c = __builtin_amdgcn_raw_buffer_load_b128(rin, lane * 16 + 1024 + so, 0, 0);
                v4 d = __builtin_amdgcn_raw_buffer_load_b128(rin, lane < 12 ? lane * 16 + 2048 + so : 0x7ffffff0, 0, 0);
                acc += a ^ c ^ d;
            }
            if (MODE & 2) {
                const int so = pr * (int)orow;
                if (ST != 2) {
                    __builtin_amdgcn_raw_buffer_store_b128(acc, rout, lane * 16, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(acc, rout, lane * 16 + 1024, so, 0);
                    if (ST != 3) __builtin_amdgcn_raw_buffer_store_b64((v2){acc.x, acc.y}, rout, lane == 0 ? 2048 : 0x7ffffff0, so, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        __builtin_amdgcn_raw_buffer_store_b32(acc.x + r, rout, lane * 4, so + 256 * r, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(acc.y + r, rout, lane * 4, so + 1028 + 256 * r, 0);
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(acc.z, rout, lane == 0 ? 1024 : 0x7ffffff0, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(acc.w, rout, lane == 0 ? 2052 : 0x7ffffff0, so, 0);
                }
            } else if (acc.x == 0x12345678u) out[0] = 1.f;
        }
    }
}

template <typename F> static float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1e3f;
}

int main() {
    const int B = 512, N = 160240, ppu = 500;
    float *in, *out;
    hipMalloc(&in, (size_t)B * N * 4 + 4096);
    hipMalloc(&out, (size_t)B * ppu * 2064 + 4096);
    hipMemset(in, 1, (size_t)B * N * 4);
    const long n4 = (long)B * ppu * 2056 / 16 * 0 + 427000000L / 16;
    float us = timeit([&] { copy4<<<4096, 256>>>((const float4 *)out, (float4 *)in + 0, 0); });
    (void)us;
    float *c0, *c1; hipMalloc(&c0, n4 * 16); hipMalloc(&c1, n4 * 16);
    us = timeit([&] { copy4<<<8192, 256>>>((const float4 *)c0, (float4 *)c1, n4); });
    printf("(a) float4 copy 427 MB + 427 MB          %8.1f us  %.2f TB/s\n", us, 2.0 * n4 * 16 / us / 1e6);
    const double rd = (double)B * N * 4, wr = (double)B * ppu * 2056;
    for (int ppw : {2, 4, 16}) {
        dim3 g((ppu + 4 * ppw - 1) / (4 * ppw), B);
        us = timeit([&] { k1pat<3, 0><<<g, 256>>>(in, out, B, N, ppu, ppw); });
        printf("(b) K1 pattern, x4 stores 8-B aligned ppw %2d %8.1f us  %.2f TB/s\n", ppw, us, (rd + wr) / us / 1e6);
        us = timeit([&] { k1pat<3, 1><<<g, 256>>>(in, out, B, N, ppu, ppw); });
        printf("(c) K1 pattern, x4 stores 16-B aligned ppw %2d %8.1f us  %.2f TB/s\n", ppw, us, (rd + wr) / us / 1e6);
        us = timeit([&] { k1pat<3, 2><<<g, 256>>>(in, out, B, N, ppu, ppw); });
        printf("(d) K1 pattern, b32 stores            ppw %2d %8.1f us  %.2f TB/s\n", ppw, us, (rd + wr) / us / 1e6);
        us = timeit([&] { k1pat<1, 0><<<g, 256>>>(in, out, B, N, ppu, ppw); });
        printf("(e) reads only                        ppw %2d %8.1f us  %.2f TB/s (of %.0f MB)\n", ppw, us, rd / us / 1e6, rd / 1e6);
        us = timeit([&] { k1pat<2, 0><<<g, 256>>>(in, out, B, N, ppu, ppw); });
        printf("(f) x4 stores only                    ppw %2d %8.1f us  %.2f TB/s (of %.0f MB)\n", ppw, us, wr / us / 1e6, wr / 1e6);
        us = timeit([&] { k1pat<2, 3><<<g, 256>>>(in, out, B, N, ppu, ppw); });
        printf("(h) dense 2048-B aligned chunks only  ppw %2d %8.1f us  %.2f TB/s\n", ppw, us, (double)B * ppu * 2048 / us / 1e6);
        us = timeit([&] { k1pat<2, 0, 1><<<g, 256>>>(in, out, B, N, ppu, ppw); });
        printf("(i) x4 stores only, consecutive pairs ppw %2d %8.1f us  %.2f TB/s\n", ppw, us, wr / us / 1e6);
        us = timeit([&] { k1pat<3, 0, 1><<<g, 256>>>(in, out, B, N, ppu, ppw); });
        printf("(j) K1 pattern x4, consecutive pairs  ppw %2d %8.1f us  %.2f TB/s\n", ppw, us, (rd + wr) / us / 1e6);
    }
    return 0;
}
