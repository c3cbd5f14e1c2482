// Microbenchmark: what does a flag hand-off between two workgroups cost, by how the consumer polls?
//   recurrent_rr.hip raises a flag word with a write-through (agent-scope) store and polls it with a VECTOR sc1 load; the poll sits in
//   the wave's in-order vector-memory queue and a half-step's S_E2 slice spends ~1 k cycles waiting for it.  Candidates:
//     mode 0: vector load sc1 (today)      mode 1: SCALAR load, glc (bypasses the scalar cache; returns out of order w.r.t. the
//     vector queue)                         mode 2: vector load, no sc bits (L2-coherent inside an XCD only -- shows the L2 hit latency)
// Ping-pong: workgroup A raises flag[0] = i, B waits for it and raises flag[64] = i, A waits, N rounds; reported: ns per one-way hop.
// Pairs are placed on the same XCD (blocks b and b + 8) or on different XCDs (b and b + 1) -- workgroups are dealt round-robin
// over the 8 XCDs.  Every spin is bounded: a mode that never sees the flag reports "TIMEOUT" instead of hanging.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/flag_latency.hip -o tools/micro/bin/flag_latency && tools/micro/bin/flag_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__device__ __forceinline__ unsigned poll_once(const unsigned *f) {
    unsigned v;
    if (MODE == 1) {
        asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(f) : "memory");
    } else if (MODE == 0) {
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(f) : "memory");
    } else {
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(f) : "memory");
    }
    return v;
}

template <int MODE>
__global__ __launch_bounds__(64) void pingpong(unsigned *flags, int rounds, int partner_stride, unsigned long long *ticks, int *timeouts) {
    // pair p = blocks (a, a + partner_stride); roles by position inside the pair
    const int b = blockIdx.x;
    const int grp = b / (2 * partner_stride), r = b % (2 * partner_stride);
    const int role = r / partner_stride;                       // 0 = A, 1 = B
    const int pair = grp * partner_stride + r % partner_stride;
    unsigned *fa = flags + (size_t)pair * 256, *fb = fa + 64;  // separate 256-byte lines
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int to = 0;
    for (int i = 1; i <= rounds && !to; ++i) {
        if (role == 0) {
            if (threadIdx.x == 0) __hip_atomic_store(fa, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (poll_once<MODE>(fb) < (unsigned)i) { if (++spins > 2000000u) { to = 1; break; } }
        } else {
            unsigned spins = 0;
            while (poll_once<MODE>(fa) < (unsigned)i) { if (++spins > 2000000u) { to = 1; break; } }
            if (threadIdx.x == 0) __hip_atomic_store(fb, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (threadIdx.x == 0) {
        ticks[b] = __builtin_amdgcn_s_memrealtime() - t0;
        if (to) atomicAdd(timeouts, 1);
        if (to) {   // release the partner
            __hip_atomic_store(fa, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fb, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int MODE>
static void run(const char *name, int pairs, int stride, int rounds) {
    unsigned *flags; unsigned long long *ticks; int *timeouts;
    const int blocks = pairs * 2;
    hipMalloc(&flags, (size_t)pairs * 1024); hipMemset(flags, 0, (size_t)pairs * 1024);
    hipMalloc(&ticks, blocks * 8); hipMalloc(&timeouts, 4); hipMemset(timeouts, 0, 4);
    hipLaunchKernelGGL(pingpong<MODE>, dim3(blocks), dim3(64), 0, 0, flags, rounds, stride, ticks, timeouts);
    hipDeviceSynchronize();
    unsigned long long *h = (unsigned long long *)malloc(blocks * 8); int to = 0;
    hipMemcpy(h, ticks, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost);
    double mx = 0, sum = 0;
    for (int i = 0; i < blocks; ++i) { double ns = h[i] * 10.0 / (2.0 * rounds); sum += ns; if (ns > mx) mx = ns; }
    printf("%-22s %3d pairs, partner %s: %7.1f ns per hop (max %7.1f)%s\n", name, pairs, stride == 8 ? "same XCD " : "other XCD",
           sum / blocks, mx, to ? "   TIMEOUT: flag never seen" : "");
    hipFree(flags); hipFree(ticks); hipFree(timeouts); free(h);
}

int main() {
    const int rounds = 2000;
    for (int rep = 0; rep < 2; ++rep)
        for (int pairs : {8, 64, 128}) {
            run<0>("vector sc1", pairs, 8, rounds); run<0>("vector sc1", pairs, 1, rounds);
            run<1>("scalar glc", pairs, 8, rounds); run<1>("scalar glc", pairs, 1, rounds);
            run<2>("vector plain", pairs, 8, rounds); run<2>("vector plain", pairs, 1, rounds);
        }
    return 0;
}
