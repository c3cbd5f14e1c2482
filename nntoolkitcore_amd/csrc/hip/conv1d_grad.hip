// conv1d_grad.hip -- first slice of the training path: Conv1dCalculateGradient (layers/conv_1d.c:185-245).
//   dW[o][i][kk] = sum_b sum_x dout[b, x, o] * in[b, x*stride + kk, i]
//   db[o]        = sum_b sum_x dout[b, x, o]
//   dX[b, t, i]  = sum_o sum_{x, kk : x*stride + kk = t} dout[b, x, o] * W[o, i, kk]
// Correct and deterministic (fixed-order two-stage reductions, no atomics); NOT tuned: plain VALU kernels with coalesced
// dout reads.  The MFMA forms (dW as a split-K GEMM over the B*Tout rows, dX as the forward kernel on flipped weights) are
// the next step once the rest of the training path exists.
#include "nntk_common.hpp"

#define GRAD_SLICES 32

// one block per (kk, i) and row slice; thread o accumulates dW[o][i][kk] over the slice's (b, x) rows
__global__ __launch_bounds__(256) void conv1d_dw_partial_kernel(const float *in, const float *dout, float *partial,
                                                                int B, int T, int Cin, int Cout, int k, int stride, int Tout) {
    const int col = blockIdx.x;                 // kk * Cin + i
    const int kk = col / Cin, i = col % Cin;
    const long rows = (long)B * Tout;
    const long r0 = rows * blockIdx.y / gridDim.y, r1 = rows * (blockIdx.y + 1) / gridDim.y;
    for (int o = threadIdx.x; o < Cout; o += blockDim.x) {
        float acc = 0.0f;
        for (long r = r0; r < r1; ++r) {
            const long b = r / Tout;
            const int x = (int)(r % Tout);
            acc = fmaf(dout[r * Cout + o], in[((size_t)b * T + (size_t)x * stride + kk) * Cin + i], acc);
        }
        partial[((size_t)blockIdx.y * gridDim.x + col) * Cout + o] = acc;
    }
}

// bias gradient partials: block y = row slice
__global__ __launch_bounds__(256) void conv1d_db_partial_kernel(const float *dout, float *partial, long rows, int Cout) {
    const long r0 = rows * blockIdx.x / gridDim.x, r1 = rows * (blockIdx.x + 1) / gridDim.x;
    for (int o = threadIdx.x; o < Cout; o += blockDim.x) {
        float acc = 0.0f;
        for (long r = r0; r < r1; ++r) acc += dout[r * Cout + o];
        partial[(size_t)blockIdx.x * Cout + o] = acc;
    }
}

// fixed-order sum of the slices; writes dW in the caller's [Cout][Cin][k] layout and db
__global__ __launch_bounds__(256) void conv1d_grad_reduce_kernel(const float *pw, const float *pb, float *dW, float *db,
                                                                 int Cin, int Cout, int k, int slices, size_t pw_slice, size_t pb_slice) {
    const int ncol = k * Cin;
    const long total = (long)ncol * Cout;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total + Cout; e += (long)gridDim.x * blockDim.x) {
        if (e < total) {
            const int col = (int)(e / Cout), o = (int)(e % Cout);
            const int kk = col / Cin, i = col % Cin;
            float s = 0.0f;
            for (int sl = 0; sl < slices; ++sl) s += pw[(size_t)sl * pw_slice + (size_t)col * Cout + o];
            dW[((size_t)o * Cin + i) * k + kk] = s;
        } else {
            const int o = (int)(e - total);
            float s = 0.0f;
            for (int sl = 0; sl < slices; ++sl) s += pb[(size_t)sl * pb_slice + o];
            db[o] = s;
        }
    }
}

// one thread per input element: gathers every (x, kk) that touched it
__global__ __launch_bounds__(256) void conv1d_dx_kernel(const float *dout, const float *W, float *dX,
                                                        int B, int T, int Cin, int Cout, int k, int stride, int Tout) {
    const long total = (long)B * T * Cin;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e % Cin);
        const long bt = e / Cin;
        const int t = (int)(bt % T);
        const long b = bt / T;
        float acc = 0.0f;
        for (int kk = 0; kk < k; ++kk) {
            const int tt = t - kk;
            if (tt < 0 || tt % stride) continue;
            const int x = tt / stride;
            if (x >= Tout) continue;
            const float *d = dout + ((size_t)b * Tout + x) * Cout;
            for (int o = 0; o < Cout; ++o) acc = fmaf(d[o], W[((size_t)o * Cin + i) * k + kk], acc);
        }
        dX[e] = acc;
    }
}

// d_in [B,T,Cin] (the forward pass's input), d_W [Cout][Cin][k] (caller layout), d_dout [B,Tout,Cout]
// -> d_dW [Cout][Cin][k], d_db [Cout], d_dX [B,T,Cin]; d_scratch >= nntk_shim_conv1d_grad_scratch_floats(...) floats
extern "C" size_t nntk_shim_conv1d_grad_scratch_floats(int Cin, int Cout, int k) {
    return (size_t)GRAD_SLICES * ((size_t)k * Cin + 1) * Cout;       // also the MFMA form's [slices <= 32][k Cin + 1][Cout]
}
extern "C" int nntk_shim_conv1d_grad(const float *d_in, const float *d_W, const float *d_dout, float *d_dW, float *d_db,
                                     float *d_dX, float *d_scratch, int B, int T, int Cin, int Cout, int k, int stride, int Tout) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || k <= 0) return 0;
    hipStream_t st = nntk_stream();
    float *pw = d_scratch, *pb = d_scratch + (size_t)GRAD_SLICES * k * Cin * Cout;
    const long rows = (long)B * (Tout > 0 ? Tout : 0);
    const int bs = Cout >= 256 ? 256 : ((Cout + 63) / 64) * 64;
    const long tot = (long)k * Cin * Cout + Cout;
    const unsigned rg = (unsigned)((tot + 255) / 256 > 2048 ? 2048 : (tot + 255) / 256);
    // large products: the row-sliced f32-MFMA form (train.hip outer_mfma_kernel) on the im2col VIEW of the input -- row (b, x) is the
    // contiguous window in[b, x * stride .. + k - 1, :], so A = [rows][k Cin] needs no buffer; column sums of dout (d_b) come with it
    const int ncol = k * Cin;
    const int ms = rows > 0 ? nntk_outer_mfma_launch(d_in, d_dout, d_scratch, rows, ncol, Cout, 0, Tout, (long)T * Cin, (long)stride * Cin) : 0;
    if (ms > 0) {
        const size_t sl = (size_t)(ncol + 1) * Cout;
        hipLaunchKernelGGL(conv1d_grad_reduce_kernel, dim3(rg), dim3(256), 0, st, d_scratch, d_scratch + (size_t)ncol * Cout, d_dW, d_db,
                           Cin, Cout, k, ms, sl, sl);
    } else {
        hipLaunchKernelGGL(conv1d_dw_partial_kernel, dim3((unsigned)(k * Cin), GRAD_SLICES), dim3(bs), 0, st, d_in, d_dout, pw,
                           B, T, Cin, Cout, k, stride, Tout > 0 ? Tout : 0);
        hipLaunchKernelGGL(conv1d_db_partial_kernel, dim3(GRAD_SLICES), dim3(bs), 0, st, d_dout, pb, rows, Cout);
        hipLaunchKernelGGL(conv1d_grad_reduce_kernel, dim3(rg), dim3(256), 0, st, pw, pb, d_dW, d_db, Cin, Cout, k, GRAD_SLICES,
                           (size_t)ncol * Cout, (size_t)Cout);
    }
    const long nx = (long)B * T * Cin;
    if (nx > 0 && d_dX)                       // d_dX == NULL: the caller computes d_X itself (MFMA form, train.hip)
        hipLaunchKernelGGL(conv1d_dx_kernel, dim3((unsigned)((nx + 255) / 256 > 8192 ? 8192 : (nx + 255) / 256)), dim3(256), 0, st,
                           d_dout, d_W, d_dX, B, T, Cin, Cout, k, stride, Tout > 0 ? Tout : 0);
    NNTK_LAUNCH_CHECK("conv1d_grad kernels");
    return 0;
}
