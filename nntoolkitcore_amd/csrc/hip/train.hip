// train.hip -- training path, second slice: activation gradients, DenseCalculateGradient, the two
// losses with their derivatives, SGD.  These are small, HBM- or latency-bound VALU kernels; what matters is that they
// follow the reference's OPERATION ORDER (separately rounded multiply and add, sequential sums), because the
// reference's results are the contract:
//   activation gradients   layers/activation_default.c:38-46 (sigmoid), :70-77 (tanh), :94-96 (identity),
//                          :118-121 (ReLU: clamp(z, 0, 1) * d_out -- NOT a step function), :169-185 (softmax Jacobian)
//   Dense gradient         layers/dense.c:164-185 with weights_private.c:43-48 (per-sample d_W = x^T dz, summed over
//                          the mini-batch IN ORDER onto the caller's gradient block; d_X = W dz)
//   losses                 train/loss.c:13-52
//   SGD                    train/optimizers.c:13-19 (buffer = g * lr, then w - buffer: two roundings, no FMA)
#include "nntk_common.hpp"

// hipcc contracts a * b + c into an FMA by default -- also through __fmul_rn / __fadd_rn, whose bodies are compiled
// under the header's contraction mode and carry the `contract` flag into this file when inlined (seen as 1-ulp
// differences in the tanh derivative).  The reference rounds every operation separately: plain operators under
// contract(off).
#pragma clang fp contract(off)
static __device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
static __device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
static __device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }

// ---- activation gradients -------------------------------------------------------------------------------------
// a may be NULL (no cached forward value): the forward function is recomputed from z, as the reference's non-cached
// derivative does (activation_default.c:48-51, :79-82).
__global__ __launch_bounds__(256) void act_grad_kernel(int kind, const float *__restrict__ z, const float *__restrict__ a,
                                                       const float *__restrict__ dout, float *__restrict__ out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = dout[i];
        float r;
        if (kind == NNTK_ACT_SIGMOID) {
            const float s = a ? a[i] : nntk_act(NNTK_ACT_SIGMOID, z[i], 1.f);
            float t = add_rn(-s, 1.0f);          // op_vec_neg, op_vec_add_sc 1
            t = mul_rn(s, t);
            r = mul_rn(t, d);
        } else if (kind == NNTK_ACT_TANH) {
            const float s = a ? a[i] : nntk_act(NNTK_ACT_TANH, z[i], 1.f);
            float t = mul_rn(s, s);
            t = add_rn(-t, 1.0f);
            r = mul_rn(t, d);
        } else if (kind == NNTK_ACT_RELU) {
            r = mul_rn(fmaxf(fminf(z[i], 1.0f), 0.0f), d);      // op_vec_clamp(z, 0, 1) * d_out; the ReLU scale a is not applied
        } else {
            r = d;                                   // identity
        }
        out[i] = r;
    }
}

// softmax: out[vec][j] = sum_i dout[i] * M[i][j],  M[i][j] = i == j ? s_i (1 - s_i) : -1 * s_i * s_j, summed over i in
// order (op_mat_mul 1 x v x v).  `calls` groups of `vpc` vectors: inside one reference call every vector reads d_out at
// the CALL's base, not at its own offset (activation_default.c:183 passes d_out, not d_out + offset) -- kept, so a
// handle created with input_size > 1 behaves like the reference's.
__global__ __launch_bounds__(256) void softmax_grad_kernel(const float *__restrict__ s, const float *__restrict__ dout,
                                                           float *__restrict__ out, long n_vec, int v, int vpc) {
    const long total = n_vec * v;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long vec = e / v;
        const int j = (int)(e % v);
        const float *sv = s + vec * v;
        const float *dv = dout + (vec / vpc) * (long)vpc * v;
        const float sj = sv[j];
        float acc = 0.0f;
        for (int i = 0; i < v; ++i) {
            const float si = sv[i];
            const float m = i == j ? mul_rn(si, add_rn(1.0f, -si)) : mul_rn(mul_rn(-1.0f, si), sj);
            acc = add_rn(acc, mul_rn(dv[i], m));
        }
        out[e] = acc;
    }
}

static unsigned grid_for(long n, int block) {
    long g = (n + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : g > 8192 ? 8192 : g);
}

// n = number of elements (softmax: vectors * vector_size).  d_z or d_a may be NULL (not both).
extern "C" int nntk_shim_activation_grad(int kind, int vector_size, int vectors_per_call, const float *d_z, const float *d_a,
                                         const float *d_dout, float *d_out, long n) {
    if (n <= 0) return 0;
    if (kind == NNTK_ACT_CUSTOM) return nntk_fail_msg("activation gradient: custom host-callback activations run on the host");
    if (kind == NNTK_ACT_NONE) kind = NNTK_ACT_IDENTITY;
    if (kind == NNTK_ACT_SOFTMAX) {
        if (vector_size <= 0 || n % vector_size) return nntk_fail_msg("softmax gradient: size must be whole vectors");
        if (!d_a) return nntk_fail_msg("softmax gradient: the forward output is required");
        hipLaunchKernelGGL(softmax_grad_kernel, dim3(grid_for(n, 256)), dim3(256), 0, nntk_stream(), d_a, d_dout, d_out,
                           n / vector_size, vector_size, vectors_per_call < 1 ? 1 : vectors_per_call);
        NNTK_LAUNCH_CHECK("softmax_grad_kernel");
        return 0;
    }
    if ((kind == NNTK_ACT_RELU && !d_z) || ((kind == NNTK_ACT_SIGMOID || kind == NNTK_ACT_TANH) && !d_z && !d_a))
        return nntk_fail_msg("activation gradient: the cached forward input is required");
    hipLaunchKernelGGL(act_grad_kernel, dim3(grid_for(n, 256)), dim3(256), 0, nntk_stream(), kind, d_z,
                       kind == NNTK_ACT_RELU ? nullptr : d_a, d_dout, d_out, n);
    NNTK_LAUNCH_CHECK("act_grad_kernel");
    return 0;
}

// ---- Dense gradient -------------------------------------------------------------------------------------------
// g_W [in, out] and g_b [out] are ACCUMULATED IN PLACE in mini-batch order, exactly default_gradient_sum's
// ((g + d_W[0]) + d_W[1]) + ...; each per-sample term is the separately rounded product x[b][i] * dz[b][o].
__global__ __launch_bounds__(256) void dense_dw_kernel(const float *__restrict__ x, const float *__restrict__ dz,
                                                       float *__restrict__ gW, float *__restrict__ gb, int B, int in, int out) {
    const long total = (long)in * out;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total + out; e += (long)gridDim.x * blockDim.x) {
        if (e < total) {
            const int i = (int)(e / out), o = (int)(e % out);
            float acc = gW[e];
            for (int b = 0; b < B; ++b) acc = add_rn(mul_rn(x[(long)b * in + i], dz[(long)b * out + o]), acc);
            gW[e] = acc;
        } else {
            const int o = (int)(e - total);
            float acc = gb[o];
            for (int b = 0; b < B; ++b) acc = add_rn(dz[(long)b * out + o], acc);
            gb[o] = acc;
        }
    }
}
// d_X[b][i] = sum_o W[i][o] * dz[b][o] in order (op_mat_mul(W, dz, ., in, 1, out)); overwritten
__global__ __launch_bounds__(256) void dense_dx_kernel(const float *__restrict__ W, const float *__restrict__ dz,
                                                       float *__restrict__ dX, int B, int in, int out) {
    const long total = (long)B * in;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int b = (int)(e / in), i = (int)(e % in);
        const float *w = W + (long)i * out, *d = dz + (long)b * out;
        float acc = 0.0f;
        for (int o = 0; o < out; ++o) acc = add_rn(acc, mul_rn(w[o], d[o]));
        dX[e] = acc;
    }
}
// d_W: caller layout [in, out] row-major (NOT the packed GEMM layout)
extern "C" int nntk_shim_dense_grad(const float *d_x, const float *d_W, const float *d_dz, float *d_gW, float *d_gb,
                                    float *d_dX, int B, int in, int out) {
    if (B <= 0 || in <= 0 || out <= 0) return 0;
    hipLaunchKernelGGL(dense_dw_kernel, dim3(grid_for((long)in * out + out, 256)), dim3(256), 0, nntk_stream(), d_x, d_dz,
                       d_gW, d_gb, B, in, out);
    NNTK_LAUNCH_CHECK("dense_dw_kernel");
    hipLaunchKernelGGL(dense_dx_kernel, dim3(grid_for((long)B * in, 256)), dim3(256), 0, nntk_stream(), d_W, d_dz, d_dX,
                       B, in, out);
    NNTK_LAUNCH_CHECK("dense_dx_kernel");
    return 0;
}

// ---- losses (train/loss.c) ------------------------------------------------------------------------------------
// One thread per sample walks its row IN ORDER (op_vec_sum is a sequential sum), so the per-sample value is the
// reference's bit for bit for MSE; the batch sum is done on the host in order.  kind 0 = MSE, 1 = categorical CE.
__global__ __launch_bounds__(64) void loss_rows_kernel(int kind, const float *__restrict__ y, const float *__restrict__ p,
                                                       float *__restrict__ per_row, int size, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const float *yb = y + (long)b * size, *pb = p + (long)b * size;
    float one = 0.0f;
    if (kind == 0) {
        for (int i = 0; i < size; ++i) {
            const float d = sub_rn(yb[i], pb[i]);
            one = add_rn(one, mul_rn(d, d));
        }
        per_row[b] = one / (float)size;
    } else {
        for (int i = 0; i < size; ++i) one = add_rn(one, mul_rn(logf(pb[i]), yb[i]));
        per_row[b] = -one;
    }
}
// kind 0: (y - p) * (-2 / (size * batch))      (loss.c:26-32)
// kind 1: (y / p) * -1                          (loss.c:47-52; the reference's loop forgets the row offset and only ever
//                                                writes row 0 -- every row is written here, row 0 identically)
__global__ __launch_bounds__(256) void loss_grad_kernel(int kind, const float *__restrict__ y, const float *__restrict__ p,
                                                        float *__restrict__ d, long n, float k) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        d[i] = kind == 0 ? mul_rn(sub_rn(y[i], p[i]), k) : mul_rn(y[i] / p[i], -1.0f);
}
extern "C" int nntk_shim_loss_rows(int kind, const float *d_y, const float *d_pred, float *d_per_row, int size, int batch) {
    if (batch <= 0) return 0;
    hipLaunchKernelGGL(loss_rows_kernel, dim3((batch + 63) / 64), dim3(64), 0, nntk_stream(), kind, d_y, d_pred, d_per_row, size, batch);
    NNTK_LAUNCH_CHECK("loss_rows_kernel");
    return 0;
}
extern "C" int nntk_shim_loss_grad(int kind, const float *d_y, const float *d_pred, float *d_out, int size, int batch) {
    const long n = (long)size * batch;
    if (n <= 0) return 0;
    const float k = -2.0f / (float)(size * batch);
    hipLaunchKernelGGL(loss_grad_kernel, dim3(grid_for(n, 256)), dim3(256), 0, nntk_stream(), kind, d_y, d_pred, d_out, n, k);
    NNTK_LAUNCH_CHECK("loss_grad_kernel");
    return 0;
}

// ---- SGD (train/optimizers.c:13-19) ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(const float *__restrict__ g, float *__restrict__ w, float lr, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        w[i] = sub_rn(w[i], mul_rn(g[i], lr));
}
extern "C" int nntk_shim_sgd(float lr, const float *d_grad, float *d_w, long n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, nntk_stream(), d_grad, d_w, lr, n);
    NNTK_LAUNCH_CHECK("sgd_kernel");
    return 0;
}
