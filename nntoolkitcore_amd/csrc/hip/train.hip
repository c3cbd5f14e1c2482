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

// ---- BatchNorm training (layers/batch_norm.c:191-386) -----------------------------------------------------------
// x, d_out are [N, F] (N = count * mini_batch rows).  Every per-feature quantity is a column sum over N.  The reference
// transposes and sums each column in order; here a column is summed in R-row slices (one lane per feature, coalesced
// across features, rows of a slice in order) and the slices are added in order by the finishing kernel: deterministic,
// and different from the reference's single chain only in where the partial sums are cut (the reference's own NEON /
// vDSP builds cut them differently again).  Elementwise operations keep the reference's order and separate roundings.
//   mode 0: sum x                                    (mean, :203-207)
//   mode 1: sum (x + -mean)^2                        (variance, :210-220)
//   mode 2: sum d_out | sum d_out * x_norm | sum (d_out * gamma) * x_mu      (d_beta, d_gamma, d_ivar, :283-326)
//   mode 3: sum d_x_mu,  d_x_mu = (d_out * gamma) / sqrt_var + (x_mu * 2) * d_var       (d_mu, :353-373)
struct BnParams {
    const float *x, *dout;       // [N, F]
    const float *gamma;          // [F]
    const float *mean, *sqrt_var, *dvar;   // [F] (as each mode needs)
    float *partial;              // [slices][nsums][F]
    long N;
    int F, rows_per_slice;
};
template <int MODE>
__global__ __launch_bounds__(64) void bn_colsum_kernel(BnParams p) {
    const int f = blockIdx.y * 64 + threadIdx.x;
    if (f >= p.F) return;
    const long r0 = (long)blockIdx.x * p.rows_per_slice;
    long r1 = r0 + p.rows_per_slice;
    if (r1 > p.N) r1 = p.N;
    constexpr int NS = MODE == 2 ? 3 : 1;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const float mean = MODE >= 1 ? p.mean[f] : 0.f;
    const float sv = MODE >= 2 ? p.sqrt_var[f] : 1.f;
    const float g = MODE >= 2 ? p.gamma[f] : 1.f;
    const float dvar = MODE == 3 ? p.dvar[f] : 0.f;
    for (long r = r0; r < r1; ++r) {
        const float x = p.x[r * p.F + f];
        if (MODE == 0) {
            s0 = add_rn(s0, x);
        } else if (MODE == 1) {
            const float d = add_rn(x, -mean);
            s0 = add_rn(s0, mul_rn(d, d));
        } else {
            const float d = p.dout[r * p.F + f];
            const float x_mu = sub_rn(x, mean);
            const float dxn = mul_rn(d, g);
            if (MODE == 2) {
                const float x_norm = x_mu / sv;
                s0 = add_rn(s0, d);
                s1 = add_rn(s1, mul_rn(d, x_norm));
                s2 = add_rn(s2, mul_rn(dxn, x_mu));
            } else {
                s0 = add_rn(s0, add_rn(dxn / sv, mul_rn(mul_rn(x_mu, 2.0f), dvar)));
            }
        }
    }
    float *out = p.partial + (size_t)blockIdx.x * NS * p.F;
    out[f] = s0;
    if (MODE == 2) { out[p.F + f] = s1; out[2 * p.F + f] = s2; }
}
// stats block (device, [8][F]): 0 mean | 1 variance | 2 var_eps | 3 sqrt_var | 4 d_beta | 5 d_gamma | 6 d_var | 7 d_mu
// step 0: mean = sum / N            step 1: variance = sum / N, var_eps = variance + eps, sqrt_var = sqrtf(var_eps)
// step 2: d_beta, d_gamma, d_var = (((d_ivar * -1) / var_eps) / sqrt_var) / 2 / N        step 3: d_mu = sum * (-1 / N)
__global__ __launch_bounds__(256) void bn_finish_kernel(int step, const float *partial, int slices, float *stats, int F, long N, float eps) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int ns = step == 2 ? 3 : 1;
    float s[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < slices; ++k)
        for (int q = 0; q < ns; ++q) s[q] = add_rn(s[q], partial[((size_t)k * ns + q) * F + f]);
    const float n = (float)N;
    if (step == 0) {
        stats[f] = s[0] / n;
    } else if (step == 1) {
        const float var = s[0] / n;
        const float ve = add_rn(var, eps);
        stats[F + f] = var; stats[2 * F + f] = ve; stats[3 * F + f] = sqrtf(ve);
    } else if (step == 2) {
        stats[4 * F + f] = s[0];
        stats[5 * F + f] = s[1];
        float d = mul_rn(s[2], -1.0f) / stats[2 * F + f];      // d_sqrt_var
        d = d / stats[3 * F + f];                              // d_var
        d = d / 2.0f;
        stats[6 * F + f] = d / n;
    } else {
        stats[7 * F + f] = mul_rn(s[0], -1.0f / n);
    }
}
// forward: out = ((x - mean) / sqrt_var) * gamma + beta       backward: d_x = ((d_out * gamma) / sqrt_var + (x_mu * 2) * d_var) + d_mu
__global__ __launch_bounds__(256) void bn_train_apply_kernel(int backward, const float *__restrict__ x, const float *__restrict__ dout,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             const float *__restrict__ stats, float *__restrict__ out, long n, int F) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i % F);
        const float x_mu = sub_rn(x[i], stats[f]);
        const float sv = stats[3 * F + f];
        if (!backward) {
            out[i] = add_rn(mul_rn(x_mu / sv, gamma[f]), beta[f]);
        } else {
            const float dxn = mul_rn(dout[i], gamma[f]);
            out[i] = add_rn(add_rn(dxn / sv, mul_rn(mul_rn(x_mu, 2.0f), stats[6 * F + f])), stats[7 * F + f]);
        }
    }
}

extern "C" int nntk_shim_bn_train_slices(long N, int *rows_per_slice) {
    long r = (N + 1023) / 1024;
    if (r < 32) r = 32;
    *rows_per_slice = (int)r;
    return (int)((N + r - 1) / r);
}
static int bn_colsum(int mode, const BnParams &p, int slices) {
    dim3 grid((unsigned)slices, (unsigned)((p.F + 63) / 64));
    switch (mode) {
    case 0: hipLaunchKernelGGL(bn_colsum_kernel<0>, grid, dim3(64), 0, nntk_stream(), p); break;
    case 1: hipLaunchKernelGGL(bn_colsum_kernel<1>, grid, dim3(64), 0, nntk_stream(), p); break;
    case 2: hipLaunchKernelGGL(bn_colsum_kernel<2>, grid, dim3(64), 0, nntk_stream(), p); break;
    default: hipLaunchKernelGGL(bn_colsum_kernel<3>, grid, dim3(64), 0, nntk_stream(), p); break;
    }
    NNTK_LAUNCH_CHECK("bn_colsum_kernel");
    return 0;
}
// d_block: gamma | beta | ... (the BatchNorm device block); d_stats [8][F]; d_partial [slices][3][F]
extern "C" int nntk_shim_bn_train_forward(const float *d_x, const float *d_block, float eps, float *d_stats, float *d_partial,
                                          float *d_out, long N, int F) {
    if (N <= 0 || F <= 0) return 0;
    BnParams p{};
    p.x = d_x; p.gamma = d_block; p.mean = d_stats; p.sqrt_var = d_stats + 3 * (size_t)F; p.partial = d_partial; p.N = N; p.F = F;
    const int slices = nntk_shim_bn_train_slices(N, &p.rows_per_slice);
    for (int step = 0; step < 2; ++step) {
        if (bn_colsum(step, p, slices)) return -1;
        hipLaunchKernelGGL(bn_finish_kernel, dim3((F + 255) / 256), dim3(256), 0, nntk_stream(), step, d_partial, slices, d_stats, F, N, eps);
        NNTK_LAUNCH_CHECK("bn_finish_kernel");
    }
    hipLaunchKernelGGL(bn_train_apply_kernel, dim3(grid_for(N * F, 256)), dim3(256), 0, nntk_stream(), 0, d_x, (const float *)nullptr,
                       d_block, d_block + F, d_stats, d_out, N * F, F);
    NNTK_LAUNCH_CHECK("bn_train_apply_kernel");
    return 0;
}
extern "C" int nntk_shim_bn_train_backward(const float *d_x, const float *d_dout, const float *d_block, float *d_stats,
                                           float *d_partial, float *d_dx, long N, int F) {
    if (N <= 0 || F <= 0) return 0;
    BnParams p{};
    p.x = d_x; p.dout = d_dout; p.gamma = d_block; p.mean = d_stats; p.sqrt_var = d_stats + 3 * (size_t)F;
    p.dvar = d_stats + 6 * (size_t)F; p.partial = d_partial; p.N = N; p.F = F;
    const int slices = nntk_shim_bn_train_slices(N, &p.rows_per_slice);
    for (int step = 2; step < 4; ++step) {
        if (bn_colsum(step, p, slices)) return -1;
        hipLaunchKernelGGL(bn_finish_kernel, dim3((F + 255) / 256), dim3(256), 0, nntk_stream(), step, d_partial, slices, d_stats, F, N, 0.f);
        NNTK_LAUNCH_CHECK("bn_finish_kernel");
    }
    hipLaunchKernelGGL(bn_train_apply_kernel, dim3(grid_for(N * F, 256)), dim3(256), 0, nntk_stream(), 1, d_x, d_dout,
                       d_block, d_block + F, d_stats, d_dx, N * F, F);
    NNTK_LAUNCH_CHECK("bn_train_apply_kernel");
    return 0;
}

// ---- GRU training (layers/gru.c:246-512) ------------------------------------------------------------------------
// Correct, deterministic, reference operation order; NOT tuned (one launch per timestep and direction, VALU dots).
// Forward step t for the whole mini-batch: thread (b, j) computes GRUCellForward's three columns j, H + j, 2H + j
// (gru.c:128-187; op_mat_mul as sums of separately rounded products, in eight k-ordered chunks -- cell_dots_chunked) and stores what the backward pass
// reads: Z_gates [B][T][6H] = Z_z | Z_r | Z_h~ | z | r | h~, h_pr_Uh [B][T][H] (= h_prev U_h + b_hh), h [B][T][H].
// The per-timestep forward dots, cut the same way as rows_times_colmat_kernel below: a workgroup is one batch row x 32
// hidden units x 8 K-chunks of the concatenated [x_t | h_{t-1}] walk; chunk sums are added in chunk order through LDS.
// Returns (in the threads with chunk index 0) xw[g] = x_t . W[:, g H + j] and hu[g] = h_{t-1} . U[:, g H + j].
#define CELL_CHUNKS 8
template <int G>
__device__ __forceinline__ bool cell_dots_chunked(const float *__restrict__ x, const float *__restrict__ hp, const float *__restrict__ W,
                                                  const float *__restrict__ U, int in, int H, int j, float (&xw)[G], float (&hu)[G]) {
    __shared__ float part[CELL_CHUNKS][2 * G][32];
    const int il = threadIdx.x & 31, c = threadIdx.x >> 5;
    const int Kt = in + (hp ? H : 0);
    const int per = (Kt + CELL_CHUNKS - 1) / CELL_CHUNKS;
    const int k0 = c * per, k1 = k0 + per < Kt ? k0 + per : Kt;
    const int GH = G * H;
#pragma unroll
    for (int g = 0; g < G; ++g) { xw[g] = 0.f; hu[g] = 0.f; }
    if (j < H) {
        for (int k = k0; k < k1 && k < in; ++k) {
            const float xv = x[k];
            const float *w = W + (size_t)k * GH + j;
#pragma unroll
            for (int g = 0; g < G; ++g) xw[g] = add_rn(xw[g], mul_rn(xv, w[g * H]));
        }
        for (int k = (k0 > in ? k0 : in); k < k1; ++k) {
            const float hv = hp[k - in];
            const float *u = U + (size_t)(k - in) * GH + j;
#pragma unroll
            for (int g = 0; g < G; ++g) hu[g] = add_rn(hu[g], mul_rn(hv, u[g * H]));
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) { part[c][g][il] = xw[g]; part[c][G + g][il] = hu[g]; }
    __syncthreads();
    if (c != 0 || j >= H) return false;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float a = part[0][g][il], b = part[0][G + g][il];
#pragma unroll
        for (int q = 1; q < CELL_CHUNKS; ++q) { a = add_rn(a, part[q][g][il]); b = add_rn(b, part[q][G + g][il]); }
        xw[g] = a; hu[g] = b;
    }
    return true;
}

struct GruTrainParams {
    const float *x;              // [B][T][in]
    const float *W, *U, *bi, *bh;    // caller layouts: W [in][3H], U [H][3H]
    float *h, *Zg, *hU;          // caches
    int B, T, in, H, t;
    int act_z, act_h, act_r;
    float sc_z, sc_h, sc_r;      // ReLU output scales
};
__global__ __launch_bounds__(256) void gru_train_fwd_step_kernel(GruTrainParams p) {
    const int b = blockIdx.y, j = blockIdx.x * 32 + (threadIdx.x & 31), H = p.H;
    const size_t row = (size_t)b * p.T + p.t;
    const float *hp = p.t > 0 ? p.h + (row - 1) * H : nullptr;      // h_0 = 0 for every sequence (gru.c:262)
    float xw[3], hu[3];
    if (!cell_dots_chunked<3>(p.x + row * p.in, hp, p.W, p.U, p.in, H, j, xw, hu)) return;
#pragma unroll
    for (int g = 0; g < 3; ++g) { xw[g] = add_rn(xw[g], p.bi[g * H + j]); hu[g] = add_rn(hu[g], p.bh[g * H + j]); }
    const float Zz = add_rn(xw[0], hu[0]), Zr = add_rn(xw[1], hu[1]);
    const float z = nntk_gate_act(p.act_z, Zz, p.sc_z), r = nntk_gate_act(p.act_r, Zr, p.sc_r);
    const float Zh = add_rn(mul_rn(r, hu[2]), xw[2]);
    const float ht = nntk_gate_act(p.act_h, Zh, p.sc_h);
    const float hprev = hp ? hp[j] : 0.0f;
    const float hn = add_rn(mul_rn(add_rn(-z, 1.0f), ht), mul_rn(z, hprev));
    float *Zg = p.Zg + row * 6 * H;
    Zg[j] = Zz; Zg[H + j] = Zr; Zg[2 * H + j] = Zh; Zg[3 * H + j] = z; Zg[4 * H + j] = r; Zg[5 * H + j] = ht;
    p.hU[row * H + j] = hu[2];
    p.h[row * H + j] = hn;
}

__device__ __forceinline__ float gate_grad(int kind, float Z, float a, float d) {     // activation_default.c derivatives
    if (kind == NNTK_ACT_SIGMOID) return mul_rn(mul_rn(a, add_rn(-a, 1.0f)), d);
    if (kind == NNTK_ACT_TANH) return mul_rn(add_rn(-mul_rn(a, a), 1.0f), d);
    if (kind == NNTK_ACT_RELU) return mul_rn(fmaxf(fminf(Z, 1.0f), 0.0f), d);
    return d;
}
// Backward step t, elementwise part of GRUCellBackward (gru.c:314-430) for thread (b, j):
//   d_h = (carry or 0) + d_out_t;  d_h_prev_1 = z d_h;  d_h~ = (-z) d_h + d_h;  d_z = (h_prev - h~) d_h;
//   d_Zh = act_h'(.) d_h~;  d_r = h_pr_Uh d_Zh;  d_Zz = act_z'(.) d_z;  d_Zr = act_r'(.) d_r
// writes d_xW [B][T][3H] = d_Zz | d_Zr | d_Zh, d_hU [B][T][3H] = d_Zz | d_Zr | r d_Zh, its copy for this step
// d_hU_step [B][3H], and d_h_prev_1 [B][H].  carry = d_h_prev_1 + d_h_prev_2 of step t + 1 (gru.c:426).
struct GruBwdParams {
    const float *dout;           // [B][T][H] or [B][H]
    const float *h, *Zg, *hU;    // forward caches
    const float *dhp1, *dhp2;    // [B][H] of step t + 1
    float *dxW, *dhU, *dhU_step, *dhp1_out;
    int B, T, H, t, return_sequences;
    int act_z, act_h, act_r;
};
__global__ __launch_bounds__(256) void gru_train_bwd_step_kernel(GruBwdParams p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.B * p.H) return;
    const int b = e / p.H, j = e % p.H, H = p.H;
    const size_t row = (size_t)b * p.T + p.t;
    float dout = 0.0f;
    if (p.return_sequences) dout = p.dout[row * H + j];
    else if (p.t == p.T - 1) dout = p.dout[(size_t)b * H + j];
    const float carry = p.t == p.T - 1 ? 0.0f : add_rn(p.dhp1[e], p.dhp2[e]);
    const float dh = add_rn(carry, dout);
    const float *Zg = p.Zg + row * 6 * H;
    const float z = Zg[3 * H + j], r = Zg[4 * H + j], ht = Zg[5 * H + j];
    const float dhp1 = mul_rn(z, dh);
    const float dht = add_rn(mul_rn(-z, dh), dh);
    const float dz = mul_rn(p.t > 0 ? sub_rn(p.h[(row - 1) * H + j], ht) : -ht, dh);
    const float dZh = gate_grad(p.act_h, Zg[2 * H + j], ht, dht);
    const float dr = mul_rn(p.hU[row * H + j], dZh);
    const float dZz = gate_grad(p.act_z, Zg[j], z, dz);
    const float dZr = gate_grad(p.act_r, Zg[H + j], r, dr);
    const float dhUh = mul_rn(r, dZh);
    float *dxW = p.dxW + row * 3 * H, *dhU = p.dhU + row * 3 * H, *ds = p.dhU_step + (size_t)b * 3 * H;
    dxW[j] = dZz; dxW[H + j] = dZr; dxW[2 * H + j] = dZh;
    dhU[j] = dZz; dhU[H + j] = dZr; dhU[2 * H + j] = dhUh;
    ds[j] = dZz; ds[H + j] = dZr; ds[2 * H + j] = dhUh;
    p.dhp1_out[e] = dhp1;
}
// out[row][i] = sum_k M[i][k] * d[row][k] in k order (op_mat_mul(M, d, ., I, 1, K)): d_h_prev_2 = U d_hU per step, and
// d_X = W d_xW for all rows at the end
__global__ __launch_bounds__(256) void rows_times_rowmat_kernel(const float *__restrict__ d, const float *__restrict__ M,
                                                                float *__restrict__ out, long rows, int I, int K) {
    const long total = rows * I;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / I;
        const int i = (int)(e % I);
        const float *m = M + (size_t)i * K, *dv = d + row * K;
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc = add_rn(acc, mul_rn(m[k], dv[k]));
        out[e] = acc;
    }
}
// the same product with the matrix given TRANSPOSED (MT [K][I]) so that lanes i read consecutive floats; used inside the
// per-timestep loops (U^T is built once per gradient call).
// One workgroup = one row x 32 outputs x 8 K-chunks: at mini-batch 64 a (row, i) thread grid is only 16-32 k threads, each
// walking all of K -- latency-bound.  The K range is cut into 8 contiguous chunks summed in order inside a chunk and then
// chunk 0 .. 7 in order through LDS: deterministic, eight times the parallelism (LSTM-512 backward step 250 -> ~60 us);
// the sum is no longer the reference's single chain (the difference is a few f32 roundings; small shapes and every other
// product keep the exact order).
#define COLMAT_CHUNKS 8
__global__ __launch_bounds__(256) void rows_times_colmat_kernel(const float *__restrict__ d, const float *__restrict__ MT,
                                                                float *__restrict__ out, long rows, int I, int K) {
    __shared__ float part[COLMAT_CHUNKS][32];
    const int il = threadIdx.x & 31, c = threadIdx.x >> 5;
    const long row = blockIdx.y;
    const int i = blockIdx.x * 32 + il;
    const int per = (K + COLMAT_CHUNKS - 1) / COLMAT_CHUNKS;
    const int k0 = c * per, k1 = k0 + per < K ? k0 + per : K;
    float acc = 0.0f;
    if (i < I) {
        const float *dv = d + row * K;
        for (int k = k0; k < k1; ++k) acc = add_rn(acc, mul_rn(MT[(size_t)k * I + i], dv[k]));
    }
    part[c][il] = acc;
    __syncthreads();
    if (c == 0 && i < I) {
        float s = part[0][il];
#pragma unroll
        for (int q = 1; q < COLMAT_CHUNKS; ++q) s = add_rn(s, part[q][il]);
        out[row * I + i] = s;
    }
}
// The same product on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32) for the per-timestep BPTT step d_h_prev = dgates U^T once
// the shapes are tile-sized (rows % 16 == 0, I % 16 == 0, K % 128 == 0: LSTM-512 / GRU-256 at mini-batch 64).  The VALU form
// above re-reads the whole U^T for every batch row (64 x 4 MB out of L2 per step at LSTM-512: 55 us of a 64-us step); here a
// workgroup owns one 16 x 16 output tile, its 8 wavefronts split K (deterministic: chunk sums added in order through LDS), and
// U^T is read once per 16 rows.  An f32 MFMA is a k-ordered fmaf chain, so the result differs from the VALU form only in
// summation order (fused multiply-adds, other chunk boundaries): same tolerance, not the same bits.
// A operand: lane (m, q) reads d[row0 + m][kb + 4 q .. + 3] (one 16-byte load per 16-deep block) and feeds component c to
// MFMA c, which therefore multiplies k = kb + 4 q + c (q = 0..3); B operand: MT[(kb + 4 q + c) * I + i0 + n].
__global__ __launch_bounds__(512) void rows_times_colmat_mfma_kernel(const float *__restrict__ d, const float *__restrict__ MT,
                                                                     float *__restrict__ out, int I, int K) {
    __shared__ float part[8][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int i0 = blockIdx.x * 16;
    const long row0 = (long)blockIdx.y * 16;
    const int per = K / 8, k0 = wv * per;
    const float *dp = d + (row0 + m) * K + k0 + 4 * q;
    const float *mp = MT + (size_t)(k0 + 4 * q) * I + i0 + m;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int kb = 0; kb < per; kb += 16) {
        const float4 a = *reinterpret_cast<const float4 *>(dp + kb);
        const float b0 = mp[(size_t)(kb + 0) * I], b1 = mp[(size_t)(kb + 1) * I], b2 = mp[(size_t)(kb + 2) * I], b3 = mp[(size_t)(kb + 3) * I];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b3, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wv][r][lane] = acc[r];
    __syncthreads();
    if (wv == 0) {
        // D layout: lane (n = lane & 15, q) holds rows 4 q + r of the tile, column n
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sum = part[0][r][lane];
#pragma unroll
            for (int c = 1; c < 8; ++c) sum += part[c][r][lane];
            out[(row0 + 4 * q + r) * I + i0 + m] = sum;
        }
    }
}
static void launch_rows_times_colmat(const float *d, const float *MT, float *out, long rows, int I, int K) {
    if (rows % 16 == 0 && I % 16 == 0 && K % 128 == 0 && rows / 16 <= 65535)
        hipLaunchKernelGGL(rows_times_colmat_mfma_kernel, dim3((unsigned)(I / 16), (unsigned)(rows / 16)), dim3(512), 0, nntk_stream(), d, MT, out, I, K);
    else
        hipLaunchKernelGGL(rows_times_colmat_kernel, dim3((unsigned)((I + 31) / 32), (unsigned)rows), dim3(256), 0, nntk_stream(), d, MT, out, rows, I, K);
}
// C[i][k] += sum_rows A[row][i] * Bm[row][k] and c[k] += sum_rows Bm[row][k]: row slices summed in order inside a slice,
// slices added in order onto C (the reference adds each (b, t) term onto the gradient block one by one, gru.c:508)
#define OUTER_SLICES 32
__global__ __launch_bounds__(256) void outer_partial_kernel(const float *__restrict__ A, const float *__restrict__ Bm,
                                                            float *__restrict__ partial, long rows, int I, int K, int a_shift_T) {
    // block (i, slice); thread k.  a_shift_T > 0: A is h [B][T][I] and row (b, t) uses h_{t-1} (zero at t = 0)
    const int i = blockIdx.x;
    const long r0 = rows * blockIdx.y / gridDim.y, r1 = rows * (blockIdx.y + 1) / gridDim.y;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float acc = 0.0f;
        for (long r = r0; r < r1; ++r) {
            float a;
            if (i == I) a = 1.0f;                                   // bias row: plain column sum
            else if (a_shift_T > 0) a = (r % a_shift_T) ? A[(r - 1) * I + i] : 0.0f;
            else a = A[r * I + i];
            acc = i == I ? add_rn(acc, Bm[r * K + k]) : add_rn(acc, mul_rn(a, Bm[r * K + k]));
        }
        partial[((size_t)blockIdx.y * (I + 1) + i) * K + k] = acc;
    }
}
__global__ __launch_bounds__(256) void outer_reduce_kernel(const float *__restrict__ partial, float *__restrict__ C, float *__restrict__ c,
                                                           int I, int K, int slices) {
    const long total = (long)(I + 1) * K;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        float *dst = e < (long)I * K ? C + e : c + (e - (long)I * K);
        float s = *dst;
        for (int sl = 0; sl < slices; ++sl) s = add_rn(s, partial[(size_t)sl * (I + 1) * K + e]);
        *dst = s;
    }
}

// the bias row alone (I == 0: column sums of Bm): outer_partial_kernel would run it on OUTER_SLICES workgroups walking all of K
// (1.25 ms for LSTM-512's [12800, 2048] dgates); here K is spread over the grid too.  Same slices, same order, same bits.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ Bm, float *__restrict__ partial, long rows, int K,
                                                             size_t slice_stride) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const long r0 = rows * blockIdx.y / gridDim.y, r1 = rows * (blockIdx.y + 1) / gridDim.y;
    float acc = 0.0f;
    for (long r = r0; r < r1; ++r) acc = add_rn(acc, Bm[r * K + k]);
    partial[(size_t)blockIdx.y * slice_stride + k] = acc;
}

// The weight-gradient product C[i][k] += sum_rows A[row][i] Bm[row][k] on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32) once it is
// large: d_W / d_U of a recurrent layer ([128 | 512] x 2048 over 12800 rows) and Dense's d_W ([512] x 1000 over 63744 rows).
// The output is small and the contraction (the ROWS) is long, so an output-tiled GEMM has 8-32 workgroups walking thousands of
// k-steps each (1.1 ms per product at LSTM-512, 5.5 ms at Dense-1000).  Here the rows are cut into `slices` (grid z) and every
// (128 x 128 tile, slice) is one workgroup: hundreds of workgroups, each a k-ordered fmaf chain over its slice; the slices are
// then added in order by outer_reduce_kernel (same structure as the VALU form above, 32x32x2 blocks instead of single dots).
// The waves of i-tile 0 also keep the column sums of Bm (the bias gradient, row I of every slice) from the B values they load.
// No operand is transposed or staged: for this MFMA shape lane l supplies A[k = l / 32][m = l % 32] and B[k = l / 32][n = l % 32],
// i.e. 32 consecutive floats of one source row -- both operands load straight from their row-major tensors, coalesced.
//   wave (wi, wk) of a workgroup owns the 64 x 64 sub-tile at (i0 + 64 wi, k0 + 64 wk): 2 x 2 MFMAs per pair of rows
//   a_shift_T > 0: A is h [B][T][I] and row (b, t) uses h_{t-1} (zero at t = 0), as in outer_partial_kernel
#define OUTER_TILE 128
//   rps / seq_pitch / row_pitch: A's row r starts at A + (r / rps) * seq_pitch + (r % rps) * row_pitch -- a plain matrix is rps = rows,
//   row_pitch = I; a channels-last convolution's im2col matrix (row (b, x) = the CONTIGUOUS window in[b, x * stride .. + k - 1, :],
//   I = k * Cin) is rps = Tout, seq_pitch = T * Cin, row_pitch = stride * Cin: Conv1d's d_W needs no im2col buffer either
__global__ __launch_bounds__(256) void outer_mfma_kernel(const float *__restrict__ A, const float *__restrict__ Bm, float *__restrict__ partial,
                                                         long rows, int I, int K, int a_shift_T, long rps, long seq_pitch, long row_pitch) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane & 31, kk = lane >> 5;
    const int i0 = blockIdx.y * OUTER_TILE + (wv >> 1) * 64, k0 = blockIdx.x * OUTER_TILE + (wv & 1) * 64;
    if (i0 >= I || k0 >= K) return;                          // no barrier below: a wave outside the matrix just leaves
    const long r0 = rows * blockIdx.z / gridDim.z, r1 = rows * (blockIdx.z + 1) / gridDim.z;
    const bool iv0 = i0 + c < I, iv1 = i0 + 32 + c < I, kv0 = k0 + c < K, kv1 = k0 + 32 + c < K;
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[x][y][v] = 0.f;
    float bs0 = 0.f, bs1 = 0.f;
    long tt = a_shift_T > 0 ? (r0 + kk) % a_shift_T : 1;    // t of this lane's row (only "is it 0" matters)
    const long ra = r0 + kk - (a_shift_T > 0 ? 1 : 0);      // (a_shift_T only with plain matrices: rps = rows, so ra = -1 stays in "sequence" 0)
    long ax = ra >= 0 ? ra % rps : ra;                       // position inside the sequence; the row's address follows it
    const float *ap = A + (ra >= 0 ? ra / rps : 0) * seq_pitch + ax * row_pitch + i0 + c;
    const long seq_step = seq_pitch - rps * row_pitch;      // pointer correction when a row pair crosses into the next sequence
    const float *bp = Bm + (r0 + kk) * (long)K + k0 + c;
    // four row pairs per trip (a constant inner trip count: a loop holding MFMAs is only unrolled without a remainder loop);
    // pairs past the slice's end load zeros
    for (long rb = r0 + kk; rb < r1 + kk; rb += 8) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool rv = rb + 2 * u < r1, av = rv && tt != 0;
            const float a0 = av && iv0 ? ap[0] : 0.f, a1 = av && iv1 ? ap[32] : 0.f;
            const float b0 = rv && kv0 ? bp[0] : 0.f, b1 = rv && kv1 ? bp[32] : 0.f;
            bs0 += b0; bs1 += b1;                                // column sums of Bm (the bias gradient), kept by the i-tile-0 waves
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            ap += 2 * row_pitch; bp += 2 * (long)K;
            ax += 2;
            if (ax >= rps) { ax -= rps; ap += seq_step; if (ax >= rps) { ax -= rps; ap += seq_step; } }       // rps = 1 needs both
            if (a_shift_T > 0) { tt += 2; if (tt >= a_shift_T) tt -= a_shift_T; if (tt >= a_shift_T) tt -= a_shift_T; }     // T = 1 needs both
        }
    }
    // D layout: lane (n = c, kk) holds rows 8 (v / 4) + 4 kk + v % 4 of each 32 x 32 block, column n
    float *dst = partial + (size_t)blockIdx.z * (I + 1) * K;
    if (blockIdx.y == 0 && (wv >> 1) == 0) {                 // row I of the slice: even rows + odd rows of the slice
        bs0 += __shfl_xor(bs0, 32, 64); bs1 += __shfl_xor(bs1, 32, 64);
        if (kk == 0) {
            if (kv0) dst[(size_t)I * K + k0 + c] = bs0;
            if (kv1) dst[(size_t)I * K + k0 + 32 + c] = bs1;
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int k = k0 + 32 * y + c;
            if (k >= K) continue;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = i0 + 32 * x + 8 * (v >> 2) + 4 * kk + (v & 3);
                if (i < I) dst[(size_t)i * K + k] = acc[x][y][v];
            }
        }
}

// The same product for PLAIN row-major operands (rps = rows), with (almost) no vector-ALU instruction in the loop: an f32 MFMA and a
// VALU instruction never overlap on gfx950, and the general kernel above spends ~25 VALU per row pair on predicates and 64-bit
// pointer arithmetic (0.39 of the MFMA pipe at LSTM-512's d_U).  Here every load is a buffer load whose per-lane offset never
// changes: column masks are baked into the offsets (out-of-range = 0), the row advance moves the descriptor's base and shrinks its
// range on the scalar unit, so rows past the slice's end fall out of range by themselves.  SHIFT (A = h_prev: row (b, t) reads
// h_{t-1}, zero at t = 0) costs a select per operand and a counter update per row pair.
template <bool SHIFT>
__global__ __launch_bounds__(256) void outer_mfma_plain_kernel(const float *__restrict__ A, const float *__restrict__ Bm, float *__restrict__ partial,
                                                               long rows, int I, int K, int T) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane & 31, kk = lane >> 5;
    const int i0 = blockIdx.y * OUTER_TILE + (wv >> 1) * 64, k0 = blockIdx.x * OUTER_TILE + (wv & 1) * 64;
    if (i0 >= I || k0 >= K) return;
    const long r0 = rows * blockIdx.z / gridDim.z, r1 = rows * (blockIdx.z + 1) / gridDim.z;
    const int OOB = 0x7ffffff0;
    const int va0 = i0 + c < I ? (kk * I + i0 + c) * 4 : OOB, va1 = i0 + 32 + c < I ? (kk * I + i0 + 32 + c) * 4 : OOB;
    const int vb0 = k0 + c < K ? (kk * K + k0 + c) * 4 : OOB, vb1 = k0 + 32 + c < K ? (kk * K + k0 + 32 + c) * 4 : OOB;
    const bool do_bs = blockIdx.y == 0 && (wv >> 1) == 0;
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[x][y][v] = 0.f;
    float bs0 = 0.f, bs1 = 0.f;
    int tt = SHIFT ? (int)((r0 + kk) % T) : 1;
    // bases of the slice (A one row earlier with SHIFT: that row is only read where t > 0); the ranges end with the slice
    const char *abase = reinterpret_cast<const char *>(A + (r0 - (SHIFT ? 1 : 0)) * (long)I);
    const char *bbase = reinterpret_cast<const char *>(Bm + r0 * (long)K);
    long aleft = (r1 - r0) * (long)I * 4, bleft = (r1 - r0) * (long)K * 4;
    const long astep = 2L * I * 4, bstep = 2L * K * 4;
    for (long rb = r0; rb < r1; rb += 8) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)abase, 0, (int)(aleft > 0 ? (aleft < 0x7fffffffL ? aleft : 0x7fffffffL) : 0), 0x00020000);
            const __amdgpu_buffer_rsrc_t rbd = __builtin_amdgcn_make_buffer_rsrc((void *)bbase, 0, (int)(bleft > 0 ? (bleft < 0x7fffffffL ? bleft : 0x7fffffffL) : 0), 0x00020000);
            const bool live = !SHIFT || tt != 0;
            const float a0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, live ? va0 : OOB, 0, 0));
            const float a1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, live ? va1 : OOB, 0, 0));
            const float b0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rbd, vb0, 0, 0));
            const float b1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rbd, vb1, 0, 0));
            if (do_bs) { bs0 += b0; bs1 += b1; }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            abase += astep; bbase += bstep; aleft -= astep; bleft -= bstep;
            if (SHIFT) { tt += 2; if (tt >= T) tt -= T; if (tt >= T) tt -= T; }     // T = 1 needs both
        }
    }
    float *dst = partial + (size_t)blockIdx.z * (I + 1) * K;
    if (do_bs) {
        bs0 += __shfl_xor(bs0, 32, 64); bs1 += __shfl_xor(bs1, 32, 64);
        if (kk == 0) {
            if (k0 + c < K) dst[(size_t)I * K + k0 + c] = bs0;
            if (k0 + 32 + c < K) dst[(size_t)I * K + k0 + 32 + c] = bs1;
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int k = k0 + 32 * y + c;
            if (k >= K) continue;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = i0 + 32 * x + 8 * (v >> 2) + 4 * kk + (v & 3);
                if (i < I) dst[(size_t)i * K + k] = acc[x][y][v];
            }
        }
}

extern "C" size_t nntk_shim_outer_scratch_floats(int I, int K) { return (size_t)OUTER_SLICES * (I + 1) * K; }
// partial [slices][I + 1][K] (row I: column sums of Bm); returns the number of slices written, 0 when the shape is not taken
int nntk_outer_mfma_launch(const float *d_A, const float *d_B, float *d_partial, long rows, int I, int K, int a_shift_T,
                           long rps, long seq_pitch, long row_pitch) {
    if (!(I >= 32 && K >= 32 && rps >= 1 && (double)rows * I * K >= (double)(1 << 27))) return 0;
    // enough (tile, slice) workgroups for three waves per SIMD, at least 64 rows per slice
    const int ti = (I + OUTER_TILE - 1) / OUTER_TILE, tk = (K + OUTER_TILE - 1) / OUTER_TILE;
    int slices = (768 + ti * tk - 1) / (ti * tk);
    if (slices > OUTER_SLICES) slices = OUTER_SLICES;
    if ((long)slices > rows / 64) slices = (int)(rows / 64);
    if (slices < 1) slices = 1;
    const bool plain = rps >= rows && row_pitch == I && (double)(rows / slices + 2) * (I > K ? I : K) * 4 < 2.0e9 && nntk_options().train_outer_plain != 0;
    if (plain && a_shift_T > 0)
        hipLaunchKernelGGL(outer_mfma_plain_kernel<true>, dim3((unsigned)tk, (unsigned)ti, (unsigned)slices), dim3(256), 0, nntk_stream(),
                           d_A, d_B, d_partial, rows, I, K, a_shift_T);
    else if (plain)
        hipLaunchKernelGGL(outer_mfma_plain_kernel<false>, dim3((unsigned)tk, (unsigned)ti, (unsigned)slices), dim3(256), 0, nntk_stream(),
                           d_A, d_B, d_partial, rows, I, K, 1);
    else
        hipLaunchKernelGGL(outer_mfma_kernel, dim3((unsigned)tk, (unsigned)ti, (unsigned)slices), dim3(256), 0, nntk_stream(),
                           d_A, d_B, d_partial, rows, I, K, a_shift_T, rps, seq_pitch, row_pitch);
    return slices;
}
extern "C" int nntk_shim_outer_accumulate(const float *d_A, const float *d_B, float *d_C, float *d_c, float *d_scratch,
                                          long rows, int I, int K, int a_shift_T) {
    if (rows <= 0 || I < 0 || K <= 0) return 0;             // I == 0: only the column sums c
    int slices = OUTER_SLICES;
    const int ms = I > 0 ? nntk_outer_mfma_launch(d_A, d_B, d_scratch, rows, I, K, a_shift_T, rows, 0, I) : 0;
    if (ms > 0) {
        slices = ms;
        NNTK_LAUNCH_CHECK("outer_mfma_kernel");
    } else if (I == 0)
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((K + 255) / 256), OUTER_SLICES), dim3(256), 0, nntk_stream(), d_B, d_scratch, rows, K,
                           (size_t)K);
    else
        hipLaunchKernelGGL(outer_partial_kernel, dim3((unsigned)(I + 1), OUTER_SLICES), dim3(K >= 256 ? 256 : ((K + 63) / 64) * 64), 0,
                           nntk_stream(), d_A, d_B, d_scratch, rows, I, K, a_shift_T);
    NNTK_LAUNCH_CHECK("outer_partial_kernel");
    hipLaunchKernelGGL(outer_reduce_kernel, dim3(grid_for((long)(I + 1) * K, 256)), dim3(256), 0, nntk_stream(), d_scratch, d_C, d_c,
                       I, K, slices);
    NNTK_LAUNCH_CHECK("outer_reduce_kernel");
    return 0;
}
extern "C" int nntk_shim_rows_times_rowmat(const float *d_d, const float *d_M, float *d_out, long rows, int I, int K) {
    if (rows <= 0 || I <= 0) return 0;
    hipLaunchKernelGGL(rows_times_rowmat_kernel, dim3(grid_for(rows * I, 256)), dim3(256), 0, nntk_stream(), d_d, d_M, d_out, rows, I, K);
    NNTK_LAUNCH_CHECK("rows_times_rowmat_kernel");
    return 0;
}
// ---- persistent BPTT (GRU and LSTM) ------------------------------------------------------------------------------------
// The per-timestep loops above cost two launches per step (elementwise cell backward, then d_h_prev = dgates U^T): 17 us per
// step at LSTM-512 / mini-batch 64, most of it launch boundaries and re-reading U^T (4 MB) from L2 every step.  This kernel
// runs the whole loop in one launch.
//   workgroup (ct, bt): hidden units [16 ct, 16 ct + 16) of batch rows [16 bt, 16 bt + 16) -- one (b, j) per thread for the
//     elementwise part, whose carries (d_c / d_h_prev_1) therefore live in a register for the whole loop;
//   the product: the same 16 x 16 tile of d_h_prev = dgates_t [16 x K] U^T [K x 16] on v_mfma_f32_16x16x4_f32, K = 3H | 4H
//     split over the 4 waves; each wave's slice of U^T ([K / 4] x 16) is loaded ONCE into registers (H / 4 <= 128 VGPRs);
//   exchange: dgates_t itself ([B][T][K], an output of the call anyway) -- every workgroup writes its 16 x (3 | 4 x 16) piece with
//     write-through stores and raises its own flag word, wave 0 of every workgroup polls the batch tile's H / 16 flags with one
//     wave-wide load, then each workgroup reads the full 16 x K rows back (sc1 loads).  Rows of different t never share an address, so there is no reuse hazard.
// Hand-off: Guideline 16 R1 (sc1 stores, vmcnt(0) in every storing wave, workgroup barrier, one agent-scope flag store; one
// polling wave, sc1 loads after a barrier); the spin is bounded and raises the runtime's fault word (runtime.hip).
// Same arithmetic as the per-step kernels: the elementwise part is their code, the product their chunked k-ordered MFMA chain
// (4 chunks of K / 4 here instead of 8 of K / 8: a few roundings apart, inside the gradient tests' tolerance).
typedef unsigned bp_v4u __attribute__((ext_vector_type(4)));
struct BpttParams {
    const float *dout;           // [B][T][H] or [B][H]
    const float *UT;             // [K][H]
    const float *c, *zifgo;      // LSTM caches
    const float *h, *Zg, *hU;    // GRU caches
    float *dG;                   // LSTM dgates [B][T][4H] / GRU d_hU [B][T][3H]: output and exchange
    float *dxW;                  // GRU only [B][T][3H]
    unsigned *count;             // [ceil(B / 16)][32] arrival flags (steps handed off per column tile), zeroed by the host
    unsigned *fault;
    unsigned long long spin_ticks;
    int B, T, H, return_sequences;
    int act[5];
    float sc_out;
};
// timing ablations (tools/bptt_ablate.sh): compile-time mask, 0 in the product.  1 no poll, 2 no operand loads, 4 no MFMAs,
// 8 no wait for the stores, 16 no stores, 32 no cache fetch
#ifndef NNTK_BPTT_DBG
#define NNTK_BPTT_DBG 0
#endif
#define BPTT_DBG(bit) ((NNTK_BPTT_DBG & (bit)) != 0)
template <int CELL, int NB>      // CELL 0: GRU, 1: LSTM;  NB: 16-deep K blocks per wave, compile-time (>= K / 64; the excess multiplies zeros)
__global__ __launch_bounds__(256) void bptt_persistent_kernel(BpttParams p) {
    constexpr int NG = CELL ? 4 : 3;
    __shared__ float part[4][4][64];
    __shared__ int s_stop;
    const int H = p.H, K = NG * H, NCT = H / 16, T = p.T;
    const int ct = blockIdx.x % NCT, bt = blockIdx.x / NCT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int i0 = ct * 16;
    const int kw = K / 4, k0 = wv * kw, nb = kw / 16;            // this wave's K range, in blocks of 16
    float ut[NB][4];
#pragma unroll
    for (int kbi = 0; kbi < NB; ++kbi)
#pragma unroll
        for (int c = 0; c < 4; ++c) ut[kbi][c] = kbi < nb ? p.UT[(size_t)(k0 + kbi * 16 + 4 * q + c) * H + i0 + n] : 0.0f;
    const int bl = tid >> 4, jl = tid & 15;
    const int b = bt * 16 + bl, j = i0 + jl;
    const bool live = b < p.B;
    const int am = bt * 16 + n;                                  // the batch row this lane feeds to the MFMA
    const bool a_live = am < p.B;
    const size_t tile_bytes = (size_t)(p.B - bt * 16 < 16 ? p.B - bt * 16 : 16) * T * K * 4;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)(p.dG + (size_t)bt * 16 * T * K), 0, (unsigned)tile_bytes, 0x00020000);
    if (tid == 0) s_stop = 0;
    float carry = 0.0f;          // LSTM: d_c carry; GRU: d_h_prev_1
    float dh2 = 0.0f;            // this thread's element of the product
    // forward caches of one step for this thread's (b, j): fetched one step ahead (the loads fly during the hand-off and the product)
    struct Cached { float z[8]; float c, cprev, dout; };
    auto fetch = [&](int t) {
        Cached f{};
        if (!live || t < 0 || (BPTT_DBG(32) && t < T - 1)) return f;
        const size_t row = (size_t)b * T + t;
        if (p.return_sequences) f.dout = p.dout[row * H + j];
        else if (t == T - 1) f.dout = p.dout[(size_t)b * H + j];
        if (CELL) {
            const float *zg = p.zifgo + row * 8 * H;
#pragma unroll
            for (int g = 0; g < 8; ++g) f.z[g] = zg[g * H + j];
            f.c = p.c[row * H + j];
            f.cprev = t > 0 ? p.c[(row - 1) * H + j] : 0.0f;
        } else {
            const float *Zg = p.Zg + row * 6 * H;
#pragma unroll
            for (int g = 0; g < 6; ++g) f.z[g] = Zg[g * H + j];
            f.c = p.hU[row * H + j];
            f.cprev = t > 0 ? p.h[(row - 1) * H + j] : 0.0f;
        }
        return f;
    };
    Cached cur = fetch(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        const bool last = t == T - 1;
        if (live) {
            const size_t row = (size_t)b * T + t;
            const float dout = cur.dout;
            const int vo = ((bl * T + t) * K + j) * 4;
            if (CELL) {
                const float dh = add_rn(last ? 0.0f : dh2, dout);
                const float it = cur.z[4], ft = cur.z[5], gt = cur.z[6], ot = cur.z[7];
                const float cv = cur.c;
                const float tc = nntk_gate_act(p.act[4], cv, p.sc_out);
                const float d_o = gate_grad(p.act[3], cur.z[3], ot, mul_rn(dh, tc));
                float dc = gate_grad(p.act[4], cv, nntk_gate_act(p.act[4], cv, 1.0f), mul_rn(dh, ot));
                if (!last) dc = add_rn(dc, carry);
                const float d_i = gate_grad(p.act[0], cur.z[0], it, mul_rn(dc, gt));
                const float d_f = t == 0 ? 0.0f : gate_grad(p.act[1], cur.z[1], ft, mul_rn(cur.cprev, dc));
                const float d_g = gate_grad(p.act[2], cur.z[2], gt, mul_rn(dc, it));
                carry = mul_rn(dc, ft);
                if (BPTT_DBG(16)) { if (d_i + d_f + d_g + d_o == 1.2345f) carry += 1.f; } else {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d_i), rg, vo, 0, 16 /* sc1 */);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d_f), rg, vo + H * 4, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d_g), rg, vo + 2 * H * 4, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d_o), rg, vo + 3 * H * 4, 0, 16);
                }
            } else {
                const float dh = add_rn(last ? 0.0f : add_rn(carry, dh2), dout);
                const float z = cur.z[3], r = cur.z[4], ht = cur.z[5];
                carry = mul_rn(z, dh);
                const float dht = add_rn(mul_rn(-z, dh), dh);
                const float dz = mul_rn(t > 0 ? sub_rn(cur.cprev, ht) : -ht, dh);
                const float dZh = gate_grad(p.act[1], cur.z[2], ht, dht);
                const float dr = mul_rn(cur.c, dZh);
                const float dZz = gate_grad(p.act[0], cur.z[0], z, dz);
                const float dZr = gate_grad(p.act[2], cur.z[1], r, dr);
                const float dhUh = mul_rn(r, dZh);
                float *dxW = p.dxW + row * 3 * H;
                dxW[j] = dZz; dxW[H + j] = dZr; dxW[2 * H + j] = dZh;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dZz), rg, vo, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dZr), rg, vo + H * 4, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dhUh), rg, vo + 2 * H * 4, 0, 16);
            }
        }
        if (t == 0) break;
        if (!BPTT_DBG(8)) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's stores have been written through
        __syncthreads();
        cur = fetch(t - 1);                                      // in flight across the hand-off and the product
        // arrival: one flag word per (batch tile, column tile) holding the number of steps handed off -- a plain write-through
        // store, no read-modify-write on a shared counter (32 workgroups adding to one word queue up at L2); wave 0 polls all of
        // the batch tile's flags with one wave-wide load
        if (wv == 0) {
            const unsigned tag = (unsigned)(T - t);
            unsigned *flags = p.count + (size_t)bt * 32;
            if (lane == 0) __hip_atomic_store(flags + ct, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            bool expired = p.spin_ticks == 0;                    // 0: fault injection (tests)
            unsigned spins = 0;
            if (!BPTT_DBG(1))
            while (!expired) {
                const unsigned v = lane < NCT ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
                if (__builtin_amdgcn_ballot_w64(v < tag) == 0) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 63u) == 0) {                      // a peer that gave up has raised the fault word: stop spinning too
                    if (__hip_atomic_load(p.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) expired = true;
                    else expired = __builtin_amdgcn_s_memrealtime() - t_start > p.spin_ticks;
                }
            }
            if (expired && lane == 0) {
                __hip_atomic_fetch_or(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_stop = 1;
            }
        }
        __syncthreads();
        if (s_stop) return;                                      // uniform: every wave of the workgroup leaves
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int ao = a_live ? ((n * T + t) * K + k0 + 4 * q) * 4 : 0x7fff0000;      // out of range (the tile is < 2e9 bytes): the load returns 0; room for + kbi * 64
        // all of the wave's operand loads are requested before the first MFMA: one memory round trip per step, not one per
        // batch of loads (the rows were written through moments ago, so every load goes out to memory)
        bp_v4u a[NB];
#pragma unroll
        for (int kbi = 0; kbi < NB; ++kbi) {
            if (BPTT_DBG(2)) a[kbi] = (bp_v4u){(unsigned)kbi, 1u, 2u, (unsigned)t};
            else a[kbi] = __builtin_amdgcn_raw_buffer_load_b128(rg, kbi < nb ? ao + kbi * 64 : 0x7ffffff0, 0, 16 /* sc1 */);
        }
        __builtin_amdgcn_sched_barrier(0);                      // keep the requests ahead of the MFMAs (the scheduler otherwise pairs them up)
#pragma unroll
        for (int kbi = 0; kbi < NB; ++kbi) {
            if (BPTT_DBG(4)) { acc[0] += __uint_as_float(a[kbi].x + a[kbi].y + a[kbi].z + a[kbi].w); continue; }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[kbi].x), ut[kbi][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[kbi].y), ut[kbi][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[kbi].z), ut[kbi][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[kbi].w), ut[kbi][3], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wv][r][lane] = acc[r];
        __syncthreads();
        // D layout: lane (n, q) holds rows 4 q + r, column n  ->  element (row bl, column jl)
        const int sl = (bl >> 2) * 16 + jl, sr = bl & 3;
        dh2 = add_rn(add_rn(add_rn(part[0][sr][sl], part[1][sr][sl]), part[2][sr][sl]), part[3][sr][sl]);
        // part is rewritten only after the next step's two barriers
    }
}
// 0: ran; 1: shape not taken (the caller falls back to the per-step loop); -1: error
template <int CELL>
static int bptt_persistent(BpttParams &p, float *d_count_words) {
    const int NG = CELL ? 4 : 3;
    const int B = p.B, T = p.T, H = p.H, K = NG * H;
    const NntkOptions &opt = nntk_options();
    if (opt.train_bptt == 0 || nntk_persistent_disabled()) return 1;         // after a fault the process keeps to per-step launches
    if (H % 16 != 0 || H > 512 || K % 64 != 0 || K > 2048 || B < 1 || T < 2) return 1;      // H / 16 <= 32 flags per batch tile
    if ((double)16 * T * K * 4 >= 2.0e9) return 1;               // 32-bit buffer offsets inside a batch tile
    const int nbt = (B + 15) / 16, grid = nbt * (H / 16);
    const int nb = K / 64;
    void (*kern)(BpttParams) = nb <= 8 ? bptt_persistent_kernel<CELL, 8> : nb <= 12 ? bptt_persistent_kernel<CELL, 12> :
                               nb <= 16 ? bptt_persistent_kernel<CELL, 16> : nb <= 24 ? bptt_persistent_kernel<CELL, 24> :
                               bptt_persistent_kernel<CELL, 32>;
    if (nntk_resident_blocks((const void *)kern, 256, 0, 8) < grid) return 1;      // every workgroup must be resident
    p.fault = nntk_fault_word();
    if (!p.fault) return 1;
    p.spin_ticks = (unsigned long long)(opt.rec_spin_us > 0 ? opt.rec_spin_us : 0) * 100ull;
    p.count = reinterpret_cast<unsigned *>(d_count_words);
    if (nntk_shim_memset(d_count_words, 0, (size_t)nbt * 32 * sizeof(unsigned))) return -1;      // B * NG * H floats available: >= nbt * 32
    nntk_persistent_launch_begin();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, nntk_stream(), p);
    const int copy_rc = nntk_fault_enqueue_copy();
    nntk_persistent_launch_end();
    if (copy_rc) return -1;
    NNTK_LAUNCH_CHECK("bptt_persistent_kernel");
    return 0;
}

// forward over all timesteps; caches: d_h [B][T][H], d_Zg [B][T][6H], d_hU [B][T][H]
extern "C" int nntk_shim_gru_train_forward(const float *d_x, const float *d_W, const float *d_U, const float *d_bi, const float *d_bh,
                                           float *d_h, float *d_Zg, float *d_hU, int B, int T, int in, int H,
                                           const int *acts /*z,h,r*/, const float *scales) {
    if (B <= 0 || T <= 0) return 0;
    if (B > 65535) return nntk_fail_msg("recurrent training: mini-batch above 65535 (one grid row per batch entry)");
    GruTrainParams p{};
    p.x = d_x; p.W = d_W; p.U = d_U; p.bi = d_bi; p.bh = d_bh; p.h = d_h; p.Zg = d_Zg; p.hU = d_hU;
    p.B = B; p.T = T; p.in = in; p.H = H;
    p.act_z = acts[0]; p.act_h = acts[1]; p.act_r = acts[2];
    p.sc_z = scales[0]; p.sc_h = scales[1]; p.sc_r = scales[2];
    for (int t = 0; t < T; ++t) {
        p.t = t;
        hipLaunchKernelGGL(gru_train_fwd_step_kernel, dim3((unsigned)((H + 31) / 32), (unsigned)B), dim3(256), 0, nntk_stream(), p);
    }
    NNTK_LAUNCH_CHECK("gru_train_fwd_step_kernel");
    return 0;
}
// backward recurrence; d_work: 2 x [B][H] (d_h_prev_1, d_h_prev_2) + [B][3H] (this step's d_hU)
extern "C" int nntk_shim_gru_train_backward(const float *d_dout, const float *d_UT /*[3H][H]*/, const float *d_h, const float *d_Zg,
                                            const float *d_hU, float *d_dxW, float *d_dhU, float *d_work, int B, int T, int H,
                                            int return_sequences, const int *acts) {
    if (B <= 0 || T <= 0) return 0;
    if (B > 65535) return nntk_fail_msg("recurrent training: mini-batch above 65535 (one grid row per batch entry)");
    float *dhp1 = d_work, *dhp2 = d_work + (size_t)B * H, *step = d_work + (size_t)2 * B * H;
    {
        BpttParams q{};
        q.dout = d_dout; q.UT = d_UT; q.h = d_h; q.Zg = d_Zg; q.hU = d_hU; q.dG = d_dhU; q.dxW = d_dxW;
        q.B = B; q.T = T; q.H = H; q.return_sequences = return_sequences;
        q.act[0] = acts[0]; q.act[1] = acts[1]; q.act[2] = acts[2];
        const int rc = bptt_persistent<0>(q, step);
        if (rc <= 0) return rc;
    }
    GruBwdParams p{};
    p.dout = d_dout; p.h = d_h; p.Zg = d_Zg; p.hU = d_hU; p.dxW = d_dxW; p.dhU = d_dhU;
    p.dhp1 = dhp1; p.dhp2 = dhp2; p.dhp1_out = dhp1; p.dhU_step = step;
    p.B = B; p.T = T; p.H = H; p.return_sequences = return_sequences;
    p.act_z = acts[0]; p.act_h = acts[1]; p.act_r = acts[2];
    for (int t = T - 1; t >= 0; --t) {
        p.t = t;
        hipLaunchKernelGGL(gru_train_bwd_step_kernel, dim3((B * H + 255) / 256), dim3(256), 0, nntk_stream(), p);
        if (t > 0)
            launch_rows_times_colmat(step, d_UT, dhp2, (long)B, H, 3 * H);
    }
    NNTK_LAUNCH_CHECK("gru_train_bwd_step_kernel");
    return 0;
}

// ---- LSTM training (layers/lstm.c:185-239 forward cell, :294-556 BPTT) --------------------------------------------
// Same structure as the GRU path.  Caches: zifgo [B][T][8H] = Z_i | Z_f | Z_g | Z_o | i | f | g | o, c [B][T][H], h [B][T][H].
struct LstmTrainParams {
    const float *x, *W, *U, *bi, *bh;      // W [in][4H], U [H][4H]
    float *h, *c, *zifgo;
    int B, T, in, H, t, v2;
    int act[5];                            // i, f, g, o, out
    float sc[5];
};
__global__ __launch_bounds__(256) void lstm_train_fwd_step_kernel(LstmTrainParams p) {
    const int b = blockIdx.y, j = blockIdx.x * 32 + (threadIdx.x & 31), H = p.H;
    const size_t row = (size_t)b * p.T + p.t;
    const float *hp = p.t > 0 ? p.h + (row - 1) * H : nullptr;      // zero state per sequence (lstm.c:441)
    float Z[4], hu[4];
    if (!cell_dots_chunked<4>(p.x + row * p.in, hp, p.W, p.U, p.in, H, j, Z, hu)) return;
    float a[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        Z[g] = add_rn(Z[g], p.bi[g * H + j]);
        if (p.v2) hu[g] = add_rn(hu[g], p.bh[g * H + j]);
        Z[g] = add_rn(Z[g], hu[g]);
        a[g] = nntk_gate_act(p.act[g], Z[g], p.sc[g]);
    }
    const float cp = p.t > 0 ? p.c[(row - 1) * H + j] : 0.0f;
    const float c = add_rn(mul_rn(a[1], cp), mul_rn(a[0], a[2]));         // f c_prev + i g
    const float h = mul_rn(a[3], nntk_gate_act(p.act[4], c, p.sc[4]));
    float *zg = p.zifgo + row * 8 * H;
#pragma unroll
    for (int g = 0; g < 4; ++g) { zg[g * H + j] = Z[g]; zg[(4 + g) * H + j] = a[g]; }
    p.c[row * H + j] = c;
    p.h[row * H + j] = h;
}
// elementwise part of LSTMCellBackward (lstm.c:294-416) for thread (b, j); d_c carry in place, d_h carry from the
// previous step's U dgates product
struct LstmBwdParams {
    const float *dout, *c, *zifgo;
    const float *dh_carry;       // [B][H] = U dgates of step t + 1
    float *dc_carry;             // [B][H], read (t < T-1) and rewritten
    float *dG, *dG_step;         // [B][T][4H], [B][4H]
    int B, T, H, t, return_sequences;
    int act[5];
    float sc_out;
};
__global__ __launch_bounds__(256) void lstm_train_bwd_step_kernel(LstmBwdParams p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.B * p.H) return;
    const int b = e / p.H, j = e % p.H, H = p.H;
    const size_t row = (size_t)b * p.T + p.t;
    float dout = 0.0f;
    if (p.return_sequences) dout = p.dout[row * H + j];
    else if (p.t == p.T - 1) dout = p.dout[(size_t)b * H + j];
    const bool last = p.t == p.T - 1;
    const float dh = add_rn(last ? 0.0f : p.dh_carry[e], dout);
    const float *zg = p.zifgo + row * 8 * H;
    const float it = zg[4 * H + j], ft = zg[5 * H + j], gt = zg[6 * H + j], ot = zg[7 * H + j];
    const float ct = p.c[row * H + j];
    const float tc = nntk_gate_act(p.act[4], ct, p.sc_out);
    const float d_o = gate_grad(p.act[3], zg[3 * H + j], ot, mul_rn(dh, tc));
    // non-cached derivative of the output activation at c_t (activation.c:49-50): forward value recomputed, UNscaled
    float dc = gate_grad(p.act[4], ct, nntk_gate_act(p.act[4], ct, 1.0f), mul_rn(dh, ot));
    if (!last) dc = add_rn(dc, p.dc_carry[e]);
    const float d_i = gate_grad(p.act[0], zg[j], it, mul_rn(dc, gt));
    const float d_f = p.t == 0 ? 0.0f : gate_grad(p.act[1], zg[H + j], ft, mul_rn(p.c[(row - 1) * H + j], dc));
    const float d_g = gate_grad(p.act[2], zg[2 * H + j], gt, mul_rn(dc, it));
    p.dc_carry[e] = mul_rn(dc, ft);
    float *dG = p.dG + row * 4 * H, *ds = p.dG_step + (size_t)b * 4 * H;
    dG[j] = d_i; dG[H + j] = d_f; dG[2 * H + j] = d_g; dG[3 * H + j] = d_o;
    ds[j] = d_i; ds[H + j] = d_f; ds[2 * H + j] = d_g; ds[3 * H + j] = d_o;
}
extern "C" int nntk_shim_lstm_train_forward(const float *d_x, const float *d_W, const float *d_U, const float *d_bi, const float *d_bh,
                                            float *d_h, float *d_c, float *d_zifgo, int B, int T, int in, int H, int v2,
                                            const int *acts /*i,f,g,o,out*/, const float *scales) {
    if (B <= 0 || T <= 0) return 0;
    if (B > 65535) return nntk_fail_msg("recurrent training: mini-batch above 65535 (one grid row per batch entry)");
    LstmTrainParams p{};
    p.x = d_x; p.W = d_W; p.U = d_U; p.bi = d_bi; p.bh = d_bh; p.h = d_h; p.c = d_c; p.zifgo = d_zifgo;
    p.B = B; p.T = T; p.in = in; p.H = H; p.v2 = v2;
    for (int g = 0; g < 5; ++g) { p.act[g] = acts[g]; p.sc[g] = scales[g]; }
    for (int t = 0; t < T; ++t) {
        p.t = t;
        hipLaunchKernelGGL(lstm_train_fwd_step_kernel, dim3((unsigned)((H + 31) / 32), (unsigned)B), dim3(256), 0, nntk_stream(), p);
    }
    NNTK_LAUNCH_CHECK("lstm_train_fwd_step_kernel");
    return 0;
}
// d_work: [B][H] d_h carry + [B][H] d_c carry + [B][4H] this step's dgates
extern "C" int nntk_shim_lstm_train_backward(const float *d_dout, const float *d_UT /*[4H][H]*/, const float *d_c, const float *d_zifgo,
                                             float *d_dG, float *d_work, int B, int T, int H, int return_sequences,
                                             const int *acts, const float *scales) {
    if (B <= 0 || T <= 0) return 0;
    if (B > 65535) return nntk_fail_msg("recurrent training: mini-batch above 65535 (one grid row per batch entry)");
    float *dh = d_work, *dc = d_work + (size_t)B * H, *step = d_work + (size_t)2 * B * H;
    {
        BpttParams q{};
        q.dout = d_dout; q.UT = d_UT; q.c = d_c; q.zifgo = d_zifgo; q.dG = d_dG;
        q.B = B; q.T = T; q.H = H; q.return_sequences = return_sequences;
        for (int g = 0; g < 5; ++g) q.act[g] = acts[g];
        q.sc_out = scales[4];
        const int rc = bptt_persistent<1>(q, step);
        if (rc <= 0) return rc;
    }
    LstmBwdParams p{};
    p.dout = d_dout; p.c = d_c; p.zifgo = d_zifgo; p.dh_carry = dh; p.dc_carry = dc; p.dG = d_dG; p.dG_step = step;
    p.B = B; p.T = T; p.H = H; p.return_sequences = return_sequences;
    for (int g = 0; g < 5; ++g) p.act[g] = acts[g];
    p.sc_out = scales[4];
    for (int t = T - 1; t >= 0; --t) {
        p.t = t;
        hipLaunchKernelGGL(lstm_train_bwd_step_kernel, dim3((B * H + 255) / 256), dim3(256), 0, nntk_stream(), p);
        if (t > 0)
            launch_rows_times_colmat(step, d_UT, dh, (long)B, H, 4 * H);
    }
    NNTK_LAUNCH_CHECK("lstm_train_bwd_step_kernel");
    return 0;
}

// ---- RNN training (layers/rnn.c:144-166 forward cell, :184-221 backward cell, :249-351) ---------------------------
// caches: gate [B][T][H] (pre-activation), h [B][T][H]
struct RnnTrainParams {
    const float *x, *W, *U, *bi, *bh;      // W [in][H], U [H][H]
    float *h, *gate;
    int B, T, in, H, t, v2, act;
    float sc;
};
__global__ __launch_bounds__(256) void rnn_train_fwd_step_kernel(RnnTrainParams p) {
    const int b = blockIdx.y, j = blockIdx.x * 32 + (threadIdx.x & 31), H = p.H;
    const size_t row = (size_t)b * p.T + p.t;
    const float *hp = p.t > 0 ? p.h + (row - 1) * H : nullptr;
    float xw[1], hu[1];
    if (!cell_dots_chunked<1>(p.x + row * p.in, hp, p.W, p.U, p.in, H, j, xw, hu)) return;
    const float xv = add_rn(xw[0], p.bi[j]);
    const float hv = p.v2 ? add_rn(hu[0], p.bh[j]) : hu[0];
    const float g = add_rn(hv, xv);
    p.gate[row * H + j] = g;
    p.h[row * H + j] = nntk_gate_act(p.act, g, p.sc);
}
__global__ __launch_bounds__(256) void rnn_train_bwd_step_kernel(const float *dout, const float *h, const float *gate,
                                                                 const float *dh_carry, float *dG, float *dG_step,
                                                                 int B, int T, int H, int t, int return_sequences, int act) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * H) return;
    const int b = e / H, j = e % H;
    const size_t row = (size_t)b * T + t;
    float d_o = 0.0f;
    if (return_sequences) d_o = dout[row * H + j];
    else if (t == T - 1) d_o = dout[(size_t)b * H + j];
    const float dh = add_rn(t == T - 1 ? 0.0f : dh_carry[e], d_o);
    const float dg = gate_grad(act, gate[row * H + j], h[row * H + j], dh);
    dG[row * H + j] = dg;
    dG_step[e] = dg;
}
extern "C" int nntk_shim_rnn_train_forward(const float *d_x, const float *d_W, const float *d_U, const float *d_bi, const float *d_bh,
                                           float *d_h, float *d_gate, int B, int T, int in, int H, int v2, int act, float scale) {
    if (B <= 0 || T <= 0) return 0;
    if (B > 65535) return nntk_fail_msg("recurrent training: mini-batch above 65535 (one grid row per batch entry)");
    RnnTrainParams p{};
    p.x = d_x; p.W = d_W; p.U = d_U; p.bi = d_bi; p.bh = d_bh; p.h = d_h; p.gate = d_gate;
    p.B = B; p.T = T; p.in = in; p.H = H; p.v2 = v2; p.act = act; p.sc = scale;
    for (int t = 0; t < T; ++t) {
        p.t = t;
        hipLaunchKernelGGL(rnn_train_fwd_step_kernel, dim3((unsigned)((H + 31) / 32), (unsigned)B), dim3(256), 0, nntk_stream(), p);
    }
    NNTK_LAUNCH_CHECK("rnn_train_fwd_step_kernel");
    return 0;
}
// d_work: [B][H] d_h carry + [B][H] this step's d_gate
extern "C" int nntk_shim_rnn_train_backward(const float *d_dout, const float *d_UT /*[H][H]*/, const float *d_h, const float *d_gate,
                                            float *d_dG, float *d_work, int B, int T, int H, int return_sequences, int act) {
    if (B <= 0 || T <= 0) return 0;
    if (B > 65535) return nntk_fail_msg("recurrent training: mini-batch above 65535 (one grid row per batch entry)");
    float *dh = d_work, *step = d_work + (size_t)B * H;
    for (int t = T - 1; t >= 0; --t) {
        hipLaunchKernelGGL(rnn_train_bwd_step_kernel, dim3((B * H + 255) / 256), dim3(256), 0, nntk_stream(), d_dout, d_h, d_gate,
                           (const float *)dh, d_dG, step, B, T, H, t, return_sequences, act);
        if (t > 0)
            launch_rows_times_colmat(step, d_UT, dh, (long)B, H, H);
    }
    NNTK_LAUNCH_CHECK("rnn_train_bwd_step_kernel");
    return 0;
}

// ---- MFMA forms of the large training products -------------------------------------------------------------------
// C [M][N] (+)= A [M][K] x Bw [N][K]^T with both operands K-contiguous: the inference GEMM kernel (conv1d.hip, k = 1)
// with Bw packed as its weights each call (zero padded to whole tiles, bf16 split images behind it).  Used where a
// training product is big enough that the VALU dots above would dominate (rows x I x K past a threshold); the small
// cases keep the reference-ordered VALU kernels.  d_X = dgates W^T needs no transpose (W [in][G*H] IS Bw); d_W = x^T dgates
// and d_U = h_prev^T dgates transpose both operands first (rows become K).
__global__ __launch_bounds__(256) void pack_rows_kernel(const float *__restrict__ src, float *__restrict__ dst, int N, int K, int N_p, int K_p) {
    const long total = (long)N_p * K_p;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e / K_p), k = (int)(e % K_p);
        dst[e] = (n < N && k < K) ? src[(size_t)n * K + k] : 0.0f;
    }
}
// dst [C][R] = src [R][C]^T through a 32 x 33 LDS tile; shift_T > 0: source row r is replaced by row r - 1, and by zeros
// where r % shift_T == 0 (h_prev of the recurrent layers: row (b, t) -> h[b][t - 1], zero state at t = 0)
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ src, float *__restrict__ dst, long R, int Cc, int shift_T) {
    __shared__ float tile[32][33];
    const long r0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const long r = r0 + j;
        const int c = c0 + tx;
        float v = 0.0f;
        if (r < R && c < Cc) {
            if (shift_T > 0) v = (r % shift_T) ? src[(r - 1) * Cc + c] : 0.0f;
            else v = src[r * Cc + c];
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const long r = r0 + tx;
        if (c < Cc && r < R) dst[(size_t)c * R + r] = tile[tx][j];
    }
}
__global__ __launch_bounds__(256) void add_into_kernel(float *__restrict__ dst, const float *__restrict__ src, long n) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) dst[e] = add_rn(dst[e], src[e]);
}
extern "C" int nntk_shim_add_into(float *d_dst, const float *d_src, long n) {      // dst += src (gradient blocks accumulate)
    if (n <= 0) return 0;
    hipLaunchKernelGGL(add_into_kernel, dim3(grid_for(n, 256)), dim3(256), 0, nntk_stream(), d_dst, d_src, n);
    NNTK_LAUNCH_CHECK("add_into_kernel");
    return 0;
}
extern "C" int nntk_shim_transpose(const float *d_src, float *d_dst, long R, int Cc, int shift_T) {
    if (R <= 0 || Cc <= 0) return 0;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((R + 31) / 32), (unsigned)((Cc + 31) / 32)), dim3(256), 0, nntk_stream(),
                       d_src, d_dst, R, Cc, shift_T);
    NNTK_LAUNCH_CHECK("transpose_kernel");
    return 0;
}
extern "C" size_t nntk_shim_gemm_nt_scratch_floats(int N, int K) {
    int K_p, N_p;
    nntk_shim_conv_pack_sizes(K, N, 1, &K_p, &N_p);
    return (size_t)N_p * K_p * 5 / 2 + 16;            // f32 matrix + three bf16 images
}
// d_C [M][N] = d_A [M][K] x d_Bw [N][K]^T  (accumulate != 0: d_C += that, through d_tmp [M][N])
extern "C" int nntk_shim_gemm_nt(const float *d_A, const float *d_Bw, float *d_C, float *d_pack, float *d_tmp,
                                 long M, int N, int K, int accumulate) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (M > 0x7fffffffL) return nntk_fail_msg("gemm_nt: too many rows");
    int K_p, N_p;
    nntk_shim_conv_pack_sizes(K, N, 1, &K_p, &N_p);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(grid_for((long)N_p * K_p, 256)), dim3(256), 0, nntk_stream(), d_Bw, d_pack, N, K, N_p, K_p);
    NNTK_LAUNCH_CHECK("pack_rows_kernel");
    if (nntk_shim_split_bf16x3(d_pack, d_pack + (size_t)N_p * K_p, N_p, K_p)) return -1;
    float *out = accumulate ? d_tmp : d_C;
    if (nntk_shim_conv1d(d_A, d_pack, nullptr, nullptr, 0.f, NNTK_ACT_IDENTITY, 1.f, out, 1, (int)M, K, N, 1, 1, (int)M, 0)) return -1;
    if (accumulate) {
        hipLaunchKernelGGL(add_into_kernel, dim3(grid_for(M * N, 256)), dim3(256), 0, nntk_stream(), d_C, (const float *)d_tmp, M * N);
        NNTK_LAUNCH_CHECK("add_into_kernel");
    }
    return 0;
}

// ---- MFMA form of the Conv1d input gradient (stride 1; large shapes) ----------------------------------------------
// d_X: the forward kernel itself on d_out padded with k - 1 zero rows on both sides, with the weights flipped in time and
//      transposed in channels: d_X[b][t][i] = sum_o sum_j dout'[b][t + j][o] W[o][i][k - 1 - j].
// (d_W stays on conv1d_grad.hip's sliced VALU dots: its output is only [Cout][Cin * k] -- two MFMA tiles at config 3 --
// with K = B * Tout rows, a shape the tile kernel cannot spread over the chip; measured as an im2col GEMM: slower.)
__global__ __launch_bounds__(256) void pad_time_kernel(const float *__restrict__ in, float *__restrict__ out, int B, int Tin, int pad, int C) {
    const int Tp = Tin + 2 * pad;
    const long total = (long)B * Tp * C;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const long bt = e / C;
        const int t = (int)(bt % Tp) - pad;
        const long b = bt / Tp;
        out[e] = (t >= 0 && t < Tin) ? in[((size_t)b * Tin + t) * C + c] : 0.0f;
    }
}
// packed forward-kernel weights of the d_X convolution: wp [Cin_p32][k][Cout_p16], wp[i][j][o] = W[o][i][k - 1 - j]
__global__ __launch_bounds__(256) void conv_flip_pack_kernel(const float *__restrict__ W, float *__restrict__ wp, int Cout, int Cin, int k,
                                                             int rows_p /*Cin padded to 32*/, int cols_p /*Cout padded to 16*/) {
    const long total = (long)rows_p * k * cols_p;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int o = (int)(e % cols_p);
        const long ij = e / cols_p;
        const int j = (int)(ij % k);
        const int i = (int)(ij / k);
        wp[e] = (i < Cin && o < Cout) ? W[((size_t)o * Cin + i) * k + (k - 1 - j)] : 0.0f;
    }
}
// d_X [B][T][Cin] from d_dout [B][Tout][Cout] (stride 1, T = Tout + k - 1); d_pad >= B * (Tout + 2k - 2) * Cout floats,
// d_wpack >= nntk_shim_conv_dx_pack_floats(Cin, Cout, k) floats
extern "C" size_t nntk_shim_conv_dx_pack_floats(int Cin, int Cout, int k) {
    int cols_p, rows_p;
    nntk_shim_conv_pack_sizes(Cout, Cin, k, &cols_p, &rows_p);
    return (size_t)rows_p * k * cols_p * 5 / 2 + 16;
}
extern "C" int nntk_shim_conv_dx_mfma(const float *d_dout, const float *d_W, float *d_dX, float *d_pad, float *d_wpack,
                                      int B, int T, int Cin, int Cout, int k, int Tout) {
    if (B <= 0 || Tout <= 0) return 0;
    int cols_p, rows_p;
    nntk_shim_conv_pack_sizes(Cout, Cin, k, &cols_p, &rows_p);       // the d_X convolution has Cin' = Cout, Cout' = Cin
    const int Tp = Tout + 2 * (k - 1);
    hipLaunchKernelGGL(pad_time_kernel, dim3(grid_for((long)B * Tp * Cout, 256)), dim3(256), 0, nntk_stream(), d_dout, d_pad, B, Tout, k - 1, Cout);
    NNTK_LAUNCH_CHECK("pad_time_kernel");
    hipLaunchKernelGGL(conv_flip_pack_kernel, dim3(grid_for((long)rows_p * k * cols_p, 256)), dim3(256), 0, nntk_stream(), d_W, d_wpack,
                       Cout, Cin, k, rows_p, cols_p);
    NNTK_LAUNCH_CHECK("conv_flip_pack_kernel");
    if (nntk_shim_split_bf16x3(d_wpack, d_wpack + (size_t)rows_p * k * cols_p, rows_p, k * cols_p)) return -1;
    return nntk_shim_conv1d(d_pad, d_wpack, nullptr, nullptr, 0.f, NNTK_ACT_IDENTITY, 1.f, d_dX, B, Tp, Cout, Cin, k, 1, T, 0);
}
