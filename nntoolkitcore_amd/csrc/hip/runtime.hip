// runtime.hip -- device/stream/memory plumbing behind nntk_shim.h, plus the
// memory-bound elementwise kernels (standalone BatchNorm, activations, softmax).
#include "nntk_common.hpp"
#include <stdio.h>
#include <string.h>
#include <vector>

#include <atomic>
#include <mutex>
#include <stdlib.h>

// ---- threading contract (SURVEY 8(b) "Threading": distinct handles are independent) ----
// The CURRENT STREAM and the error string are per host thread; options, the profiling switch and the
// device fault word are process-wide and either atomic or mutex-guarded.  A handle is not re-entrant
// (it owns scratch and recurrent state, like the reference's: conv_1d.c:41, gru.c:86, lstm.c:97), and is
// used on one stream at a time; two threads may drive two handles on two streams concurrently.
static thread_local hipStream_t t_stream = nullptr;
static thread_local char g_err[512] = "";

hipStream_t nntk_stream() { return t_stream; }

int nntk_fail(const char *what, hipError_t err) {
    snprintf(g_err, sizeof(g_err), "HIP error in %s: %s", what, hipGetErrorString(err));
    return -1;
}
int nntk_fail_msg(const char *what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return -1;
}

// ---- options: tuning / diagnostics knobs.  Defaults come from the environment (NNTK_<NAME>) ONCE, at first
//      use; nntk_hip_set_option() changes them at run time.  Nothing on an Apply path calls getenv(). ----
static NntkOptions g_opt;
static std::once_flag g_opt_once;
static std::mutex g_opt_mutex;

struct OptDesc { const char *name; const char *env; int NntkOptions::*field; };
static const OptDesc g_opt_table[] = {
    {"rec_persistent", "NNTK_REC_PERSISTENT", &NntkOptions::rec_persistent},
    {"rec_xw", "NNTK_REC_XW", &NntkOptions::rec_xw},
    {"rec_pingpong", "NNTK_REC_PINGPONG", &NntkOptions::rec_pingpong},
    {"rec_groups", "NNTK_REC_GROUPS", &NntkOptions::rec_groups},
    {"rec_spin_us", "NNTK_REC_SPIN_US", &NntkOptions::rec_spin_us},
    {"rec_stream", "NNTK_REC_STREAM", &NntkOptions::rec_stream},
    {"rec_fused2", "NNTK_REC_FUSED2", &NntkOptions::rec_fused2},
    {"rec_rr", "NNTK_REC_RR", &NntkOptions::rec_rr},
    {"rec_xf", "NNTK_REC_XF", &NntkOptions::rec_xf},
    {"rec_fk", "NNTK_REC_FK", &NntkOptions::rec_fk},
    {"rec_hf", "NNTK_REC_HF", &NntkOptions::rec_hf},
    {"dense_frag3", "NNTK_DENSE_FRAG3", &NntkOptions::dense_frag3},
    {"dense_f16x2", "NNTK_DENSE_F16X2", &NntkOptions::dense_f16x2},
    {"train_bptt", "NNTK_TRAIN_BPTT", &NntkOptions::train_bptt},
    {"train_outer_plain", "NNTK_TRAIN_OUTER_PLAIN", &NntkOptions::train_outer_plain},
    {"spec_ppw", "NNTK_SPEC_PPW", &NntkOptions::spec_ppw},
    {"spec_variant", "NNTK_SPEC_VARIANT", &NntkOptions::spec_variant},
#ifdef NNTK_VARIANT_SPEC_DMA
    {"spec_dma", "NNTK_SPEC_DMA", &NntkOptions::spec_dma},
#endif
    {"bn_fast", "NNTK_BN_FAST", &NntkOptions::bn_fast},
    {"gemm_tm_batch", "NNTK_GEMM_TM_BATCH", &NntkOptions::gemm_tm_batch},
    {"gemm_split_bf16", "NNTK_GEMM_SPLIT_BF16", &NntkOptions::gemm_split_bf16},
    {"conv_store", "NNTK_CONV_STORE", &NntkOptions::conv_store},
    {"conv_frag3_out", "NNTK_CONV_FRAG3_OUT", &NntkOptions::conv_frag3_out},
    {"conv_flatk", "NNTK_CONV_FLATK", &NntkOptions::conv_flatk},
    {"conv_a4", "NNTK_CONV_A4", &NntkOptions::conv_a4},
    {"gemm_wide", "NNTK_GEMM_WIDE", &NntkOptions::gemm_wide},
    {"conv_dbg", "NNTK_CONV_DBG", &NntkOptions::conv_dbg},
    {"weights_check", "NNTK_WEIGHTS_CHECK", &NntkOptions::weights_check},
};
static void opt_init() {
    for (const OptDesc &d : g_opt_table) {
        const char *e = getenv(d.env);
        if (e && *e && strcmp(e, "auto") != 0 && strcmp(e, "default") != 0) g_opt.*(d.field) = atoi(e);    // "auto" keeps the default
    }
}
const NntkOptions &nntk_options() {
    std::call_once(g_opt_once, opt_init);
    return g_opt;
}

// ---- optional HIP-event spans around launch sequences (bench.py's live roofline) ----
struct ProfSpan { hipEvent_t a, b; long launches, units; int kind; hipStream_t stream; };
static std::vector<ProfSpan> g_spans;
static std::mutex g_span_mutex;
static std::atomic<bool> g_prof{false};

int nntk_prof_span_begin(int kind) {
    if (!g_prof.load(std::memory_order_relaxed)) return -1;
    ProfSpan s;
    s.launches = 0;
    s.units = 0;
    s.kind = kind;
    s.stream = t_stream;
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return -1;
    (void)hipEventRecord(s.a, t_stream);
    std::lock_guard<std::mutex> lk(g_span_mutex);
    g_spans.push_back(s);
    return (int)g_spans.size() - 1;
}
void nntk_prof_span_end(int idx, long launches, long units) {
    std::lock_guard<std::mutex> lk(g_span_mutex);
    if (idx < 0 || idx >= (int)g_spans.size()) return;
    (void)hipEventRecord(g_spans[idx].b, g_spans[idx].stream);
    g_spans[idx].launches = launches;
    g_spans[idx].units = units;
}

// ---- device fault word: the persistent recurrent kernel's bounded spins cannot return an error, they OR a bit
//      into ONE sticky word in device memory (never reset per launch).  After each such launch the word is copied
//      (async, same stream) to pinned memory; every host-visible sync point looks at the copy.  Nothing is lost when
//      many launches go by without a sync, and device-pointer callers can poll nntk_hip_device_status().
//      The word (and its mirror) belong to the launching THREAD, so one thread's check never consumes another's fault.
static thread_local unsigned *t_fault_dev = nullptr;              // device word
static thread_local volatile unsigned *t_fault_host = nullptr;    // pinned copy
static std::atomic<int> g_persistent_off{0};         // set after a fault: later recurrent calls take the per-step kernels

unsigned *nntk_fault_word() {
    if (!t_fault_dev) {
        unsigned *d = nullptr, *h = nullptr;
        if (hipMalloc((void **)&d, 256) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipHostMalloc((void **)&h, 256, hipHostMallocDefault) != hipSuccess) { (void)hipFree(d); (void)hipGetLastError(); return nullptr; }
        (void)hipMemset(d, 0, 256);
        h[0] = 0;
        t_fault_host = h;
        t_fault_dev = d;
    }
    return t_fault_dev;
}
int nntk_fault_enqueue_copy() {
    if (!t_fault_dev) return 0;
    NNTK_HIP_TRY(hipMemcpyAsync((void *)t_fault_host, t_fault_dev, sizeof(unsigned), hipMemcpyDeviceToHost, t_stream));
    return 0;
}
int nntk_persistent_disabled() { return g_persistent_off.load(std::memory_order_relaxed); }

// Looks at the pinned copy (valid after the calling thread synchronised its stream).  On a fault: clears the
// device word, switches the process to the per-step recurrent kernels (bit-compatible, no co-residency needed)
// and reports it ONCE on stderr.  Returns the fault bits seen (0 = healthy).
static unsigned fault_take() {
    if (!t_fault_host || t_fault_host[0] == 0) return 0;
    const unsigned bits = t_fault_host[0];
    (void)hipMemsetAsync(t_fault_dev, 0, sizeof(unsigned), t_stream);
    (void)hipStreamSynchronize(t_stream);
    t_fault_host[0] = 0;
    if (!g_persistent_off.exchange(1))
        fprintf(stderr, "nntoolkitcore_hip: a persistent recurrent launch timed out waiting for its peer workgroups "
                        "(another kernel held the CUs?); switching this process to the per-timestep recurrent kernels\n");
    return bits;
}
static int post_sync() {
    if (fault_take())
        return nntk_fail_msg("persistent recurrent kernel: a workgroup timed out waiting for its peers; results of the "
                             "recurrent calls since the last successful synchronisation are invalid. Later calls use "
                             "the per-timestep kernels.");
    return 0;
}

// ---- persistent launches are co-residency-critical (grid <= CU count, one workgroup per CU): two of them running
//      at once on two streams could each hold CUs the other is waiting for.  Inside one process they are therefore
//      ordered across streams with an event chain (no host blocking); other processes are covered by the bounded
//      spins + fault word above. ----
static std::mutex g_persist_mutex;
static hipStream_t g_persist_last = nullptr;
static bool g_persist_any = false;
static hipEvent_t g_persist_ev = nullptr;
void nntk_persistent_launch_begin() {
    g_persist_mutex.lock();
    if (g_persist_any && g_persist_last != t_stream) {
        if (!g_persist_ev && hipEventCreateWithFlags(&g_persist_ev, hipEventDisableTiming) != hipSuccess) g_persist_ev = nullptr;
        if (g_persist_ev && hipEventRecord(g_persist_ev, g_persist_last) == hipSuccess)
            (void)hipStreamWaitEvent(t_stream, g_persist_ev, 0);
        (void)hipGetLastError();      // a destroyed previous stream is not this call's error
    }
}
void nntk_persistent_launch_end() {
    g_persist_last = t_stream;
    g_persist_any = true;
    g_persist_mutex.unlock();
}

// ---- per-device facts and per-kernel attributes, looked up once ----
int nntk_cu_count() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int c = cus[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        cus[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}
int nntk_set_max_dynamic_lds(const void *kernel, size_t bytes) {
    struct Seen { const void *k; size_t bytes; int dev; };
    static std::vector<Seen> seen;
    static std::mutex m;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(m);
    for (const Seen &s : seen)
        if (s.k == kernel && s.dev == dev && s.bytes >= bytes) return 0;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return nntk_fail("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
    seen.push_back({kernel, bytes, dev});
    return 0;
}

// How many workgroups of `kernel` (threads per workgroup, dynamic LDS bytes) are resident at once on the current device:
// the runtime's occupancy answer per CU (clamped to `max_per_cu`, the number the kernel's protocol was designed for)
// times the CU count.  The persistent recurrent kernels size their grids from this instead of assuming "one per CU":
// 0 means the kernel does not fit at all (another LDS carve-out, a different part) and the caller takes the per-step
// path.  Cached per (kernel, device, LDS size).
int nntk_resident_blocks(const void *kernel, int threads, size_t lds, int max_per_cu) {
    struct Seen { const void *k; size_t lds; int dev, threads, per_cu; };
    static std::vector<Seen> seen;
    static std::mutex m;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int per_cu = -1;
    {
        std::lock_guard<std::mutex> lk(m);
        for (const Seen &s : seen)
            if (s.k == kernel && s.dev == dev && s.lds == lds && s.threads == threads) { per_cu = s.per_cu; break; }
    }
    if (per_cu < 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) != hipSuccess) { (void)hipGetLastError(); n = 0; }
        per_cu = n;
        std::lock_guard<std::mutex> lk(m);
        seen.push_back({kernel, lds, dev, threads, per_cu});
    }
    if (per_cu > max_per_cu) per_cu = max_per_cu;
    return per_cu * nntk_cu_count();
}

// name of the recurrent kernel the calling thread launched last (bench.py labels its roofline line with it)
static thread_local const char *t_last_rec_kernel = "";
void nntk_set_last_rec_kernel(const char *name) { t_last_rec_kernel = name; }
static thread_local const char *t_last_conv_kernel = "";
void nntk_set_last_conv_kernel(const char *name) { t_last_conv_kernel = name; }

extern "C" {

int nntk_shim_set_option(const char *name, const char *value) {
    if (!name || !value) return nntk_fail_msg("nntk_hip_set_option: NULL name or value");
    (void)nntk_options();
    std::lock_guard<std::mutex> lk(g_opt_mutex);
    for (const OptDesc &d : g_opt_table)
        if (strcmp(d.name, name) == 0) {
            g_opt.*(d.field) = (strcmp(value, "auto") == 0 || strcmp(value, "default") == 0) ? NntkOptions().*(d.field) : atoi(value);
            // asking for the persistent kernel again re-arms it after a fault had switched the process away from it
            if (d.field == &NntkOptions::rec_persistent && g_opt.rec_persistent != 0) g_persistent_off.store(0);
            return 0;
        }
    return nntk_fail_msg("nntk_hip_set_option: unknown option");
}
int nntk_shim_get_option(const char *name, int *value) {
    if (!name || !value) return nntk_fail_msg("nntk_hip_get_option: NULL argument");
    const NntkOptions &o = nntk_options();
    for (const OptDesc &d : g_opt_table)
        if (strcmp(d.name, name) == 0) { *value = o.*(d.field); return 0; }
    return nntk_fail_msg("nntk_hip_get_option: unknown option");
}

void nntk_shim_profile_enable(int on) { g_prof.store(on != 0); }
// Sums and clears the recorded spans of one kind ("rec_step": recurrent kernels, "spectrogram": K1).  `launches` =
// kernel launches inside the spans, `units` = timesteps (or frames) they covered.
int nntk_shim_profile_get(const char *name, double *total_ms, long *launches, long *units) {
    const int kind = (name && strcmp(name, "spectrogram") == 0) ? NNTK_SPAN_SPEC : NNTK_SPAN_REC;
    *total_ms = 0.0;
    *launches = 0;
    *units = 0;
    std::lock_guard<std::mutex> lk(g_span_mutex);
    std::vector<ProfSpan> keep;
    for (auto &s : g_spans) {
        if (s.kind != kind) { keep.push_back(s); continue; }
        float ms = 0.f;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            *total_ms += ms;
            *launches += s.launches;
            *units += s.units;
        }
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    g_spans.swap(keep);
    return 0;
}

const char *nntk_shim_error(void) { return g_err; }
void nntk_shim_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }
void nntk_shim_clear_error(void) { g_err[0] = 0; }

int nntk_shim_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int nntk_shim_set_device(int device) {
    NNTK_HIP_TRY(hipSetDevice(device));
    return 0;
}
const char *nntk_shim_last_rec_kernel(void) { return t_last_rec_kernel; }
const char *nntk_shim_last_conv_kernel(void) { return t_last_conv_kernel; }
int nntk_shim_get_device(void) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return dev;
}
void nntk_shim_set_stream(void *stream) { t_stream = (hipStream_t)stream; }
void *nntk_shim_get_stream(void) { return (void *)t_stream; }
int nntk_shim_synchronize(void) {
    NNTK_HIP_TRY(hipStreamSynchronize(t_stream));
    return post_sync();
}
// After a stream sync: 1 if a persistent recurrent launch faulted since the last call (and clears it), else 0.
// Used by the host-pointer recurrent Apply calls, which then simply repeat the call on the per-step kernels.
int nntk_shim_take_fault(void) { return fault_take() ? 1 : 0; }
int nntk_shim_persistent_disabled(void) { return nntk_persistent_disabled(); }
// Non-blocking health query for device-pointer callers that synchronise by their own means: 0 healthy,
// 1 a recurrent launch completed so far has faulted (sticky until nntk_hip_synchronize() reports it).
int nntk_shim_device_status(void) { return (t_fault_host && t_fault_host[0]) ? 1 : 0; }

void *nntk_shim_malloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { nntk_fail("hipMalloc", e); return nullptr; }
    return p;
}
void nntk_shim_free(void *p) {
    if (!p) return;
    nntk_shim_weights_exact_only(p, 0);       // a later allocation may reuse the address
    (void)hipFree(p);
}

void *nntk_shim_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { nntk_fail("hipHostMalloc", e); return nullptr; }
    memset(p, 0, bytes);
    return p;
}
void nntk_shim_host_free(void *p) { if (p) (void)hipHostFree(p); }

int nntk_shim_upload(void *d_dst, const void *h_src, size_t bytes) {
    if (!bytes) return 0;
    NNTK_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, t_stream));
    NNTK_HIP_TRY(hipStreamSynchronize(t_stream));
    return 0;
}
/* `height` rows of `width` bytes, source rows `spitch` bytes apart, packed at the destination (the last timestep of every
 * sequence of a [B][T][H] tensor in one copy) */
int nntk_shim_download_rows(void *h_dst, const void *d_src, size_t spitch, size_t width, size_t height) {
    if (!width || !height) return 0;
    NNTK_HIP_TRY(hipMemcpy2DAsync(h_dst, width, d_src, spitch, width, height, hipMemcpyDeviceToHost, t_stream));
    NNTK_HIP_TRY(hipStreamSynchronize(t_stream));
    return post_sync();
}
int nntk_shim_copy_rows_d2d(void *d_dst, const void *d_src, size_t spitch, size_t width, size_t height) {
    if (!width || !height) return 0;
    NNTK_HIP_TRY(hipMemcpy2DAsync(d_dst, width, d_src, spitch, width, height, hipMemcpyDeviceToDevice, t_stream));
    return 0;
}
int nntk_shim_upload_async(void *d_dst, const void *h_src, size_t bytes) {
    if (!bytes) return 0;
    NNTK_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, t_stream));
    return 0;
}
int nntk_shim_download(void *h_dst, const void *d_src, size_t bytes) {
    if (!bytes) return 0;
    NNTK_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, t_stream));
    NNTK_HIP_TRY(hipStreamSynchronize(t_stream));
    return post_sync();
}
// like nntk_shim_download but leaves a recurrent fault for the caller to take (nntk_shim_take_fault)
int nntk_shim_download_nocheck(void *h_dst, const void *d_src, size_t bytes) {
    if (bytes) NNTK_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, t_stream));
    NNTK_HIP_TRY(hipStreamSynchronize(t_stream));
    return 0;
}
int nntk_shim_copy_d2d(void *d_dst, const void *d_src, size_t bytes) {
    if (!bytes) return 0;
    NNTK_HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, t_stream));
    return 0;
}
int nntk_shim_memset(void *d_ptr, int value, size_t bytes) {
    if (!bytes) return 0;
    NNTK_HIP_TRY(hipMemsetAsync(d_ptr, value, bytes, t_stream));
    return 0;
}

}  // extern "C"

// ----------------------------------------------------------------------------
// Standalone BatchNorm (inference): layers/batch_norm.c:140-163 in its op order
//   ((x - mean) / sqrt(var + eps)) * gamma + beta
// HBM-bound: one read + one write of the [rows, C] tensor; the 4*C parameter
// block stays in L1/L2.  Grid-stride over float elements; when C % 4 == 0 each
// lane moves 16 B.
// ----------------------------------------------------------------------------
// (contraction off: the reference rounds the multiply by gamma and the add of beta separately; hipcc would fuse them)
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void bn_kernel_vec4(const float4 *in, const float *__restrict__ bn,
                                                      float eps, float4 *out, long n4, int C) {
    const float *gamma = bn, *beta = bn + C, *mean = bn + 2 * C, *var = bn + 3 * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        int c = (int)((i * 4) % C);
        float4 x = in[i];
        float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = sqrtf(var[c + j] + eps);
            v[j] = ((v[j] - mean[c + j]) / s) * gamma[c + j] + beta[c + j];
        }
        out[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// The same arithmetic with everything per-channel hoisted out of the element loop: when the grid stride is a multiple of the
// row (256 % (C / 4) == 0) a thread stays on ONE channel quad, so gamma / beta / mean and sd = sqrtf(var + eps), 1 / sd are
// computed once per thread, and the division is a reciprocal multiply refined by one FMA step -- the correctly rounded
// quotient (the fused conv epilogue's form, conv1d_kernels.hpp) without the 11-instruction IEEE sequence.  bn_kernel_vec4 spent
// its time there (IEEE sqrt + divide + four parameter loads per ELEMENT): 0.50 ms = 2.1 TB/s on config 3's [1024 x 996, 128]
// tensor against 0.19 ms for the activations (profiles/r03_elementwise.log).  Same bits as bn_kernel_vec4.
__global__ __launch_bounds__(256) void bn_kernel_vec4_rows(const float4 *in, const float *__restrict__ bn,
                                                           float eps, float4 *out, long n4, int C) {
    const long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int c = (int)((i0 * 4) % C);
    float ga[4], be[4], mu[4], sd[4], rsd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ga[j] = bn[c + j]; be[j] = bn[C + c + j]; mu[j] = bn[2 * C + c + j];
        sd[j] = sqrtf(bn[3 * C + c + j] + eps);
        rsd[j] = 1.0f / sd[j];
    }
    for (long i = i0; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 x = in[i];
        float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[j] - mu[j];
            float q = d * rsd[j];
            q = __builtin_fmaf(__builtin_fmaf(-q, sd[j], d), rsd[j], q);       // = d / sd, correctly rounded
            v[j] = q * ga[j] + be[j];                                            // two roundings (contract off)
        }
        out[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__global__ __launch_bounds__(256) void bn_kernel(const float *in, const float *__restrict__ bn, float eps,
                                                 float *out, long n, int C) {
    const float *gamma = bn, *beta = bn + C, *mean = bn + 2 * C, *var = bn + 3 * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        float s = sqrtf(var[c] + eps);
        out[i] = ((in[i] - mean[c]) / s) * gamma[c] + beta[c];
    }
}
#pragma clang fp contract(fast)

// elementwise activations (activation_default.c), 16 B per lane where aligned
__global__ __launch_bounds__(256) void act_kernel(int kind, float relu_a, const float *in,
                                                  float *out, long n) {
    long n4 = n / 4;
    const float4 *in4 = reinterpret_cast<const float4 *>(in);
    float4 *out4 = reinterpret_cast<float4 *>(out);
    long stride = (long)gridDim.x * blockDim.x;
    long tid = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (long i = tid; i < n4; i += stride) {
        float4 x = in4[i];
        out4[i] = make_float4(nntk_act(kind, x.x, relu_a), nntk_act(kind, x.y, relu_a),
                              nntk_act(kind, x.z, relu_a), nntk_act(kind, x.w, relu_a));
    }
    for (long i = n4 * 4 + tid; i < n; i += stride) out[i] = nntk_act(kind, in[i], relu_a);
}

__global__ __launch_bounds__(256) void act_kernel_scalar(int kind, float relu_a, const float *in,
                                                         float *out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = nntk_act(kind, in[i], relu_a);
}

// softmax WITHOUT max subtraction (activation_default.c:149-154): exp, sum, divide.
// One wavefront per vector; 64-lane shuffle reduction of the exp sum.
__global__ __launch_bounds__(256) void softmax_kernel(const float *in, float *out,
                                                      long vectors, int vsize) {
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long v = wave; v < vectors; v += nwaves) {
        const float *x = in + v * vsize;
        float *y = out + v * vsize;
        float sum = 0.0f;
        for (int i = lane; i < vsize; i += 64) {
            float e = expf(x[i]);
            y[i] = e;
            sum += e;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        for (int i = lane; i < vsize; i += 64) y[i] = y[i] / sum;
    }
}

// Short vectors (vsize = 4 * LPV, LPV a power of two <= 64, 16-byte aligned): LPV lanes hold one vector as one float4 each,
// 64 / LPV vectors per wavefront, the exponentials stay in registers -- one 16-byte load and one 16-byte store per lane where
// the generic kernel writes the exponentials, reads them back and writes again with 4- or 8-byte accesses (0.34 ms against
// 0.19 ms for a plain activation on [1024 x 996] vectors of 128).  Same operations: expf, sum (shuffle tree), true divide.
template <int LPV>
__global__ __launch_bounds__(256) void softmax_short_kernel(const float4 *in, float4 *out, long vectors) {
    const long g = (blockIdx.x * (long)blockDim.x + threadIdx.x) / LPV;        // vector index of this lane group
    const int l = threadIdx.x % LPV;
    const long ngroups = ((long)gridDim.x * blockDim.x) / LPV;
    for (long v = g; v < vectors; v += ngroups) {
        const float4 x = in[v * LPV + l];
        float4 e = make_float4(expf(x.x), expf(x.y), expf(x.z), expf(x.w));
        float sum = (e.x + e.y) + (e.z + e.w);
#pragma unroll
        for (int off = LPV / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        out[v * LPV + l] = make_float4(e.x / sum, e.y / sum, e.z / sum, e.w / sum);
    }
}

// Bidirectional helpers (layers/bidirectional.c): time reversal of [B, T, F] rows, and the two merges.
// HBM-bound row copies; one thread per element, rows are contiguous so neighbouring lanes are too.
__global__ __launch_bounds__(256) void reverse_time_kernel(const float *in, float *out, long B, int T, int F) {
    const long total = B * (long)T * F;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int f = (int)(e % F);
        const long bt = e / F;
        const int t = (int)(bt % T);
        const long b = bt / T;
        out[e] = in[(b * T + (T - 1 - t)) * F + f];          // bidirectional.c:11-15
    }
}

// out[r][0:C] = a[r][:], out[r][C:2C] = b[r][:]   (bd_merge_concat, bidirectional.c:40-56, without its transposes)
__global__ __launch_bounds__(256) void concat2_kernel(const float *a, const float *b, float *out, long rows, int C) {
    const long total = rows * 2L * C;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % (2 * C));
        const long r = e / (2 * C);
        out[e] = c < C ? a[r * C + c] : b[r * C + (c - C)];
    }
}

// a[r][:] = in[r][0:C], b[r][:] = in[r][C:2C]   (bd_merge_concat_gradient, bidirectional.c:58-74, without its transposes)
__global__ __launch_bounds__(256) void split2_kernel(const float *in, float *a, float *b, long rows, int C) {
    const long total = rows * 2L * C;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % (2 * C));
        const long r = e / (2 * C);
        if (c < C) a[r * C + c] = in[e]; else b[r * C + (c - C)] = in[e];
    }
}

__global__ __launch_bounds__(256) void add2_kernel(const float *a, const float *b, float *out, long n) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
        out[e] = a[e] + b[e];                                  // bd_merge_sum: op_vec_add (bidirectional.c:76-85)
}

// ---- signal/dft.h: one complex DFT of any size (signal/dft.c:34-47 runs kissfft; un-normalised, inverse = conjugate
// kernel).  A direct O(n^2) transform: the public DFT API is a building block the spectrogram kernels do not go through
// (they carry their own FFTs), so this favours exactness over speed -- twiddles from a table filled with sincospi in
// double, accumulation in double, one output bin per thread.
__global__ __launch_bounds__(256) void dft_twiddle_kernel(float2 *tw, int n) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < n) {
        double sv, cv;
        sincospi(2.0 * (double)m / (double)n, &sv, &cv);
        tw[m] = make_float2((float)cv, (float)sv);
    }
}
__global__ __launch_bounds__(256) void dft_direct_kernel(const float *__restrict__ re, const float *__restrict__ im,
                                                         const float2 *__restrict__ tw, float *__restrict__ ore,
                                                         float *__restrict__ oim, int n, int inverse) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double ar = 0.0, ai = 0.0;
    int m = 0;                                   // (k * j) mod n
    for (int j = 0; j < n; ++j) {
        const float2 w = tw[m];
        const double c = w.x, s = inverse ? (double)w.y : -(double)w.y;       // forward: exp(-2 pi i k j / n)
        const double xr = re[j], xi = im[j];
        ar += xr * c - xi * s;
        ai += xr * s + xi * c;
        m += k;
        if (m >= n) m -= n;
    }
    ore[k] = (float)ar;
    oim[k] = (float)ai;
}

static int grid_for(long work_items, int block) {
    long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;   // 256 CUs x 8 blocks; grid-stride beyond
    return (int)g;
}

extern "C" {

int nntk_shim_batch_norm(const float *d_in, const float *d_bn, float eps, float *d_out, long rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    long n = rows * C;
    bool vec = (C % 4 == 0) && (((size_t)d_in | (size_t)d_out) % 16 == 0);
    if (vec && 256 % (C / 4) == 0) {          // a thread keeps its channel quad: per-channel work hoisted
        hipLaunchKernelGGL(bn_kernel_vec4_rows, dim3(grid_for(n / 4, 256)), dim3(256), 0, nntk_stream(),
                           (const float4 *)d_in, d_bn, eps, (float4 *)d_out, n / 4, C);
    } else if (vec) {
        hipLaunchKernelGGL(bn_kernel_vec4, dim3(grid_for(n / 4, 256)), dim3(256), 0, nntk_stream(),
                           (const float4 *)d_in, d_bn, eps, (float4 *)d_out, n / 4, C);
    } else {
        hipLaunchKernelGGL(bn_kernel, dim3(grid_for(n, 256)), dim3(256), 0, nntk_stream(), d_in, d_bn, eps, d_out, n, C);
    }
    NNTK_LAUNCH_CHECK("bn_kernel");
    return 0;
}

int nntk_shim_reverse_time(const float *d_in, float *d_out, long B, int T, int F) {
    if (B <= 0 || T <= 0 || F <= 0) return 0;
    if (d_in == d_out) return nntk_fail_msg("reverse_time: in-place reversal is not supported");
    hipLaunchKernelGGL(reverse_time_kernel, dim3(grid_for(B * T * F, 256)), dim3(256), 0, nntk_stream(), d_in, d_out, B, T, F);
    NNTK_LAUNCH_CHECK("reverse_time_kernel");
    return 0;
}

int nntk_shim_concat2(const float *d_a, const float *d_b, float *d_out, long rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(concat2_kernel, dim3(grid_for(rows * 2 * C, 256)), dim3(256), 0, nntk_stream(), d_a, d_b, d_out, rows, C);
    NNTK_LAUNCH_CHECK("concat2_kernel");
    return 0;
}

int nntk_shim_split2(const float *d_in, float *d_a, float *d_b, long rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(split2_kernel, dim3(grid_for(rows * 2 * C, 256)), dim3(256), 0, nntk_stream(), d_in, d_a, d_b, rows, C);
    NNTK_LAUNCH_CHECK("split2_kernel");
    return 0;
}

int nntk_shim_add2(const float *d_a, const float *d_b, float *d_out, long n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(add2_kernel, dim3(grid_for(n, 256)), dim3(256), 0, nntk_stream(), d_a, d_b, d_out, n);
    NNTK_LAUNCH_CHECK("add2_kernel");
    return 0;
}

int nntk_shim_dft_twiddles(float *d_tw /*2n floats*/, int n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dft_twiddle_kernel, dim3((n + 255) / 256), dim3(256), 0, nntk_stream(), (float2 *)d_tw, n);
    NNTK_LAUNCH_CHECK("dft_twiddle_kernel");
    return 0;
}
int nntk_shim_dft(const float *d_re, const float *d_im, const float *d_tw, float *d_ore, float *d_oim, int n, int inverse) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dft_direct_kernel, dim3((n + 255) / 256), dim3(256), 0, nntk_stream(), d_re, d_im, (const float2 *)d_tw,
                       d_ore, d_oim, n, inverse);
    NNTK_LAUNCH_CHECK("dft_direct_kernel");
    return 0;
}

int nntk_shim_activation(int kind, float relu_a, int softmax_vector_size, const float *d_in, float *d_out,
                         long n_elems) {
    if (n_elems <= 0) return 0;
    if (kind == NNTK_ACT_SOFTMAX) {
        if (softmax_vector_size <= 0) return nntk_fail_msg("softmax: vector_size must be > 0");
        long vectors = n_elems / softmax_vector_size;
        const int lpv = softmax_vector_size / 4;
        if (softmax_vector_size % 4 == 0 && lpv >= 1 && lpv <= 64 && (lpv & (lpv - 1)) == 0 && (((size_t)d_in | (size_t)d_out) % 16) == 0) {
            const unsigned grid = (unsigned)grid_for(vectors * lpv, 256);
#define NNTK_SM(L) hipLaunchKernelGGL(softmax_short_kernel<L>, dim3(grid), dim3(256), 0, nntk_stream(), (const float4 *)d_in, (float4 *)d_out, vectors)
            switch (lpv) { case 1: NNTK_SM(1); break; case 2: NNTK_SM(2); break; case 4: NNTK_SM(4); break; case 8: NNTK_SM(8); break;
                           case 16: NNTK_SM(16); break; case 32: NNTK_SM(32); break; default: NNTK_SM(64); break; }
#undef NNTK_SM
            NNTK_LAUNCH_CHECK("softmax_short_kernel");
            return 0;
        }
        hipLaunchKernelGGL(softmax_kernel, dim3(grid_for(vectors * 64, 256)), dim3(256), 0, nntk_stream(),
                           d_in, d_out, vectors, softmax_vector_size);
        NNTK_LAUNCH_CHECK("softmax_kernel");
        return 0;
    }
    if (kind == NNTK_ACT_IDENTITY || kind == NNTK_ACT_NONE) {
        if (d_in != d_out) return nntk_shim_copy_d2d(d_out, d_in, (size_t)n_elems * sizeof(float));
        return 0;
    }
    if (kind != NNTK_ACT_SIGMOID && kind != NNTK_ACT_TANH && kind != NNTK_ACT_RELU)
        return nntk_fail_msg("activation: custom host-callback activations cannot run on the device");
    bool aligned = (((size_t)d_in | (size_t)d_out) % 16 == 0);
    if (aligned)
        hipLaunchKernelGGL(act_kernel, dim3(grid_for(n_elems / 4 + 1, 256)), dim3(256), 0, nntk_stream(),
                           kind, relu_a, d_in, d_out, n_elems);
    else
        hipLaunchKernelGGL(act_kernel_scalar, dim3(grid_for(n_elems, 256)), dim3(256), 0, nntk_stream(),
                           kind, relu_a, d_in, d_out, n_elems);
    NNTK_LAUNCH_CHECK("act_kernel");
    return 0;
}

}  // extern "C"
