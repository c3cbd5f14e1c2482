/*
 * spectrogram.c -- window generators and the Spectrogram host layer.  Reference:
 * signal/window.c:13-54, signal/spectrogram.c:36-140.  The window and the scale
 * factor are setup-time host work (kept in the reference's arithmetic: window in
 * double then rounded, scale factors as left-to-right fp32 sums); frames are
 * transformed on the GPU (csrc/hip/spectrogram.hip).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "nntk_internal.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------- windows ---------------------------------- */

/* window.c:13-17: alpha - (1 - alpha) * cos(2*pi*i / denominator), double math */
static void cosine_sum_window(float *v, int size, int denominator, float alpha) {
    for (int i = 0; i < size; ++i) v[i] = alpha - (1 - alpha) * cos(2 * M_PI * i / denominator);
}
void ones(float *vector, int size) { for (int i = 0; i < size; ++i) vector[i] = 1.0f; }
void hann_window(float *vector, int size) { cosine_sum_window(vector, size, size - 1, 0.5f); }
void hamming_window(float *vector, int size) { cosine_sum_window(vector, size, size - 1, 0.54f); }
void periodic_hann_window(float *vector, int size) { cosine_sum_window(vector, size, size, 0.5f); }
void periodic_hamming_window(float *vector, int size) { cosine_sum_window(vector, size, size, 0.54f); }
/* window.c:49-54: the angle is held in a float */
void blackman_window(float *vector, int size) {
    for (int i = 0; i < size; ++i) {
        float angle = 2.f * M_PI * i / size;
        vector[i] = .42f - .5 * cos(angle) + .08 * cos(2 * angle);
    }
}

/* ------------------------------ spectrogram ------------------------------- */

struct SpectrogramStruct {
    SpectrogramConfig config;
    int mode;                /* 0 magnitude, 1 psd */
    int fs;
    float *window;           /* host, [window_size] */
    float scale_factor;
    int window_dirty;
    float *d_window, *d_twiddle;
    nntk_devbuf d_in, d_out;
};

/* spectrogram.c:59-70 */
SpectrogramConfig SpectrogramConfigCreate(int nfft, int window_size, int noverlap, int input_size,
                                          float fft_normalization_factor) {
    SpectrogramConfig c;
    c.nfft = nfft;
    c.window_size = window_size;
    c.noverlap = noverlap;
    c.input_size = input_size;
    c.step = window_size - noverlap;
    c.nfreq = nfft / 2 + 1;
    c.ntime_series = (input_size - noverlap) / c.step;
    c.fft_normalization_factor = fft_normalization_factor;
    return c;
}

/* spectrogram.c:36-38 (sum w) and :49-57 (fs * sum w^2), fp32 left to right */
static void recompute_scale(Spectrogram f) {
    float s = 0.0f;
    if (f->mode == 0) {
        for (int i = 0; i < f->config.window_size; ++i) s += f->window[i];
        f->scale_factor = s;
    } else {
        for (int i = 0; i < f->config.window_size; ++i) {
            float p = f->window[i] * f->window[i];
            s += p;
        }
        f->scale_factor = s * f->fs;
    }
}

static Spectrogram spectrogram_create(SpectrogramConfig config, int mode, int fs) {
    nntk_shim_clear_error();
    Spectrogram f = (Spectrogram)calloc(1, sizeof(struct SpectrogramStruct));
    if (!f) return NULL;
    f->config = config;
    f->mode = mode;
    f->fs = fs;
    f->window = (float *)calloc((size_t)(config.window_size > 0 ? config.window_size : 1), sizeof(float));
    if (!f->window) { free(f); return NULL; }
    SpectrogramSetWindowFunc(f, ones);      /* spectrogram.c:88, :96 */
    return f;
}

Spectrogram SpectrogramCreatePSD(SpectrogramConfig config, int fs) { return spectrogram_create(config, 1, fs); }
Spectrogram SpectrogramCreateMagnitude(SpectrogramConfig config) { return spectrogram_create(config, 0, 0); }
SpectrogramConfig SpectrogramGetConfig(Spectrogram filter) { return filter->config; }

/* spectrogram.c:100-107 */
void SpectrogramSetScaleFactor(Spectrogram filter, float factor) { filter->scale_factor = factor; }
void SpectrogramSetWindowFunc(Spectrogram filter, window_fn fn) {
    fn(filter->window, filter->config.window_size);
    recompute_scale(filter);
    filter->window_dirty = 1;
}

void SpectrogramDestroy(Spectrogram filter) {
    if (!filter) return;
    nntk_shim_synchronize();
    nntk_shim_free(filter->d_window);
    nntk_shim_free(filter->d_twiddle);
    nntk_devbuf_free(&filter->d_in);
    nntk_devbuf_free(&filter->d_out);
    free(filter->window);
    free(filter);
}

static int spectrogram_ensure(Spectrogram f) {
    if (!f->d_twiddle) {
        int n = f->config.nfft;
        float *tw = (float *)malloc(sizeof(float) * 2 * (size_t)n);
        if (!tw) NNTK_FAIL("out of host memory for the twiddle table");
        for (int m = 0; m < n; ++m) {
            double phase = -2.0 * M_PI * m / n;
            tw[2 * m] = (float)cos(phase);
            tw[2 * m + 1] = (float)sin(phase);
        }
        int rc = nntk_upload_floats(&f->d_twiddle, tw, 2 * (size_t)n);
        free(tw);
        if (rc) return rc;
    }
    if (f->window_dirty || !f->d_window) {
        if (f->d_window) nntk_shim_synchronize();
        if (nntk_upload_floats(&f->d_window, f->window, (size_t)f->config.window_size)) return -1;
        f->window_dirty = 0;
    }
    return 0;
}

int SpectrogramApplyDevice(Spectrogram filter, const float *d_input, float *d_output, int batch) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("SpectrogramApplyDevice: NULL handle");
    if (spectrogram_ensure(filter)) return -1;
    const SpectrogramConfig *c = &filter->config;
    return nntk_shim_spectrogram(d_input, filter->d_window, filter->d_twiddle, d_output, batch, c->input_size, c->nfft,
                                 c->window_size, c->step, c->nfreq, c->ntime_series, c->fft_normalization_factor,
                                 filter->mode, filter->scale_factor);
}

/* K1 with the mel projection fused (mel.c); 1 = not taken for this configuration */
int nntk_spectrogram_apply_mel_device(Spectrogram filter, const float *d_input, float *d_output, int batch,
                                      const int *d_mel_tab, const float *d_mel_w, int n_mels, float eps, int do_log) {
    if (spectrogram_ensure(filter)) return -1;
    const SpectrogramConfig *c = &filter->config;
    return nntk_shim_spectrogram_mel(d_input, filter->d_window, filter->d_twiddle, d_output, batch, c->input_size, c->nfft,
                                     c->window_size, c->step, c->nfreq, c->ntime_series, c->fft_normalization_factor,
                                     filter->mode, filter->scale_factor, d_mel_tab, d_mel_w, n_mels, eps, do_log);
}

int SpectrogramApplyBatch(Spectrogram filter, const float *input, float *output, int batch) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("SpectrogramApplyBatch: NULL handle");
    if (batch <= 0) return 0;
    const SpectrogramConfig *c = &filter->config;
    size_t n_in = (size_t)batch * c->input_size;
    size_t n_out = (size_t)batch * c->ntime_series * c->nfreq;
    float *d_in = nntk_devbuf_reserve(&filter->d_in, n_in);
    float *d_out = nntk_devbuf_reserve(&filter->d_out, n_out);
    if (!d_in || !d_out) return -1;
    if (nntk_shim_upload(d_in, input, n_in * sizeof(float))) return -1;
    if (SpectrogramApplyDevice(filter, d_in, d_out, batch)) return -1;
    return nntk_shim_download(output, d_out, n_out * sizeof(float));
}

/* spectrogram.c:113-135: void in the reference; failures are reported through
 * nntk_last_error() and leave `output` untouched. */
void SpectrogramApply(Spectrogram filter, const float *input, float *output) {
    (void)SpectrogramApplyBatch(filter, input, output, 1);
}


/* ============================== signal/dft.h ==============================
 * The reference's public complex DFT (dft.h:15-47, dft.c:23-92: kissfft of size nfft, un-normalised; `complex` is
 * carried in the config and not used there either).  Host split-complex buffers in and out, as the reference. */
struct DFTSetupStruct {
    DFTConfig config;
    float *d_tw;              /* [nfft] (cos, sin) */
    float *d_io;              /* re | im | out re | out im */
};
DFTConfig DFTConfigCreate(int nfft, bool forward, bool complex) {
    DFTConfig c;
    memset(&c, 0, sizeof(c));
    c.nfft = nfft;
    c.forward = forward;
    c.complex = complex;
    return c;
}
DFTSetup DFTSetupCreate(DFTConfig config) {
    nntk_shim_clear_error();
    if (config.nfft <= 0) { nntk_set_error("DFTSetupCreate: nfft must be positive"); return NULL; }
    DFTSetup s = (DFTSetup)calloc(1, sizeof(struct DFTSetupStruct));
    if (!s) return NULL;
    s->config = config;
    s->d_tw = (float *)nntk_shim_malloc((size_t)2 * config.nfft * sizeof(float));
    s->d_io = (float *)nntk_shim_malloc((size_t)4 * config.nfft * sizeof(float));
    if (!s->d_tw || !s->d_io || nntk_shim_dft_twiddles(s->d_tw, config.nfft)) { DFTSetupDestroy(s); return NULL; }
    return s;
}
void DFTSetupDestroy(DFTSetup setup) {
    if (!setup) return;
    nntk_shim_synchronize();
    nntk_shim_free(setup->d_tw);
    nntk_shim_free(setup->d_io);
    free(setup);
}
void DFTPerform(DFTSetup setup, ComplexFloatSplit *input, ComplexFloatSplit *output) {
    nntk_shim_clear_error();
    if (!setup || !input || !output) { nntk_set_error("DFTPerform: NULL argument"); return; }
    const size_t n = (size_t)setup->config.nfft;
    float *d = setup->d_io;
    if (nntk_shim_upload(d, input->real_p, n * sizeof(float)) || nntk_shim_upload(d + n, input->imag_p, n * sizeof(float))) return;
    if (nntk_shim_dft(d, d + n, setup->d_tw, d + 2 * n, d + 3 * n, (int)n, setup->config.forward ? 0 : 1)) return;
    if (nntk_shim_download(output->real_p, d + 2 * n, n * sizeof(float))) return;
    (void)nntk_shim_download(output->imag_p, d + 3 * n, n * sizeof(float));
}
/* dft.c:62-75, :85-92: pure data movement between the interleaved and the split layout */
void split_complex(const ComplexFloat *complex, ComplexFloatSplit *split, int size) {
    for (int i = 0; i < size; ++i) { split->real_p[i] = complex[i].real; split->imag_p[i] = complex[i].imag; }
}
void join_complex_split(const ComplexFloatSplit *split, ComplexFloat *complex, int size) {
    for (int i = 0; i < size; ++i) { complex[i].real = split->real_p[i]; complex[i].imag = split->imag_p[i]; }
}
