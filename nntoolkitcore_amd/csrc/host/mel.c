/*
 * mel.c -- MelFilterBank and LogMelSpectrogram host layer (the bridge from the
 * 257 spectrogram bins to the 40-feature tensor that Conv1d(40 -> ...) consumes).  Reference:
 * signal/mel_filterbank.c:11-23 (HTK mel <-> Hz), :43-102 (triangular filters with slopes
 * evaluated in Hz, min(lower, upper) clamped at 0, bin 0 forced to 0), :116-118 (apply =
 * spec[T,nbins] x W[nbins,n_mels]), signal/log_mel_spectrogram.c:31-36 (log(mel + 1.5849e-13)).
 * The filter matrix is setup-time host work in the reference's fp32 arithmetic; the apply is
 * the MFMA GEMM of csrc/hip/conv1d.hip (k = 1) with the log fused into its epilogue.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "nntk_internal.h"

struct MelFilterBankStruct {
    MelFilterBankConfig config;
    float *weights;          /* host [nbins, n_mels] */
    float *d_wp;             /* packed for the GEMM kernel */
    int uploaded;
    /* sparse form for the fused K1 epilogue: every filter is one run of nonzero weights */
    int *d_tab;              /* [3][n_mels]: first bin, run length, offset into d_runs */
    float *d_runs;
    int sparse_state;        /* 0 not built, 1 ready, -1 not representable (a filter with a gap) */
    nntk_devbuf d_in, d_out;
};

/* mel_filterbank.c:32-40 */
MelFilterBankConfig MelFilterBankConfigCreate(int n_mels, int n_fft, int sample_rate, float lower_hz, float upper_hz) {
    MelFilterBankConfig c;
    c.n_mels = n_mels;
    c.n_fft = n_fft;
    c.sample_rate = sample_rate;
    c.lower_hz = lower_hz;
    c.upper_hz = upper_hz;
    return c;
}

/* mel_filterbank.c:11-23: each step a separate fp32 rounding, like the op_* chain */
static float hz_to_mel(float hz) { float m = hz / 700.0f; m = m + 1.0f; m = logf(m); return m * 1127.0f; }
static float mel_to_hz(float mel) { float h = mel / 1127.0f; h = expf(h); h = h + -1.0f; return h * 700.0f; }

/* mel_filterbank.c:43-102 */
static void build_weights(const MelFilterBankConfig *c, float *w) {
    const int nbins = c->n_fft / 2 + 1, n_mels = c->n_mels;
    float *edge = (float *)calloc((size_t)n_mels + 2, sizeof(float));
    const float m0 = hz_to_mel(c->lower_hz), m1 = hz_to_mel(c->upper_hz);
    const float mstep = (m1 - m0) / (float)(n_mels + 1);
    for (int i = 0; i < n_mels + 2; ++i) edge[i] = mel_to_hz(m0 + (mstep * i));
    const float hstep = (float)c->sample_rate / (float)c->n_fft;
    for (int m = 0; m < n_mels; ++m) {
        const float lo = edge[m], ce = edge[m + 1], up = edge[m + 2];
        for (int j = 0; j < nbins; ++j) {
            const float f = hstep * (float)j;
            float ls = f + (-1.0f * lo);
            ls = ls / (ce - lo);
            float us = -f;
            us = us + up;
            us = us / (up - ce);
            float r = fmaxf(fminf(us, ls), 0.0f);
            if (j == 0) r = 0.0f;
            w[(size_t)j * n_mels + m] = r;
        }
    }
    free(edge);
}

MelFilterBank MelFilterBankCreate(MelFilterBankConfig config) {
    nntk_shim_clear_error();
    MelFilterBank b = (MelFilterBank)calloc(1, sizeof(struct MelFilterBankStruct));
    if (!b) return NULL;
    b->config = config;
    const int nbins = config.n_fft / 2 + 1;
    b->weights = (float *)calloc((size_t)nbins * config.n_mels, sizeof(float));
    if (!b->weights) { free(b); return NULL; }
    build_weights(&config, b->weights);
    return b;
}

void MelFilterBankDestroy(MelFilterBank bank) {
    if (!bank) return;
    nntk_shim_synchronize();
    nntk_shim_free(bank->d_wp);
    nntk_shim_free(bank->d_tab);
    nntk_shim_free(bank->d_runs);
    nntk_devbuf_free(&bank->d_in);
    nntk_devbuf_free(&bank->d_out);
    free(bank->weights);
    free(bank);
}

const float *nntk_mel_weights(MelFilterBank bank) { return bank->weights; }

static int mel_ensure(MelFilterBank b) {
    if (b->uploaded) return 0;
    if (nntk_upload_gemm_weights(&b->d_wp, b->weights, b->config.n_fft / 2 + 1, b->config.n_mels)) return -1;
    b->uploaded = 1;
    return 0;
}

/* run-length form of the filter matrix: filter m's nonzero weights are bins [k0, k0 + len) */
static int mel_ensure_sparse(MelFilterBank b) {
    if (b->sparse_state) return b->sparse_state > 0 ? 0 : 1;
    const int nb = b->config.n_fft / 2 + 1, nm = b->config.n_mels;
    int *tab = (int *)calloc((size_t)3 * nm, sizeof(int));
    float *runs = (float *)calloc((size_t)nb * 2 + 8, sizeof(float));
    if (!tab || !runs) { free(tab); free(runs); NNTK_FAIL("out of host memory for the mel run table"); }
    int total = 0, ok = 1;
    for (int m = 0; m < nm && ok; ++m) {
        int k0 = -1, k1 = -1;
        for (int k = 0; k < nb; ++k)
            if (b->weights[(size_t)k * nm + m] != 0.0f) { if (k0 < 0) k0 = k; k1 = k; }
        const int len = k0 < 0 ? 0 : k1 - k0 + 1;
        if (total + len > nb * 2) { ok = 0; break; }          /* triangles overlap two-fold at most */
        tab[m] = k0 < 0 ? 0 : k0; tab[nm + m] = len; tab[2 * nm + m] = total;
        for (int i = 0; i < len; ++i) runs[total + i] = b->weights[(size_t)(k0 + i) * nm + m];     /* zeros inside a run stay */
        total += len;
    }
    int rc = 1;
    if (ok) {
        b->d_tab = (int *)nntk_shim_malloc((size_t)3 * nm * sizeof(int));
        rc = b->d_tab ? nntk_shim_upload(b->d_tab, tab, (size_t)3 * nm * sizeof(int)) : -1;
        if (!rc) rc = nntk_upload_floats(&b->d_runs, runs, (size_t)(total > 0 ? total : 1));
    }
    free(tab); free(runs);
    if (rc < 0) return -1;
    b->sparse_state = ok ? 1 : -1;
    return ok ? 0 : 1;
}

static int mel_rows_device(MelFilterBank b, const float *d_spec, float *d_out, long rows, int log_eps) {
    if (rows <= 0) return 0;
    if (rows > 0x7fffffffL) NNTK_FAIL("mel filterbank: too many rows");
    if (mel_ensure(b)) return -1;
    return nntk_shim_conv1d(d_spec, b->d_wp, NULL, NULL, 0.f, log_eps ? NNTK_ACT_LOG_EPS : NNTK_ACT_IDENTITY,
                            1.5849e-13f, d_out, 1, (int)rows, b->config.n_fft / 2 + 1, b->config.n_mels, 1, 1,
                            (int)rows, 0);
}

int MelFilterBankApplyDevice(MelFilterBank bank, const float *d_spectrogram, float *d_mel, int rows) {
    nntk_shim_clear_error();
    if (!bank) NNTK_FAIL("MelFilterBankApplyDevice: NULL handle");
    return mel_rows_device(bank, d_spectrogram, d_mel, rows, 0);
}

/* mel_filterbank.c:116-118 (void there; errors through nntk_last_error()) */
void MelFilterBankApply(MelFilterBank bank, const float *spectrogram, float *mel_spectrogram, int timesteps) {
    nntk_shim_clear_error();
    if (!bank) { nntk_set_error("MelFilterBankApply: NULL handle"); return; }
    if (timesteps <= 0) return;
    const size_t n_in = (size_t)timesteps * (bank->config.n_fft / 2 + 1), n_out = (size_t)timesteps * bank->config.n_mels;
    float *d_in = nntk_devbuf_reserve(&bank->d_in, n_in), *d_out = nntk_devbuf_reserve(&bank->d_out, n_out);
    if (!d_in || !d_out) return;
    if (nntk_shim_upload(d_in, spectrogram, n_in * sizeof(float))) return;
    if (mel_rows_device(bank, d_in, d_out, timesteps, 0)) return;
    nntk_shim_download(mel_spectrogram, d_out, n_out * sizeof(float));
}

/* ============================ LogMelSpectrogram =========================== */

struct LogMelSpectrogramStruct {
    Spectrogram spectrogram;     /* owned by the caller, like the reference (log_mel_spectrogram.c:38-42) */
    MelFilterBank bank;
    nntk_devbuf d_in, d_spec, d_out;
};

/* log_mel_spectrogram.c:18-29 */
LogMelSpectrogram LogMelSpectrogramCreate(Spectrogram spectrogram, MelFilterBankConfig mel_filter_bank_config) {
    nntk_shim_clear_error();
    if (!spectrogram) { nntk_set_error("LogMelSpectrogramCreate: NULL spectrogram"); return NULL; }
    if (SpectrogramGetConfig(spectrogram).nfreq != mel_filter_bank_config.n_fft / 2 + 1) {
        nntk_set_error("LogMelSpectrogramCreate: mel n_fft does not match the spectrogram's nfft");
        return NULL;
    }
    LogMelSpectrogram f = (LogMelSpectrogram)calloc(1, sizeof(struct LogMelSpectrogramStruct));
    if (!f) return NULL;
    f->spectrogram = spectrogram;
    f->bank = MelFilterBankCreate(mel_filter_bank_config);
    if (!f->bank) { free(f); return NULL; }
    return f;
}

void LogMelSpectrogramDestroy(LogMelSpectrogram filter) {
    if (!filter) return;
    MelFilterBankDestroy(filter->bank);
    nntk_devbuf_free(&filter->d_in);
    nntk_devbuf_free(&filter->d_spec);
    nntk_devbuf_free(&filter->d_out);
    free(filter);
}

int LogMelSpectrogramApplyDevice(LogMelSpectrogram filter, const float *d_input, float *d_output, int batch) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("LogMelSpectrogramApplyDevice: NULL handle");
    if (batch <= 0) return 0;
    const SpectrogramConfig c = SpectrogramGetConfig(filter->spectrogram);
    const long rows = (long)batch * c.ntime_series;
    /* fused form (nfft = 512): mel + log inside K1's output stage, the [rows, 257] tensor never exists */
    int fused_off = 0;
    (void)nntk_shim_get_option("spec_variant", &fused_off);          /* spec_variant = 1: force the two-kernel form (A/B, tests) */
    if (fused_off != 1) {
        int rs = mel_ensure_sparse(filter->bank);
        if (rs < 0) return -1;
        if (rs == 0) {
            int rc = nntk_spectrogram_apply_mel_device(filter->spectrogram, d_input, d_output, batch, filter->bank->d_tab,
                                                       filter->bank->d_runs, filter->bank->config.n_mels, 1.5849e-13f, 1);
            if (rc <= 0) return rc;
        }
    }
    float *d_spec = nntk_devbuf_reserve(&filter->d_spec, (size_t)rows * c.nfreq);
    if (!d_spec) return -1;
    if (SpectrogramApplyDevice(filter->spectrogram, d_input, d_spec, batch)) return -1;
    return mel_rows_device(filter->bank, d_spec, d_output, rows, 1);
}

int LogMelSpectrogramApplyBatch(LogMelSpectrogram filter, const float *input, float *output, int batch) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("LogMelSpectrogramApplyBatch: NULL handle");
    if (batch <= 0) return 0;
    const SpectrogramConfig c = SpectrogramGetConfig(filter->spectrogram);
    const size_t n_in = (size_t)batch * c.input_size;
    const size_t n_out = (size_t)batch * c.ntime_series * filter->bank->config.n_mels;
    float *d_in = nntk_devbuf_reserve(&filter->d_in, n_in), *d_out = nntk_devbuf_reserve(&filter->d_out, n_out);
    if (!d_in || !d_out) return -1;
    if (nntk_shim_upload(d_in, input, n_in * sizeof(float))) return -1;
    if (LogMelSpectrogramApplyDevice(filter, d_in, d_out, batch)) return -1;
    return nntk_shim_download(output, d_out, n_out * sizeof(float));
}

/* log_mel_spectrogram.c:31-36 */
void LogMelSpectrogramApply(LogMelSpectrogram filter, const float *input, float *output) {
    (void)LogMelSpectrogramApplyBatch(filter, input, output, 1);
}
