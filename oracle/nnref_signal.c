/*
 * oracle/nnref_signal.c -- CPU ORACLE (test infrastructure, not product code).
 * Restates signal/window.c, signal/dft.c (+ kissfft's published algorithm),
 * signal/spectrogram.c, signal/mel_filterbank.c, signal/log_mel_spectrogram.c
 * and the scalar op_* helpers of core/default_ops.cc they use.
 * PARITY: windows pinned by oracle/_ref (reference window.c compiled in place);
 * FFT/spectrogram "parity unpinned" (kissfft absent) -- see nnref.h header.
 */
#include "nnref.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ ops --- */

/* core/default_ops.cc:224-231 (op_vec_dot_c): left-to-right fp32 sum */
float ref_op_vec_dot(const float *a, const float *b, int size) {
    float sum = 0.0f;
    for (int i = 0; i < size; ++i) sum += a[i] * b[i];
    return sum;
}

/* core/default_ops.cc:707-717 (op_mat_mul_c): a [M,K], b [K,N], c [M,N] */
void ref_op_mat_mul(const float *a, const float *b, float *c, int M, int N, int K) {
    for (int m = 0; m < M; ++m) {
        for (int n = 0; n < N; ++n) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc += a[m * K + k] * b[k * N + n];
            c[m * N + n] = acc;
        }
    }
}

/* core/default_ops.cc:719-727 (op_mat_transp_c): a is [N,M], b is [M,N] */
void ref_op_mat_transp(const float *a, float *b, int M, int N) {
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) b[i * N + j] = a[j * M + i];
}

/* -------------------------------------------------------------- windows --- */

/* signal/window.c:13-17: computed in double, stored float; float alpha */
static void hann_family(float *v, int size, int denominator, float alpha) {
    for (int i = 0; i < size; ++i)
        v[i] = alpha - (1 - alpha) * cos(2 * M_PI * i / denominator);
}

/* signal/window.c:27-54 */
void ref_window(int kind, float *v, int size) {
    switch (kind) {
    case REF_WIN_ONES:
        for (int i = 0; i < size; ++i) v[i] = 1.0f;
        break;
    case REF_WIN_HANN:             hann_family(v, size, size - 1, 0.5f);  break;
    case REF_WIN_HAMMING:          hann_family(v, size, size - 1, 0.54f); break;
    case REF_WIN_PERIODIC_HANN:    hann_family(v, size, size, 0.5f);      break;
    case REF_WIN_PERIODIC_HAMMING: hann_family(v, size, size, 0.54f);     break;
    case REF_WIN_BLACKMAN:
        /* window.c:49-54: the angle itself is a float */
        for (int i = 0; i < size; ++i) {
            float angle = 2.f * M_PI * i / size;
            v[i] = .42f - .5 * cos(angle) + .08 * cos(2 * angle);
        }
        break;
    default: break;
    }
}

/* -------------------------------------------------------------- kissfft --- */
/* Published algorithm of kissfft's kiss_fft.c (float build, no fixed point). */

typedef struct { float r, i; } cpx;

#define MAXFACTORS 32
typedef struct {
    int nfft;
    int inverse;
    int factors[2 * MAXFACTORS];
    cpx *twiddles;
} kiss_state;

#define C_MUL(m, a, b) do { (m).r = (a).r * (b).r - (a).i * (b).i; \
                            (m).i = (a).r * (b).i + (a).i * (b).r; } while (0)
#define C_ADD(res, a, b) do { (res).r = (a).r + (b).r; (res).i = (a).i + (b).i; } while (0)
#define C_SUB(res, a, b) do { (res).r = (a).r - (b).r; (res).i = (a).i - (b).i; } while (0)
#define C_ADDTO(res, a) do { (res).r += (a).r; (res).i += (a).i; } while (0)
#define C_MULBYSCALAR(c, s) do { (c).r *= (s); (c).i *= (s); } while (0)
#define HALF_OF(x) ((x) * .5f)

static void kf_bfly2(cpx *Fout, const size_t fstride, const kiss_state *st, int m) {
    cpx *Fout2 = Fout + m;
    const cpx *tw1 = st->twiddles;
    cpx t;
    do {
        C_MUL(t, *Fout2, *tw1);
        tw1 += fstride;
        C_SUB(*Fout2, *Fout, t);
        C_ADDTO(*Fout, t);
        ++Fout2;
        ++Fout;
    } while (--m);
}

static void kf_bfly4(cpx *Fout, const size_t fstride, const kiss_state *st, const size_t m) {
    const cpx *tw1, *tw2, *tw3;
    cpx scratch[6];
    size_t k = m;
    const size_t m2 = 2 * m, m3 = 3 * m;
    tw3 = tw2 = tw1 = st->twiddles;
    do {
        C_MUL(scratch[0], Fout[m], *tw1);
        C_MUL(scratch[1], Fout[m2], *tw2);
        C_MUL(scratch[2], Fout[m3], *tw3);

        C_SUB(scratch[5], *Fout, scratch[1]);
        C_ADDTO(*Fout, scratch[1]);
        C_ADD(scratch[3], scratch[0], scratch[2]);
        C_SUB(scratch[4], scratch[0], scratch[2]);
        C_SUB(Fout[m2], *Fout, scratch[3]);
        tw1 += fstride;
        tw2 += fstride * 2;
        tw3 += fstride * 3;
        C_ADDTO(*Fout, scratch[3]);

        if (st->inverse) {
            Fout[m].r = scratch[5].r - scratch[4].i;
            Fout[m].i = scratch[5].i + scratch[4].r;
            Fout[m3].r = scratch[5].r + scratch[4].i;
            Fout[m3].i = scratch[5].i - scratch[4].r;
        } else {
            Fout[m].r = scratch[5].r + scratch[4].i;
            Fout[m].i = scratch[5].i - scratch[4].r;
            Fout[m3].r = scratch[5].r - scratch[4].i;
            Fout[m3].i = scratch[5].i + scratch[4].r;
        }
        ++Fout;
    } while (--k);
}

static void kf_bfly3(cpx *Fout, const size_t fstride, const kiss_state *st, size_t m) {
    size_t k = m;
    const size_t m2 = 2 * m;
    const cpx *tw1, *tw2;
    cpx scratch[5];
    cpx epi3 = st->twiddles[fstride * m];
    tw1 = tw2 = st->twiddles;
    do {
        C_MUL(scratch[1], Fout[m], *tw1);
        C_MUL(scratch[2], Fout[m2], *tw2);

        C_ADD(scratch[3], scratch[1], scratch[2]);
        C_SUB(scratch[0], scratch[1], scratch[2]);
        tw1 += fstride;
        tw2 += fstride * 2;

        Fout[m].r = Fout->r - HALF_OF(scratch[3].r);
        Fout[m].i = Fout->i - HALF_OF(scratch[3].i);

        C_MULBYSCALAR(scratch[0], epi3.i);

        C_ADDTO(*Fout, scratch[3]);

        Fout[m2].r = Fout[m].r + scratch[0].i;
        Fout[m2].i = Fout[m].i - scratch[0].r;

        Fout[m].r -= scratch[0].i;
        Fout[m].i += scratch[0].r;

        ++Fout;
    } while (--k);
}

static void kf_bfly5(cpx *Fout, const size_t fstride, const kiss_state *st, int m) {
    cpx *Fout0, *Fout1, *Fout2, *Fout3, *Fout4;
    int u;
    cpx scratch[13];
    const cpx *twiddles = st->twiddles;
    const cpx *tw;
    cpx ya, yb;
    ya = twiddles[fstride * m];
    yb = twiddles[fstride * 2 * m];

    Fout0 = Fout;
    Fout1 = Fout0 + m;
    Fout2 = Fout0 + 2 * m;
    Fout3 = Fout0 + 3 * m;
    Fout4 = Fout0 + 4 * m;

    tw = st->twiddles;
    for (u = 0; u < m; ++u) {
        scratch[0] = *Fout0;

        C_MUL(scratch[1], *Fout1, tw[u * fstride]);
        C_MUL(scratch[2], *Fout2, tw[2 * u * fstride]);
        C_MUL(scratch[3], *Fout3, tw[3 * u * fstride]);
        C_MUL(scratch[4], *Fout4, tw[4 * u * fstride]);

        C_ADD(scratch[7], scratch[1], scratch[4]);
        C_SUB(scratch[10], scratch[1], scratch[4]);
        C_ADD(scratch[8], scratch[2], scratch[3]);
        C_SUB(scratch[9], scratch[2], scratch[3]);

        Fout0->r += scratch[7].r + scratch[8].r;
        Fout0->i += scratch[7].i + scratch[8].i;

        scratch[5].r = scratch[0].r + scratch[7].r * ya.r + scratch[8].r * yb.r;
        scratch[5].i = scratch[0].i + scratch[7].i * ya.r + scratch[8].i * yb.r;

        scratch[6].r = scratch[10].i * ya.i + scratch[9].i * yb.i;
        scratch[6].i = -(scratch[10].r * ya.i) - scratch[9].r * yb.i;

        C_SUB(*Fout1, scratch[5], scratch[6]);
        C_ADD(*Fout4, scratch[5], scratch[6]);

        scratch[11].r = scratch[0].r + scratch[7].r * yb.r + scratch[8].r * ya.r;
        scratch[11].i = scratch[0].i + scratch[7].i * yb.r + scratch[8].i * ya.r;
        scratch[12].r = -(scratch[10].i * yb.i) + scratch[9].i * ya.i;
        scratch[12].i = scratch[10].r * yb.i - scratch[9].r * ya.i;

        C_ADD(*Fout2, scratch[11], scratch[12]);
        C_SUB(*Fout3, scratch[11], scratch[12]);

        ++Fout0; ++Fout1; ++Fout2; ++Fout3; ++Fout4;
    }
}

static void kf_bfly_generic(cpx *Fout, const size_t fstride, const kiss_state *st, int m, int p) {
    int u, k, q1, q;
    const cpx *twiddles = st->twiddles;
    cpx t;
    int Norig = st->nfft;
    cpx *scratch = (cpx *)malloc(sizeof(cpx) * p);
    for (u = 0; u < m; ++u) {
        k = u;
        for (q1 = 0; q1 < p; ++q1) {
            scratch[q1] = Fout[k];
            k += m;
        }
        k = u;
        for (q1 = 0; q1 < p; ++q1) {
            int twidx = 0;
            Fout[k] = scratch[0];
            for (q = 1; q < p; ++q) {
                twidx += (int)fstride * k;
                if (twidx >= Norig) twidx -= Norig;
                C_MUL(t, scratch[q], twiddles[twidx]);
                C_ADDTO(Fout[k], t);
            }
            k += m;
        }
    }
    free(scratch);
}

static void kf_work(cpx *Fout, const cpx *f, const size_t fstride, int in_stride,
                    const int *factors, const kiss_state *st) {
    cpx *Fout_beg = Fout;
    const int p = *factors++;
    const int m = *factors++;
    const cpx *Fout_end = Fout + p * m;

    if (m == 1) {
        do {
            *Fout = *f;
            f += fstride * in_stride;
        } while (++Fout != Fout_end);
    } else {
        do {
            kf_work(Fout, f, fstride * p, in_stride, factors, st);
            f += fstride * in_stride;
        } while ((Fout += m) != Fout_end);
    }

    Fout = Fout_beg;
    switch (p) {
    case 2: kf_bfly2(Fout, fstride, st, m); break;
    case 3: kf_bfly3(Fout, fstride, st, m); break;
    case 4: kf_bfly4(Fout, fstride, st, m); break;
    case 5: kf_bfly5(Fout, fstride, st, m); break;
    default: kf_bfly_generic(Fout, fstride, st, m, p); break;
    }
}

/* facbuf is populated by p1,m1,p2,m2,... where p[i]*m[i] = m[i-1], m0 = n */
static void kf_factor(int n, int *facbuf) {
    int p = 4;
    double floor_sqrt = floor(sqrt((double)n));
    do {
        while (n % p) {
            switch (p) {
            case 4: p = 2; break;
            case 2: p = 3; break;
            default: p += 2; break;
            }
            if (p > floor_sqrt) p = n;
        }
        n /= p;
        *facbuf++ = p;
        *facbuf++ = n;
    } while (n > 1);
}

int ref_kiss_fft(int nfft, int inverse, const float *in, float *out) {
    kiss_state st;
    st.nfft = nfft;
    st.inverse = inverse;
    st.twiddles = (cpx *)malloc(sizeof(cpx) * (size_t)nfft);
    if (!st.twiddles) return -1;
    for (int i = 0; i < nfft; ++i) {
        const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
        double phase = -2 * pi * i / nfft;
        if (inverse) phase *= -1;
        st.twiddles[i].r = (float)cos(phase);
        st.twiddles[i].i = (float)sin(phase);
    }
    kf_factor(nfft, st.factors);
    if (in == out) {
        cpx *tmp = (cpx *)malloc(sizeof(cpx) * (size_t)nfft);
        kf_work(tmp, (const cpx *)in, 1, 1, st.factors, &st);
        memcpy(out, tmp, sizeof(cpx) * (size_t)nfft);
        free(tmp);
    } else {
        kf_work((cpx *)out, (const cpx *)in, 1, 1, st.factors, &st);
    }
    free(st.twiddles);
    return 0;
}

/* ---------------------------------------------------------- spectrogram --- */

/* signal/spectrogram.c:59-70 */
void ref_spectrogram_geometry(int nfft, int window_size, int noverlap, int input_size,
                              int *step, int *nfreq, int *ntime_series) {
    *step = window_size - noverlap;
    *nfreq = nfft / 2 + 1;
    *ntime_series = (input_size - noverlap) / *step;
}

/* signal/spectrogram.c:36-38: op_vec_sum of the window, fp32 left to right */
float ref_spectrogram_scale_magnitude(const float *window, int window_size) {
    float s = 0.0f;
    for (int i = 0; i < window_size; ++i) s += window[i];
    return s;
}

/* signal/spectrogram.c:49-57: sum(w*w) * fs */
float ref_spectrogram_scale_psd(const float *window, int window_size, int fs) {
    float s = 0.0f;
    for (int i = 0; i < window_size; ++i) {
        float p = window[i] * window[i];
        s += p;
    }
    return s * fs;
}

/* signal/spectrogram.c:113-135 with DFTPerform (dft.c:34-47) and the two
 * finishers (spectrogram.c:29-34 magnitude, :41-47 psd). */
int ref_spectrogram(const float *input, const float *window, float *out,
                    int nfft, int window_size, int noverlap, int input_size,
                    float fft_norm, int mode, float scale_factor) {
    int step, nfreq, nts;
    ref_spectrogram_geometry(nfft, window_size, noverlap, input_size, &step, &nfreq, &nts);
    float *re_im = (float *)malloc(sizeof(float) * 2 * (size_t)nfft);   /* split: re[nfft] | im[nfft] */
    float *spec  = (float *)malloc(sizeof(float) * 2 * (size_t)nfft);
    float *cin   = (float *)malloc(sizeof(float) * 2 * (size_t)nfft);   /* interleaved */
    float *cout  = (float *)malloc(sizeof(float) * 2 * (size_t)nfft);
    if (!re_im || !spec || !cin || !cout) return -1;
    for (int t = 0; t < nts; ++t) {
        memset(re_im, 0, sizeof(float) * 2 * (size_t)nfft);
        for (int i = 0; i < window_size; ++i) re_im[i] = window[i] * input[t * step + i];
        /* dft.c:79-84 join, kiss_fft, dft.c:59-69 split */
        for (int i = 0; i < nfft; ++i) { cin[2 * i] = re_im[i]; cin[2 * i + 1] = re_im[nfft + i]; }
        ref_kiss_fft(nfft, 0, cin, cout);
        for (int i = 0; i < nfft; ++i) { spec[i] = cout[2 * i]; spec[nfft + i] = cout[2 * i + 1]; }
        if (fft_norm != 1.0f)
            for (int i = 0; i < 2 * nfft; ++i) spec[i] = spec[i] * fft_norm;
        float *o = out + (size_t)t * nfreq;
        const float *re = spec, *im = spec + nfft;
        for (int i = 0; i < nfreq; ++i) o[i] = re[i] * re[i] + im[i] * im[i];
        if (mode == 0) {
            for (int i = 0; i < nfreq; ++i) o[i] = sqrtf(o[i]);
            for (int i = 0; i < nfreq; ++i) o[i] = o[i] / scale_factor;
        } else {
            float two_over = 2.0f / scale_factor;
            for (int i = 1; i < nfreq - 1; ++i) o[i] = o[i] * two_over;
            o[0] = o[0] / scale_factor;
            o[nfreq - 1] = o[nfreq - 1] / scale_factor;
        }
    }
    free(re_im); free(spec); free(cin); free(cout);
    return 0;
}

/* ------------------------------------------------------- mel filterbank --- */

/* signal/mel_filterbank.c:11-23 */
static float hz_to_mel(float hz) {
    float m = hz / 700.0f;
    m = m + 1.0f;
    m = logf(m);
    return m * 1127.0f;
}
static float mel_to_hz(float mel) {
    float h = mel / 1127.0f;
    h = expf(h);
    h = h + -1.0f;
    return h * 700.0f;
}

/* signal/mel_filterbank.c:43-102; output [nbins, n_mels] */
void ref_mel_filterbank_weights(int n_mels, int n_fft, int sample_rate, float lower_hz, float upper_hz,
                                float *weights) {
    int nbins = n_fft / 2 + 1;
    float *band = (float *)calloc((size_t)n_mels + 2, sizeof(float));
    float *bin_hz = (float *)calloc((size_t)nbins, sizeof(float));
    float e0 = hz_to_mel(lower_hz), e1 = hz_to_mel(upper_hz);
    float stepm = (e1 - e0) / (float)(n_mels + 1);
    for (int i = 0; i < n_mels + 2; ++i) band[i] = e0 + (stepm * i);
    for (int i = 0; i < n_mels + 2; ++i) band[i] = mel_to_hz(band[i]);
    float steph = (float)sample_rate / (float)n_fft;
    for (int i = 0; i < nbins; ++i) bin_hz[i] = steph * (float)i;
    for (int m = 0; m < n_mels; ++m) {
        float lo = band[m], ce = band[m + 1], up = band[m + 2];
        for (int j = 0; j < nbins; ++j) {
            float ls = bin_hz[j] + (-1.0f * lo);
            ls = ls / (ce - lo);
            float us = -bin_hz[j];
            us = us + up;
            us = us / (up - ce);
            float r = fminf(us, ls);
            r = fmaxf(r, 0.0f);
            if (j == 0) r = 0.0f;
            weights[(size_t)j * n_mels + m] = r;
        }
    }
    free(band); free(bin_hz);
}

/* signal/mel_filterbank.c:116-118 + log_mel_spectrogram.c:31-36 */
void ref_log_mel(const float *spec, const float *weights, float *out, int ts, int nbins, int n_mels) {
    ref_op_mat_mul(spec, weights, out, ts, n_mels, nbins);
    const float eps = 1.5849e-13;
    for (int i = 0; i < ts * n_mels; ++i) out[i] = logf(out[i] + eps);
}
