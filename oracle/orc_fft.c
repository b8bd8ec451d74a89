/*
 * orc_fft.c -- ORACLE (test infrastructure only; see orc.h).
 *
 * Restates the real-FFT the reference obtains from kissfft (github.com/mborgerding/kissfft,
 * un-vendored submodule lib/kissfft, version unpinned: .gitmodules:1-3; compiled with
 * kiss_fft_scalar=float: build.zig:150-183).  The reference's call sites are
 * src/FFT.zig:52-57 (kiss_fftr_alloc), :108-112 (kiss_fftr), :129-133 (kiss_fftri), :79 (free).
 *
 * The algorithm follows kissfft's published structure so that float rounding happens in the
 * same places: a mixed-radix decimation-in-time complex FFT of length nfft/2 whose radix
 * schedule takes 4s first, then 2, 3, 5, then odd numbers; twiddles evaluated in double and
 * rounded to float once; the real transform packs even/odd samples into one complex sequence
 * and un-mixes it with "super twiddles" exp(-i*pi*((k+1)/ncfft + 1/2)).  Forward and inverse are
 * both unscaled (inverse(forward(x)) == nfft * x), which is why NSNet2.zig:323,335 divides by
 * n_fft itself.
 *
 * PARITY UNPINNED: no kissfft source or golden vector is available to check bit-equality; the
 * tests cross-check against numpy.fft in float64 only.
 */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_FACTORS 32
static const double ORC_PI = 3.14159265358979323846264338327;

typedef struct {
    int n;
    int inverse;
    int radix[ORC_MAX_FACTORS];  /* radix of each stage, outermost first */
    int remain[ORC_MAX_FACTORS]; /* sub-transform length after peeling that radix */
    int n_stages;
    orc_cpx *tw; /* tw[j] = exp(-+ 2 pi i j / n) */
} cfft_plan;

struct orc_fftr {
    int nfft;  /* real length */
    int ncfft; /* nfft / 2 */
    int inverse;
    cfft_plan sub;
    orc_cpx *tmp;   /* ncfft */
    orc_cpx *super; /* ncfft / 2 */
};

static inline orc_cpx c_mul(orc_cpx a, orc_cpx b)
{
    orc_cpx m;
    m.r = a.r * b.r - a.i * b.i;
    m.i = a.r * b.i + a.i * b.r;
    return m;
}
static inline orc_cpx c_add(orc_cpx a, orc_cpx b) { orc_cpx m = { a.r + b.r, a.i + b.i }; return m; }
static inline orc_cpx c_sub(orc_cpx a, orc_cpx b) { orc_cpx m = { a.r - b.r, a.i - b.i }; return m; }

/* radix schedule: 4,4,...,2,3,5,7,... ; once p*p > n the remainder itself is the radix */
static void plan_factors(cfft_plan *pl)
{
    int n = pl->n;
    int p = 4;
    double floor_sqrt = floor(sqrt((double)n));
    pl->n_stages = 0;
    do {
        while (n % p) {
            if (p == 4) p = 2;
            else if (p == 2) p = 3;
            else p += 2;
            if (p > floor_sqrt) p = n;
        }
        n /= p;
        pl->radix[pl->n_stages] = p;
        pl->remain[pl->n_stages] = n;
        pl->n_stages++;
    } while (n > 1);
}

static int plan_init(cfft_plan *pl, int n, int inverse)
{
    pl->n = n;
    pl->inverse = inverse;
    pl->tw = (orc_cpx *)malloc(sizeof(orc_cpx) * (size_t)n);
    if (!pl->tw) return -1;
    for (int j = 0; j < n; ++j) {
        double phase = -2.0 * ORC_PI * (double)j / (double)n;
        if (inverse) phase *= -1.0;
        pl->tw[j].r = (float)cos(phase);
        pl->tw[j].i = (float)sin(phase);
    }
    plan_factors(pl);
    return 0;
}

static void radix2(const cfft_plan *pl, orc_cpx *out, size_t stride, int m)
{
    const orc_cpx *tw = pl->tw;
    orc_cpx *hi = out + m;
    for (int k = 0; k < m; ++k) {
        orc_cpx t = c_mul(hi[k], tw[(size_t)k * stride]);
        hi[k] = c_sub(out[k], t);
        out[k] = c_add(out[k], t);
    }
}

static void radix3(const cfft_plan *pl, orc_cpx *out, size_t stride, int m)
{
    const orc_cpx *tw = pl->tw;
    const float epi3_i = tw[stride * (size_t)m].i;
    for (int k = 0; k < m; ++k) {
        orc_cpx *o0 = out + k, *o1 = out + k + m, *o2 = out + k + 2 * m;
        orc_cpx s1 = c_mul(*o1, tw[(size_t)k * stride]);
        orc_cpx s2 = c_mul(*o2, tw[(size_t)k * stride * 2]);
        orc_cpx s3 = c_add(s1, s2);
        orc_cpx s0 = c_sub(s1, s2);
        o1->r = o0->r - s3.r * 0.5f;
        o1->i = o0->i - s3.i * 0.5f;
        s0.r *= epi3_i;
        s0.i *= epi3_i;
        o0->r += s3.r;
        o0->i += s3.i;
        o2->r = o1->r + s0.i;
        o2->i = o1->i - s0.r;
        o1->r -= s0.i;
        o1->i += s0.r;
    }
}

static void radix4(const cfft_plan *pl, orc_cpx *out, size_t stride, int m)
{
    const orc_cpx *tw = pl->tw;
    const int m2 = 2 * m, m3 = 3 * m;
    for (int k = 0; k < m; ++k) {
        orc_cpx *o = out + k;
        orc_cpx s0 = c_mul(o[m], tw[(size_t)k * stride]);
        orc_cpx s1 = c_mul(o[m2], tw[(size_t)k * stride * 2]);
        orc_cpx s2 = c_mul(o[m3], tw[(size_t)k * stride * 3]);
        orc_cpx s5 = c_sub(o[0], s1);
        o[0] = c_add(o[0], s1);
        orc_cpx s3 = c_add(s0, s2);
        orc_cpx s4 = c_sub(s0, s2);
        o[m2] = c_sub(o[0], s3);
        o[0] = c_add(o[0], s3);
        if (pl->inverse) {
            o[m].r = s5.r - s4.i;
            o[m].i = s5.i + s4.r;
            o[m3].r = s5.r + s4.i;
            o[m3].i = s5.i - s4.r;
        } else {
            o[m].r = s5.r + s4.i;
            o[m].i = s5.i - s4.r;
            o[m3].r = s5.r - s4.i;
            o[m3].i = s5.i + s4.r;
        }
    }
}

static void radix5(const cfft_plan *pl, orc_cpx *out, size_t stride, int m)
{
    const orc_cpx *tw = pl->tw;
    const orc_cpx ya = tw[stride * (size_t)m];
    const orc_cpx yb = tw[stride * 2 * (size_t)m];
    for (int u = 0; u < m; ++u) {
        orc_cpx *o0 = out + u, *o1 = o0 + m, *o2 = o0 + 2 * m, *o3 = o0 + 3 * m, *o4 = o0 + 4 * m;
        orc_cpx s0 = *o0;
        orc_cpx s1 = c_mul(*o1, tw[(size_t)u * stride]);
        orc_cpx s2 = c_mul(*o2, tw[2 * (size_t)u * stride]);
        orc_cpx s3 = c_mul(*o3, tw[3 * (size_t)u * stride]);
        orc_cpx s4 = c_mul(*o4, tw[4 * (size_t)u * stride]);
        orc_cpx s7 = c_add(s1, s4);
        orc_cpx s10 = c_sub(s1, s4);
        orc_cpx s8 = c_add(s2, s3);
        orc_cpx s9 = c_sub(s2, s3);

        o0->r += s7.r + s8.r;
        o0->i += s7.i + s8.i;

        orc_cpx s5, s6, s11, s12;
        s5.r = s0.r + s7.r * ya.r + s8.r * yb.r;
        s5.i = s0.i + s7.i * ya.r + s8.i * yb.r;
        s6.r = s10.i * ya.i + s9.i * yb.i;
        s6.i = -(s10.r * ya.i) - s9.r * yb.i;
        *o1 = c_sub(s5, s6);
        *o4 = c_add(s5, s6);

        s11.r = s0.r + s7.r * yb.r + s8.r * ya.r;
        s11.i = s0.i + s7.i * yb.r + s8.i * ya.r;
        s12.r = -(s10.i * yb.i) + s9.i * ya.i;
        s12.i = s10.r * yb.i - s9.r * ya.i;
        *o2 = c_add(s11, s12);
        *o3 = c_sub(s11, s12);
    }
}

/* any other prime radix: direct p-point DFT per column */
static void radix_generic(const cfft_plan *pl, orc_cpx *out, size_t stride, int m, int p)
{
    const orc_cpx *tw = pl->tw;
    const int n = pl->n;
    orc_cpx *scratch = (orc_cpx *)malloc(sizeof(orc_cpx) * (size_t)p);
    for (int u = 0; u < m; ++u) {
        int k = u;
        for (int q1 = 0; q1 < p; ++q1) {
            scratch[q1] = out[k];
            k += m;
        }
        k = u;
        for (int q1 = 0; q1 < p; ++q1) {
            size_t twidx = 0;
            out[k] = scratch[0];
            for (int q = 1; q < p; ++q) {
                twidx += stride * (size_t)k;
                if (twidx >= (size_t)n) twidx -= (size_t)n;
                out[k] = c_add(out[k], c_mul(scratch[q], tw[twidx]));
            }
            k += m;
        }
    }
    free(scratch);
}

/* decimation in time: gather the p interleaved sub-sequences, transform each recursively,
 * then combine with one radix-p butterfly pass */
static void cfft_work(const cfft_plan *pl, orc_cpx *out, const orc_cpx *in, size_t stride,
                      int stage)
{
    const int p = pl->radix[stage];
    const int m = pl->remain[stage];
    if (m == 1) {
        for (int q = 0; q < p; ++q) out[q] = in[(size_t)q * stride];
    } else {
        for (int q = 0; q < p; ++q)
            cfft_work(pl, out + (size_t)q * m, in + (size_t)q * stride, stride * (size_t)p,
                      stage + 1);
    }
    switch (p) {
    case 2: radix2(pl, out, stride, m); break;
    case 3: radix3(pl, out, stride, m); break;
    case 4: radix4(pl, out, stride, m); break;
    case 5: radix5(pl, out, stride, m); break;
    default: radix_generic(pl, out, stride, m, p); break;
    }
}

static void cfft_run(const cfft_plan *pl, const orc_cpx *in, orc_cpx *out)
{
    if (in == out) {
        orc_cpx *tmp = (orc_cpx *)malloc(sizeof(orc_cpx) * (size_t)pl->n);
        cfft_work(pl, tmp, in, 1, 0);
        memcpy(out, tmp, sizeof(orc_cpx) * (size_t)pl->n);
        free(tmp);
    } else {
        cfft_work(pl, out, in, 1, 0);
    }
}

orc_fftr *orc_fftr_alloc(int nfft, int inverse)
{
    if (nfft <= 0 || (nfft & 1)) return NULL; /* FFT.zig:41-43 rejects odd / zero sizes */
    orc_fftr *c = (orc_fftr *)calloc(1, sizeof(orc_fftr));
    if (!c) return NULL;
    c->nfft = nfft;
    c->ncfft = nfft / 2;
    c->inverse = inverse ? 1 : 0;
    if (plan_init(&c->sub, c->ncfft, c->inverse)) {
        free(c);
        return NULL;
    }
    c->tmp = (orc_cpx *)malloc(sizeof(orc_cpx) * (size_t)c->ncfft);
    c->super = (orc_cpx *)malloc(sizeof(orc_cpx) * (size_t)(c->ncfft / 2 + 1));
    for (int i = 0; i < c->ncfft / 2; ++i) {
        double phase = -ORC_PI * ((double)(i + 1) / (double)c->ncfft + 0.5);
        if (c->inverse) phase *= -1.0;
        c->super[i].r = (float)cos(phase);
        c->super[i].i = (float)sin(phase);
    }
    return c;
}

void orc_fftr_free(orc_fftr *c)
{
    if (!c) return;
    free(c->sub.tw);
    free(c->tmp);
    free(c->super);
    free(c);
}

void orc_fftr_forward(orc_fftr *c, const float *timedata, orc_cpx *freq)
{
    const int ncfft = c->ncfft;
    orc_cpx *tmp = c->tmp;
    /* even samples -> real parts, odd samples -> imaginary parts */
    cfft_run(&c->sub, (const orc_cpx *)timedata, tmp);

    const orc_cpx tdc = tmp[0];
    freq[0].r = tdc.r + tdc.i;
    freq[ncfft].r = tdc.r - tdc.i;
    freq[0].i = 0.0f;
    freq[ncfft].i = 0.0f;

    for (int k = 1; k <= ncfft / 2; ++k) {
        orc_cpx fpk = tmp[k];
        orc_cpx fpnk = { tmp[ncfft - k].r, -tmp[ncfft - k].i };
        orc_cpx f1k = c_add(fpk, fpnk);
        orc_cpx f2k = c_sub(fpk, fpnk);
        orc_cpx tw = c_mul(f2k, c->super[k - 1]);
        freq[k].r = (f1k.r + tw.r) * 0.5f;
        freq[k].i = (f1k.i + tw.i) * 0.5f;
        freq[ncfft - k].r = (f1k.r - tw.r) * 0.5f;
        freq[ncfft - k].i = (tw.i - f1k.i) * 0.5f;
    }
}

void orc_fftr_inverse(orc_fftr *c, const orc_cpx *freq, float *timedata)
{
    const int ncfft = c->ncfft;
    orc_cpx *tmp = c->tmp;
    tmp[0].r = freq[0].r + freq[ncfft].r;
    tmp[0].i = freq[0].r - freq[ncfft].r;
    for (int k = 1; k <= ncfft / 2; ++k) {
        orc_cpx fk = freq[k];
        orc_cpx fnkc = { freq[ncfft - k].r, -freq[ncfft - k].i };
        orc_cpx fek = c_add(fk, fnkc);
        orc_cpx t = c_sub(fk, fnkc);
        orc_cpx fok = c_mul(t, c->super[k - 1]);
        tmp[k] = c_add(fek, fok);
        tmp[ncfft - k] = c_sub(fek, fok);
        tmp[ncfft - k].i *= -1.0f;
    }
    cfft_run(&c->sub, tmp, (orc_cpx *)timedata);
}

/* ------------------------------------------------------------------ FFT.zig wrapper */

int orc_fft_bin_count(int n_fft) { return n_fft / 2 + 1; } /* FFT.zig:137-139 */

int orc_fft_fft(orc_fftr *c, const float *first, size_t n_first, const float *second,
                size_t n_second, const float *window, size_t n_window, orc_cpx *bins,
                size_t n_bins)
{
    /* FFT.zig:91-102, checks in the reference's order */
    if (n_first + n_second != (size_t)c->nfft) return ORC_ERR_INVALID_SAMPLES_LENGTH;
    if (n_window != (size_t)c->nfft) return ORC_ERR_INVALID_WINDOW_LENGTH;
    if (n_bins != (size_t)orc_fft_bin_count(c->nfft)) return ORC_ERR_INVALID_RESULT_LENGTH;
    /* loadSamplesFwd, FFT.zig:183-199: buf_real[i] = sample * window[i] */
    float *buf = (float *)malloc(sizeof(float) * (size_t)c->nfft);
    for (size_t i = 0; i < n_first; ++i) buf[i] = first[i] * window[i];
    for (size_t i = 0; i < n_second; ++i) buf[n_first + i] = second[i] * window[n_first + i];
    orc_fftr_forward(c, buf, bins);
    free(buf);
    return ORC_OK;
}

long orc_fft_freq_to_bin(int n_fft, int sample_rate, float freq)
{
    /* FFT.zig:142-167 */
    const float sample_rate_f = (float)sample_rate;
    const float n_fft_f = (float)n_fft;
    const float nyquist = sample_rate_f / 2;
    if (freq > nyquist) return ORC_ERR_OUT_OF_RANGE;
    if (freq < 0) return ORC_ERR_NEGATIVE_FREQUENCY;
    const float bin_width = sample_rate_f / n_fft_f;
    const float bin_f = roundf(freq / bin_width); /* Zig @round: half away from zero */
    return (long)bin_f;
}
