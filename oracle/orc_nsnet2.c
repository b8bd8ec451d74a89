/*
 * orc_nsnet2.c -- ORACLE (test infrastructure only; see orc.h).
 *
 * Restates src/NSNet2.zig.  The network itself runs inside ONNX Runtime in the reference
 * (NSNet2.zig:220, un-vendored submodule lib/onnxruntime.zig, unpinned; model
 * data/nsnet2-20ms-baseline.onnx is a missing LFS blob) -- orc_nsnet2_forward restates the
 * graph from the ONNX operator definitions: MatMul+Add (fc1), GRU, GRU, MatMul+Add+Relu (fc2,
 * fc3), MatMul+Add+Sigmoid (fc4), with the GRU in its PyTorch-export form
 * (linear_before_reset = 1, gate order z,r,h, initial_h absent = zeros).  ORT's CPU kernels use
 * MLAS GEMMs and polynomial logistic/tanh approximations whose rounding cannot be reproduced
 * here; this file uses k-ascending fmaf chains and libm expf/tanhf.  PARITY UNPINNED.
 */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { N_FFT = 320, N_HOP = 160, CHUNK = 50 * N_HOP, MITIGATION = 4 }; /* NSNet2.zig:12-16 */
enum { N_FRAMES = CHUNK / N_HOP, N_BINS = N_FFT / 2 + 1 };               /* :365-381 */

/* y[o] = (sum_k x[k] * W[o][k]) + b[o]: MatMul first (accumulator starts at 0, one fmaf per k in
 * ascending k), then Add.  Wt is W transposed to [n_in][n_out] so the loop over output units is
 * the inner, vectorisable one; every output is still its own k-ascending fmaf chain, so the
 * result is bit-identical to the row-by-row scalar form. */
static void dense(const float *restrict x, const float *restrict Wt, const float *restrict b,
                  int n_in, int n_out, float *restrict y)
{
    for (int o = 0; o < n_out; ++o) y[o] = 0.0f;
    for (int k = 0; k < n_in; ++k) {
        const float xk = x[k];
        const float *row = Wt + (size_t)k * n_out;
        for (int o = 0; o < n_out; ++o) y[o] = fmaf(xk, row[o], y[o]);
    }
    for (int o = 0; o < n_out; ++o) y[o] = y[o] + b[o];
}

static float *transpose(const float *W, int n_out, int n_in)
{
    float *t = (float *)malloc(sizeof(float) * (size_t)n_out * (size_t)n_in);
    for (int o = 0; o < n_out; ++o)
        for (int k = 0; k < n_in; ++k) t[(size_t)k * n_out + o] = W[(size_t)o * n_in + k];
    return t;
}

static inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

/* ONNX GRU, one direction, linear_before_reset=1:
 *   z = sigmoid(Wz x + Wbz + Rz h + Rbz)
 *   r = sigmoid(Wr x + Wbr + Rr h + Rbr)
 *   n = tanh   (Wh x + Wbh + r * (Rh h + Rbh))
 *   h' = (1 - z) * n + z * h
 */
static void gru_layer(const float *x, int T, int n_in, int H, const float *W_, const float *R_,
                      const float *B, float *out)
{
    float *W = transpose(W_, 3 * H, n_in);
    float *R = transpose(R_, 3 * H, H);
    float *h = (float *)calloc((size_t)H, sizeof(float));
    float *gi = (float *)malloc(sizeof(float) * 3 * (size_t)H);
    float *gh = (float *)malloc(sizeof(float) * 3 * (size_t)H);
    for (int t = 0; t < T; ++t) {
        dense(x + (size_t)t * n_in, W, B, n_in, 3 * H, gi);
        dense(h, R, B + 3 * H, H, 3 * H, gh);
        float *ht = out + (size_t)t * H;
        for (int j = 0; j < H; ++j) {
            const float z = sigmoidf(gi[j] + gh[j]);
            const float r = sigmoidf(gi[H + j] + gh[H + j]);
            const float n = tanhf(gi[2 * H + j] + r * gh[2 * H + j]);
            ht[j] = (1.0f - z) * n + z * h[j];
        }
        memcpy(h, ht, sizeof(float) * (size_t)H);
    }
    free(h);
    free(gi);
    free(gh);
    free(W);
    free(R);
}

void orc_nsnet2_forward(const orc_nsnet2_weights *w, const float *features, int T, float *gains)
{
    const int B = w->n_bins, F1 = w->n_fc1, H = w->n_hidden, F2 = w->n_fc2, F3 = w->n_fc3;
    float *a1 = (float *)malloc(sizeof(float) * (size_t)T * F1);
    float *h1 = (float *)malloc(sizeof(float) * (size_t)T * H);
    float *h2 = (float *)malloc(sizeof(float) * (size_t)T * H);
    float *a2 = (float *)malloc(sizeof(float) * (size_t)F2);
    float *a3 = (float *)malloc(sizeof(float) * (size_t)F3);
    float *fc1 = transpose(w->fc1_w, F1, B), *fc2 = transpose(w->fc2_w, F2, H);
    float *fc3 = transpose(w->fc3_w, F3, F2), *fc4 = transpose(w->fc4_w, B, F3);
    for (int t = 0; t < T; ++t)
        dense(features + (size_t)t * B, fc1, w->fc1_b, B, F1, a1 + (size_t)t * F1);
    gru_layer(a1, T, F1, H, w->gru1_w, w->gru1_r, w->gru1_b, h1);
    gru_layer(h1, T, H, H, w->gru2_w, w->gru2_r, w->gru2_b, h2);
    for (int t = 0; t < T; ++t) {
        dense(h2 + (size_t)t * H, fc2, w->fc2_b, H, F2, a2);
        for (int j = 0; j < F2; ++j) a2[j] = fmaxf(a2[j], 0.0f);
        dense(a2, fc3, w->fc3_b, F2, F3, a3);
        for (int j = 0; j < F3; ++j) a3[j] = fmaxf(a3[j], 0.0f);
        float *g = gains + (size_t)t * B;
        dense(a3, fc4, w->fc4_b, F3, B, g);
        for (int j = 0; j < B; ++j) g[j] = sigmoidf(g[j]);
    }
    free(a1);
    free(h1);
    free(h2);
    free(a2);
    free(a3);
    free(fc1); free(fc2); free(fc3); free(fc4);
}

/* ------------------------------------------------------------------ NSNet2 object */

struct orc_nsnet2 {
    int in_sample_rate;
    orc_nsnet2_weights w;
    orc_fftr *fwd_fft, *inv_fft;
    float window[N_FFT];
    float audio_input[CHUNK + N_HOP];  /* NSNet2.zig:115-116 */
    float audio_output[CHUNK + N_HOP]; /* :119-120 */
    orc_cpx specgram[N_FRAMES * N_BINS];
    float inv_fft_buffer[N_FFT];
    float features[(N_FRAMES + MITIGATION) * N_BINS]; /* :71-79, zero-initialised */
    float gains[(N_FRAMES + MITIGATION) * N_BINS];
    float last_sample; /* :33 */
};

size_t orc_nsnet2_chunk_size(int sample_rate)
{
    if (sample_rate % 16000 != 0) abort();          /* resample.zig:4-7 @panic */
    return (size_t)CHUNK * (size_t)(sample_rate / 16000); /* NSNet2.zig:157-159 */
}

orc_nsnet2 *orc_nsnet2_create(int sample_rate, const orc_nsnet2_weights *w)
{
    if (w->n_bins != N_BINS) return NULL;
    orc_nsnet2 *d = (orc_nsnet2 *)calloc(1, sizeof(orc_nsnet2));
    d->in_sample_rate = sample_rate;
    d->w = *w;
    d->fwd_fft = orc_fftr_alloc(N_FFT, 0); /* NSNet2.zig:40 */
    d->inv_fft = orc_fftr_alloc(N_FFT, 1); /* :43 */
    orc_nsnet2_create_window(d->window);   /* :46 */
    d->last_sample = 0;
    return d;
}

void orc_nsnet2_destroy(orc_nsnet2 *d)
{
    if (!d) return;
    orc_fftr_free(d->fwd_fft);
    orc_fftr_free(d->inv_fft);
    free(d);
}

/* NSNet2.zig:239-264 */
static void calc_spectrogram(orc_fftr *fft, const float *audio_chunk, const float *window,
                             orc_cpx *spec)
{
    for (int f = 0; f < N_FRAMES; ++f) {
        const float *frame = audio_chunk + (size_t)f * N_HOP;
        orc_fft_fft(fft, frame, N_FFT, NULL, 0, window, N_FFT, spec + (size_t)f * N_BINS, N_BINS);
    }
}

/* NSNet2.zig:266-287.  p_min = std.math.pow(f32, 10, -12): Zig's pow squares up 10^12 in f32
 * and takes the reciprocal, i.e. 1.0f / 1e12f (not the literal 1e-12f). */
static void calc_features(const orc_cpx *spec, float *features, size_t n)
{
    const float p_min = 1.0f / 1e12f;
    for (size_t i = 0; i < n; ++i) {
        const float pow_spec = spec[i].r * spec[i].r + spec[i].i * spec[i].i;
        const float p_out = fmaxf(pow_spec, p_min);
        features[i] = log10f(p_out);
    }
}

void orc_nsnet2_spec_features(const float *audio_input8160, orc_cpx *spec, float *features)
{
    float window[N_FFT];
    orc_nsnet2_create_window(window);
    orc_fftr *fft = orc_fftr_alloc(N_FFT, 0);
    calc_spectrogram(fft, audio_input8160, window, spec);
    calc_features(spec, features, (size_t)N_FRAMES * N_BINS);
    orc_fftr_free(fft);
}

/* NSNet2.zig:289-310 */
static void apply_specgram_gain(orc_cpx *spec, const float *gains, size_t n)
{
    const float p_min = -80, p_max = 1;
    for (size_t i = 0; i < n; ++i) {
        float g = gains[i];
        if (g < p_min) g = p_min;
        else if (g > p_max) g = p_max;
        spec[i].r *= g;
        spec[i].i *= g;
    }
}

/* NSNet2.zig:312-339 */
static void reconstruct_audio(orc_fftr *fft, const orc_cpx *spec, const float *window,
                              float *inv_buf, float *audio_output)
{
    const float vol_norm_factor = 1 / (float)N_FFT;
    for (int f = 0; f < N_FRAMES; ++f) {
        orc_fftr_inverse(fft, spec + (size_t)f * N_BINS, inv_buf);
        const size_t out_start = (size_t)f * N_HOP;
        for (int i = 0; i < N_FFT; ++i) {
            inv_buf[i] *= window[i] * vol_norm_factor;
            audio_output[out_start + i] += inv_buf[i];
        }
    }
}

int orc_nsnet2_denoise(orc_nsnet2 *d, const float *first, size_t n_first, const float *second,
                       size_t n_second, float *denoised, size_t n_denoised)
{
    const size_t rate = (size_t)(d->in_sample_rate / 16000); /* NSNet2.zig:162 */
    if (n_first + n_second != (size_t)CHUNK * rate) return ORC_ERR_INVALID_INPUT_LENGTH; /* :166-169 */
    if (n_denoised != (size_t)CHUNK * rate) abort(); /* resample.zig:38-40 @panic */

    /* NSNet2.zig:175-192: views into the carry buffers */
    float *in_last_hop = d->audio_input + CHUNK;
    float *in_first_hop = d->audio_input;
    float *in_read_slice = d->audio_input + N_HOP;
    float *out_last_hop = d->audio_output + CHUNK;
    float *out_first_hop = d->audio_output;
    float *out_after_first_hop = d->audio_output + N_HOP;
    const size_t cur = sizeof(d->features) / sizeof(float) - (size_t)N_FRAMES * N_BINS; /* :188 */
    float *gains_cur = d->gains + cur;
    float *features_cur = d->features + cur;
    const float *features_copy_src = d->features + (size_t)N_FRAMES * N_BINS;
    float *features_copy_dst = d->features;

    memcpy(in_first_hop, in_last_hop, sizeof(float) * N_HOP);   /* :196 */
    memcpy(out_first_hop, out_last_hop, sizeof(float) * N_HOP); /* :197 */
    memset(out_after_first_hop, 0, sizeof(float) * CHUNK);      /* :201 */
    memmove(features_copy_dst, features_copy_src, sizeof(float) * cur); /* :203 copyBackwards */

    orc_downsample(first, n_first, second, n_second, in_read_slice, CHUNK, rate); /* :205-209 */
    calc_spectrogram(d->fwd_fft, d->audio_input, d->window, d->specgram);        /* :211-217 */
    calc_features(d->specgram, features_cur, (size_t)N_FRAMES * N_BINS);          /* :219 */
    orc_nsnet2_forward(&d->w, d->features, N_FRAMES + MITIGATION, d->gains);      /* :220 */
    apply_specgram_gain(d->specgram, gains_cur, (size_t)N_FRAMES * N_BINS);       /* :221 */
    reconstruct_audio(d->inv_fft, d->specgram, d->window, d->inv_fft_buffer,
                      d->audio_output);                                           /* :223-229 */
    d->last_sample =
        orc_upsample(d->audio_output, CHUNK, denoised, n_denoised, d->last_sample, rate); /* :231-236 */
    return ORC_OK;
}

const float *orc_nsnet2_features(const orc_nsnet2 *d) { return d->features; }
const float *orc_nsnet2_gains(const orc_nsnet2 *d) { return d->gains; }
const orc_cpx *orc_nsnet2_specgram(const orc_nsnet2 *d) { return d->specgram; }
const float *orc_nsnet2_audio_output(const orc_nsnet2 *d) { return d->audio_output; }
