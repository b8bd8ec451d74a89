/*
 * orc_dsp.c -- ORACLE (test infrastructure only; see orc.h).
 * Restates src/audio_utils/window_fn.zig, src/audio_utils/resample.zig and
 * src/audio_utils.zig:14-24 of the reference.  All arithmetic is f32 in the reference's order.
 * Build with -ffp-contract=off: Zig's default float mode does not contract a*b+c.
 */
#include "orc.h"

#include <math.h>
#include <stdlib.h>

/* Zig coerces the comptime_float `2 * pi` to f32 before it meets a runtime f32 operand
 * (window_fn.zig:35 and :65). */
static const float TWO_PI_F = (float)(2.0 * 3.14159265358979323846264338327950288);

/* window_fn.zig:30-41 hannWindowSymmetric */
void orc_hann_window_symmetric(float *result, size_t n)
{
    const float a0 = 0.5f, a1 = 0.5f;
    const float N = (float)n;
    const float step = TWO_PI_F / (N - 1);
    for (size_t i = 0; i < n; ++i) {
        const float x = (float)i;
        result[i] = a0 - a1 * cosf(x * step);
    }
}

/* window_fn.zig:22-28 hannWindowPeriodic -> :51-68 cosineSumWindowPeriodic(K=1, {0.5, 0.5}) */
void orc_hann_window_periodic(float *result, size_t n)
{
    const float alphas[2] = { 0.5f, 1.0f - 0.5f };
    const float N = (float)n;
    for (size_t i = 0; i < n; ++i) {
        const float x = (float)i;
        result[i] = 0;
        for (int k_idx = 0; k_idx < 2; ++k_idx) {
            const float k = (float)k_idx;
            const float sign = (k_idx & 1) ? -1.0f : 1.0f; /* pow(-1, k), window_fn.zig:65 */
            result[i] += sign * alphas[k_idx] * cosf((TWO_PI_F * k * x) / N);
        }
    }
}

/* window_fn.zig:8-16 windowNormFactor */
float orc_window_norm_factor(const float *window, size_t n)
{
    float sum = 0;
    for (size_t i = 0; i < n; ++i) sum += window[i];
    return (float)n / sum;
}

/* NSNet2.zig:384-396 createWindow: sqrt of the symmetric Hann */
void orc_nsnet2_create_window(float *window320)
{
    orc_hann_window_symmetric(window320, 320);
    for (size_t i = 0; i < 320; ++i) window320[i] = sqrtf(window320[i]);
}

/* resample.zig:9-29 downsampleAudio: out[i] = in[i*rate], no filter */
void orc_downsample(const float *first, size_t n_first, const float *second, size_t n_second,
                    float *out, size_t n_out, size_t rate)
{
    const size_t n_in = n_first + n_second;
    if (n_in != n_out * rate) abort(); /* resample.zig:14-16 @panic */
    const size_t n_steps = n_in / rate;
    for (size_t i = 0; i < n_steps; ++i) {
        const size_t src = i * rate;
        if (src < n_first) out[i] = first[src];
        else out[i] = second[src - n_first];
    }
}

/* resample.zig:67-79 interpolate; std.math.lerp(a,b,t) = @mulAdd(b - a, t, a) (Zig std) */
static void interpolate(float first, float second, float *dest, size_t n_dest)
{
    for (size_t i = 0; i < n_dest; ++i) {
        const float fill_idx_f = (float)(i + 1);
        const float fill_count_f = (float)(n_dest + 1);
        const float frac = fill_idx_f / fill_count_f;
        dest[i] = fmaf(second - first, frac, first);
    }
}

/* resample.zig:32-65 upsampleAudio */
float orc_upsample(const float *in, size_t n_in, float *out, size_t n_out, float prev_last_sample,
                   size_t rate)
{
    if (n_in * rate != n_out) abort(); /* resample.zig:38-40 @panic */
    const size_t n_interpolate = rate - 1;
    interpolate(prev_last_sample, in[0], out, n_interpolate);
    out[n_interpolate] = in[0];
    float last_sample = in[0];
    for (size_t i = 1; i < n_in; ++i) {
        const float prev_in = in[i - 1];
        const float curr_in = in[i];
        const size_t from = i * rate;
        const size_t to = from + n_interpolate;
        interpolate(prev_in, curr_in, out + from, n_interpolate);
        out[to] = curr_in;
        last_sample = curr_in;
    }
    return last_sample;
}

/* audio_utils.zig:14-24 rmsVolume: sequential f32 sum of squares, first then second */
float orc_rms_volume(const float *first, size_t n_first, const float *second, size_t n_second)
{
    float sum = 0.0f;
    for (size_t i = 0; i < n_first; ++i) sum += first[i] * first[i];
    for (size_t i = 0; i < n_second; ++i) sum += second[i] * second[i];
    const float mean = sum / (float)(n_first + n_second);
    return sqrtf(mean);
}
