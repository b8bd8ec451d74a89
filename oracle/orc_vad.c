/*
 * orc_vad.c -- ORACLE (test infrastructure only; see orc.h).
 * Restates src/structures/RollingAverage.zig, src/AudioPipeline/VADMetadata.zig and
 * src/AudioPipeline/VADMachine.zig.  f64 averages and integer sample math exactly as written.
 */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ RollingAverage */

struct orc_rolling_average {
    double *data;
    size_t len;
    int has_last_avg;
    double last_avg;
    size_t write_idx;
    size_t written_count;
};

/* RollingAverage.zig:45-56: recomputed from scratch in index order on every call */
static double ra_avg(orc_rolling_average *ra)
{
    double avg = 0.0;
    const double scalar = 1.0 / (double)ra->written_count;
    for (size_t i = 0; i < ra->written_count; ++i) avg += ra->data[i] * scalar;
    ra->last_avg = avg;
    ra->has_last_avg = 1;
    return avg;
}

/* RollingAverage.zig:11-28 */
orc_rolling_average *orc_ra_create(size_t count, int has_initial, double initial_val)
{
    orc_rolling_average *ra = (orc_rolling_average *)calloc(1, sizeof(*ra));
    ra->data = (double *)calloc(count ? count : 1, sizeof(double));
    ra->len = count;
    if (has_initial) {
        for (size_t i = 0; i < count; ++i) ra->data[i] = initial_val;
        ra->written_count = count;
        ra_avg(ra);
    }
    return ra;
}

void orc_ra_destroy(orc_rolling_average *ra)
{
    if (!ra) return;
    free(ra->data);
    free(ra);
}

/* RollingAverage.zig:34-43 */
double orc_ra_push(orc_rolling_average *ra, float sample)
{
    ra->data[ra->write_idx] = (double)sample;
    ra->write_idx = (ra->write_idx + 1) % ra->len;
    if (ra->written_count < ra->len) ra->written_count += 1;
    return ra_avg(ra);
}

int orc_ra_last_avg(const orc_rolling_average *ra, double *out)
{
    if (ra->has_last_avg && out) *out = ra->last_avg;
    return ra->has_last_avg;
}

/* ------------------------------------------------------------------ VADMetadata */

void orc_meta_reset(orc_meta *m) { memset(m, 0, sizeof(*m)); }

/* VADMetadata.zig:29-60; an integer weight is converted with @floatFromInt (:30-33) */
void orc_meta_push(orc_meta *m, const orc_meta_result *v, float weight)
{
    if (v->has_ratio) {
        if (!m->has_ratio) {
            m->has_ratio = 1;
            m->ratio_sum = 0.0f;
            m->ratio_weight = 0.0f;
        }
        m->ratio_sum += v->volume_ratio * weight;
        m->ratio_weight += weight;
    }
    if (v->has_min) {
        if (!m->has_min || v->volume_min < m->volume_min) {
            m->has_min = 1;
            m->volume_min = v->volume_min;
        }
    }
    if (v->has_max) {
        if (!m->has_max || v->volume_max > m->volume_max) {
            m->has_max = 1;
            m->volume_max = v->volume_max;
        }
    }
}

/* VADMetadata.zig:16-27 */
orc_meta_result orc_meta_to_result(const orc_meta *m)
{
    orc_meta_result r;
    memset(&r, 0, sizeof(r));
    r.has_min = m->has_min;
    r.volume_min = m->volume_min;
    r.has_max = m->has_max;
    r.volume_max = m->volume_max;
    if (m->has_ratio) {
        r.has_ratio = 1;
        r.volume_ratio = m->ratio_sum / m->ratio_weight;
    }
    return r;
}

/* ------------------------------------------------------------------ VADMachine */

enum { ST_CLOSED = 0, ST_OPENING = 1, ST_OPEN = 2, ST_CLOSING = 3 }; /* VADMachine.zig:11-16 */

struct orc_vad {
    orc_vad_config cfg;
    int sample_rate, n_channels, fft_size;
    int state;
    orc_rolling_average *long_term, *short_term, *ch_ratio;
    int has_start, has_end;
    uint64_t speech_start_index, speech_end_index;
    float channel_vol_ratio_sum;
    size_t channel_vol_ratio_count;
    float vad_threshold_met_cumulative_sec;
    orc_speech_segment *segments;
    size_t n_segments, cap_segments;
    orc_vad_trace *trace;
    size_t n_trace, cap_trace;
};

void orc_vad_config_default(orc_vad_config *c)
{
    /* VADMachine.zig:30-51 */
    c->speech_min_freq = 500;
    c->speech_max_freq = 2000;
    c->long_term_speech_avg_sec = 180;
    c->has_initial_long_term_avg = 1;
    c->initial_long_term_avg = 0.005;
    c->short_term_speech_avg_sec = 0.2f;
    c->speech_threshold_factor = 10;
    c->channel_vol_ratio_avg_sec = 0.5f;
    c->channel_vol_ratio_threshold = 0.5f;
    c->min_consecutive_sec_to_open = 0.2f;
    c->max_speech_gap_sec = 2;
    c->min_vad_duration_sec = 0.7f;
}

/* VADMachine.zig:75-128 */
orc_vad *orc_vad_create(const orc_vad_config *cfg, int sample_rate, int n_channels, int fft_size)
{
    orc_vad *v = (orc_vad *)calloc(1, sizeof(*v));
    v->cfg = *cfg;
    v->sample_rate = sample_rate;
    v->n_channels = n_channels;
    v->fft_size = fft_size;
    const float sample_rate_f = (float)sample_rate;
    const float fft_size_f = (float)fft_size;
    const float eval_per_sec = sample_rate_f / fft_size_f;
    /* @intFromFloat truncates toward zero (:83-85) */
    size_t long_len = (size_t)(eval_per_sec * cfg->long_term_speech_avg_sec);
    size_t short_len = (size_t)(eval_per_sec * cfg->short_term_speech_avg_sec);
    size_t ratio_len = (size_t)(eval_per_sec * cfg->channel_vol_ratio_avg_sec);
    if (long_len < 1) long_len = 1;   /* @max(1, ..) :89 */
    if (short_len < 1) short_len = 1; /* :96 */
    v->long_term = orc_ra_create(long_len, cfg->has_initial_long_term_avg,
                                 cfg->initial_long_term_avg);
    v->short_term = orc_ra_create(short_len, 0, 0.0);
    v->ch_ratio = orc_ra_create(ratio_len, 0, 0.0); /* no @max here (:101-105) */
    return v;
}

void orc_vad_destroy(orc_vad *v)
{
    if (!v) return;
    orc_ra_destroy(v->long_term);
    orc_ra_destroy(v->short_term);
    orc_ra_destroy(v->ch_ratio);
    free(v->segments);
    free(v->trace);
    free(v);
}

/* VADMachine.zig:311-325 */
static uint64_t offset_recording_start(const orc_vad *v, uint64_t vad_from)
{
    const float sample_rate_f = (float)v->sample_rate;
    const uint64_t start_buffer = (uint64_t)(sample_rate_f * 2);
    return vad_from - (start_buffer < vad_from ? start_buffer : vad_from);
}
static uint64_t offset_recording_end(const orc_vad *v, uint64_t vad_to)
{
    const float sample_rate_f = (float)v->sample_rate;
    const uint64_t end_buffer = (uint64_t)(sample_rate_f * 2);
    return vad_to + end_buffer;
}

/* VADMachine.zig:265-309 */
static orc_vad_result on_speech_end(orc_vad *v)
{
    const float sample_rate_f = (float)v->sample_rate;
    const uint64_t sample_from = v->speech_start_index;
    const uint64_t sample_to = v->speech_end_index;
    const uint64_t length_samples = sample_to - sample_from;
    const float length_sec = (float)length_samples / sample_rate_f;
    const int speech_duration_met = length_sec >= v->cfg.min_vad_duration_sec;
    const float avg_channel_vol_ratio =
        v->channel_vol_ratio_sum / (float)v->channel_vol_ratio_count;
    orc_vad_result res;
    if (speech_duration_met) {
        if (v->n_segments == v->cap_segments) {
            v->cap_segments = v->cap_segments ? v->cap_segments * 2 : 100;
            v->segments = (orc_speech_segment *)realloc(
                v->segments, sizeof(orc_speech_segment) * v->cap_segments);
        }
        orc_speech_segment *s = &v->segments[v->n_segments++];
        s->sample_from = offset_recording_start(v, sample_from);
        s->sample_to = offset_recording_end(v, sample_to);
        s->avg_channel_vol_ratio = avg_channel_vol_ratio;
        s->vad_met_sec = v->vad_threshold_met_cumulative_sec;
        res.recording_state = ORC_REC_COMPLETED;
        res.sample_number = offset_recording_end(v, sample_to);
    } else {
        res.recording_state = ORC_REC_ABORTED;
        res.sample_number = 0;
    }
    return res;
}

/* VADMachine.zig:241-263 */
static void track_speech_stats(orc_vad *v, int has_ratio, float ratio, int threshold_met,
                               int from_state, int to_state)
{
    const float sample_rate_f = (float)v->sample_rate;
    const float input_length_sec = (float)v->fft_size / sample_rate_f;
    const float r = has_ratio ? ratio : 0;
    if (from_state == ST_CLOSED && to_state == ST_OPENING) {
        v->channel_vol_ratio_sum = r;
        v->channel_vol_ratio_count = 1;
        v->vad_threshold_met_cumulative_sec = input_length_sec;
    } else if (from_state == ST_OPEN) {
        v->channel_vol_ratio_sum += r;
        v->channel_vol_ratio_count += 1;
        if (threshold_met) v->vad_threshold_met_cumulative_sec += input_length_sec;
    }
}

/* VADMachine.zig:138-239 */
orc_vad_result orc_vad_run(orc_vad *v, uint64_t index, const float *channel_volumes,
                           int has_ratio, float volume_ratio)
{
    const float sample_rate_f = (float)v->sample_rate;
    const orc_vad_config *config = &v->cfg;

    float min_volume = 999;
    float max_volume = 0;
    for (int c = 0; c < v->n_channels; ++c) {
        const float volume = channel_volumes[c];
        if (volume < min_volume) min_volume = volume;
        if (volume > max_volume) max_volume = volume;
    }

    const uint64_t min_consecutive_to_open =
        (uint64_t)(sample_rate_f * config->min_consecutive_sec_to_open);   /* :161 */
    const uint64_t max_gap_samples = (uint64_t)(sample_rate_f * config->max_speech_gap_sec); /* :163 */

    const double short_term = orc_ra_push(v->short_term, min_volume);       /* :166 */
    const double channel_vol_ratio = orc_ra_push(v->ch_ratio, has_ratio ? volume_ratio : 0); /* :167 */

    /* :169  last_avg orelse initial_long_term_avg orelse short_term */
    double threshold_base;
    if (!orc_ra_last_avg(v->long_term, &threshold_base)) {
        threshold_base = config->has_initial_long_term_avg ? config->initial_long_term_avg
                                                           : short_term;
    }
    /* f64 * f32: Zig peer-type resolution widens the f32 factor to f64 (:170) */
    const double threshold = threshold_base * (double)config->speech_threshold_factor;
    const int threshold_met = short_term > threshold &&
                              channel_vol_ratio > (double)config->channel_vol_ratio_threshold; /* :171 */

    if (!threshold_met) orc_ra_push(v->long_term, min_volume); /* :176-178 */

    orc_vad_result result = { ORC_REC_NONE, 0 };
    const int from_state = v->state;

    switch (v->state) { /* :189-233 */
    case ST_CLOSED:
        if (threshold_met) {
            v->state = ST_OPENING;
            v->speech_start_index = index;
            v->has_start = 1;
        }
        break;
    case ST_OPENING: {
        const uint64_t samples_since_opening = index - v->speech_start_index;
        const int opening_duration_met = samples_since_opening >= min_consecutive_to_open;
        if (threshold_met && opening_duration_met) {
            v->state = ST_OPEN;
            result.recording_state = ORC_REC_STARTED;
            result.sample_number = offset_recording_start(v, v->speech_start_index);
        } else if (!threshold_met) {
            v->state = ST_CLOSED;
        }
        break;
    }
    case ST_OPEN:
        if (!threshold_met) {
            v->state = ST_CLOSING;
            v->speech_end_index = index;
            v->has_end = 1;
        }
        break;
    case ST_CLOSING: {
        const uint64_t samples_since_closing = index - v->speech_end_index;
        const int closing_duration_met = samples_since_closing >= max_gap_samples;
        if (threshold_met) {
            v->state = ST_OPEN;
        } else if (closing_duration_met) {
            v->state = ST_CLOSED;
            result = on_speech_end(v);
        }
        break;
    }
    }

    const int to_state = v->state;
    track_speech_stats(v, has_ratio, volume_ratio, threshold_met, from_state, to_state); /* :236 */

    if (v->n_trace == v->cap_trace) {
        v->cap_trace = v->cap_trace ? v->cap_trace * 2 : 1024;
        v->trace = (orc_vad_trace *)realloc(v->trace, sizeof(orc_vad_trace) * v->cap_trace);
    }
    orc_vad_trace *tr = &v->trace[v->n_trace++];
    tr->index = index;
    tr->min_volume = min_volume;
    tr->short_term = short_term;
    tr->channel_vol_ratio = channel_vol_ratio;
    tr->threshold = threshold;
    tr->threshold_met = threshold_met;
    tr->state_after = to_state;
    return result;
}

size_t orc_vad_n_segments(const orc_vad *v) { return v->n_segments; }
const orc_speech_segment *orc_vad_segments(const orc_vad *v) { return v->segments; }
size_t orc_vad_n_trace(const orc_vad *v) { return v->n_trace; }
const orc_vad_trace *orc_vad_traces(const orc_vad *v) { return v->trace; }
