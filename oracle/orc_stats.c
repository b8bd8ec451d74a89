/*
 * orc_stats.c -- ORACLE (test infrastructure only; see orc.h).
 * Restates src/Evaluator.zig:90-156 (initAndRun), src/Evaluator/SpeechSegment.zig and
 * src/Evaluator/statistics.zig.  All f32, reference summation order.
 */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float from_sec, to_sec;
    size_t *opp; /* indices into the opposite (sorted) array */
    size_t n_opp;
} eseg;

static float seg_duration(float from, float to) { return to - from; } /* SpeechSegment.zig:18-20 */

/* SpeechSegment.zig:22-27 */
static float overlap_with(float a_from, float a_to, float b_from, float b_to)
{
    const float max_from = a_from > b_from ? a_from : b_from; /* @max */
    const float min_to = a_to < b_to ? a_to : b_to;           /* @min */
    return min_to - max_from;
}

/* std.mem.sort is a stable insertion/block sort; equal keys keep their order
 * (Evaluator.zig:110-111, SpeechSegment.zig:54-57 sortByStart: lhs.from < rhs.from) */
static void stable_sort_by_start(orc_seg_sec *s, size_t n)
{
    for (size_t i = 1; i < n; ++i) {
        orc_seg_sec key = s[i];
        size_t j = i;
        while (j > 0 && key.from_sec < s[j - 1].from_sec) {
            s[j] = s[j - 1];
            --j;
        }
        s[j] = key;
    }
}

/* SpeechSegment.zig:41-52 findOverlapping: overlap > 0 strictly */
static void find_overlapping(eseg *t, const orc_seg_sec *others, size_t n_others)
{
    t->opp = (size_t *)malloc(sizeof(size_t) * (n_others ? n_others : 1));
    t->n_opp = 0;
    for (size_t i = 0; i < n_others; ++i)
        if (overlap_with(t->from_sec, t->to_sec, others[i].from_sec, others[i].to_sec) > 0.0f)
            t->opp[t->n_opp++] = i;
}

/* statistics.zig:229-256 extrudeSegments on a clone of the matched refs */
static void extrude_segments(orc_seg_sec *cloned, size_t n, const orc_stat_config *cfg)
{
    if (n == 0) return;
    cloned[0].from_sec -= cfg->extrude_start;
    cloned[n - 1].to_sec += cfg->extrude_end;
    for (size_t i = 0; i + 1 < n; ++i)
        if (cloned[i + 1].from_sec - cloned[i].to_sec <= cfg->fill_gaps)
            cloned[i].to_sec = cloned[i + 1].from_sec;
}

/* statistics.zig:191-203 */
float orc_calc_false_positive_sec(orc_seg_sec vad, const orc_seg_sec *matched, size_t n,
                                  const orc_stat_config *cfg)
{
    orc_seg_sec *cl = (orc_seg_sec *)malloc(sizeof(orc_seg_sec) * (n ? n : 1));
    memcpy(cl, matched, sizeof(orc_seg_sec) * n);
    extrude_segments(cl, n, cfg);
    float overlap = 0.0f; /* calcOverlapMany, statistics.zig:280-284 */
    for (size_t i = 0; i < n; ++i) {
        const float o = overlap_with(vad.from_sec, vad.to_sec, cl[i].from_sec, cl[i].to_sec);
        overlap += o > 0.0f ? o : 0.0f;
    }
    free(cl);
    const float fp = seg_duration(vad.from_sec, vad.to_sec) - overlap;
    return fp > 0.0f ? fp : 0.0f;
}

static float f_score(float beta, float precision, float recall) /* statistics.zig:175-177 */
{
    const float b2 = beta * beta; /* pow(f32, beta, 2) */
    return (1 + b2) * (precision * recall) / (b2 * precision + recall);
}
static float fm_index(float precision, float recall) { return sqrtf(precision * recall); } /* :180-182 */

orc_single_stats orc_stats_from_segments(const orc_seg_sec *vad_in, size_t n_vad,
                                         const orc_seg_sec *ref_in, size_t n_ref,
                                         const orc_stat_config *cfg)
{
    /* Evaluator.initAndRun: copy, sort both sides by start, match (Evaluator.zig:95-153) */
    orc_seg_sec *vad = (orc_seg_sec *)malloc(sizeof(orc_seg_sec) * (n_vad ? n_vad : 1));
    orc_seg_sec *ref = (orc_seg_sec *)malloc(sizeof(orc_seg_sec) * (n_ref ? n_ref : 1));
    memcpy(vad, vad_in, sizeof(orc_seg_sec) * n_vad);
    memcpy(ref, ref_in, sizeof(orc_seg_sec) * n_ref);
    stable_sort_by_start(vad, n_vad);
    stable_sort_by_start(ref, n_ref);

    orc_single_stats st;
    memset(&st, 0, sizeof(st));

    /* statistics.fromEvaluator, statistics.zig:85-114 */
    for (size_t i = 0; i < n_vad; ++i) {
        eseg e = { vad[i].from_sec, vad[i].to_sec, NULL, 0 };
        find_overlapping(&e, ref, n_ref);
        orc_seg_sec *matched = (orc_seg_sec *)malloc(sizeof(orc_seg_sec) * (e.n_opp ? e.n_opp : 1));
        for (size_t k = 0; k < e.n_opp; ++k) matched[k] = ref[e.opp[k]];
        const float fp = orc_calc_false_positive_sec(vad[i], matched, e.n_opp, cfg);
        st.false_positives_sec += fp;
        /* calcTruePositiveSec (:205-214) recomputes fp and takes max(0, duration - fp) */
        const float fp2 = orc_calc_false_positive_sec(vad[i], matched, e.n_opp, cfg);
        float tp = seg_duration(vad[i].from_sec, vad[i].to_sec) - fp2;
        if (!(tp > 0.0f)) tp = 0.0f;
        st.true_positives_sec += tp;
        st.total_positives_sec += tp;
        free(matched);
        free(e.opp);
    }
    for (size_t i = 0; i < n_ref; ++i) {
        if (seg_duration(ref[i].from_sec, ref[i].to_sec) < cfg->ignore_shorter_than_sec) continue;
        eseg e = { ref[i].from_sec, ref[i].to_sec, NULL, 0 };
        find_overlapping(&e, vad, n_vad);
        float overlap = 0.0f; /* calcOverlapWithMatches, :274-278 */
        for (size_t k = 0; k < e.n_opp; ++k) {
            const float o = overlap_with(ref[i].from_sec, ref[i].to_sec, vad[e.opp[k]].from_sec,
                                         vad[e.opp[k]].to_sec);
            overlap += o > 0.0f ? o : 0.0f;
        }
        float fn = seg_duration(ref[i].from_sec, ref[i].to_sec) - overlap; /* :216-227 */
        if (!(fn > 0.0f)) fn = 0.0f;
        st.false_negatives_sec += fn;
        st.total_positives_sec += fn;
        free(e.opp);
    }
    st.true_positive_rate = st.true_positives_sec / st.total_positives_sec;
    st.false_negative_rate = st.false_negatives_sec / st.total_positives_sec;
    st.false_discovery_rate = st.false_positives_sec / (st.false_positives_sec + st.true_positives_sec);
    st.precision = st.true_positives_sec / (st.true_positives_sec + st.false_positives_sec);
    st.f_score_beta = 0.7f;
    st.f_score = f_score(st.f_score_beta, st.precision, st.true_positive_rate);
    st.fm_index = fm_index(st.precision, st.true_positive_rate);
    free(vad);
    free(ref);
    return st;
}

/* statistics.zig:116-172: in-order f32 sums; min starts at 2, max at -2 (:57-66) */
static void agg_update(orc_agg_stat *a, float v)
{
    if (v < a->min) a->min = v;
    if (v > a->max) a->max = v;
}

orc_aggregate_stats orc_stats_aggregate(const orc_single_stats *stats, size_t n)
{
    orc_aggregate_stats agg;
    memset(&agg, 0, sizeof(agg));
    orc_agg_stat *all[4] = { &agg.true_positive_rate, &agg.false_negative_rate,
                             &agg.false_discovery_rate, &agg.precision };
    for (int i = 0; i < 4; ++i) { all[i]->min = 2; all[i]->max = -2; }
    float sum_tpr = 0, sum_fnr = 0, sum_fdr = 0, sum_ppv = 0;
    for (size_t i = 0; i < n; ++i) {
        const orc_single_stats *s = &stats[i];
        agg.total_positives_sec += s->total_positives_sec;
        agg.true_positives_sec += s->true_positives_sec;
        agg.false_positives_sec += s->false_positives_sec;
        agg.false_negatives_sec += s->false_negatives_sec;
        sum_tpr += s->true_positive_rate;   agg_update(&agg.true_positive_rate, s->true_positive_rate);
        sum_fnr += s->false_negative_rate;  agg_update(&agg.false_negative_rate, s->false_negative_rate);
        sum_fdr += s->false_discovery_rate; agg_update(&agg.false_discovery_rate, s->false_discovery_rate);
        sum_ppv += s->precision;            agg_update(&agg.precision, s->precision);
    }
    const float n_stats_f = (float)n;
    agg.true_positive_rate.overall = agg.true_positives_sec / agg.total_positives_sec;
    agg.false_negative_rate.overall = agg.false_negatives_sec / agg.total_positives_sec;
    agg.false_discovery_rate.overall = agg.false_positives_sec / (agg.false_positives_sec + agg.true_positives_sec);
    agg.precision.overall = agg.true_positives_sec / (agg.true_positives_sec + agg.false_positives_sec);
    agg.true_positive_rate.avg = sum_tpr / n_stats_f;
    agg.false_negative_rate.avg = sum_fnr / n_stats_f;
    agg.false_discovery_rate.avg = sum_fdr / n_stats_f;
    agg.precision.avg = sum_ppv / n_stats_f;
    agg.f_score_beta = 0.7f;
    agg.f_score = f_score(agg.f_score_beta, agg.precision.overall, agg.true_positive_rate.overall);
    agg.fm_index = fm_index(agg.precision.overall, agg.true_positive_rate.overall);
    return agg;
}

/* SimulationInstance.zig:237-238 */
orc_seg_sec orc_segment_to_sec(const orc_speech_segment *s, int sample_rate)
{
    orc_seg_sec r;
    r.from_sec = (float)s->sample_from / (float)sample_rate;
    r.to_sec = (float)s->sample_to / (float)sample_rate;
    return r;
}
