// host_stats.cpp -- Evaluator + statistics + Audacity label parsing (host).
// Mirrors src/Evaluator.zig:90-156, src/Evaluator/SpeechSegment.zig, src/Evaluator/statistics.zig
// and src/Evaluator/formats.zig:7-36 of the reference; all f32, the reference's summation order.
// The per-stream SingleStats (11 floats: fvad_single_stats) is what ranks exchange in the multi-GPU run; the
// aggregate is then formed on rank 0 in plan order (statistics.zig:124-129).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fvad.h"

namespace {

inline float duration(const fvad_segment_sec& s) { return s.to_sec - s.from_sec; } // SpeechSegment.zig:18-20

inline float overlap_with(const fvad_segment_sec& a, const fvad_segment_sec& b) // :22-27
{
    const float max_from = std::max(a.from_sec, b.from_sec);
    const float min_to = std::min(a.to_sec, b.to_sec);
    return min_to - max_from;
}

// SpeechSegment.findOverlapping (:41-52): strictly positive overlap
std::vector<fvad_segment_sec> find_overlapping(const fvad_segment_sec& t, const std::vector<fvad_segment_sec>& others)
{
    std::vector<fvad_segment_sec> out;
    for (const auto& o : others)
        if (overlap_with(t, o) > 0.0f) out.push_back(o);
    return out;
}

// statistics.extrudeSegments (statistics.zig:229-256), on a clone
void extrude(std::vector<fvad_segment_sec>& c, const fvad_stat_config& cfg)
{
    if (c.empty()) return;
    c.front().from_sec -= cfg.extrude_start;
    c.back().to_sec += cfg.extrude_end;
    for (size_t i = 0; i + 1 < c.size(); ++i)
        if (c[i + 1].from_sec - c[i].to_sec <= cfg.fill_gaps) c[i].to_sec = c[i + 1].from_sec;
}

// statistics.calcFalsePositiveSec (:191-203)
float false_positive_sec(const fvad_segment_sec& vad, std::vector<fvad_segment_sec> matched, const fvad_stat_config& cfg)
{
    extrude(matched, cfg);
    float overlap = 0.0f; // calcOverlapMany :280-284
    for (const auto& o : matched) overlap += std::max(0.0f, overlap_with(vad, o));
    return std::max(0.0f, duration(vad) - overlap);
}

float f_score(float beta, float precision, float recall) // :175-177
{
    const float b2 = beta * beta;
    return (1 + b2) * (precision * recall) / (b2 * precision + recall);
}
float fm_index(float precision, float recall) { return std::sqrt(precision * recall); } // :180-182

} // namespace

extern "C" {

fvad_segment_sec fvad_segment_to_sec(const fvad_speech_segment* s, size_t sample_rate)
{
    // SimulationInstance.zig:237-238: @floatFromInt(u64) / @floatFromInt(usize), both f32
    fvad_segment_sec r;
    r.from_sec = (float)s->sample_from / (float)sample_rate;
    r.to_sec = (float)s->sample_to / (float)sample_rate;
    return r;
}

int fvad_stats_from_segments(const fvad_segment_sec* vad_in, size_t n_vad, const fvad_segment_sec* ref_in,
                             size_t n_ref, const fvad_stat_config* cfg, fvad_single_stats* out)
{
    if (!cfg || !out || (n_vad && !vad_in) || (n_ref && !ref_in)) return FVAD_ERR_INVALID_ARGUMENT;
    // Evaluator.initAndRun: copies, stable sort by start (std.mem.sort is stable), Evaluator.zig:95-111
    std::vector<fvad_segment_sec> vad(vad_in, vad_in + n_vad), ref(ref_in, ref_in + n_ref);
    auto by_start = [](const fvad_segment_sec& a, const fvad_segment_sec& b) { return a.from_sec < b.from_sec; };
    std::stable_sort(vad.begin(), vad.end(), by_start);
    std::stable_sort(ref.begin(), ref.end(), by_start);

    fvad_single_stats st;
    memset(&st, 0, sizeof st);
    for (const auto& seg : vad) { // statistics.zig:88-94
        const auto matched = find_overlapping(seg, ref);
        st.false_positives_sec += false_positive_sec(seg, matched, *cfg);
        const float fp = false_positive_sec(seg, matched, *cfg);     // calcTruePositiveSec :205-214
        const float tp = std::max(0.0f, duration(seg) - fp);
        st.true_positives_sec += tp;
        st.total_positives_sec += tp;
    }
    for (const auto& r : ref) { // :96-102
        if (duration(r) < cfg->ignore_shorter_than_sec) continue;
        float overlap = 0.0f; // calcOverlapWithMatches :274-278
        for (const auto& o : find_overlapping(r, vad)) overlap += std::max(0.0f, overlap_with(r, o));
        const float fn = std::max(0.0f, duration(r) - overlap);
        st.false_negatives_sec += fn;
        st.total_positives_sec += fn;
    }
    st.true_positive_rate = st.true_positives_sec / st.total_positives_sec;
    st.false_negative_rate = st.false_negatives_sec / st.total_positives_sec;
    st.false_discovery_rate = st.false_positives_sec / (st.false_positives_sec + st.true_positives_sec);
    st.precision = st.true_positives_sec / (st.true_positives_sec + st.false_positives_sec);
    st.f_score_beta = 0.7f;
    st.f_score = f_score(st.f_score_beta, st.precision, st.true_positive_rate);
    st.fm_index = fm_index(st.precision, st.true_positive_rate);
    *out = st;
    return FVAD_OK;
}

int fvad_stats_aggregate(const fvad_single_stats* stats, size_t n, fvad_aggregate_stats* out)
{
    if (!out || (n && !stats)) return FVAD_ERR_INVALID_ARGUMENT;
    fvad_aggregate_stats agg;
    memset(&agg, 0, sizeof agg);
    fvad_agg_stat* all[4] = {&agg.true_positive_rate, &agg.false_negative_rate, &agg.false_discovery_rate, &agg.precision};
    for (auto* a : all) { a->min = 2; a->max = -2; } // statistics.zig:57-66
    float sum[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; ++i) { // :124-153, slice order
        const fvad_single_stats& s = stats[i];
        agg.total_positives_sec += s.total_positives_sec;
        agg.true_positives_sec += s.true_positives_sec;
        agg.false_positives_sec += s.false_positives_sec;
        agg.false_negatives_sec += s.false_negatives_sec;
        const float v[4] = {s.true_positive_rate, s.false_negative_rate, s.false_discovery_rate, s.precision};
        for (int k = 0; k < 4; ++k) {
            sum[k] += v[k];
            if (v[k] < all[k]->min) all[k]->min = v[k];
            if (v[k] > all[k]->max) all[k]->max = v[k];
        }
    }
    const float n_stats_f = (float)n;
    agg.true_positive_rate.overall = agg.true_positives_sec / agg.total_positives_sec;
    agg.false_negative_rate.overall = agg.false_negatives_sec / agg.total_positives_sec;
    agg.false_discovery_rate.overall = agg.false_positives_sec / (agg.false_positives_sec + agg.true_positives_sec);
    agg.precision.overall = agg.true_positives_sec / (agg.true_positives_sec + agg.false_positives_sec);
    for (int k = 0; k < 4; ++k) all[k]->avg = sum[k] / n_stats_f;
    agg.f_score_beta = 0.7f;
    agg.f_score = f_score(agg.f_score_beta, agg.precision.overall, agg.true_positive_rate.overall);
    agg.fm_index = fm_index(agg.precision.overall, agg.true_positive_rate.overall);
    *out = agg;
    return FVAD_OK;
}

// formats.parseAudacitySegments (formats.zig:7-36): lines "from\tto\tlabel"; lines with fewer than
// two tab-separated fields are skipped; a field that is not a float is an error.  The reference
// parses the ORIGINAL text (not its CR-stripped copy, formats.zig:11-14), so a trailing '\r' only
// matters if it lands inside one of the first two fields.
int fvad_parse_audacity(const char* txt, size_t len, fvad_segment_sec* out, size_t cap, size_t* n)
{
    if (!txt || !n) return FVAD_ERR_INVALID_ARGUMENT;
    size_t count = 0;
    size_t pos = 0;
    while (pos <= len) {
        size_t eol = pos;
        while (eol < len && txt[eol] != '\n') ++eol;
        const std::string line(txt + pos, eol - pos);
        pos = eol + 1;
        const size_t t1 = line.find('\t');
        if (t1 == std::string::npos) { if (eol >= len) break; continue; }
        size_t t2 = line.find('\t', t1 + 1);
        if (t2 == std::string::npos) t2 = line.size();
        const std::string a = line.substr(0, t1), b = line.substr(t1 + 1, t2 - t1 - 1);
        char* e1 = nullptr; char* e2 = nullptr;
        const float from = strtof(a.c_str(), &e1), to = strtof(b.c_str(), &e2);
        if (a.empty() || b.empty() || *e1 != '\0' || *e2 != '\0') return FVAD_ERR_INVALID_ARGUMENT;
        if (out && count < cap) { out[count].from_sec = from; out[count].to_sec = to; }
        ++count;
        if (eol >= len) break;
    }
    *n = count;
    return (out && count > cap) ? FVAD_ERR_BUFFER_TOO_SMALL : FVAD_OK;
}

} // extern "C"
