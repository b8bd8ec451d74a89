// host_vad.cpp -- the sequential tail of the path, on the host by design.
//
// Mirrors src/structures/RollingAverage.zig, src/AudioPipeline/VADMetadata.zig and
// src/AudioPipeline/VADMachine.zig of the reference: f64 rolling averages re-summed in index
// order on every push (RollingAverage.zig:45-56), integer sample arithmetic, @intFromFloat
// truncations.  Segment boundaries are integers decided by `short_term > threshold`
// (VADMachine.zig:171), so this code keeps the reference's exact operation order; the GPU only
// supplies the per-frame band sums and per-chunk RMS values that feed it.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "host_vad.h"

namespace fvad {

// ------------------------------------------------------------------ RollingAverage
RollingAverage::RollingAverage(size_t count, bool has_initial, double initial_val)
    : data(count ? count : 1, 0.0), len(count)
{
    if (has_initial) { // RollingAverage.zig:20-26
        std::fill(data.begin(), data.begin() + (long)count, initial_val);
        written_count = count;
        avg();
        if (count) enter_steady();
    }
}

double RollingAverage::avg() // RollingAverage.zig:45-56
{
    double a = 0.0;
    const double scalar = 1.0 / (double)written_count;
    const double* d = data.data();
    for (size_t i = 0; i < written_count; ++i) a += d[i] * scalar;
    last_avg = a;
    has_last_avg = true;
    return a;
}

// The reference re-sums the whole ring on every push (RollingAverage.zig:45-56): a chain of `len`
// dependent f64 adds.  Once the ring is full the terms below the write index have not changed since
// the previous push, so the running sum up to (not including) the write index -- `pref` -- is still
// exactly what the reference's loop would have in its accumulator at that point.  Resuming the
// chain from there performs the very same additions in the very same order for the remaining
// terms: bit-identical result, half the work on average.
double RollingAverage::push(float sample) // RollingAverage.zig:34-43
{
    if (steady) {
        const size_t w = write_idx;
        data[w] = (double)sample;
        q[w] = data[w] * scalar;
        double acc = (w == 0) ? 0.0 : pref;
        acc += q[w];
        const double new_pref = acc;
        const double* qq = q.data();
        for (size_t i = w + 1; i < len; ++i) acc += qq[i];
        last_avg = acc;
        has_last_avg = true;
        write_idx = (w + 1) % len;
        pref = (write_idx == 0) ? 0.0 : new_pref;
        return acc;
    }
    data[write_idx] = (double)sample;
    write_idx = (write_idx + 1) % len;
    if (written_count < len) written_count += 1;
    const double a = avg();
    if (written_count == len && write_idx == 0) enter_steady();
    return a;
}

void RollingAverage::enter_steady()
{
    scalar = 1.0 / (double)len;
    q.resize(len);
    for (size_t i = 0; i < len; ++i) q[i] = data[i] * scalar;
    pref = 0.0;
    for (size_t i = 0; i < write_idx; ++i) pref += q[i]; // same chain the full loop would run
    steady = true;
}

// ------------------------------------------------------------------ VADMetadata
void Metadata::push(const MetaResult& v, float weight) // VADMetadata.zig:29-60
{
    if (v.has_ratio) {
        if (!has_ratio) { has_ratio = true; ratio_sum = 0.0f; ratio_weight = 0.0f; }
        ratio_sum += v.volume_ratio * weight;
        ratio_weight += weight;
    }
    if (v.has_min && (!has_min || v.volume_min < volume_min)) { has_min = true; volume_min = v.volume_min; }
    if (v.has_max && (!has_max || v.volume_max > volume_max)) { has_max = true; volume_max = v.volume_max; }
}

MetaResult Metadata::to_result() const // VADMetadata.zig:16-27
{
    MetaResult r;
    r.has_min = has_min; r.volume_min = volume_min;
    r.has_max = has_max; r.volume_max = volume_max;
    if (has_ratio) { r.has_ratio = true; r.volume_ratio = ratio_sum / ratio_weight; }
    return r;
}

// BufferedVolumeAnalyzer.analyseVolume (BufferedVolumeAnalyzer.zig:48-69) from per-channel RMS
MetaResult analyse_volume(const float* channel_rms, size_t n_channels)
{
    float vol_min = 1, vol_max = 0;
    for (size_t c = 0; c < n_channels; ++c) {
        const float vol = channel_rms[c];
        if (vol < vol_min) vol_min = vol;
        if (vol > vol_max) vol_max = vol;
    }
    MetaResult r;
    r.has_ratio = r.has_min = r.has_max = true;
    r.volume_ratio = (vol_max == 0) ? 0 : vol_min / vol_max;
    r.volume_min = vol_min;
    r.volume_max = vol_max;
    return r;
}

// ------------------------------------------------------------------ VADMachine
VadMachine::VadMachine(const fvad_vad_config& c, size_t sample_rate_, size_t n_channels_, size_t fft_size_)
    : cfg(c), sample_rate(sample_rate_), n_channels(n_channels_), fft_size(fft_size_),
      long_term(1, false, 0), short_term(1, false, 0), ch_ratio(1, false, 0)
{
    // VADMachine.zig:75-106
    const float sample_rate_f = (float)sample_rate;
    const float fft_size_f = (float)fft_size;
    const float eval_per_sec = sample_rate_f / fft_size_f;
    size_t long_len = (size_t)(eval_per_sec * cfg.long_term_speech_avg_sec);
    size_t short_len = (size_t)(eval_per_sec * cfg.short_term_speech_avg_sec);
    const size_t ratio_len = (size_t)(eval_per_sec * cfg.channel_vol_ratio_avg_sec);
    long_len = std::max<size_t>(1, long_len);
    short_len = std::max<size_t>(1, short_len);
    long_term = RollingAverage(long_len, cfg.has_initial_long_term_avg != 0, cfg.initial_long_term_avg);
    short_term = RollingAverage(short_len, false, 0);
    ch_ratio = RollingAverage(ratio_len, false, 0);
    segments.reserve(100); // :111
    audit.min_rel_threshold_margin = INFINITY;
    audit.min_abs_ratio_margin = INFINITY;
    audit.n_frames = 0;
    const char* eager = getenv("FVAD_VAD_EAGER");
    lt_lazy = !(eager && eager[0] == '1');
}

uint64_t VadMachine::offset_start(uint64_t vad_from) const // :312-317
{
    const uint64_t start_buffer = (uint64_t)((float)sample_rate * 2);
    return vad_from - std::min(start_buffer, vad_from);
}
uint64_t VadMachine::offset_end(uint64_t vad_to) const // :320-325
{
    const uint64_t end_buffer = (uint64_t)((float)sample_rate * 2);
    return vad_to + end_buffer;
}

fvad_vad_result VadMachine::on_speech_end() // :265-309
{
    const float sample_rate_f = (float)sample_rate;
    const uint64_t sample_from = speech_start_index, sample_to = speech_end_index;
    const uint64_t length_samples = sample_to - sample_from;
    const float length_sec = (float)length_samples / sample_rate_f;
    const bool speech_duration_met = length_sec >= cfg.min_vad_duration_sec;
    const float avg_ratio = channel_vol_ratio_sum / (float)channel_vol_ratio_count;
    if (speech_duration_met) {
        fvad_speech_segment s;
        s.sample_from = offset_start(sample_from);
        s.sample_to = offset_end(sample_to);
        s.avg_channel_vol_ratio = avg_ratio;
        s.vad_met_sec = vad_threshold_met_cumulative_sec;
        segments.push_back(s);
        return {FVAD_REC_COMPLETED, offset_end(sample_to)};
    }
    return {FVAD_REC_ABORTED, 0};
}

fvad_vad_result VadMachine::finish_step(uint64_t index, bool threshold_met, bool has_ratio, float ratio)
{
    const float sample_rate_f = (float)sample_rate;
    const uint64_t min_consecutive_to_open = (uint64_t)(sample_rate_f * cfg.min_consecutive_sec_to_open); // :161
    const uint64_t max_gap_samples = (uint64_t)(sample_rate_f * cfg.max_speech_gap_sec);                 // :163
    fvad_vad_result result = {FVAD_REC_NONE, 0};
    const State from_state = state;
    switch (state) { // :189-233
    case CLOSED:
        if (threshold_met) { state = OPENING; speech_start_index = index; }
        break;
    case OPENING: {
        const uint64_t since = index - speech_start_index;
        const bool met = since >= min_consecutive_to_open;
        if (threshold_met && met) {
            state = OPEN;
            result = {FVAD_REC_STARTED, offset_start(speech_start_index)};
        } else if (!threshold_met) {
            state = CLOSED;
        }
        break;
    }
    case OPEN:
        if (!threshold_met) { state = CLOSING; speech_end_index = index; }
        break;
    case CLOSING: {
        const uint64_t since = index - speech_end_index;
        const bool met = since >= max_gap_samples;
        if (threshold_met) state = OPEN;
        else if (met) { state = CLOSED; result = on_speech_end(); }
        break;
    }
    }
    // trackSpeechStats, :241-263
    const float input_length_sec = (float)fft_size / sample_rate_f;
    const float r = has_ratio ? ratio : 0;
    if (from_state == CLOSED && state == OPENING) {
        channel_vol_ratio_sum = r;
        channel_vol_ratio_count = 1;
        vad_threshold_met_cumulative_sec = input_length_sec;
    } else if (from_state == OPEN) {
        channel_vol_ratio_sum += r;
        channel_vol_ratio_count += 1;
        if (threshold_met) vad_threshold_met_cumulative_sec += input_length_sec;
    }
    return result;
}

// ---- lazily exact long-term average
// The long-term average feeds exactly one thing: the comparison `short_term > long_term * factor`
// (VADMachine.zig:169-171; plus this build's margin audit).  Re-running the reference's chain of `len`
// dependent f64 adds on every push (RollingAverage.zig:45-56) makes a stream cost ~4.5 us per frame and a
// two-hour stream three seconds, however many cores there are.  Instead the machine keeps
//   lt_approx = the chain's last exact value, updated as fl(fl(lt_approx + q_new) - q_old) per push,
//   lt_err    = a running bound on the rounding error of those updates,
//   lt_abs    = (approximately) sum |q_i|,
// and bounds the distance to what the chain would return *now*:
//   |lt_approx - chain| <= gamma_N sum|q_i|(anchor) + lt_err + gamma_N sum|q_i|(now)
// (gamma_N = N u / (1 - N u), u = 2^-53: the chain's error against the real-number sum when lt_approx was
// anchored on it, the updates' rounding, the chain's own error now).
// decide() evaluates the comparison with the threshold interval this gives; only if `short_term` falls
// inside the interval, or the frame could lower the audit's minimum margin, is the chain run for real
// (long_term_exact: the reference's additions in the reference's order).  Every decision and every
// audited number is therefore the one the eager evaluation produces; the tests compare whole runs
// bit for bit with the oracle.
static constexpr double kU = 1.1102230246251565e-16; // 2^-53

void VadMachine::long_term_exact()
{
    RollingAverage& a = long_term;
    const double* qq = a.q.data();
    double acc = 0.0, pref = 0.0, abs_sum = 0.0;
    for (size_t i = 0; i < a.len; ++i) {
        if (i == a.write_idx) pref = acc;
        acc += qq[i]; // == a += data[i] * scalar (RollingAverage.zig:50-53), products cached in q
        abs_sum += std::fabs(qq[i]);
    }
    a.last_avg = acc;
    a.has_last_avg = true;
    a.pref = (a.write_idx == 0) ? 0.0 : pref;
    lt_approx = acc;
    lt_abs = abs_sum;
    lt_abs_anchor = abs_sum;
    lt_anchored = true;
    lt_err = 0.0;
    lt_stale = false;
    lt_updates = 0;
    ++lt_exact_evals;
}

void VadMachine::long_term_push(float mv) // RollingAverage.push for the long-term ring
{
    RollingAverage& a = long_term;
    if (!a.steady || !lt_lazy) { // ring not full yet (or eager mode): the reference's path as is
        const bool was_steady = a.steady;
        a.push(mv);
        if (lt_lazy && !was_steady && a.steady) long_term_exact();
        return;
    }
    if (!lt_anchored) long_term_exact(); // first lazy push: anchor on the chain's current value
    const size_t w = a.write_idx;
    const double qn = (double)mv * a.scalar, qo = a.q[w];
    a.data[w] = (double)mv;
    a.q[w] = qn;
    a.write_idx = (w + 1) % a.len;
    const double s1 = lt_approx + qn, s2 = s1 - qo;
    lt_err += 2.0 * kU * (std::fabs(s1) + std::fabs(s2)); // each rounding <= u |result|; doubled
    lt_abs += std::fabs(qn) - std::fabs(qo);
    lt_approx = s2;
    lt_stale = true;
    a.has_last_avg = true;
    ++lt_lazy_pushes;
    if (++lt_updates >= 4096) long_term_exact(); // keep the bound tight
}

bool VadMachine::decide(double short_term_avg, double ratio_avg, double* threshold_out)
{
    const double f = (double)cfg.speech_threshold_factor;
    if (long_term.steady && lt_stale && threshold_out) long_term_exact();
    if (long_term.steady && lt_stale) {
        const double n = (double)long_term.len;
        const double gamma = n * kU / (1.0 - n * kU);
        // lt_abs is itself updated in floating point: widen it by its own drift
        const double abs_now = std::fabs(lt_abs) * (1.0 + 1e-9) + 8192.0 * 2.0 * kU * (std::fabs(lt_abs) + lt_abs_anchor);
        const double delta = lt_err + 2.0 * gamma * (abs_now + lt_abs_anchor);
        double t0 = (lt_approx - delta) * f, t1 = (lt_approx + delta) * f;
        if (t0 > t1) std::swap(t0, t1);
        const double lo = t0 - std::fabs(t0) * 4.0 * kU - 1e-300, hi = t1 + std::fabs(t1) * 4.0 * kU + 1e-300;
        const bool sure_true = short_term_avg > hi, sure_false = short_term_avg <= lo;
        bool need_exact = !(sure_true || sure_false);
        if (!need_exact && hi > 0) {
            // smallest relative margin |st - thr| / thr any threshold in [lo, hi] could give
            const double gap = sure_true ? short_term_avg - hi : lo - short_term_avg;
            const double m_lb = gap / (sure_true ? hi : std::max(lo, hi));
            if (!(m_lb * (1.0 - 1e-9) > audit.min_rel_threshold_margin)) need_exact = true;
        }
        if (!need_exact) {
            const double rm = std::fabs(ratio_avg - (double)cfg.channel_vol_ratio_threshold);
            if (rm < audit.min_abs_ratio_margin) audit.min_abs_ratio_margin = rm;
            audit.n_frames++;
            return sure_true && ratio_avg > (double)cfg.channel_vol_ratio_threshold;
        }
        long_term_exact();
    }
    double base; // :169  last_avg orelse initial_long_term_avg orelse short_term
    if (long_term.has_last_avg) base = long_term.last_avg;
    else if (cfg.has_initial_long_term_avg) base = cfg.initial_long_term_avg;
    else base = short_term_avg;
    const double threshold = base * f; // :170
    const bool met = short_term_avg > threshold && ratio_avg > (double)cfg.channel_vol_ratio_threshold; // :171
    // margin audit: how close was this frame to flipping?
    if (threshold > 0) {
        const double m = std::fabs(short_term_avg - threshold) / threshold;
        if (m < audit.min_rel_threshold_margin) audit.min_rel_threshold_margin = m;
    }
    const double rm = std::fabs(ratio_avg - (double)cfg.channel_vol_ratio_threshold);
    if (rm < audit.min_abs_ratio_margin) audit.min_abs_ratio_margin = rm;
    audit.n_frames++;
    if (threshold_out) *threshold_out = threshold;
    return met;
}

float VadMachine::min_volume(const float* channel_volumes) const // :153-158
{
    float min_v = 999, max_v = 0;
    for (size_t c = 0; c < n_channels; ++c) {
        const float v = channel_volumes[c];
        if (v < min_v) min_v = v;
        if (v > max_v) max_v = v;
    }
    (void)max_v;
    return min_v;
}

fvad_vad_result VadMachine::run(uint64_t index, const float* channel_volumes, bool has_ratio, float ratio)
{
    const float mv = min_volume(channel_volumes);
    const double st = short_term.push(mv);                       // :166
    const double cr = ch_ratio.push(has_ratio ? ratio : 0);      // :167
    const bool met = decide(st, cr, nullptr);
    if (!met) long_term_push(mv);                                // :176-178
    return finish_step(index, met, has_ratio, ratio);
}

// ------------------------------------------------------------------ many streams
// Streams are independent (one pipeline per file, simulator.zig:225-231); with the lazily exact
// long-term average a frame costs ~0.1 us, so the streams are simply dealt to threads.
void run_many(VadMachine* const* vads, size_t n_streams, const float* const* band,
              const float* const* ratio, const size_t* n_frames, size_t n_channels,
              const uint64_t* first_index, size_t fft_size, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    auto run_stream = [&](size_t s) {
        VadMachine* m = vads[s];
        for (size_t k = 0; k < n_frames[s]; ++k) {
            const float r = ratio[s][k];
            const bool has_ratio = !std::isnan(r);
            m->run(first_index[s] + (uint64_t)k * fft_size, band[s] + k * n_channels, has_ratio, r);
        }
    };
    if (n_threads == 1 || n_streams <= 1) {
        for (size_t s = 0; s < n_streams; ++s) run_stream(s);
        return;
    }
    std::vector<std::thread> th;
    std::atomic<size_t> next{0};
    const int nt = (int)std::min<size_t>((size_t)n_threads, n_streams);
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n_streams) break;
                run_stream(i);
            }
        });
    for (auto& t : th) t.join();
}

} // namespace fvad

// ------------------------------------------------------------------ C ABI
struct fvad_vad { fvad::VadMachine m; fvad_vad(const fvad_vad_config& c, size_t sr, size_t nc, size_t fs) : m(c, sr, nc, fs) {} };
struct fvad_rolling_average { fvad::RollingAverage ra; fvad_rolling_average(size_t n, bool h, double v) : ra(n, h, v) {} };

extern "C" {

void fvad_vad_config_default(fvad_vad_config* c)
{
    c->speech_min_freq = 500; c->speech_max_freq = 2000;
    c->long_term_speech_avg_sec = 180; c->has_initial_long_term_avg = 1; c->initial_long_term_avg = 0.005;
    c->short_term_speech_avg_sec = 0.2f; c->speech_threshold_factor = 10;
    c->channel_vol_ratio_avg_sec = 0.5f; c->channel_vol_ratio_threshold = 0.5f;
    c->min_consecutive_sec_to_open = 0.2f; c->max_speech_gap_sec = 2; c->min_vad_duration_sec = 0.7f;
}

int fvad_vad_create(const fvad_vad_config* cfg, size_t sample_rate, size_t n_channels, size_t fft_size, fvad_vad** out)
{
    if (!cfg || !out || n_channels == 0 || fft_size == 0 || sample_rate == 0) return FVAD_ERR_INVALID_ARGUMENT;
    // a zero-length channel_vol_ratio ring would divide by zero in the reference (RollingAverage.zig:36)
    if ((size_t)(((float)sample_rate / (float)fft_size) * cfg->channel_vol_ratio_avg_sec) == 0) return FVAD_ERR_INVALID_ARGUMENT;
    *out = new (std::nothrow) fvad_vad(*cfg, sample_rate, n_channels, fft_size);
    return *out ? FVAD_OK : FVAD_ERR_ALLOC_FAILED;
}
void fvad_vad_destroy(fvad_vad* v) { delete v; }

int fvad_vad_run(fvad_vad* v, uint64_t index, const float* channel_volumes, int has_ratio, float volume_ratio, fvad_vad_result* out)
{
    if (!v || !channel_volumes) return FVAD_ERR_INVALID_ARGUMENT;
    const fvad_vad_result r = v->m.run(index, channel_volumes, has_ratio != 0, volume_ratio);
    if (out) *out = r;
    return FVAD_OK;
}
size_t fvad_vad_segment_count(const fvad_vad* v) { return v ? v->m.segments.size() : 0; }
int fvad_vad_segments(const fvad_vad* v, fvad_speech_segment* out, size_t cap, size_t* n)
{
    if (!v || !n) return FVAD_ERR_INVALID_ARGUMENT;
    *n = v->m.segments.size();
    if (cap < *n) return FVAD_ERR_BUFFER_TOO_SMALL;
    if (*n) memcpy(out, v->m.segments.data(), *n * sizeof(fvad_speech_segment));
    return FVAD_OK;
}
int fvad_vad_lazy_stats(const fvad_vad* v, uint64_t* exact_evaluations, uint64_t* lazy_pushes)
{
    if (!v) return FVAD_ERR_INVALID_ARGUMENT;
    if (exact_evaluations) *exact_evaluations = v->m.lt_exact_evals;
    if (lazy_pushes) *lazy_pushes = v->m.lt_lazy_pushes;
    return FVAD_OK;
}

int fvad_vad_audit_get(const fvad_vad* v, fvad_vad_audit* out)
{
    if (!v || !out) return FVAD_ERR_INVALID_ARGUMENT;
    *out = v->m.audit;
    return FVAD_OK;
}

int fvad_vad_run_many(fvad_vad* const* vads, size_t n_streams, const float* const* band, const float* const* ratio,
                      const size_t* n_frames, size_t n_channels, const uint64_t* first_index, size_t fft_size, int n_threads)
{
    if (!vads || !band || !ratio || !n_frames || !first_index) return FVAD_ERR_INVALID_ARGUMENT;
    std::vector<fvad::VadMachine*> ms(n_streams);
    for (size_t i = 0; i < n_streams; ++i) {
        if (!vads[i] || vads[i]->m.n_channels != n_channels) return FVAD_ERR_CHANNEL_COUNT_MISMATCH;
        ms[i] = &vads[i]->m;
    }
    fvad::run_many(ms.data(), n_streams, band, ratio, n_frames, n_channels, first_index, fft_size, n_threads);
    return FVAD_OK;
}

int fvad_ra_create(size_t count, int has_initial, double initial_val, fvad_rolling_average** out)
{
    if (!out || count == 0) return FVAD_ERR_INVALID_ARGUMENT;
    *out = new (std::nothrow) fvad_rolling_average(count, has_initial != 0, initial_val);
    return *out ? FVAD_OK : FVAD_ERR_ALLOC_FAILED;
}
void fvad_ra_destroy(fvad_rolling_average* ra) { delete ra; }
double fvad_ra_push(fvad_rolling_average* ra, float sample) { return ra->ra.push(sample); }
int fvad_ra_last_avg(const fvad_rolling_average* ra, double* out)
{
    if (ra->ra.has_last_avg && out) *out = ra->ra.last_avg;
    return ra->ra.has_last_avg ? 1 : 0;
}

// ------------------------------------------------------------------ host stage for a whole batch
// What VADPipeline does between the kernels' outputs and the segment list, for many streams at once and
// straight from the engine's lane-major buffers: per-chunk volume ratio (BufferedVolumeAnalyzer.zig:48-69) ->
// the two metadata hand-overs (BufferedVolumeAnalyzer.zig:33-45, BufferedDenoiser.zig:83-86,115) -> the
// sample-weighted ratio of every FFT frame (BufferedFFT.zig:137-140,153) -> VADMachine.run per frame
// (VADMachine.zig:138-239).  Streams are dealt to threads like simulator.zig:221-232 deals files.
struct fvad_vad_batch {
    fvad_vad_config cfg;
    size_t sample_rate, n_channels, fft_size, n_streams;
    std::vector<std::vector<fvad_speech_segment>> segs;
    std::vector<fvad_vad_audit> audits;
    // a run in parts (fvad_vad_batch_run_part): the streams' machines live on between the parts
    std::vector<std::unique_ptr<fvad::VadMachine>> machines;
    uint64_t next_frame = 0;
};

int fvad_vad_batch_create(const fvad_vad_config* cfg, size_t sample_rate, size_t n_channels, size_t fft_size, size_t n_streams,
                          fvad_vad_batch** out)
{
    if (!cfg || !out || n_channels == 0 || fft_size == 0 || sample_rate == 0 || n_streams == 0) return FVAD_ERR_INVALID_ARGUMENT;
    if ((size_t)(((float)sample_rate / (float)fft_size) * cfg->channel_vol_ratio_avg_sec) == 0) return FVAD_ERR_INVALID_ARGUMENT;
    auto* b = new (std::nothrow) fvad_vad_batch();
    if (!b) return FVAD_ERR_ALLOC_FAILED;
    b->cfg = *cfg; b->sample_rate = sample_rate; b->n_channels = n_channels; b->fft_size = fft_size; b->n_streams = n_streams;
    b->segs.resize(n_streams);
    b->audits.resize(n_streams);
    *out = b;
    return FVAD_OK;
}
void fvad_vad_batch_destroy(fvad_vad_batch* b) { delete b; }

int fvad_vad_batch_run_part(fvad_vad_batch* b, const float* band, size_t band_stride, size_t n_frames, const float* chunk_rms,
                            size_t rms_stride, size_t n_chunks, size_t chunk_size, uint64_t first_frame, int n_threads)
{
    if (!b || (n_frames && !band) || (n_chunks && !chunk_rms) || chunk_size == 0) return FVAD_ERR_INVALID_ARGUMENT;
    // parts follow each other without gaps, and a part starts where a chunk starts (its first chunk is chunk_rms' first column)
    if (first_frame != 0 && (first_frame != b->next_frame || b->machines.size() != b->n_streams)) return FVAD_ERR_INVALID_ARGUMENT;
    if ((first_frame * b->fft_size) % chunk_size) return FVAD_ERR_INVALID_ARGUMENT;
    const uint64_t first_chunk = first_frame * b->fft_size / chunk_size;
    if ((first_frame + n_frames) * b->fft_size > (first_chunk + n_chunks) * chunk_size) return FVAD_ERR_INVALID_ARGUMENT; // a frame without its chunk's ratio
    const size_t C = b->n_channels;
    if (first_frame == 0) { // fresh machines (VADMachine.init per pipeline, VADPipeline.zig:60-75)
        b->machines.clear();
        for (size_t s = 0; s < b->n_streams; ++s) b->machines.emplace_back(new fvad::VadMachine(b->cfg, b->sample_rate, C, b->fft_size));
    }
    auto run_stream = [&](size_t s) {
        fvad::VadMachine& m = *b->machines[s];
        std::vector<float> ratio(n_chunks), ch(C), vols(C);
        for (size_t k = 0; k < n_chunks; ++k) {
            for (size_t c = 0; c < C; ++c) ch[c] = chunk_rms[(s * C + c) * rms_stride + k];
            const fvad::MetaResult va = fvad::analyse_volume(ch.data(), C);
            fvad::Metadata m1; m1.push(va, (float)chunk_size);
            const fvad::MetaResult r1 = m1.to_result();
            fvad::Metadata m2; m2.push(r1, (float)chunk_size);
            ratio[k] = m2.to_result().volume_ratio;
        }
        for (size_t f = 0; f < n_frames; ++f) {
            fvad::Metadata md;
            const uint64_t from = (first_frame + f) * b->fft_size, to = from + b->fft_size;
            for (uint64_t c = from / chunk_size; c * chunk_size < to; ++c) {
                const uint64_t lo = std::max<uint64_t>(from, c * chunk_size), hi = std::min<uint64_t>(to, (c + 1) * chunk_size);
                fvad::MetaResult r;
                r.has_ratio = true;
                r.volume_ratio = ratio[(size_t)(c - first_chunk)];
                md.push(r, (float)(hi - lo));
            }
            const fvad::MetaResult fr = md.to_result();
            for (size_t c = 0; c < C; ++c) vols[c] = band[(s * C + c) * band_stride + f];
            m.run(from, vols.data(), fr.has_ratio, fr.volume_ratio);
        }
        b->segs[s] = m.segments; // (everything so far: a segment is appended when it closes)
        b->audits[s] = m.audit;
    };
    const int nt = (int)std::min<size_t>((size_t)std::max(n_threads, 1), b->n_streams);
    if (nt <= 1) { for (size_t s = 0; s < b->n_streams; ++s) run_stream(s); }
    else {
        std::vector<std::thread> th;
        std::atomic<size_t> next{0};
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= b->n_streams) break; run_stream(i); } });
        for (auto& t : th) t.join();
    }
    b->next_frame = first_frame + n_frames;
    return FVAD_OK;
}

int fvad_vad_batch_run(fvad_vad_batch* b, const float* band, size_t band_stride, size_t n_frames, const float* chunk_rms,
                       size_t rms_stride, size_t n_chunks, size_t chunk_size, int n_threads)
{
    return fvad_vad_batch_run_part(b, band, band_stride, n_frames, chunk_rms, rms_stride, n_chunks, chunk_size, 0, n_threads);
}

size_t fvad_vad_batch_total_segments(const fvad_vad_batch* b)
{
    size_t n = 0;
    if (b) for (const auto& v : b->segs) n += v.size();
    return n;
}

int fvad_vad_batch_segments(const fvad_vad_batch* b, fvad_speech_segment* out, size_t cap, size_t* offsets)
{
    if (!b || !offsets) return FVAD_ERR_INVALID_ARGUMENT;
    size_t n = 0;
    for (size_t s = 0; s < b->n_streams; ++s) { offsets[s] = n; n += b->segs[s].size(); }
    offsets[b->n_streams] = n;
    if (cap < n) return FVAD_ERR_BUFFER_TOO_SMALL;
    if (n && !out) return FVAD_ERR_INVALID_ARGUMENT;
    for (size_t s = 0; s < b->n_streams; ++s)
        if (!b->segs[s].empty()) memcpy(out + offsets[s], b->segs[s].data(), b->segs[s].size() * sizeof(fvad_speech_segment));
    return FVAD_OK;
}

int fvad_vad_batch_audit(const fvad_vad_batch* b, size_t stream, fvad_vad_audit* out)
{
    if (!b || !out || stream >= b->n_streams) return FVAD_ERR_INVALID_ARGUMENT;
    *out = b->audits[stream];
    return FVAD_OK;
}

} // extern "C"
