// host_vad.h -- host-side VAD state machine (mirrors the reference's VADMachine / RollingAverage)
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/fvad.h"

namespace fvad {

struct RollingAverage {            // src/structures/RollingAverage.zig
    std::vector<double> data;
    size_t len = 0;
    bool has_last_avg = false;
    double last_avg = 0;
    size_t write_idx = 0;
    size_t written_count = 0;
    // steady state (ring full): products data[i] * (1/len) and the running sum below write_idx
    bool steady = false;
    double scalar = 0, pref = 0;
    std::vector<double> q;
    RollingAverage(size_t count, bool has_initial, double initial_val);
    double push(float sample);
    double avg();
    void enter_steady();
};

struct MetaResult {                // VADMetadata.Result, VADMetadata.zig:5-9 (optionals)
    bool has_ratio = false, has_min = false, has_max = false;
    float volume_ratio = 0, volume_min = 0, volume_max = 0;
};
struct Metadata {                  // VADMetadata.zig:11-60
    bool has_ratio = false, has_min = false, has_max = false;
    float ratio_sum = 0, ratio_weight = 0, volume_min = 0, volume_max = 0;
    void push(const MetaResult& v, float weight);
    MetaResult to_result() const;
    void reset() { *this = Metadata(); }
};
MetaResult analyse_volume(const float* channel_rms, size_t n_channels);

struct VadMachine {                // src/AudioPipeline/VADMachine.zig
    enum State { CLOSED, OPENING, OPEN, CLOSING };
    fvad_vad_config cfg;
    size_t sample_rate, n_channels, fft_size;
    State state = CLOSED;
    RollingAverage long_term, short_term, ch_ratio;
    uint64_t speech_start_index = 0, speech_end_index = 0;
    float channel_vol_ratio_sum = 0;
    size_t channel_vol_ratio_count = 0;
    float vad_threshold_met_cumulative_sec = 0;
    std::vector<fvad_speech_segment> segments;
    fvad_vad_audit audit;

    // Lazily exact long-term average (see host_vad.cpp): between exact evaluations of the reference's
    // 8437-term chain the machine carries an incrementally updated value and a rigorous bound on its
    // distance from what the chain would give; the chain is run only when that bound cannot settle a
    // comparison.  long_term.last_avg is current only while !lt_stale.
    double lt_approx = 0, lt_err = 0, lt_abs = 0, lt_abs_anchor = 0;
    bool lt_stale = false, lt_anchored = false;
    bool lt_lazy = true; // FVAD_VAD_EAGER=1 in the environment: run the chain on every push (testing aid)
    unsigned lt_updates = 0;
    uint64_t lt_exact_evals = 0, lt_lazy_pushes = 0; // statistics
    void long_term_push(float mv);
    void long_term_exact();

    VadMachine(const fvad_vad_config& c, size_t sample_rate, size_t n_channels, size_t fft_size);
    fvad_vad_result run(uint64_t index, const float* channel_volumes, bool has_ratio, float ratio);
    float min_volume(const float* channel_volumes) const;
    bool decide(double short_term_avg, double ratio_avg, double* threshold_out);
    fvad_vad_result finish_step(uint64_t index, bool threshold_met, bool has_ratio, float ratio);
    fvad_vad_result on_speech_end();
    uint64_t offset_start(uint64_t vad_from) const;
    uint64_t offset_end(uint64_t vad_to) const;
};

// ratio[s][k] is NaN where the frame carries no volume_ratio (null in the reference)
void run_many(VadMachine* const* vads, size_t n_streams, const float* const* band,
              const float* const* ratio, const size_t* n_frames, size_t n_channels,
              const uint64_t* first_index, size_t fft_size, int n_threads);

} // namespace fvad
