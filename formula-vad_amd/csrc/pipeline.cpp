// pipeline.cpp -- host-side mirrors of the reference's three seams above the batched engine:
//   B3  fvad_fft       <-> src/FFT.zig            (init / fft / invFft / bin helpers)
//   B2  fvad_nsnet2    <-> src/NSNet2.zig         (init / denoise / getChunkSize)
//   B1  fvad_pipeline  <-> src/AudioPipeline.zig  (init / pushSamples / vad_segments)
// Same names, argument meaning and error behaviour; the arithmetic runs in the HIP kernels, the
// sequential VAD state machine (src/AudioPipeline/VADMachine.zig) on the host (host_vad.cpp).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <new>

#include "host_vad.h"
#include "internal.h"

using namespace fvad;

// ------------------------------------------------------------------ B3: FFT
struct fvad_fft {
    fvad_ctx* ctx;
    size_t n_fft, sample_rate;
    bool inverse;
    float* d_in = nullptr;   // n_fft
    float* d_win = nullptr;  // n_fft
    float* d_out = nullptr;  // (n_fft/2+1)*2
    std::vector<float> h_in;
    VadFftPlan plan{};       // tables of every size but 320 (which has its own in the context)
};

extern "C" {

int fvad_fft_create(fvad_ctx* ctx, size_t n_fft, size_t sample_rate, int mode_inverse, fvad_fft** out)
{
    if (!ctx || !out) return FVAD_ERR_INVALID_ARGUMENT;
    if (n_fft == 0 || n_fft % 2 != 0) return set_err(ctx, FVAD_ERR_INVALID_FFT_SIZE, "n_fft must be even and non-zero"); // FFT.zig:41-43
    // 320 (NSNet2's frame: forward and inverse) and 512 / 1024 / 2048 forward have wavefront kernels; every other even size
    // kissfft would take (and the inverse of any size but 320) runs on the generic mixed-radix kernel, up to 16384 points
    if (n_fft != 320 && !fvad_fft_size_ok(n_fft))
        return set_err(ctx, FVAD_ERR_INVALID_FFT_SIZE, "n_fft must be even, at least 4 and at most 16384");
    hipSetDevice(ctx->device);
    auto* f = new (std::nothrow) fvad_fft();
    if (!f) return FVAD_ERR_ALLOC_FAILED;
    f->ctx = ctx; f->n_fft = n_fft; f->sample_rate = sample_rate; f->inverse = mode_inverse != 0;
    if (n_fft != 320) {
        const int prc = get_vad_plan(ctx, n_fft, &f->plan, /*force_generic=*/mode_inverse != 0);
        if (prc) { delete f; return prc; }
    }
    f->h_in.resize(n_fft);
    if (hipMalloc((void**)&f->d_in, n_fft * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&f->d_win, n_fft * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&f->d_out, (n_fft / 2 + 1) * 2 * sizeof(float)) != hipSuccess) {
        fvad_fft_destroy(f);
        return set_err(ctx, FVAD_ERR_ALLOC_FAILED, "hipMalloc failed"); // KissFFTAllocFailed, FFT.zig:58-60
    }
    *out = f;
    return FVAD_OK;
}

void fvad_fft_destroy(fvad_fft* f)
{
    if (!f) return;
    if (f->d_in) hipFree(f->d_in);
    if (f->d_win) hipFree(f->d_win);
    if (f->d_out) hipFree(f->d_out);
    delete f;
}

size_t fvad_fft_bin_count(const fvad_fft* f) { return f->n_fft / 2 + 1; }                      // FFT.zig:137-139
float fvad_fft_bin_width(const fvad_fft* f) { return (float)f->sample_rate / (float)f->n_fft; } // :142-147
float fvad_fft_nyquist_freq(const fvad_fft* f) { return (float)f->sample_rate / 2; }            // :150-153

int fvad_fft_freq_to_bin(const fvad_fft* f, float freq, size_t* bin) // FFT.zig:156-167
{
    if (!f || !bin) return FVAD_ERR_INVALID_ARGUMENT;
    if (freq > fvad_fft_nyquist_freq(f)) return FVAD_ERR_OUT_OF_RANGE;
    if (freq < 0) return FVAD_ERR_NEGATIVE_FREQUENCY;
    *bin = (size_t)roundf(freq / fvad_fft_bin_width(f)); // @round: half away from zero
    return FVAD_OK;
}

int fvad_fft_bin_to_freq(const fvad_fft* f, size_t bin, float* freq) // FFT.zig:170-180
{
    if (!f || !freq) return FVAD_ERR_INVALID_ARGUMENT;
    if (bin > fvad_fft_bin_count(f) - 1) return FVAD_ERR_OUT_OF_RANGE;
    *freq = (float)bin * fvad_fft_bin_width(f);
    return FVAD_OK;
}

int fvad_fft_forward(fvad_fft* f, const float* first, size_t n_first, const float* second, size_t n_second,
                     const float* window, size_t n_window, fvad_complex* bins, size_t n_bins)
{
    if (!f) return FVAD_ERR_INVALID_ARGUMENT;
    fvad_ctx* ctx = f->ctx;
    // the reference's checks, in its order (FFT.zig:91-102)
    if (n_first + n_second != f->n_fft) return set_err(ctx, FVAD_ERR_INVALID_SAMPLES_LENGTH, "samples.len != n_fft");
    if (n_window != f->n_fft) return set_err(ctx, FVAD_ERR_INVALID_WINDOW_LENGTH, "window.len != n_fft");
    if (n_bins != fvad_fft_bin_count(f)) return set_err(ctx, FVAD_ERR_INVALID_RESULT_LENGTH, "bins.len != binCount()");
    if (f->inverse) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "forward transform on an inverse-mode FFT");
    if ((n_first && !first) || (n_second && !second) || !window || !bins) return FVAD_ERR_INVALID_ARGUMENT;
    hipSetDevice(ctx->device);
    if (n_first) memcpy(f->h_in.data(), first, n_first * sizeof(float));
    if (n_second) memcpy(f->h_in.data() + n_first, second, n_second * sizeof(float));
    hipStream_t st = ctx->stream;
    FVAD_HIP(ctx, hipMemcpyAsync(f->d_in, f->h_in.data(), f->n_fft * sizeof(float), hipMemcpyHostToDevice, st));
    FVAD_HIP(ctx, hipMemcpyAsync(f->d_win, window, f->n_fft * sizeof(float), hipMemcpyHostToDevice, st));
    FVAD_HIP(ctx, (hipError_t)fvad_launch_rfft_batch(f->d_in, 1, (int)f->n_fft, f->d_win, ctx->tb, f->plan, f->d_out, nullptr, st));
    FVAD_HIP(ctx, hipMemcpyAsync(bins, f->d_out, n_bins * sizeof(fvad_complex), hipMemcpyDeviceToHost, st));
    FVAD_HIP(ctx, hipStreamSynchronize(st));
    return FVAD_OK;
}

int fvad_fft_inverse(fvad_fft* f, const fvad_complex* bins, size_t n_bins, float* result, size_t n_result)
{
    if (!f) return FVAD_ERR_INVALID_ARGUMENT;
    fvad_ctx* ctx = f->ctx;
    if (n_bins != fvad_fft_bin_count(f)) return set_err(ctx, FVAD_ERR_INVALID_BINS_LENGTH, "bins.len != binCount()");   // FFT.zig:120-122
    if (n_result != f->n_fft) return set_err(ctx, FVAD_ERR_INVALID_RESULT_LENGTH, "result.len != n_fft");                // :124-126
    if (!f->inverse) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "inverse transform on a forward-mode FFT");
    if (!bins || !result) return FVAD_ERR_INVALID_ARGUMENT;
    hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    FVAD_HIP(ctx, hipMemcpyAsync(f->d_out, bins, n_bins * sizeof(fvad_complex), hipMemcpyHostToDevice, st));
    if (f->n_fft == 320) fvad_launch_irfft_batch(f->d_out, 1, ctx->tb, f->d_in, st);
    else FVAD_HIP(ctx, (hipError_t)fvad_launch_irfft_generic(f->d_out, 1, f->plan, f->d_in, st));
    FVAD_HIP(ctx, hipMemcpyAsync(result, f->d_in, f->n_fft * sizeof(float), hipMemcpyDeviceToHost, st));
    FVAD_HIP(ctx, hipStreamSynchronize(st));
    return FVAD_OK;
}

int fvad_fft_forward_batch(fvad_fft* f, const float* frames, size_t n_frames, const float* window,
                           fvad_complex* bins, float* magnitudes, int on_device)
{
    if (!f || !frames || !window) return FVAD_ERR_INVALID_ARGUMENT;
    fvad_ctx* ctx = f->ctx;
    if (f->inverse) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "forward transform on an inverse-mode FFT");
    if (n_frames == 0) return FVAD_OK;
    hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    const size_t nb = fvad_fft_bin_count(f);
    if (on_device) {
        FVAD_HIP(ctx, (hipError_t)fvad_launch_rfft_batch(frames, (long)n_frames, (int)f->n_fft, window, ctx->tb, f->plan, (float*)bins, magnitudes, st));
        return FVAD_OK;
    }
    float *d_fr = nullptr, *d_bins = nullptr, *d_mag = nullptr;
    int rc = FVAD_OK;
    auto cleanup = [&]() { if (d_fr) hipFree(d_fr); if (d_bins) hipFree(d_bins); if (d_mag) hipFree(d_mag); };
    if (hipMalloc((void**)&d_fr, n_frames * f->n_fft * sizeof(float)) != hipSuccess ||
        (bins && hipMalloc((void**)&d_bins, n_frames * nb * 2 * sizeof(float)) != hipSuccess) ||
        (magnitudes && hipMalloc((void**)&d_mag, n_frames * nb * sizeof(float)) != hipSuccess)) {
        cleanup();
        return set_err(ctx, FVAD_ERR_ALLOC_FAILED, "hipMalloc failed");
    }
    hipMemcpyAsync(d_fr, frames, n_frames * f->n_fft * sizeof(float), hipMemcpyHostToDevice, st);
    hipMemcpyAsync(f->d_win, window, f->n_fft * sizeof(float), hipMemcpyHostToDevice, st);
    const int launch_rc = fvad_launch_rfft_batch(d_fr, (long)n_frames, (int)f->n_fft, f->d_win, ctx->tb, f->plan, d_bins, d_mag, st);
    if (bins) hipMemcpyAsync(bins, d_bins, n_frames * nb * 2 * sizeof(float), hipMemcpyDeviceToHost, st);
    if (magnitudes) hipMemcpyAsync(magnitudes, d_mag, n_frames * nb * sizeof(float), hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess || launch_rc != (int)hipSuccess || hipGetLastError() != hipSuccess) rc = set_err(ctx, FVAD_ERR_HIP, "batched FFT failed");
    cleanup();
    return rc;
}

} // extern "C"

// ------------------------------------------------------------------ B2: NSNet2
struct fvad_nsnet2 {
    fvad_ctx* ctx;
    size_t sample_rate;
    fvad_lane_state* state = nullptr;
    std::vector<float> staged;
    std::vector<float> band, rms;
    // input rates other than 48 kHz (any multiple of 16 kHz, NSNet2.zig:157-162 / resample.zig:4-7): see fvad_nsnet2_denoise
    std::vector<float> den48;
    float last_sample = 0.0f; // NSNet2.zig:33
};

extern "C" {

size_t fvad_nsnet2_chunk_size(size_t in_sample_rate)
{
    // chunk_size * calcDownsampleRate(in_sample_rate, 16000)  (NSNet2.zig:157-159, resample.zig:4-7)
    if (in_sample_rate == 0 || in_sample_rate % 16000 != 0) return 0;
    return (size_t)(kFramesPerChunk * kNHop) * (in_sample_rate / 16000);
}

int fvad_nsnet2_create(fvad_ctx* ctx, size_t sample_rate, fvad_nsnet2** out)
{
    if (!ctx || !out) return FVAD_ERR_INVALID_ARGUMENT;
    // NSNet2.init takes any multiple of 16 kHz (calcDownsampleRate @panics otherwise, resample.zig:4-7); the pipeline
    // itself only ever runs at 48 kHz (VADPipeline.zig:55-58), which is the ratio the kernels decimate by
    if (sample_rate == 0 || sample_rate % 16000 != 0) return set_err(ctx, FVAD_ERR_INVALID_SAMPLE_RATE, "the input rate must be a multiple of 16000 Hz");
    if (!ctx->dm.loaded) return set_err(ctx, FVAD_ERR_NO_MODEL, "load the NSNet2 model into the context first");
    auto* d = new (std::nothrow) fvad_nsnet2();
    if (!d) return FVAD_ERR_ALLOC_FAILED;
    d->ctx = ctx; d->sample_rate = sample_rate;
    const int rc = fvad_lane_state_create(ctx, &d->state);
    if (rc) { delete d; return rc; }
    d->staged.assign(kChunk48, 0.0f);
    if (sample_rate != 48000) d->den48.resize(kChunk48);
    d->band.resize(64);
    d->rms.resize(4);
    *out = d;
    return FVAD_OK;
}

void fvad_nsnet2_destroy(fvad_nsnet2* d)
{
    if (!d) return;
    fvad_lane_state_destroy(d->state);
    delete d;
}

int fvad_nsnet2_denoise(fvad_nsnet2* d, const float* first, size_t n_first, const float* second, size_t n_second,
                        float* denoised_result, size_t n_result)
{
    if (!d) return FVAD_ERR_INVALID_ARGUMENT;
    fvad_ctx* ctx = d->ctx;
    const size_t chunk = fvad_nsnet2_chunk_size(d->sample_rate);
    if (n_first + n_second != chunk) return set_err(ctx, FVAD_ERR_INVALID_INPUT_LENGTH, "samples.len != chunk size"); // NSNet2.zig:166-169
    if (n_result != chunk) return set_err(ctx, FVAD_ERR_INVALID_RESULT_LENGTH, "denoised_result.len != chunk size");   // resample.zig:38-40 (@panic there)
    if ((n_first && !first) || (n_second && !second) || !denoised_result) return FVAD_ERR_INVALID_ARGUMENT;
    const size_t rate = d->sample_rate / 16000; // calcDownsampleRate (resample.zig:4-7)
    fvad_lane lane;
    memset(&lane, 0, sizeof lane);
    lane.pcm = d->staged.data();
    lane.n_samples = kChunk48;
    lane.state = d->state;
    lane.band_sum = d->band.data();
    lane.band_sum_capacity = d->band.size();
    lane.chunk_rms = d->rms.data();
    lane.chunk_rms_capacity = d->rms.size();
    if (rate == 3) {
        if (n_first) memcpy(d->staged.data(), first, n_first * sizeof(float));
        if (n_second) memcpy(d->staged.data() + n_first, second, n_second * sizeof(float));
        lane.denoised = denoised_result;
        return fvad_engine_run(ctx, &lane, 1, nullptr);
    }
    // Another input rate.  Everything between the two resamplers runs at 16 kHz whatever the input rate is
    // (NSNet2.zig:205-236), and both resamplers are index arithmetic: downsampleAudio keeps in[rate i] (resample.zig:9-29),
    // upsampleAudio puts a 16 kHz sample at out[rate i + rate - 1] and lerps in between (resample.zig:32-79).  The kernels
    // decimate by 3, so the chunk is presented to them as the 48 kHz chunk with the same decimation -- x48[3 i] = in[rate i],
    // zeros elsewhere (they would only enter the chunk RMS, which this interface does not report) -- the 16 kHz output is
    // read back from where the x3 upsampler puts it, y48[3 i + 2], and upsampled by `rate` here with the reference's
    // arithmetic (std.math.lerp = one fused multiply-add).  A per-chunk streaming call: its cost is the launch chain, not
    // these 8000 samples.
    constexpr size_t n16 = (size_t)kFramesPerChunk * kNHop; // 8000
    for (size_t i = 0; i < n16; ++i) {
        const size_t src = i * rate;
        d->staged[3 * i] = src < n_first ? first[src] : second[src - n_first];
    }
    lane.denoised = d->den48.data();
    const int rc = fvad_engine_run(ctx, &lane, 1, nullptr);
    if (rc) return rc;
    const size_t n_interp = rate - 1;
    float prev = d->last_sample;
    for (size_t i = 0; i < n16; ++i) {
        const float cur = d->den48[3 * i + 2];
        for (size_t j = 0; j < n_interp; ++j) // resample.zig:67-79 interpolate
            denoised_result[i * rate + j] = std::fmaf(cur - prev, (float)(j + 1) / (float)(n_interp + 1), prev);
        denoised_result[i * rate + n_interp] = cur;
        prev = cur;
    }
    d->last_sample = prev;
    return FVAD_OK;
}

} // extern "C"

// ------------------------------------------------------------------ B1: AudioPipeline
struct fvad_pipeline {
    fvad_ctx* ctx;
    fvad_pipeline_config cfg;
    fvad_callbacks cb{};
    bool has_cb = false;
    size_t chunk_size;
    std::vector<std::vector<float>> pending;  // per channel: pushed, not yet processed
    uint64_t total_write_count = 0;           // AudioPipeline.totalWriteCount
    uint64_t pipeline_read_count = 0;         // VADPipeline.pipeline_read_count
    std::vector<fvad_lane_state*> states;     // per channel (one NSNet2 per channel, BufferedDenoiser.zig:38-41)
    std::unique_ptr<VadMachine> vad;
    std::vector<std::unique_ptr<VadMachine>> alt;
    long min_bin = 0, max_bin = 0;
    // metadata carried between the stages (VADMetadata.zig)
    std::vector<float> chunk_ratio;           // per chunk from chunk_ratio_base on: volume_ratio after the denoiser stage
    uint64_t chunk_ratio_base = 0;            // chunks before this one are no longer covered by any future frame
    uint64_t frames_done = 0;                 // FFT frames handed to the state machine so far
    // traces (parity tests; off unless fvad_pipeline_enable_trace: a live pipeline must not grow without bound)
    bool keep_trace = false;
    std::vector<float> trace_band, trace_ratio;
    // recorders (only when callbacks were given), [0] over the original audio, [1] over the denoised audio
    // (AudioPipeline.zig:30-33): every WRITTEN original sample / every denoised sample from hist_base on, and
    // the state MRBRecorder + Recorder keep (MRBRecorder.zig:26-36, Recorder.zig:12-17)
    std::vector<std::vector<float>> hist_orig, hist_den;
    uint64_t hist_base = 0;
    struct Rec {
        bool recording = false; uint64_t start = 0;   // Recorder.status / startIndex
        bool has_end = false; uint64_t end = 0;       // MRBRecorder.end_recording_on_sample
    } rec[2];
    std::vector<std::vector<float>> den_host; // per-channel D2H landing buffers
    // scratch
    std::vector<std::vector<float>> band, rms;
};

// BufferedFFT.write's metadata for frame k (BufferedFFT.zig:137-140,153): weighted mean of the
// chunk ratios over the chunks whose samples the 1024-sample window covers, accumulated in f32 in
// chunk order exactly like VADMetadata.push / toResult.  chunk_ratio[0] belongs to chunk `base`.
static MetaResult frame_metadata(const std::vector<float>& chunk_ratio, uint64_t base, uint64_t frame, size_t fft_size, size_t chunk_size)
{
    Metadata m;
    const uint64_t from = frame * fft_size, to = from + fft_size;
    for (uint64_t c = from / chunk_size; c * chunk_size < to; ++c) {
        const uint64_t lo = std::max<uint64_t>(from, c * chunk_size);
        const uint64_t hi = std::min<uint64_t>(to, (c + 1) * chunk_size);
        MetaResult r;
        r.has_ratio = true;
        r.volume_ratio = chunk_ratio[(size_t)(c - base)];
        m.push(r, (float)(hi - lo)); // weight = n_written, an integer -> @floatFromInt
    }
    return m.to_result();
}

// Recorder.findBestChannel (Recorder.zig:113-129) over rmsVolume (audio_utils.zig:14-24: sequential
// f32 sum of squares): the first channel with the strictly smallest RMS
static size_t best_channel(const std::vector<std::vector<float>>& hist, size_t off, size_t len)
{
    size_t best = 0;
    float best_vol = 9999;
    for (size_t c = 0; c < hist.size(); ++c) {
        float sum = 0.0f;
        const float* x = hist[c].data() + off;
        for (size_t i = 0; i < len; ++i) sum += x[i] * x[i];
        const float vol = std::sqrt(sum / (float)len);
        if (vol < best_vol) { best = c; best_vol = vol; }
    }
    return best;
}

// MRBRecorder.maybeFinalizeRecording (MRBRecorder.zig:160-192): recorder `which` finalises once its buffer holds
// the samples up to end_recording_on_sample (`available` = what has been written to that buffer at this point
// of the reference's schedule) -> Recorder.finalize / segmentToAudioBuffer (Recorder.zig:73-164) -> callback.
static void maybe_finalize_recording(fvad_pipeline* p, int which, uint64_t available)
{
    fvad_pipeline::Rec& r = p->rec[which];
    if (!r.recording || !r.has_end || available < r.end) return;
    r.has_end = false;
    r.recording = false;
    const fvad_recording_cb cb = which == 0 ? p->cb.on_original_recording : p->cb.on_denoised_recording;
    const std::vector<std::vector<float>>& src = which == 0 ? p->hist_orig : p->hist_den;
    if (!cb || r.end < r.start || r.start < p->hist_base) return; // (the history window covers every possible start)
    const size_t off = (size_t)(r.start - p->hist_base);
    const size_t len = (size_t)(r.end - r.start);
    if (off + len > src[0].size()) return;
    const size_t best = best_channel(src, off, len);
    const float* chan = src[best].data() + off;
    fvad_audio_buffer ab;
    ab.channel_pcm = &chan;
    ab.n_channels = 1;
    ab.length = len;
    ab.sample_rate = p->cfg.sample_rate;
    ab.duration_seconds = (float)len / (float)p->cfg.sample_rate;
    ab.global_start_frame_number = r.start;
    cb(p->cb.ctx, &ab);
}

// VADPipeline.stateMachineStep's recorder calls (VADPipeline.zig:215-229 -> AudioPipeline.zig:181-191 ->
// MRBRecorder.startRecording / stopRecording, MRBRecorder.zig:76-118), original recorder first
static void recorder_event(fvad_pipeline* p, const fvad_vad_result& res, uint64_t avail_orig, uint64_t avail_den)
{
    for (int which = 0; which < 2; ++which) {
        fvad_pipeline::Rec& r = p->rec[which];
        if (res.recording_state == FVAD_REC_STARTED) {
            r.has_end = false;            // "scheduled to stop ... but has been restarted": the pending clip is dropped
            r.recording = true;
            r.start = res.sample_number;
        } else if (res.recording_state == FVAD_REC_COMPLETED) {
            if (!r.recording || r.start > res.sample_number) continue; // error.NotRecording / EndIndexBeforeStart
            r.has_end = true;
            r.end = res.sample_number;
            maybe_finalize_recording(p, which, which == 0 ? avail_orig : avail_den);
        } else if (res.recording_state == FVAD_REC_ABORTED) {
            r.has_end = false;
            r.recording = false;
        }
    }
}

extern "C" {

void fvad_pipeline_config_default(fvad_pipeline_config* c)
{
    memset(c, 0, sizeof *c);
    c->sample_rate = 48000;
    c->n_channels = 1;
    c->buffer_length = 0;
    c->skip_processing = 0;
    c->fft_size = 1024; // VADPipeline.zig:21
    fvad_vad_config_default(&c->vad_machine_config);
}

int fvad_pipeline_create(fvad_ctx* ctx, const fvad_pipeline_config* cfg, const fvad_callbacks* callbacks, fvad_pipeline** out)
{
    if (!ctx || !cfg || !out) return FVAD_ERR_INVALID_ARGUMENT;
    if (cfg->sample_rate != 48000) return set_err(ctx, FVAD_ERR_INVALID_SAMPLE_RATE, "VADPipeline needs 48000 Hz"); // VADPipeline.zig:55-58
    if (cfg->fft_size == 0 || cfg->fft_size % 2 != 0) return set_err(ctx, FVAD_ERR_INVALID_FFT_SIZE, "fft_size must be even"); // FFT.zig:41-43
    if (!fvad_fft_size_ok(cfg->fft_size)) // FFT.init's own check (FFT.zig:41-43) + the generic kernel's size limit
        return set_err(ctx, FVAD_ERR_INVALID_FFT_SIZE, "fft_size must be even, at least 4 and at most 16384");
    if (cfg->n_channels == 0) return FVAD_ERR_INVALID_ARGUMENT;
    // pushSamples writes buffer_length / 2 samples per step (AudioPipeline.zig:121-140): below 2 the step is 0 and its
    // loop never ends, and a ring shorter than one 24000-sample chunk cannot hand VADPipeline.collectInputStep its
    // slice (MultiRingBuffer.readSlice fails, MultiRingBuffer.zig:175-183)
    if (cfg->buffer_length && cfg->buffer_length < fvad_nsnet2_chunk_size(cfg->sample_rate))
        return set_err(ctx, FVAD_ERR_OUT_OF_RANGE, "buffer_length must be 0 (default: 10 s) or at least one 24000-sample chunk");
    if (!ctx->dm.loaded) return set_err(ctx, FVAD_ERR_NO_MODEL, "load the NSNet2 model into the context first");
    auto p = std::unique_ptr<fvad_pipeline>(new (std::nothrow) fvad_pipeline());
    if (!p) return FVAD_ERR_ALLOC_FAILED;
    p->ctx = ctx;
    p->cfg = *cfg;
    if (callbacks) { p->cb = *callbacks; p->has_cb = true; }
    p->chunk_size = fvad_nsnet2_chunk_size(cfg->sample_rate);
    p->pending.resize(cfg->n_channels);
    p->band.resize(cfg->n_channels);
    p->rms.resize(cfg->n_channels);
    if (p->has_cb) {
        p->hist_orig.resize(cfg->n_channels);
        p->hist_den.resize(cfg->n_channels);
        p->den_host.resize(cfg->n_channels);
    }
    // band edges: FFT.freqToBin on the 1024-point / 48 kHz transform (BufferedFFT.zig:192-193)
    const float bin_width = (float)cfg->sample_rate / (float)cfg->fft_size;
    const float nyq = (float)cfg->sample_rate / 2;
    const float fmin = cfg->vad_machine_config.speech_min_freq, fmax = cfg->vad_machine_config.speech_max_freq;
    if (fmin > nyq || fmax > nyq) return set_err(ctx, FVAD_ERR_OUT_OF_RANGE, "speech band above Nyquist");
    if (fmin < 0 || fmax < 0) return set_err(ctx, FVAD_ERR_NEGATIVE_FREQUENCY, "negative speech band edge");
    p->min_bin = (long)roundf(fmin / bin_width);
    p->max_bin = (long)roundf(fmax / bin_width);
    if (p->max_bin < p->min_bin) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "speech_max_freq < speech_min_freq");
    for (size_t c = 0; c < cfg->n_channels; ++c) {
        fvad_lane_state* s = nullptr;
        const int rc = fvad_lane_state_create(ctx, &s);
        if (rc) { for (auto* t : p->states) fvad_lane_state_destroy(t); return rc; }
        p->states.push_back(s);
    }
    // a zero-length channel_vol_ratio ring would divide by zero (RollingAverage.zig:36; same check as fvad_vad_create)
    auto ratio_ring_ok = [&](const fvad_vad_config& vc) {
        return (size_t)(((float)cfg->sample_rate / (float)cfg->fft_size) * vc.channel_vol_ratio_avg_sec) != 0;
    };
    bool rings_ok = ratio_ring_ok(cfg->vad_machine_config);
    for (size_t i = 0; i < cfg->n_alt_vad_machine_configs; ++i) rings_ok = rings_ok && ratio_ring_ok(cfg->alt_vad_machine_configs[i]);
    if (!rings_ok) {
        for (auto* t : p->states) fvad_lane_state_destroy(t);
        return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "channel_vol_ratio_avg_sec is shorter than one FFT frame");
    }
    p->vad.reset(new VadMachine(cfg->vad_machine_config, cfg->sample_rate, cfg->n_channels, cfg->fft_size));
    for (size_t i = 0; i < cfg->n_alt_vad_machine_configs; ++i)
        p->alt.emplace_back(new VadMachine(cfg->alt_vad_machine_configs[i], cfg->sample_rate, cfg->n_channels, cfg->fft_size));
    p->cfg.alt_vad_machine_configs = nullptr; // not retained
    *out = p.release();
    return FVAD_OK;
}

void fvad_pipeline_destroy(fvad_pipeline* p)
{
    if (!p) return;
    for (auto* s : p->states) fvad_lane_state_destroy(s);
    delete p;
}

uint64_t fvad_pipeline_total_write_count(const fvad_pipeline* p) { return p ? p->total_write_count : 0; }

int fvad_pipeline_push_samples(fvad_pipeline* p, const float* const* channel_pcm, size_t n_samples, uint64_t* first_sample_index)
{
    if (!p || (n_samples && !channel_pcm)) return FVAD_ERR_INVALID_ARGUMENT;
    fvad_ctx* ctx = p->ctx;
    const size_t C = p->cfg.n_channels;
    if (first_sample_index) *first_sample_index = p->total_write_count; // AudioPipeline.zig:119
    for (size_t c = 0; c < C; ++c)
        if (n_samples && !channel_pcm[c]) return set_err(ctx, FVAD_ERR_CHANNEL_COUNT_MISMATCH, "missing channel");
    if (p->cfg.skip_processing) { // AudioPipeline.zig:212: samples are written but never read
        p->total_write_count += n_samples;
        return FVAD_OK;
    }
    const uint64_t w0 = p->total_write_count;   // written before this push
    const uint64_t d0 = p->pipeline_read_count; // processed (= denoised) before this push
    for (size_t c = 0; c < C; ++c) p->pending[c].insert(p->pending[c].end(), channel_pcm[c], channel_pcm[c] + n_samples);
    // VADPipeline.collectInputStep (VADPipeline.zig:144-166): every complete chunk, in order -- here
    // all of them in one batched engine call
    const size_t n_chunks = (size_t)((w0 + n_samples - d0) / p->chunk_size);
    std::vector<fvad_lane> lanes(C);
    if (n_chunks) {
        const size_t max_frames = (n_chunks * p->chunk_size + p->cfg.fft_size) / p->cfg.fft_size + 1;
        for (size_t c = 0; c < C; ++c) {
            p->band[c].resize(max_frames);
            p->rms[c].resize(n_chunks);
            fvad_lane& L = lanes[c];
            memset(&L, 0, sizeof L);
            L.pcm = p->pending[c].data();
            L.n_samples = n_chunks * p->chunk_size;
            L.state = p->states[c];
            L.band_sum = p->band[c].data();
            L.band_sum_capacity = max_frames;
            L.chunk_rms = p->rms[c].data();
            L.chunk_rms_capacity = n_chunks;
            if (p->has_cb) {
                p->den_host[c].resize(n_chunks * p->chunk_size);
                L.denoised = p->den_host[c].data();
            }
        }
        fvad_engine_opts opts;
        fvad_engine_opts_default(&opts);
        opts.min_bin = (int32_t)p->min_bin;
        opts.max_bin = (int32_t)p->max_bin;
        opts.fft_size = (int32_t)p->cfg.fft_size;
        const int rc = fvad_engine_run(ctx, lanes.data(), C, &opts);
        if (rc) { // nothing was consumed: the lane states only advance when the call succeeds
            for (size_t c = 0; c < C; ++c) p->pending[c].resize(p->pending[c].size() - n_samples);
            return rc;
        }
    }
    // ---- the push is accepted from here on
    p->total_write_count = w0 + n_samples;
    p->pipeline_read_count = d0 + (uint64_t)n_chunks * p->chunk_size;
    if (p->has_cb)
        for (size_t c = 0; c < C; ++c) {
            p->hist_orig[c].insert(p->hist_orig[c].end(), channel_pcm[c], channel_pcm[c] + n_samples);
            if (n_chunks) p->hist_den[c].insert(p->hist_den[c].end(), p->den_host[c].begin(), p->den_host[c].end());
        }
    for (auto& v : p->pending) v.erase(v.begin(), v.begin() + (long)(n_chunks * p->chunk_size));

    // per-chunk metadata: BufferedVolumeAnalyzer.write then BufferedDenoiser.write each push the
    // ratio with weight 24000 and divide it out again (BufferedVolumeAnalyzer.zig:33-45,
    // BufferedDenoiser.zig:83-86,115)
    std::vector<float> ch(C);
    for (size_t k = 0; k < n_chunks; ++k) {
        for (size_t c = 0; c < C; ++c) ch[c] = p->rms[c][k];
        const MetaResult va = analyse_volume(ch.data(), C);
        Metadata m1; m1.push(va, (float)p->chunk_size);
        const MetaResult r1 = m1.to_result();
        Metadata m2; m2.push(r1, (float)p->chunk_size);
        p->chunk_ratio.push_back(m2.to_result().volume_ratio);
    }

    // ---- the state machine, frame by frame (VADPipeline.stateMachineStep, VADPipeline.zig:209-237), replayed
    // on the reference's schedule: pushSamples writes <= capacity / 2 samples, runs the pipeline over every
    // complete chunk, repeats, and ends with the first short write (AudioPipeline.zig:121-140).  The schedule
    // matters to the recorders only: each looks for its pending end in front of every write to ITS buffer
    // (recordBeforeMRBWrite: a write step for the original audio, a 0.5 s chunk for the denoised audio).
    const size_t n_frames = n_chunks ? lanes[0].n_fft_frames : 0;
    const uint64_t first_index = n_chunks ? lanes[0].first_frame_index : 0;
    const size_t capacity = p->cfg.buffer_length ? p->cfg.buffer_length : p->cfg.sample_rate * 10; // AudioPipeline.zig:46
    const size_t write_chunk = capacity / 2;
    std::vector<float> vols(C);
    size_t k = 0; // next frame of this push
    auto run_frame = [&](uint64_t avail_orig, uint64_t avail_den) {
        for (size_t c = 0; c < C; ++c) vols[c] = p->band[c][k];
        const MetaResult md = frame_metadata(p->chunk_ratio, p->chunk_ratio_base, p->frames_done + k, p->cfg.fft_size, p->chunk_size);
        const uint64_t index = first_index + (uint64_t)k * p->cfg.fft_size;
        const fvad_vad_result res = p->vad->run(index, vols.data(), md.has_ratio, md.volume_ratio);
        if (p->has_cb && res.recording_state != FVAD_REC_NONE) recorder_event(p, res, avail_orig, avail_den);
        for (auto& a : p->alt) a->run(index, vols.data(), md.has_ratio, md.volume_ratio);
        if (p->keep_trace) {
            p->trace_band.insert(p->trace_band.end(), vols.begin(), vols.end());
            p->trace_ratio.push_back(md.has_ratio ? md.volume_ratio : NAN);
        }
        ++k;
    };
    uint64_t w = w0, d = d0;
    size_t off = 0;
    for (;;) {
        const size_t step = std::min(write_chunk, n_samples - off);
        if (p->has_cb) maybe_finalize_recording(p, 0, w);
        w += step;
        off += step;
        while (d + p->chunk_size <= w) {
            if (p->has_cb) maybe_finalize_recording(p, 1, d);
            d += p->chunk_size;
            // the frames this chunk completes: BufferedFFT emits a frame when its last sample has been denoised
            while (k < n_frames && first_index + (uint64_t)(k + 1) * p->cfg.fft_size <= d) run_frame(w, d);
        }
        if (step < write_chunk) break;
    }
    while (k < n_frames) run_frame(w, d); // (none are left: every frame is completed by a chunk of this push)
    p->frames_done += n_frames;

    // chunk ratios no future frame covers
    {
        const uint64_t next_frame_chunk = (p->frames_done * (uint64_t)p->cfg.fft_size) / p->chunk_size;
        if (next_frame_chunk > p->chunk_ratio_base) {
            const size_t drop = (size_t)std::min<uint64_t>(next_frame_chunk - p->chunk_ratio_base, p->chunk_ratio.size());
            p->chunk_ratio.erase(p->chunk_ratio.begin(), p->chunk_ratio.begin() + (long)drop);
            p->chunk_ratio_base += drop;
        }
    }
    if (p->has_cb) {
        // keep what a recording can still start at: a `started` carries getOffsetRecordingStart(speech start)
        // = speech start - 2 s (VADMachine.zig:309-315); the earliest speech start still to come is the one of
        // an opening in progress, else the next frame
        const uint64_t next_index = p->frames_done * (uint64_t)p->cfg.fft_size;
        uint64_t earliest = p->vad->state != VadMachine::CLOSED ? std::min(next_index, p->vad->speech_start_index) : next_index;
        uint64_t keep_from = p->vad->offset_start(earliest);
        for (const auto& r : p->rec) if (r.recording) keep_from = std::min(keep_from, r.start);
        if (keep_from > p->hist_base) {
            const size_t drop = (size_t)(keep_from - p->hist_base);
            for (size_t c = 0; c < C; ++c) {
                p->hist_orig[c].erase(p->hist_orig[c].begin(), p->hist_orig[c].begin() + (long)std::min(drop, p->hist_orig[c].size()));
                p->hist_den[c].erase(p->hist_den[c].begin(), p->hist_den[c].begin() + (long)std::min(drop, p->hist_den[c].size()));
            }
            p->hist_base = keep_from;
        }
    }
    return FVAD_OK;
}

int fvad_pipeline_enable_trace(fvad_pipeline* p, int on)
{
    if (!p) return FVAD_ERR_INVALID_ARGUMENT;
    p->keep_trace = on != 0;
    return FVAD_OK;
}

size_t fvad_pipeline_segment_count(const fvad_pipeline* p) { return p ? p->vad->segments.size() : 0; }

static int copy_segments(const std::vector<fvad_speech_segment>& v, fvad_speech_segment* out, size_t cap, size_t* n)
{
    if (!n) return FVAD_ERR_INVALID_ARGUMENT;
    *n = v.size();
    if (cap < v.size()) return FVAD_ERR_BUFFER_TOO_SMALL;
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(fvad_speech_segment));
    return FVAD_OK;
}

int fvad_pipeline_segments(const fvad_pipeline* p, fvad_speech_segment* out, size_t cap, size_t* n)
{
    if (!p) return FVAD_ERR_INVALID_ARGUMENT;
    return copy_segments(p->vad->segments, out, cap, n);
}

int fvad_pipeline_alt_segments(const fvad_pipeline* p, size_t alt_index, fvad_speech_segment* out, size_t cap, size_t* n)
{
    if (!p || alt_index >= p->alt.size()) return FVAD_ERR_INVALID_ARGUMENT;
    return copy_segments(p->alt[alt_index]->segments, out, cap, n);
}

int fvad_pipeline_audit(const fvad_pipeline* p, fvad_vad_audit* out)
{
    if (!p || !out) return FVAD_ERR_INVALID_ARGUMENT;
    *out = p->vad->audit;
    return FVAD_OK;
}

size_t fvad_pipeline_n_fft_frames(const fvad_pipeline* p) { return p ? (size_t)p->frames_done : 0; }

int fvad_pipeline_trace(const fvad_pipeline* p, float* band_volumes, float* vol_ratio, size_t cap_frames)
{
    if (!p) return FVAD_ERR_INVALID_ARGUMENT;
    if (cap_frames * p->cfg.n_channels < p->trace_band.size() || cap_frames < p->trace_ratio.size()) return FVAD_ERR_BUFFER_TOO_SMALL;
    if (band_volumes && !p->trace_band.empty()) memcpy(band_volumes, p->trace_band.data(), p->trace_band.size() * sizeof(float));
    if (vol_ratio && !p->trace_ratio.empty()) memcpy(vol_ratio, p->trace_ratio.data(), p->trace_ratio.size() * sizeof(float));
    return FVAD_OK;
}

} // extern "C"
