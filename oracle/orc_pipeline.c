/*
 * orc_pipeline.c -- ORACLE (test infrastructure only; see orc.h).
 * Restates the streaming order of src/AudioPipeline.zig:118-143 (pushSamples chunking) and
 * src/AudioPipeline/{VADPipeline,BufferedVolumeAnalyzer,BufferedDenoiser,BufferedFFT,
 * SegmentWriter}.zig.  The ring buffer (structures/MultiRingBuffer.zig) is replaced by a
 * pending-sample FIFO: it holds no arithmetic and, with the default 10 s capacity, a 24000-sample
 * chunk never wraps (480000 % 24000 == 0), so SplitSlice.second is always empty on this path.
 */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ SegmentWriter */

typedef struct {
    int n_channels;
    size_t length;      /* capacity == segment.length */
    float **chan;       /* [n_channels][length] */
    size_t write_index; /* SegmentWriter.zig:14 */
    uint64_t index;     /* segment.index */
} seg_writer;

static void sw_init(seg_writer *sw, int n_channels, size_t length)
{
    sw->n_channels = n_channels;
    sw->length = length;
    sw->chan = (float **)calloc((size_t)n_channels, sizeof(float *));
    for (int c = 0; c < n_channels; ++c) sw->chan[c] = (float *)calloc(length, sizeof(float));
    sw->write_index = 0;
    sw->index = 0; /* SegmentWriter.zig:24 */
}
static void sw_free(seg_writer *sw)
{
    for (int c = 0; c < sw->n_channels; ++c) free(sw->chan[c]);
    free(sw->chan);
}
static int sw_is_full(const seg_writer *sw) { return sw->write_index == sw->length; } /* :38-40 */

/* SegmentWriter.zig:46-109 (max_write = null).  The source segment has `other_length` samples
 * per channel split as first/second. */
static size_t sw_write(seg_writer *sw, const float *const *first, size_t n_first,
                       const float *const *second, size_t other_length, size_t read_offset)
{
    const size_t capacity = sw->length;
    const size_t wi = sw->write_index < capacity ? sw->write_index : capacity;
    const size_t remaining_capacity = capacity - wi;
    if (remaining_capacity == 0) return 0;
    const size_t other_rem = other_length - read_offset;
    const size_t to_write = remaining_capacity < other_rem ? remaining_capacity : other_rem;
    for (int c = 0; c < sw->n_channels; ++c) {
        float *dst = sw->chan[c];
        size_t n_from_first = 0;
        if (n_first > read_offset) {
            n_from_first = n_first - read_offset;
            if (to_write < n_from_first) n_from_first = to_write;
        }
        if (n_from_first > 0)
            memcpy(dst + sw->write_index, first[c] + read_offset, sizeof(float) * n_from_first);
        if (n_from_first < to_write) {
            const size_t rem = to_write - n_from_first;
            const size_t dst_from = sw->write_index + n_from_first;
            const size_t src_from = read_offset - (n_first < read_offset ? n_first : read_offset);
            memcpy(dst + dst_from, second[c] + src_from, sizeof(float) * rem);
        }
    }
    sw->write_index += to_write;
    return to_write;
}
static void sw_reset(seg_writer *sw, uint64_t new_index) /* :111-114 */
{
    sw->write_index = 0;
    sw->index = new_index;
}

struct orc_segment_writer { seg_writer sw; };
orc_segment_writer *orc_sw_create(size_t length)
{
    orc_segment_writer *s = (orc_segment_writer *)calloc(1, sizeof(*s));
    sw_init(&s->sw, 1, length);
    return s;
}
void orc_sw_destroy(orc_segment_writer *s) { if (s) { sw_free(&s->sw); free(s); } }
size_t orc_sw_write(orc_segment_writer *s, const float *first, size_t n_first,
                    const float *second, size_t n_second, size_t read_offset)
{
    const float *f[1] = { first }, *g[1] = { second };
    return sw_write(&s->sw, f, n_first, g, n_first + n_second, read_offset);
}
int orc_sw_is_full(const orc_segment_writer *s) { return sw_is_full(&s->sw); }
void orc_sw_reset(orc_segment_writer *s, uint64_t idx) { sw_reset(&s->sw, idx); }
const float *orc_sw_data(const orc_segment_writer *s) { return s->sw.chan[0]; }
size_t orc_sw_write_index(const orc_segment_writer *s) { return s->sw.write_index; }
uint64_t orc_sw_index(const orc_segment_writer *s) { return s->sw.index; }

/* ------------------------------------------------------------------ BufferedFFT arithmetic */

/* BufferedFFT.zig:95-99 + :162-181 for one channel: periodic Hann, kiss_fftr, |X| * norm */
void orc_buffered_fft_frame(const float *samples, int fft_size, float *bins_out)
{
    const int n_bins = orc_fft_bin_count(fft_size);
    float *window = (float *)malloc(sizeof(float) * (size_t)fft_size);
    orc_cpx *cbuf = (orc_cpx *)malloc(sizeof(orc_cpx) * (size_t)n_bins);
    orc_hann_window_periodic(window, (size_t)fft_size);
    const float norm_factor = orc_window_norm_factor(window, (size_t)fft_size) / (float)fft_size;
    orc_fftr *fft = orc_fftr_alloc(fft_size, 0);
    orc_fft_fft(fft, samples, (size_t)fft_size, NULL, 0, window, (size_t)fft_size, cbuf,
                (size_t)n_bins);
    for (int i = 0; i < n_bins; ++i)
        bins_out[i] = sqrtf(cbuf[i].r * cbuf[i].r + cbuf[i].i * cbuf[i].i) * norm_factor;
    orc_fftr_free(fft);
    free(window);
    free(cbuf);
}

/* BufferedFFT.zig:183-202: a SUM over [min_bin, max_bin], in index order */
float orc_band_sum(const float *bins, long min_bin, long max_bin)
{
    float acc = 0.0f;
    for (long i = min_bin; i < max_bin + 1; ++i) acc += bins[i];
    return acc;
}

/* ------------------------------------------------------------------ pipeline */

struct orc_pipeline {
    orc_pipeline_config cfg;
    size_t chunk_size; /* BufferedDenoiser.getChunkSize */
    /* pending input (stands in for original_audio_buffer) */
    float **pending;
    size_t n_pending, cap_pending;
    uint64_t total_write_count;   /* AudioPipeline.totalWriteCount */
    uint64_t pipeline_read_count; /* VADPipeline.zig:41 */
    /* BufferedDenoiser */
    orc_nsnet2 **denoisers;
    seg_writer den_buffer;
    orc_meta den_meta;
    float **den_result; /* temp_result_segment */
    /* BufferedFFT */
    orc_fftr *fft;
    float *window;
    float norm_factor;
    orc_cpx *complex_buffer;
    seg_writer fft_buffer;
    orc_meta fft_meta;
    float **channel_bins;
    long min_bin, max_bin;
    /* VADMachine */
    orc_vad *vad;
    float *temp_channel_volumes;
    /* traces */
    float *band_volumes; size_t n_frames, cap_frames;
    float *frame_ratio;
    float *chunk_rms; size_t n_chunks, cap_chunks;
    float **denoised; size_t n_denoised, cap_denoised;
    float *all_bins; /* [frame][channel][n_bins] when keep_denoised */
    size_t cap_all_bins;
    /* recorders (keep_denoised only): [0] over the original audio, [1] over the denoised audio
     * (AudioPipeline.zig:30-33); MRBRecorder state = {recorder.status/startIndex, end_recording_on_sample} */
    float **original; size_t n_original, cap_original; /* every WRITTEN original sample */
    struct {
        int recording; uint64_t start;
        int has_end; uint64_t end;           /* MRBRecorder.end_recording_on_sample */
        struct { uint64_t start; size_t length; int best; float *pcm; } *recs;
        size_t n_recs, cap_recs;
    } rec[2];
};

void orc_pipeline_config_default(orc_pipeline_config *c)
{
    c->sample_rate = 48000;
    c->n_channels = 1;
    c->fft_size = 1024; /* VADPipeline.zig:21 */
    c->keep_denoised = 0;
    c->buffer_length = 0;
    orc_vad_config_default(&c->vad);
}

orc_pipeline *orc_pipeline_create(const orc_pipeline_config *cfg, const orc_nsnet2_weights *w,
                                  int *err)
{
    if (cfg->sample_rate != 48000) { /* VADPipeline.zig:55-58 */
        if (err) *err = ORC_ERR_INVALID_SAMPLE_RATE;
        return NULL;
    }
    if (cfg->fft_size == 0 || (cfg->fft_size & 1)) { /* FFT.zig:41-43 */
        if (err) *err = ORC_ERR_INVALID_FFT_SIZE;
        return NULL;
    }
    /* Not in the reference: AudioPipeline.pushSamples writes buffer_length / 2 samples per step
     * (AudioPipeline.zig:121-140), so a buffer_length of 1 never ends its loop there, and a ring shorter than one
     * chunk fails in MultiRingBuffer.readSlice.  The checker refuses such a configuration instead of hanging a test. */
    if (cfg->buffer_length != 0 && (size_t)cfg->buffer_length < orc_nsnet2_chunk_size(cfg->sample_rate)) {
        if (err) *err = ORC_ERR_OUT_OF_RANGE;
        return NULL;
    }
    orc_pipeline *p = (orc_pipeline *)calloc(1, sizeof(*p));
    p->cfg = *cfg;
    const int C = cfg->n_channels;
    p->chunk_size = orc_nsnet2_chunk_size(cfg->sample_rate);
    p->pending = (float **)calloc((size_t)C, sizeof(float *));
    p->denoisers = (orc_nsnet2 **)calloc((size_t)C, sizeof(orc_nsnet2 *));
    p->den_result = (float **)calloc((size_t)C, sizeof(float *));
    p->channel_bins = (float **)calloc((size_t)C, sizeof(float *));
    p->denoised = (float **)calloc((size_t)C, sizeof(float *));
    p->original = (float **)calloc((size_t)C, sizeof(float *));
    const int n_bins = orc_fft_bin_count(cfg->fft_size);
    for (int c = 0; c < C; ++c) {
        p->denoisers[c] = orc_nsnet2_create(cfg->sample_rate, w); /* BufferedDenoiser.zig:38-41 */
        p->den_result[c] = (float *)calloc(p->chunk_size, sizeof(float));
        p->channel_bins[c] = (float *)calloc((size_t)n_bins, sizeof(float));
    }
    sw_init(&p->den_buffer, C, p->chunk_size);           /* BufferedDenoiser.zig:47 */
    sw_init(&p->fft_buffer, C, (size_t)cfg->fft_size);   /* BufferedFFT.zig:83-86 */
    orc_meta_reset(&p->den_meta);
    orc_meta_reset(&p->fft_meta);
    p->fft = orc_fftr_alloc(cfg->fft_size, 0);           /* BufferedFFT.zig:75-80 */
    p->window = (float *)malloc(sizeof(float) * (size_t)cfg->fft_size);
    orc_hann_window_periodic(p->window, (size_t)cfg->fft_size); /* :97 */
    p->norm_factor = orc_window_norm_factor(p->window, (size_t)cfg->fft_size) /
                     (float)cfg->fft_size;               /* :99 */
    p->complex_buffer = (orc_cpx *)malloc(sizeof(orc_cpx) * (size_t)n_bins);
    p->min_bin = orc_fft_freq_to_bin(cfg->fft_size, cfg->sample_rate, cfg->vad.speech_min_freq);
    p->max_bin = orc_fft_freq_to_bin(cfg->fft_size, cfg->sample_rate, cfg->vad.speech_max_freq);
    p->vad = orc_vad_create(&cfg->vad, cfg->sample_rate, C, cfg->fft_size);
    p->temp_channel_volumes = (float *)calloc((size_t)C, sizeof(float));
    return p;
}

void orc_pipeline_destroy(orc_pipeline *p)
{
    if (!p) return;
    const int C = p->cfg.n_channels;
    for (int c = 0; c < C; ++c) {
        free(p->pending[c]);
        orc_nsnet2_destroy(p->denoisers[c]);
        free(p->den_result[c]);
        free(p->channel_bins[c]);
        free(p->denoised[c]);
        free(p->original[c]);
    }
    for (int which = 0; which < 2; ++which) {
        for (size_t i = 0; i < p->rec[which].n_recs; ++i) free(p->rec[which].recs[i].pcm);
        free(p->rec[which].recs);
    }
    free(p->original);
    free(p->pending); free(p->denoisers); free(p->den_result); free(p->channel_bins);
    free(p->denoised);
    sw_free(&p->den_buffer); sw_free(&p->fft_buffer);
    orc_fftr_free(p->fft);
    free(p->window); free(p->complex_buffer);
    orc_vad_destroy(p->vad);
    free(p->temp_channel_volumes);
    free(p->band_volumes); free(p->frame_ratio); free(p->chunk_rms); free(p->all_bins);
    free(p);
}

/* Recorder.findBestChannel (Recorder.zig:113-129): first channel with the strictly smallest RMS */
static int find_best_channel(float *const *chan, int C, uint64_t from, size_t len)
{
    int best = 0;
    float best_vol = 9999;
    for (int c = 0; c < C; ++c) {
        const float vol = orc_rms_volume(chan[c] + from, len, NULL, 0);
        if (vol < best_vol) { best = c; best_vol = vol; }
    }
    return best;
}

/* samples recorder `which` can see: everything written to its ring buffer so far */
static uint64_t rec_available(const orc_pipeline *p, int which)
{
    return which == 0 ? (uint64_t)p->n_original : (uint64_t)p->n_denoised;
}

/* MRBRecorder.maybeFinalizeRecording (MRBRecorder.zig:160-192): finalise once the samples up to
 * end_recording_on_sample have arrived -> Recorder.finalize / segmentToAudioBuffer (Recorder.zig:73-164) */
static void maybe_finalize_recording(orc_pipeline *p, int which)
{
    if (!p->rec[which].recording || !p->rec[which].has_end) return;
    const uint64_t to = p->rec[which].end;
    if (rec_available(p, which) < to) return;
    p->rec[which].has_end = 0;
    p->rec[which].recording = 0;
    const uint64_t from = p->rec[which].start;
    if (to < from) return; /* error.InvalidEndIndex */
    const int C = p->cfg.n_channels;
    if (p->rec[which].n_recs == p->rec[which].cap_recs) {
        p->rec[which].cap_recs = p->rec[which].cap_recs ? p->rec[which].cap_recs * 2 : 16;
        p->rec[which].recs = realloc(p->rec[which].recs, sizeof(*p->rec[which].recs) * p->rec[which].cap_recs);
    }
    const size_t len = (size_t)(to - from);
    float *const *src = which == 0 ? p->original : p->denoised;
    const int best = find_best_channel(src, C, from, len);
    const size_t i = p->rec[which].n_recs++;
    p->rec[which].recs[i].start = from;
    p->rec[which].recs[i].length = len;
    p->rec[which].recs[i].best = best;
    p->rec[which].recs[i].pcm = (float *)malloc(sizeof(float) * (len ? len : 1));
    memcpy(p->rec[which].recs[i].pcm, src[best] + from, sizeof(float) * len);
}

/* MRBRecorder.startRecording (MRBRecorder.zig:76-86): a pending end is dropped ("has been restarted") */
static void start_recording(orc_pipeline *p, int which, uint64_t from)
{
    p->rec[which].has_end = 0;
    p->rec[which].recording = 1; /* Recorder.start, Recorder.zig:54-60 */
    p->rec[which].start = from;
}

/* MRBRecorder.stopRecording (MRBRecorder.zig:88-118) */
static void stop_recording(orc_pipeline *p, int which, uint64_t to, int keep)
{
    if (!p->rec[which].recording) return; /* error.NotRecording */
    if (keep) {
        if (p->rec[which].start > to) return; /* error.EndIndexBeforeStart */
        p->rec[which].has_end = 1;
        p->rec[which].end = to;
        maybe_finalize_recording(p, which);
    } else {
        p->rec[which].has_end = 0;
        p->rec[which].recording = 0;
    }
}

/* VADPipeline.stateMachineStep (VADPipeline.zig:209-237) + VADMachine.run's first step,
 * averageVolumeInBand (VADMachine.zig:146-151 -> BufferedFFT.zig:183-202) */
static void state_machine_step(orc_pipeline *p, uint64_t index, const orc_meta_result *meta)
{
    const int C = p->cfg.n_channels;
    for (int c = 0; c < C; ++c)
        p->temp_channel_volumes[c] = orc_band_sum(p->channel_bins[c], p->min_bin, p->max_bin);
    if (p->n_frames == p->cap_frames) {
        p->cap_frames = p->cap_frames ? p->cap_frames * 2 : 1024;
        p->band_volumes = (float *)realloc(p->band_volumes, sizeof(float) * p->cap_frames * (size_t)C);
        p->frame_ratio = (float *)realloc(p->frame_ratio, sizeof(float) * p->cap_frames);
    }
    memcpy(p->band_volumes + p->n_frames * (size_t)C, p->temp_channel_volumes, sizeof(float) * (size_t)C);
    p->frame_ratio[p->n_frames] = meta->has_ratio ? meta->volume_ratio : NAN;
    if (p->cfg.keep_denoised) {
        const size_t n_bins = (size_t)orc_fft_bin_count(p->cfg.fft_size);
        const size_t need = (p->n_frames + 1) * (size_t)C * n_bins;
        if (need > p->cap_all_bins) {
            p->cap_all_bins = need * 2;
            p->all_bins = (float *)realloc(p->all_bins, sizeof(float) * p->cap_all_bins);
        }
        for (int c = 0; c < C; ++c)
            memcpy(p->all_bins + (p->n_frames * (size_t)C + (size_t)c) * n_bins, p->channel_bins[c],
                   sizeof(float) * n_bins);
    }
    p->n_frames++;
    const orc_vad_result r =
        orc_vad_run(p->vad, index, p->temp_channel_volumes, meta->has_ratio, meta->volume_ratio);
    if (p->cfg.keep_denoised) { /* VADPipeline.zig:215-229 -> AudioPipeline.zig:181-191: original, then denoised */
        for (int which = 0; which < 2; ++which) {
            if (r.recording_state == ORC_REC_STARTED) start_recording(p, which, r.sample_number);
            else if (r.recording_state == ORC_REC_COMPLETED) stop_recording(p, which, r.sample_number, 1);
            else if (r.recording_state == ORC_REC_ABORTED) stop_recording(p, which, r.sample_number, 0);
        }
    }
}

/* VADPipeline.fftStep (VADPipeline.zig:191-207) driving BufferedFFT.write (BufferedFFT.zig:129-160) */
static void fft_step(orc_pipeline *p, float *const *segment, size_t seg_length, uint64_t seg_index,
                     const orc_meta_result *in_meta)
{
    const int C = p->cfg.n_channels;
    size_t input_offset = 0;
    for (;;) {
        const size_t n_written = sw_write(&p->fft_buffer, (const float *const *)segment, seg_length,
                                          NULL, seg_length, input_offset);
        const size_t n_remaining_input = seg_length - input_offset - n_written;
        orc_meta_push(&p->fft_meta, in_meta, (float)n_written); /* :137-140 */
        if (!sw_is_full(&p->fft_buffer)) return;                /* :142-147 */

        /* BufferedFFT.fft, :162-181 */
        for (int c = 0; c < C; ++c) {
            orc_fft_fft(p->fft, p->fft_buffer.chan[c], (size_t)p->cfg.fft_size, NULL, 0, p->window,
                        (size_t)p->cfg.fft_size, p->complex_buffer,
                        (size_t)orc_fft_bin_count(p->cfg.fft_size));
            const int n_bins = orc_fft_bin_count(p->cfg.fft_size);
            for (int i = 0; i < n_bins; ++i) {
                const orc_cpx b = p->complex_buffer[i];
                p->channel_bins[c][i] = sqrtf(b.r * b.r + b.i * b.i) * p->norm_factor; /* FFT.zig:16-18 */
            }
        }
        const uint64_t result_index = p->fft_buffer.index; /* :152 */
        const orc_meta_result meta = orc_meta_to_result(&p->fft_meta); /* :153 */
        orc_meta_reset(&p->fft_meta);                                   /* :155 */
        sw_reset(&p->fft_buffer, seg_index + input_offset + n_written); /* :149 defer */

        state_machine_step(p, result_index, &meta);

        if (n_remaining_input == 0) return;                 /* VADPipeline.zig:204 */
        input_offset = seg_length - n_remaining_input;      /* :205 */
    }
}

/* BufferedVolumeAnalyzer.analyseVolume (BufferedVolumeAnalyzer.zig:48-69) */
static orc_meta_result analyse_volume(orc_pipeline *p, float *const *chunk, size_t n)
{
    const int C = p->cfg.n_channels;
    float vol_min = 1, vol_max = 0;
    if (p->n_chunks == p->cap_chunks) {
        p->cap_chunks = p->cap_chunks ? p->cap_chunks * 2 : 256;
        p->chunk_rms = (float *)realloc(p->chunk_rms, sizeof(float) * p->cap_chunks * (size_t)C);
    }
    for (int c = 0; c < C; ++c) {
        const float vol = orc_rms_volume(chunk[c], n, NULL, 0);
        p->chunk_rms[p->n_chunks * (size_t)C + (size_t)c] = vol;
        if (vol < vol_min) vol_min = vol;
        if (vol > vol_max) vol_max = vol;
    }
    p->n_chunks++;
    orc_meta_result r;
    r.has_ratio = r.has_min = r.has_max = 1;
    r.volume_ratio = (vol_max == 0) ? 0 : vol_min / vol_max;
    r.volume_min = vol_min;
    r.volume_max = vol_max;
    return r;
}

/* one iteration of VADPipeline.collectInputStep's loop body (VADPipeline.zig:150-165) */
static void process_chunk(orc_pipeline *p, float *const *chunk, uint64_t from)
{
    const int C = p->cfg.n_channels;
    const size_t n = p->chunk_size;

    /* BufferedVolumeAnalyzer.write (BufferedVolumeAnalyzer.zig:29-46): push with weight =
     * segment.length, toResult, reset */
    orc_meta m;
    orc_meta_reset(&m);
    const orc_meta_result va = analyse_volume(p, chunk, n);
    orc_meta_push(&m, &va, (float)n);
    const orc_meta_result analyzed = orc_meta_to_result(&m);

    /* VADPipeline.denoiserStep (:168-189) -> BufferedDenoiser.write (BufferedDenoiser.zig:75-120) */
    size_t input_offset = 0;
    for (;;) {
        const size_t n_written =
            sw_write(&p->den_buffer, (const float *const *)chunk, n, NULL, n, input_offset);
        const size_t n_remaining_input = n - input_offset - n_written;
        orc_meta_push(&p->den_meta, &analyzed, (float)n_written); /* :83-86 */
        if (!sw_is_full(&p->den_buffer)) return;                  /* :88-95 */

        const uint64_t result_index = p->den_buffer.index;        /* :103 */
        for (int c = 0; c < C; ++c)                               /* :105-110 */
            orc_nsnet2_denoise(p->denoisers[c], p->den_buffer.chan[c], n, NULL, 0,
                               p->den_result[c], n);
        const orc_meta_result den_meta = orc_meta_to_result(&p->den_meta); /* :115 */
        sw_reset(&p->den_buffer, from + n_written);               /* :97-100 defer */
        orc_meta_reset(&p->den_meta);

        /* pipeline.pushDenoisedSamples (VADPipeline.zig:183 -> AudioPipeline.zig:145-166): the denoised
         * recorder looks at its pending end BEFORE the new samples are written (recordBeforeMRBWrite) */
        if (p->cfg.keep_denoised) {
            maybe_finalize_recording(p, 1);
            if (p->n_denoised + n > p->cap_denoised) {
                p->cap_denoised = (p->n_denoised + n) * 2;
                for (int c = 0; c < C; ++c)
                    p->denoised[c] = (float *)realloc(p->denoised[c], sizeof(float) * p->cap_denoised);
            }
            for (int c = 0; c < C; ++c)
                memcpy(p->denoised[c] + p->n_denoised, p->den_result[c], sizeof(float) * n);
            p->n_denoised += n;
        }

        fft_step(p, p->den_result, n, result_index, &den_meta);   /* :184 */

        if (n_remaining_input == 0) return;                       /* :186 */
        input_offset = n - n_remaining_input;                     /* :187 */
    }
}

uint64_t orc_pipeline_push_samples(orc_pipeline *p, const float *const *channel_pcm, size_t n)
{
    const int C = p->cfg.n_channels;
    const uint64_t first_sample_index = p->total_write_count; /* AudioPipeline.zig:119 */
    /* AudioPipeline.zig:121-140: write at most capacity / 2 samples, run the pipeline, repeat; the loop ends
     * with the first short write (so a push of an exact multiple makes one more, empty, round).  The chunk
     * sequence does not depend on this granularity, the moment a recording is finalised does. */
    const size_t capacity = p->cfg.buffer_length ? (size_t)p->cfg.buffer_length : (size_t)p->cfg.sample_rate * 10; /* :46 */
    const size_t write_chunk_size = capacity / 2;
    float **view = (float **)malloc(sizeof(float *) * (size_t)C);
    size_t read_offset = 0;
    for (;;) {
        const size_t step = (n - read_offset < write_chunk_size) ? n - read_offset : write_chunk_size;
        if (p->cfg.keep_denoised) maybe_finalize_recording(p, 0); /* original_audio_recorder.recordBeforeMRBWrite */
        if (p->n_pending + step > p->cap_pending) {
            p->cap_pending = (p->n_pending + step) * 2 + p->chunk_size;
            for (int c = 0; c < C; ++c)
                p->pending[c] = (float *)realloc(p->pending[c], sizeof(float) * p->cap_pending);
        }
        for (int c = 0; c < C; ++c)
            memcpy(p->pending[c] + p->n_pending, channel_pcm[c] + read_offset, sizeof(float) * step);
        p->n_pending += step;
        if (p->cfg.keep_denoised) {
            if (p->n_original + step > p->cap_original) {
                p->cap_original = (p->n_original + step) * 2;
                for (int c = 0; c < C; ++c)
                    p->original[c] = (float *)realloc(p->original[c], sizeof(float) * p->cap_original);
            }
            for (int c = 0; c < C; ++c)
                memcpy(p->original[c] + p->n_original, channel_pcm[c] + read_offset, sizeof(float) * step);
            p->n_original += step;
        }
        p->total_write_count += step;
        read_offset += step;

        /* maybeRunPipeline -> VADPipeline.collectInputStep, VADPipeline.zig:144-166 */
        size_t consumed = 0;
        while (p->total_write_count - p->pipeline_read_count >= p->chunk_size) {
            const uint64_t from = p->pipeline_read_count;
            p->pipeline_read_count = from + p->chunk_size;
            for (int c = 0; c < C; ++c) view[c] = p->pending[c] + consumed;
            process_chunk(p, view, from);
            consumed += p->chunk_size;
        }
        if (consumed) {
            for (int c = 0; c < C; ++c)
                memmove(p->pending[c], p->pending[c] + consumed, sizeof(float) * (p->n_pending - consumed));
            p->n_pending -= consumed;
        }
        if (step < write_chunk_size) break;
    }
    free(view);
    return first_sample_index;
}

size_t orc_pipeline_n_segments(const orc_pipeline *p) { return orc_vad_n_segments(p->vad); }
const orc_speech_segment *orc_pipeline_segments(const orc_pipeline *p) { return orc_vad_segments(p->vad); }
size_t orc_pipeline_n_fft_frames(const orc_pipeline *p) { return p->n_frames; }
const float *orc_pipeline_band_volumes(const orc_pipeline *p) { return p->band_volumes; }
const float *orc_pipeline_frame_vol_ratio(const orc_pipeline *p) { return p->frame_ratio; }
const orc_vad_trace *orc_pipeline_vad_traces(const orc_pipeline *p) { return orc_vad_traces(p->vad); }
const float *orc_pipeline_chunk_rms(const orc_pipeline *p) { return p->chunk_rms; }
size_t orc_pipeline_n_chunks(const orc_pipeline *p) { return p->n_chunks; }
const float *orc_pipeline_denoised(const orc_pipeline *p, int channel) { return p->denoised[channel]; }
size_t orc_pipeline_n_denoised(const orc_pipeline *p) { return p->n_denoised; }
size_t orc_pipeline_n_recordings(const orc_pipeline *p, int which) { return p->rec[which ? 1 : 0].n_recs; }
const float *orc_pipeline_recording(const orc_pipeline *p, int which, size_t i, uint64_t *start,
                                    size_t *length, int *best_channel)
{
    which = which ? 1 : 0;
    if (i >= p->rec[which].n_recs) return NULL;
    if (start) *start = p->rec[which].recs[i].start;
    if (length) *length = p->rec[which].recs[i].length;
    if (best_channel) *best_channel = p->rec[which].recs[i].best;
    return p->rec[which].recs[i].pcm;
}
const float *orc_pipeline_fft_bins(const orc_pipeline *p, size_t frame, int channel)
{
    if (!p->all_bins) return NULL;
    const size_t n_bins = (size_t)orc_fft_bin_count(p->cfg.fft_size);
    return p->all_bins + (frame * (size_t)p->cfg.n_channels + (size_t)channel) * n_bins;
}
