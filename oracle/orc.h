/*
 * orc.h -- CPU ORACLE for the Formula-VAD spectral front end + NSNet2 + VAD path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liborc.so.  The product (libfvad_hip.so) never
 * links, loads or calls anything in this directory.
 *
 * PARITY UNPINNED for all floating-point arithmetic: the reference (recursiveGecko/Formula-VAD,
 * Zig 0.11-dev) delegates every FFT to kissfft and the network to ONNX Runtime, both un-vendored
 * git submodules that are absent from /root/reference (.gitmodules:1-9, no pinned commit), the
 * model blob is a missing LFS object (.MISSING_LARGE_BLOBS:1), and the reference's own tests hold
 * no golden vector for FFT.zig / NSNet2.zig / BufferedFFT / VADMachine.  The
 * reference cannot be compiled here (no zig).  What IS pinned: the literal unit cases of
 * SegmentWriter.zig:130-181, VADMetadata.zig:70-110 and statistics.zig:286-360 (tests/golden),
 * and the FFT against numpy.fft in float64 as a mathematical cross-check.
 *
 * Every function cites the reference file:line it restates.  Paths are relative to
 * /root/reference/.  Third-party algorithms restated from their published form:
 *   kissfft  (github.com/mborgerding/kissfft, unpinned)  -> orc_fft.c
 *   ONNX GRU/MatMul/Add/Relu/Sigmoid operator semantics (onnx.ai operator spec; the graph is
 *   Microsoft DNS-Challenge NSNet2-baseline) -> orc_nsnet2.c
 */
#ifndef ORC_H
#define ORC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- FFT (src/FFT.zig) */

typedef struct { float r, i; } orc_cpx; /* FFT.zig:12-19 `Complex`, == kiss_fft_cpx */

typedef struct orc_fftr orc_fftr;

/* kiss_fftr_alloc(nfft, inverse, NULL, NULL)  -- call site FFT.zig:52-57 */
orc_fftr *orc_fftr_alloc(int nfft, int inverse);
void orc_fftr_free(orc_fftr *cfg);                                   /* FFT.zig:79 */
/* kiss_fftr: nfft reals -> nfft/2+1 complex, forward, unscaled        FFT.zig:108-112 */
void orc_fftr_forward(orc_fftr *cfg, const float *timedata, orc_cpx *freqdata);
/* kiss_fftri: nfft/2+1 complex -> nfft reals, unscaled (== nfft * x)  FFT.zig:129-133 */
void orc_fftr_inverse(orc_fftr *cfg, const orc_cpx *freqdata, float *timedata);

/* FFT.fft: window-multiply (loadSamplesFwd, FFT.zig:183-199) then kiss_fftr.
 * `first`/`second` model SplitSlice (structures/SplitSlice.zig:12-14). Returns 0 or a
 * negative ORC_ERR_* mirroring FFT.zig:92,96,101. */
int orc_fft_fft(orc_fftr *cfg, const float *first, size_t n_first, const float *second,
                size_t n_second, const float *window, size_t n_window, orc_cpx *bins,
                size_t n_bins);
int orc_fft_bin_count(int n_fft);                                    /* FFT.zig:137-139 */
/* FFT.freqToBin (FFT.zig:156-167): returns bin >= 0, or ORC_ERR_OUT_OF_RANGE /
 * ORC_ERR_NEGATIVE_FREQUENCY */
long orc_fft_freq_to_bin(int n_fft, int sample_rate, float freq);

enum {
    ORC_OK = 0,
    ORC_ERR_INVALID_FFT_SIZE = -1,      /* FFT.zig:42 */
    ORC_ERR_INVALID_SAMPLES_LENGTH = -2,/* FFT.zig:92 */
    ORC_ERR_INVALID_WINDOW_LENGTH = -3, /* FFT.zig:96 */
    ORC_ERR_INVALID_RESULT_LENGTH = -4, /* FFT.zig:101,125 */
    ORC_ERR_INVALID_BINS_LENGTH = -5,   /* FFT.zig:121 */
    ORC_ERR_OUT_OF_RANGE = -6,          /* FFT.zig:158 */
    ORC_ERR_NEGATIVE_FREQUENCY = -7,    /* FFT.zig:162 */
    ORC_ERR_INVALID_INPUT_LENGTH = -8,  /* NSNet2.zig:168 */
    ORC_ERR_INVALID_SAMPLE_RATE = -9,   /* VADPipeline.zig:57 */
    ORC_ERR_CHANNEL_COUNT_MISMATCH = -10/* SegmentWriter.zig:70 */
};

/* ------------------------------------------- windows (src/audio_utils/window_fn.zig) */
void orc_hann_window_symmetric(float *result, size_t n);             /* window_fn.zig:30-41 */
void orc_hann_window_periodic(float *result, size_t n);              /* window_fn.zig:22-28,51-68 */
float orc_window_norm_factor(const float *window, size_t n);         /* window_fn.zig:8-16 */
void orc_nsnet2_create_window(float *window320);                     /* NSNet2.zig:384-396 */

/* ------------------------------------------ resample (src/audio_utils/resample.zig) */
void orc_downsample(const float *first, size_t n_first, const float *second, size_t n_second,
                    float *out, size_t n_out, size_t rate);          /* resample.zig:9-29 */
float orc_upsample(const float *in, size_t n_in, float *out, size_t n_out,
                   float prev_last_sample, size_t rate);             /* resample.zig:32-79 */

/* audio_utils.zig:14-24 rmsVolume */
float orc_rms_volume(const float *first, size_t n_first, const float *second, size_t n_second);

/* ---------------------------------------------------------------- NSNet2 (src/NSNet2.zig) */

/* NSNet2-baseline weights.  All matrices are [out][in] row-major (y = W x + b); GRU tensors
 * use the ONNX operator layout and gate order z,r,h: W [3H][in], R [3H][H],
 * B [6H] = {Wb_z, Wb_r, Wb_h, Rb_z, Rb_r, Rb_h}. */
typedef struct {
    int32_t n_bins;   /* 161 */
    int32_t n_fc1;    /* 400 */
    int32_t n_hidden; /* 400 */
    int32_t n_fc2;    /* 600 */
    int32_t n_fc3;    /* 600 */
    const float *fc1_w, *fc1_b;
    const float *gru1_w, *gru1_r, *gru1_b;
    const float *gru2_w, *gru2_r, *gru2_b;
    const float *fc2_w, *fc2_b;
    const float *fc3_w, *fc3_b;
    const float *fc4_w, *fc4_b;
} orc_nsnet2_weights;

/* The ONNX graph behind onnx_instance.run() (NSNet2.zig:220): features [T][161] -> gains
 * [T][161], GRU hidden state zero at row 0 of every call (no state tensors exist on the
 * session, NSNet2.zig:57-58,71-112). */
void orc_nsnet2_forward(const orc_nsnet2_weights *w, const float *features, int T, float *gains);

typedef struct orc_nsnet2 orc_nsnet2;
/* NSNet2.init (NSNet2.zig:35-142).  sample_rate must be a multiple of 16000. */
orc_nsnet2 *orc_nsnet2_create(int sample_rate, const orc_nsnet2_weights *w);
void orc_nsnet2_destroy(orc_nsnet2 *d);                              /* NSNet2.zig:144-155 */
size_t orc_nsnet2_chunk_size(int sample_rate);                       /* NSNet2.zig:157-159 */
/* NSNet2.denoise (NSNet2.zig:161-237) */
int orc_nsnet2_denoise(orc_nsnet2 *d, const float *first, size_t n_first, const float *second,
                       size_t n_second, float *denoised, size_t n_denoised);
/* internals, for traces: features [54][161], gains [54][161], specgram [50][161] (after the
 * gain was applied), audio_input/audio_output [8160] */
const float *orc_nsnet2_features(const orc_nsnet2 *d);
const float *orc_nsnet2_gains(const orc_nsnet2 *d);
const orc_cpx *orc_nsnet2_specgram(const orc_nsnet2 *d);
const float *orc_nsnet2_audio_output(const orc_nsnet2 *d);
/* NSNet2.calcSpectrogram + calcFeatures only (NSNet2.zig:239-287) on an 8160-sample buffer */
void orc_nsnet2_spec_features(const float *audio_input8160, orc_cpx *spec, float *features);

/* ------------------------------------- RollingAverage (src/structures/RollingAverage.zig) */
typedef struct orc_rolling_average orc_rolling_average;
orc_rolling_average *orc_ra_create(size_t count, int has_initial, double initial_val); /* :11-28 */
void orc_ra_destroy(orc_rolling_average *ra);
double orc_ra_push(orc_rolling_average *ra, float sample);           /* :34-43 */
int orc_ra_last_avg(const orc_rolling_average *ra, double *out);     /* 1 if non-null */

/* ---------------------------------- VADMetadata (src/AudioPipeline/VADMetadata.zig:16-60) */
typedef struct {
    int has_ratio, has_min, has_max;
    float volume_ratio, volume_min, volume_max;
} orc_meta_result;                                                   /* VADMetadata.zig:5-9 */
typedef struct {
    int has_ratio, has_min, has_max;
    float ratio_sum, ratio_weight, volume_min, volume_max;
} orc_meta;                                                          /* VADMetadata.zig:11-14 */
void orc_meta_reset(orc_meta *m);
void orc_meta_push(orc_meta *m, const orc_meta_result *values, float weight); /* :29-60 */
orc_meta_result orc_meta_to_result(const orc_meta *m);               /* :16-27 */

/* ----------------------------- VADMachine (src/AudioPipeline/VADMachine.zig) */
typedef struct {
    float speech_min_freq;             /* 500 */
    float speech_max_freq;             /* 2000 */
    float long_term_speech_avg_sec;    /* 180 */
    int32_t has_initial_long_term_avg; /* 1 */
    double initial_long_term_avg;      /* 0.005 */
    float short_term_speech_avg_sec;   /* 0.2 */
    float speech_threshold_factor;     /* 10 */
    float channel_vol_ratio_avg_sec;   /* 0.5 */
    float channel_vol_ratio_threshold; /* 0.5 */
    float min_consecutive_sec_to_open; /* 0.2 */
    float max_speech_gap_sec;          /* 2 */
    float min_vad_duration_sec;        /* 0.7 */
} orc_vad_config;                                                    /* VADMachine.zig:30-51 */
void orc_vad_config_default(orc_vad_config *c);

typedef struct {
    uint64_t sample_from, sample_to;
    float avg_channel_vol_ratio, vad_met_sec;
} orc_speech_segment;                                                /* VADPipeline.zig:28-33 */

enum { ORC_REC_NONE = 0, ORC_REC_STARTED = 1, ORC_REC_COMPLETED = 2, ORC_REC_ABORTED = 3 };
typedef struct { int32_t recording_state; uint64_t sample_number; } orc_vad_result; /* :18-28 */

/* one record per VADMachine.run call, for the margin audit and the traces */
typedef struct {
    uint64_t index;
    float min_volume;
    double short_term, channel_vol_ratio, threshold;
    int32_t threshold_met, state_after;
} orc_vad_trace;

typedef struct orc_vad orc_vad;
orc_vad *orc_vad_create(const orc_vad_config *cfg, int sample_rate, int n_channels, int fft_size);
void orc_vad_destroy(orc_vad *v);
/* VADMachine.run (VADMachine.zig:138-239) given the per-channel band volumes already summed
 * (averageVolumeInBand) and the frame's vad_metadata.volume_ratio (has_ratio=0 -> null). */
orc_vad_result orc_vad_run(orc_vad *v, uint64_t index, const float *channel_volumes,
                           int has_ratio, float volume_ratio);
size_t orc_vad_n_segments(const orc_vad *v);
const orc_speech_segment *orc_vad_segments(const orc_vad *v);
size_t orc_vad_n_trace(const orc_vad *v);
const orc_vad_trace *orc_vad_traces(const orc_vad *v);

/* ------------------- whole pipeline (AudioPipeline.zig + AudioPipeline/VADPipeline.zig) */
typedef struct orc_pipeline orc_pipeline;
typedef struct {
    int32_t sample_rate;      /* AudioPipeline.zig:20-26 */
    int32_t n_channels;
    int32_t fft_size;         /* VADPipeline.zig:21 */
    int32_t keep_denoised;    /* oracle-only: retain the denoised PCM that pushDenoisedSamples
                                 (VADPipeline.zig:183) hands back, for parity traces */
    int32_t buffer_length;    /* AudioPipeline.Config.buffer_length (:24,46); 0 = sample_rate * 10 */
    orc_vad_config vad;
} orc_pipeline_config;
void orc_pipeline_config_default(orc_pipeline_config *c);
/* AudioPipeline.init -> VADPipeline.init (VADPipeline.zig:51-126); NULL + *err on failure */
orc_pipeline *orc_pipeline_create(const orc_pipeline_config *cfg, const orc_nsnet2_weights *w,
                                  int *err);
void orc_pipeline_destroy(orc_pipeline *p);
/* AudioPipeline.pushSamples (AudioPipeline.zig:118-143): channel-planar; returns the absolute
 * index of the first pushed sample */
uint64_t orc_pipeline_push_samples(orc_pipeline *p, const float *const *channel_pcm, size_t n);
size_t orc_pipeline_n_segments(const orc_pipeline *p);
const orc_speech_segment *orc_pipeline_segments(const orc_pipeline *p);
/* traces: one entry per BufferedFFT result */
size_t orc_pipeline_n_fft_frames(const orc_pipeline *p);
const float *orc_pipeline_band_volumes(const orc_pipeline *p);   /* [n_fft_frames][n_channels] */
const float *orc_pipeline_frame_vol_ratio(const orc_pipeline *p);/* [n_fft_frames] */
const orc_vad_trace *orc_pipeline_vad_traces(const orc_pipeline *p);
const float *orc_pipeline_chunk_rms(const orc_pipeline *p);      /* [n_chunks][n_channels] */
size_t orc_pipeline_n_chunks(const orc_pipeline *p);
const float *orc_pipeline_denoised(const orc_pipeline *p, int channel); /* keep_denoised only */
size_t orc_pipeline_n_denoised(const orc_pipeline *p);
/* Recordings (MRBRecorder.zig:76-203 + Recorder.zig:60-164), kept only when keep_denoised: on a
 * `started` result both recorders begin at its sample_number (dropping an end that is still pending);
 * on `completed` each recorder notes end_recording_on_sample and finalises as soon as ITS buffer holds
 * the samples up to it -- at once, or in front of a later write (recordBeforeMRBWrite: before every
 * <= capacity/2 write step of pushSamples for the original audio, before every 0.5 s chunk of
 * pushDenoisedSamples for the denoised audio).  A clip is the quietest channel
 * (Recorder.findBestChannel, :113-129) over [start, end).  which: 0 = original, 1 = denoised. */
size_t orc_pipeline_n_recordings(const orc_pipeline *p, int which);
const float *orc_pipeline_recording(const orc_pipeline *p, int which, size_t i, uint64_t *start,
                                    size_t *length, int *best_channel);
/* full 513-bin magnitudes of BufferedFFT result `frame` (kept only when keep_denoised) */
const float *orc_pipeline_fft_bins(const orc_pipeline *p, size_t frame, int channel);

/* BufferedFFT.fft on one 1024-sample frame (BufferedFFT.zig:162-181) + band sum (:183-202) */
void orc_buffered_fft_frame(const float *samples, int fft_size, float *bins_out);
float orc_band_sum(const float *bins, long min_bin, long max_bin);

/* SegmentWriter semantics (SegmentWriter.zig:46-114), single channel, for the literal test */
typedef struct orc_segment_writer orc_segment_writer;
orc_segment_writer *orc_sw_create(size_t length);
void orc_sw_destroy(orc_segment_writer *sw);
size_t orc_sw_write(orc_segment_writer *sw, const float *first, size_t n_first,
                    const float *second, size_t n_second, size_t read_offset);
int orc_sw_is_full(const orc_segment_writer *sw);
void orc_sw_reset(orc_segment_writer *sw, uint64_t new_index);
const float *orc_sw_data(const orc_segment_writer *sw);
size_t orc_sw_write_index(const orc_segment_writer *sw);
uint64_t orc_sw_index(const orc_segment_writer *sw);

/* ------------------------------------- Evaluator + statistics (src/Evaluator*) */
typedef struct {
    float total_positives_sec, true_positives_sec, false_positives_sec, false_negatives_sec;
    float true_positive_rate, false_negative_rate, false_discovery_rate, precision;
    float fm_index, f_score, f_score_beta;
} orc_single_stats;                                                  /* statistics.zig:8-37 */
typedef struct { float overall, min, max, avg; } orc_agg_stat;       /* statistics.zig:39-44 */
typedef struct {
    float total_positives_sec, true_positives_sec, false_positives_sec, false_negatives_sec;
    orc_agg_stat true_positive_rate, false_negative_rate, false_discovery_rate, precision;
    float fm_index, f_score, f_score_beta;
} orc_aggregate_stats;                                               /* statistics.zig:46-75 */
typedef struct {
    float ignore_shorter_than_sec, extrude_start, extrude_end, fill_gaps;
} orc_stat_config;                                                   /* statistics.zig:77-83 */
typedef struct { float from_sec, to_sec; } orc_seg_sec;

/* Evaluator.initAndRun (Evaluator.zig:90-156) + statistics.fromEvaluator (statistics.zig:85-114)
 */
orc_single_stats orc_stats_from_segments(const orc_seg_sec *vad, size_t n_vad,
                                         const orc_seg_sec *ref, size_t n_ref,
                                         const orc_stat_config *cfg);
orc_aggregate_stats orc_stats_aggregate(const orc_single_stats *stats, size_t n); /* :116-172 */
/* statistics.calcFalsePositiveSec (statistics.zig:191-203) with explicit matched refs */
float orc_calc_false_positive_sec(orc_seg_sec vad, const orc_seg_sec *matched_refs, size_t n,
                                  const orc_stat_config *cfg);
/* SimulationInstance.storeResult sample->seconds (SimulationInstance.zig:237-238) */
orc_seg_sec orc_segment_to_sec(const orc_speech_segment *s, int sample_rate);

#ifdef __cplusplus
}
#endif
#endif
