/*
 * fvad.h -- C ABI of libfvad_hip.so: the MI355X (gfx950) implementation of Formula-VAD's
 * per-frame spectral front end + NSNet2 denoiser + VAD decision path.
 *
 * The reference (recursiveGecko/Formula-VAD) is Zig; its hot path sits behind three nested Zig
 * seams: B1 AudioPipeline (src/AudioPipeline.zig), B2 NSNet2
 * (src/NSNet2.zig), B3 FFT (src/FFT.zig), which bottom out in two C-ABI dependencies, kissfft
 * and ONNX Runtime.  Every entry point below names the reference interface it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the Zig `extern` block a
 * maintainer adds (bindings/fvad.zig).
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns an int
 * status (0 = FVAD_OK, negative = the reference's Zig error of the same name) unless it is a
 * pure query; nothing throws or aborts across this boundary.  The caller owns every sample
 * buffer and the callee never keeps a pointer past the call (same contract as the ring-buffer
 * slices the reference hands around, src/structures/MultiRingBuffer.zig:159-161).  Objects are
 * thread-confined like the reference's (one AudioPipeline per OS thread,
 * src/simulator.zig:225-231): one fvad_ctx = one HIP device + one HIP stream.
 *
 * Audio is channel-planar f32 (`[][]f32`, src/AudioPipeline.zig:118); `Complex` is
 * {f32 r; f32 i} (src/FFT.zig:12-14, == kiss_fft_cpx); sample indices are u64
 * (src/AudioPipeline/Segment.zig:22).
 *
 * The functions that need the GPU fail with FVAD_ERR_NO_DEVICE when no gfx950 device is
 * present; there is no CPU fallback.  The VAD state machine / Evaluator entry points are host
 * code by design (src/AudioPipeline/VADMachine.zig is sequential per stream) and work without
 * a device.
 */
#ifndef FVAD_H
#define FVAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FVAD_ABI_VERSION 3

/* ------------------------------------------------------------------ status codes */
enum {
    FVAD_OK = 0,
    FVAD_ERR_INVALID_FFT_SIZE = -1,       /* error.InvalidFFTSize        FFT.zig:42 */
    FVAD_ERR_INVALID_SAMPLES_LENGTH = -2, /* error.InvalidSamplesLength  FFT.zig:92 */
    FVAD_ERR_INVALID_WINDOW_LENGTH = -3,  /* error.InvalidWindowLength   FFT.zig:96 */
    FVAD_ERR_INVALID_RESULT_LENGTH = -4,  /* error.InvalidResultLength   FFT.zig:101,125 */
    FVAD_ERR_INVALID_BINS_LENGTH = -5,    /* error.InvalidBinsLength     FFT.zig:121 */
    FVAD_ERR_OUT_OF_RANGE = -6,           /* error.OutOfRange            FFT.zig:158,174 */
    FVAD_ERR_NEGATIVE_FREQUENCY = -7,     /* error.NegativeFrequency     FFT.zig:162 */
    FVAD_ERR_INVALID_INPUT_LENGTH = -8,   /* error.InvalidInputLength    NSNet2.zig:168 */
    FVAD_ERR_INVALID_SAMPLE_RATE = -9,    /* error.InvalidSampleRate     VADPipeline.zig:57 */
    FVAD_ERR_CHANNEL_COUNT_MISMATCH = -10,/* error.ChannelCountMismatch  SegmentWriter.zig:70 */
    FVAD_ERR_ALLOC_FAILED = -11,          /* error.KissFFTAllocFailed / OutOfMemory FFT.zig:59 */
    /* errors with no reference counterpart */
    FVAD_ERR_INVALID_ARGUMENT = -100,
    FVAD_ERR_NO_DEVICE = -101,            /* no gfx950 device / HIP runtime unavailable */
    FVAD_ERR_HIP = -102,                  /* a HIP call failed; text in fvad_last_error */
    FVAD_ERR_NO_MODEL = -103,             /* NSNet2 weights not loaded */
    FVAD_ERR_MODEL_FORMAT = -104,         /* ONNX file unreadable / not the NSNet2 graph */
    FVAD_ERR_IO = -105,
    FVAD_ERR_BUFFER_TOO_SMALL = -106
};

const char *fvad_status_name(int status);
int fvad_abi_version(void);

/* ------------------------------------------------------------------ context */
typedef struct fvad_ctx fvad_ctx;

/* Binds HIP device `device` (hipSetDevice) and creates the context's stream and constant
 * tables (windows, twiddles).  Replaces the allocator argument every reference init takes. */
int fvad_ctx_create(int device, fvad_ctx **out);
void fvad_ctx_destroy(fvad_ctx *ctx);
/* Text of the last failure on this context ("" if none).  Valid until the next call. */
const char *fvad_last_error(const fvad_ctx *ctx);
/* Blocks until all work queued on the context's stream has finished. */
int fvad_ctx_synchronize(fvad_ctx *ctx);
/* The context's hipStream_t as an opaque pointer, so a caller can time it with HIP events. */
void *fvad_ctx_stream(fvad_ctx *ctx);
/* hipMemcpyAsync(device -> host) on the context's stream: ordered after everything queued so
 * far; the bytes are valid after fvad_ctx_synchronize. */
int fvad_ctx_copy_to_host(fvad_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);
/* Page-locked host memory for audio buffers handed to fvad_engine_run / fvad_pipeline_push_samples.
 * Optional: any host pointer is accepted, but pageable memory has to be staged through pinned rings
 * (the host side of that moves ~25-45 GB/s), while buffers from this allocator are copied by the DMA
 * engine directly (57 GB/s measured).  The allocator replaces the std.mem.Allocator the reference
 * injects for its sample buffers (AudioPipeline.zig:40-44).  Free with fvad_host_free. */
int fvad_host_alloc(fvad_ctx *ctx, size_t bytes, void **out);
void fvad_host_free(fvad_ctx *ctx, void *p);
/* Device (HBM) memory on the context's device, for callers that keep audio resident on the GPU
 * (fvad_engine_enqueue_device, `on_device` lanes) without linking a HIP runtime themselves: a Zig or C
 * host needs nothing but this library.  fvad_ctx_copy_to_device is a hipMemcpyAsync(host -> device)
 * on the context's stream: `src_host` must stay valid until fvad_ctx_synchronize. */
int fvad_device_alloc(fvad_ctx *ctx, size_t bytes, void **out);
void fvad_device_free(fvad_ctx *ctx, void *p);
int fvad_ctx_copy_to_device(fvad_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);

/* ------------------------------------------------------------------ NSNet2 model
 * Replaces onnx.OnnxInstance.init(allocator, .{ .model_path = ... }) (NSNet2.zig:53-61): the
 * model is loaded once per context and shared by every denoiser/pipeline made from it. */

/* Host-side weights in the ONNX operator layout: all matrices [out][in] row-major; GRU tensors
 * W [3H][in], R [3H][H], B [6H] = {Wb_z,Wb_r,Wb_h,Rb_z,Rb_r,Rb_h}, gate order z,r,h,
 * linear_before_reset = 1. */
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
} fvad_nsnet2_weights;

/* Reads nsnet2-20ms-baseline.onnx (the file NSNet2.zig:56 names) with a built-in protobuf
 * reader; dimensions are taken from the file. */
int fvad_load_nsnet2_onnx(fvad_ctx *ctx, const char *onnx_path);
/* Takes weights from host memory (copied). */
int fvad_load_nsnet2_weights(fvad_ctx *ctx, const fvad_nsnet2_weights *w);
/* Seeded synthetic weights of the NSNet2-baseline architecture (for benchmarks and tests: the
 * real model file is not redistributable here). */
int fvad_load_nsnet2_synth(fvad_ctx *ctx, uint64_t seed);
/* Borrow the host copy of the loaded weights (valid until the next load / ctx destroy). */
int fvad_get_nsnet2_weights(const fvad_ctx *ctx, fvad_nsnet2_weights *out);
/* Host-only helpers (no device needed): */
int fvad_onnx_read_nsnet2(const char *onnx_path, fvad_nsnet2_weights *out, void **owner);
int fvad_synth_nsnet2(uint64_t seed, fvad_nsnet2_weights *out, void **owner);
void fvad_weights_free(void *owner);

/* ------------------------------------------------------------------ B3: FFT  (src/FFT.zig)
 * Replaces FFT.init/fft/invFft/deinit and, underneath, kiss_fftr_alloc / kiss_fftr /
 * kiss_fftri / kiss_fftr_free (FFT.zig:52-57,108-112,129-133,79). */
typedef struct { float r, i; } fvad_complex; /* FFT.zig:12-14 */
typedef struct fvad_fft fvad_fft;

/* FFT.init(allocator, n_fft, sample_rate, mode_inverse)  FFT.zig:35-76.  Any even n_fft from 4 to 16384, forward or inverse,
 * like kiss_fftr_alloc (odd or zero: FVAD_ERR_INVALID_FFT_SIZE, FFT.zig:41-43; so are 2 and sizes past 16384, this library's
 * limits).  The sizes the pipeline runs at have wavefront kernels -- 320 (forward and inverse: NSNet2's STFT), 512 / 1024 /
 * 2048 (forward: the VAD-side transform, VADPipeline.Config.fft_size); every other case runs on a generic mixed-radix kernel
 * (one workgroup per frame, any radix; correct, not tuned). */
int fvad_fft_create(fvad_ctx *ctx, size_t n_fft, size_t sample_rate, int mode_inverse,
                    fvad_fft **out);
void fvad_fft_destroy(fvad_fft *fft);                                   /* FFT.deinit :78-83 */
/* FFT.fft(samples: SplitSlice, window, bins)  FFT.zig:85-113.  Host pointers. */
int fvad_fft_forward(fvad_fft *fft, const float *first, size_t n_first, const float *second,
                     size_t n_second, const float *window, size_t n_window, fvad_complex *bins,
                     size_t n_bins);
/* FFT.invFft(bins, result)  FFT.zig:115-134: unscaled inverse (== n_fft * x). */
int fvad_fft_inverse(fvad_fft *fft, const fvad_complex *bins, size_t n_bins, float *result,
                     size_t n_result);
size_t fvad_fft_bin_count(const fvad_fft *fft);                         /* FFT.zig:137-139 */
float fvad_fft_bin_width(const fvad_fft *fft);                          /* :142-147 */
float fvad_fft_nyquist_freq(const fvad_fft *fft);                       /* :150-153 */
int fvad_fft_freq_to_bin(const fvad_fft *fft, float freq, size_t *bin); /* :156-167 */
int fvad_fft_bin_to_freq(const fvad_fft *fft, size_t bin, float *freq); /* :170-180 */
/* The batched form the GPU wants (BASELINE config 2): n_frames contiguous frames of n_fft
 * samples -> bins [n_frames][n_fft/2+1] and/or magnitudes [n_frames][n_fft/2+1]; either output
 * may be NULL.  `on_device` != 0: all pointers are device pointers and the call only enqueues. */
int fvad_fft_forward_batch(fvad_fft *fft, const float *frames, size_t n_frames,
                           const float *window, fvad_complex *bins, float *magnitudes,
                           int on_device);

/* window_fn.zig:22-41, 8-16 and NSNet2.zig:384-396 (host; same f32 arithmetic as the reference) */
void fvad_hann_window_periodic(float *result, size_t n);
void fvad_hann_window_symmetric(float *result, size_t n);
float fvad_window_norm_factor(const float *window, size_t n);
void fvad_nsnet2_window(float *window320);

/* ------------------------------------------------------------------ B2: NSNet2 (src/NSNet2.zig) */
typedef struct fvad_nsnet2 fvad_nsnet2;
/* NSNet2.init(allocator, sample_rate, model_path)  NSNet2.zig:35-142.  The model comes from the
 * context.  One object per channel, like BufferedDenoiser.zig:38-41.  sample_rate: any multiple of
 * 16000 Hz like the reference (resample.zig:4-7: calcDownsampleRate), FVAD_ERR_INVALID_SAMPLE_RATE
 * otherwise; chunks are fvad_nsnet2_chunk_size(sample_rate) = 8000 * (sample_rate / 16000) samples. */
int fvad_nsnet2_create(fvad_ctx *ctx, size_t sample_rate, fvad_nsnet2 **out);
void fvad_nsnet2_destroy(fvad_nsnet2 *d);                               /* NSNet2.zig:144-155 */
size_t fvad_nsnet2_chunk_size(size_t in_sample_rate);                   /* NSNet2.zig:157-159 */
/* NSNet2.denoise(samples: SplitSlice, denoised_result)  NSNet2.zig:161-237.  Host pointers;
 * carries the same cross-chunk state (input hop, overlap-add tail, 4 feature rows,
 * last_sample: NSNet2.zig:27-33). */
int fvad_nsnet2_denoise(fvad_nsnet2 *d, const float *first, size_t n_first, const float *second,
                        size_t n_second, float *denoised_result, size_t n_result);

/* ------------------------------------------------------------------ batched engine
 * What simulator.zig would call with preload_audio = true: whole streams in, per-stream VAD
 * inputs out.  A "lane" is one channel of one stream.  For every lane the engine runs, for all
 * complete 24000-sample chunks at once: chunk RMS (BufferedVolumeAnalyzer.zig:48-69), decimate +
 * sqrt-Hann STFT-320 + log-power features (NSNet2.zig:205-219), NSNet2 (NSNet2.zig:220), gain +
 * inverse STFT overlap-add + x3 upsample (NSNet2.zig:221-236), then the 1024-point periodic-Hann
 * rFFT magnitude and 500-2000 Hz band sum of the denoised audio (BufferedFFT.zig:162-202). */
typedef struct fvad_lane_state fvad_lane_state; /* cross-call carry of one lane (device) */
int fvad_lane_state_create(fvad_ctx *ctx, fvad_lane_state **out);
void fvad_lane_state_reset(fvad_lane_state *s);
void fvad_lane_state_destroy(fvad_lane_state *s);
/* Time-split sharding of ONE long stream over several GPUs (BASELINE config 5).  The ONNX
 * session carries no state across chunks (NSNet2.zig:57-58,71-112), so what crosses a chunk edge is short: the
 * 160-sample input hop, the 4 warm-up feature rows, the overlap-add tail and the upsampler's last sample
 * (NSNet2.zig:27-33,188-203), all functions of the previous chunk and of the 4 last frames of the one before.  A
 * lane that starts TWO chunks early from zero history (this call, sample_index = 24000 * (c0 - 2)) is therefore
 * bit-identical to the unsplit stream from chunk c0 on -- when both runs select the same kernels (see
 * fvad_ctx_set_option: "reproducible"; ~1e-6 apart otherwise) --; the VAD FFT's frame grid stays anchored at sample 0
 * (first_frame_index of the next fvad_engine_run says where the lane's first frame starts).  The caller drops
 * the two warm-up chunks and the frames that start before 24000 * c0.  sample_index: a multiple of 24000. */
int fvad_lane_state_seek(fvad_lane_state *s, uint64_t sample_index, size_t fft_size /* 0 = 1024 */);

typedef struct {
    const float *pcm;        /* n_samples f32 @48 kHz (host or device, see on_device); NULL: use pcm_i16 */
    size_t n_samples;        /* only floor(n/24000) chunks are consumed */
    fvad_lane_state *state;  /* NULL = fresh stream (zero history), state not kept */
    float *denoised;         /* out, optional: n_chunks*24000 f32 (same memory space as pcm) */
    float *band_sum;         /* out: one f32 per completed 1024-sample frame */
    size_t band_sum_capacity;
    float *chunk_rms;        /* out: one f32 per chunk */
    size_t chunk_rms_capacity;
    float *fft_bins;         /* out, optional (parity/debug): [n_fft_frames][513] magnitudes */
    /* 16-bit transport: the reference decodes PCM16 files to f32 on the host (AudioFileStream.zig:56-102 through
     * libsndfile: s / 32768); here the samples can cross PCIe and HBM as PCM16 and are converted by the kernel
     * that reads them, bit-identical to converting first. */
    const int16_t *pcm_i16;  /* used when pcm == NULL: n_samples PCM16 samples (same memory space as pcm would be) */
    int16_t *denoised_i16;   /* out, optional: n_chunks*24000 samples, rint(clamp(y * 32768, -32768, 32767)) */
    float *spectrogram;      /* out, optional (parity/debug, host): [n_chunks][50][161] {r,i} -- NSNet2.calcSpectrogram's
                                bins before the gain (NSNet2.zig:239-264) */
    float *features;         /* out, optional (parity/debug, host): [n_chunks][54][161] -- the ONNX input rows: 4 warm-up
                                rows (previous chunk's last 4, zeros at t = 0) + calcFeatures (NSNet2.zig:188-203,266-287) */
    /* filled by the call: */
    size_t n_chunks;         /* chunks consumed */
    size_t n_fft_frames;     /* band sums written */
    uint64_t first_frame_index; /* absolute sample index of the first FFT frame's window */
} fvad_lane;

typedef struct {
    int32_t on_device;       /* pcm/denoised are device pointers; outputs band_sum/chunk_rms/
                                fft_bins are always host pointers */
    int32_t min_bin;         /* band edges, inclusive; default 11..43 = freqToBin(500/2000) */
    int32_t max_bin;
    int32_t max_chunks_per_launch; /* 0 = default (49152) */
    int32_t fft_size;        /* frame length of the VAD-side FFT (VADPipeline.Config.fft_size, VADPipeline.zig:21): any even
                                size from 4 to 16384 (512, 1024, 2048: wavefront kernels; others: the generic kernel);
                                0 = 1024.  min_bin / max_bin index that transform's bins and fft_bins rows have
                                fft_size / 2 + 1 entries */
    int32_t no_wait;         /* fvad_engine_enqueue_device* only: 1 = return as soon as the work is queued on the
                                context's stream (results valid after fvad_ctx_synchronize or an event the caller
                                records on fvad_ctx_stream); the next call may be made at once -- descriptor and job
                                tables are double-buffered -- so a caller can keep one batch queued behind the running
                                one.  0 (default) = return when the work has completed */
    int32_t use_graph;       /* fvad_engine_enqueue_device* only: 1 = the call's launch sequence (K1, the NSNet2 kernels,
                                K3 per launch, then K4) is captured into a hipGraph the first time and replayed while the
                                arguments, the model and the workspace stay the same -- the "hipGraph-captured steady-state
                                frame loop" of a long corpus processed batch after batch through the same buffers.  Results
                                are bit-identical to direct launches.  The call returns when the work has completed. */
} fvad_engine_opts;
void fvad_engine_opts_default(fvad_engine_opts *o);

int fvad_engine_run(fvad_ctx *ctx, fvad_lane *lanes, size_t n_lanes, const fvad_engine_opts *opts);

/* Device-resident form: `d_pcm` holds n_lanes lanes of n_samples (lane l at d_pcm + l * lane_stride);
 * d_denoised [n_lanes][n_chunks*24000] (NULL: kept in the context's workspace), d_band_sum
 * [n_lanes][n_chunks*24000/1024] and d_chunk_rms [n_lanes][n_chunks] (may be NULL) are device buffers
 * too; nothing is copied to the host.  Every lane starts from zero history.  The call returns once
 * the work has COMPLETED on the context's stream, unless opts->no_wait is set. */
int fvad_engine_enqueue_device(fvad_ctx *ctx, const float *d_pcm, size_t n_lanes,
                               size_t lane_stride, size_t n_samples, float *d_denoised,
                               float *d_band_sum, float *d_chunk_rms,
                               const fvad_engine_opts *opts);
/* The same with PCM16 device buffers in (and optionally out): half the HBM footprint and transport. */
int fvad_engine_enqueue_device_i16(fvad_ctx *ctx, const int16_t *d_pcm16, size_t n_lanes,
                                   size_t lane_stride, size_t n_samples, int16_t *d_denoised16,
                                   float *d_band_sum, float *d_chunk_rms,
                                   const fvad_engine_opts *opts);
/* NSNet2 graph only: features [n_seq][T][161] -> gains [n_seq][T][161] (host pointers).
 * Replaces onnx_instance.run() (NSNet2.zig:220) for n_seq independent sequences. */
int fvad_nsnet2_forward(fvad_ctx *ctx, const float *features, size_t n_seq, size_t T,
                        float *gains);
/* Arithmetic of the NSNet2 matrix products -- a property of the CONTEXT (and of the loaded model), never of a
 * launch's size: every launch of a context, large or small, uses the same one.
 *   FVAD_NN_MATH_F32 (default): v_mfma_f32_16x16x4_f32 throughout (kernels_nn.hip, kernels_ws.hip): f32 operands,
 *     f32 accumulation, each output a k-ordered chain of f32 fmas -- the arithmetic of the reference's ONNX Runtime
 *     CPU kernels (NSNet2.zig:220);
 *   FVAD_NN_MATH_F16X3 (opt-in, an EMULATION that is narrower than f32): every f32 operand as two f16 pieces of a
 *     power-of-two scaled value (22 significand bits, f32 has 24), three f16 MFMAs with f32 accumulation per product
 *     (kernels_h3.hip), batches padded to 128 sequences.  Measured against float64 as close as the f32 kernels on the
 *     models tried (tests), 2.1 x their speed at saturating batches; differs from them by ~1e-6 in the gains.  A
 *     model whose weights are not finite or whose l1 activation bounds exceed 2^17 is not eligible and keeps f32.
 *   FVAD_NN_MATH_BF16X3 (opt-in, an emulation that is NOT narrower than f32): the five dense layers with every f32
 *     operand as three bf16 pieces (x = h + m + l exactly: all 24 significand bits, f32's exponent range, no scales
 *     or bounds) and the six significant cross terms as six bf16 MFMAs with f32 accumulation per product
 *     (kernels_b3.hip); the two GRU recurrences stay on the f32 matrix cores.  Batches padded to 128 sequences.
 *     NSNet2-baseline dimensions only (other models keep f32).
 * fvad_ctx_set_nn_math returns the previous setting or a negative status.  fvad_ctx_nn_math_effective returns what
 * the context actually uses with the model it has loaded (the request, demoted to F32 for an ineligible model or
 * while an f32 kernel variant is forced through fvad_ctx_set_option); fvad_ctx_last_nn_path names the kernels the
 * last NSNet2 pass ran, e.g. "f32: panel_gemm3 (fc1 folded) + gru_rec3<12>". */
enum { FVAD_NN_MATH_F32 = 0, FVAD_NN_MATH_F16X3 = 1, FVAD_NN_MATH_BF16X3 = 2 };
int fvad_ctx_set_nn_math(fvad_ctx *ctx, int mode);
int fvad_ctx_nn_math_effective(const fvad_ctx *ctx);
const char *fvad_ctx_last_nn_path(const fvad_ctx *ctx);
/* Bit-reproducibility.  Two launches that select the same NSNet2 kernels (fvad_ctx_last_nn_path names them) give a
 * chunk the same bits wherever in the batch it sits and however lanes and chunks are split.  With FVAD_NN_MATH_F32
 * the engine selects by launch size: up to 1536 sequences the pipelined two-layer weight-stationary recurrence (it
 * computes layer 2's input projection itself, and for 65..96 sequences layer 1's too, in the accumulation order of the GEMM
 * that otherwise runs in front: one set of bits for the whole range);
 * up to 2047 the narrow-block GEMMs with a weight-stationary or the low-latency recurrence; from 2048 the persistent GEMM
 * with the low-latency or the multi-wavefront recurrence (by a cost model over the CU count).  A call whose launch size
 * is left to the engine (max_chunks_per_launch = 0) is cut into launches that fill the chip: 1537..3400 chunks run as two or
 * three equal launches, larger calls as launches of 4096 / 8192 / 12288 / 16384 / 32768 / 49152 chunks and a remainder where
 * the measured curve says that pays (5120 = 4096 + 1024).  Every selection runs f32 operands and f32 accumulation; they differ in
 * accumulation order and agree to ~1e-6 in the gains, not bit for bit.  The option "reproducible" = "1" makes every
 * launch use one selection (persistent GEMM + multi-wavefront recurrence; small launches are padded to 128 sequences
 * and lose their low-latency kernels), so that a stream pushed in any pieces, split over any number of launches or
 * time-split over ranks gives the same bits.  FVAD_NN_MATH_F16X3 and FVAD_NN_MATH_BF16X3 have one selection each.
 *
 * Testing / tuning aids, none needed in production: name = "reproducible" | "nn_math" ("f32" | "f16x3" | "bf16x3": overrides
 * fvad_ctx_set_nn_math) | "gru_kernel" ("v3w12" | "v3w8" | "v3w4" | "v4w8" | "v5w0" | "v6w0") | "gemm_kernel" ("v1" |
 * "v3" | "v3nofold") | "h3_waves" ("8" | "12") | "max_chunks" | "copy_threads" | "no_pipeline" | "run_groups" ("1,3,4,8": the lane groups of
 * fvad_engine_run's host-buffer pipeline in sixteenths of the call, at most seven, instead of the planned ones) | "trace_run" (a timeline of every fvad_engine_run call on stderr) | "trace_kernels" |
 * "ws_spin_ticks" | "ws2_variant" (diagnostic bit mask; the timing-only bits exist in the diagnostics build alone) |
 * "ws2_waits" | "ws2_calibrate" (below) | "k4_plain_loads" (the band FFT's staging path of unaligned frames) | "gru_lat_tiles" ("1" | "2" | "3": row tiles per
 * workgroup of the low-latency recurrence instead of the cost model's choice; same bits); value NULL or ""
 * restores the default.  The environment variables FVAD_<NAME> are read ONCE, by
 * fvad_ctx_create, as initial values (a bad value fails the creation); the data path never reads the environment. */
int fvad_ctx_set_option(fvad_ctx *ctx, const char *name, const char *value);
/* Network passes in which a weight-stationary small-batch recurrence (gru_ws_kernel, or the pipelined gru_ws2k / gru_ws2m /
 * gru_ws2) gave up waiting for a peer workgroup -- the chip was shared with another process, or with a long kernel of another
 * context of this process -- and the low-latency kernel redid the GRU layers.  A spin gives up after max(2 ms, 20 x the
 * launch's own expected duration) (option "ws_spin_ticks" overrides).  Bits: gru_ws and its fallback accumulate in the same
 * order (same bits); the pipelined kernels compute layer 2's input projection in the kernel and their fallback does not, so
 * a pass that fell back differs from one that did not by round-off (<= 2e-6 in the gains): default-mode results of small
 * launches are load-dependent within that bound.  "reproducible" = "1" never runs these kernels.  Waits for the context's
 * stream. */
int fvad_ctx_ws_fallbacks(fvad_ctx *ctx, uint64_t *n);
/* The pipelined recurrence of launches up to 96 sequences (gru_ws2k) waits a fixed interval before a step's first poll of
 * its peers' flags -- a poll made too early is a wasted round trip and traffic on the flag lines.  The intervals are a
 * built-in table per group shape (wait_class 1: groups of 25 + 25 workgroups, 1..80 sequences; 2 and 3: groups of 13 + 25,
 * without / with layer 1's input projection in the kernel), swept on one MI355X.  fvad_ctx_set_option(ctx, "ws2_calibrate",
 * "1") measures them on THIS device (about 0.2 s, the model must be loaded; the table's entry stays unless a candidate is
 * more than 1.5 % faster); "ws2_waits" = layer 1's wait | layer 2's << 16, in 10 ns ticks, sets them by hand for every class.
 * Timing only: results do not depend on them.  Returns the waits in effect for a class, packed like "ws2_waits"; 0 for an
 * unknown class. */
uint32_t fvad_ctx_ws2_waits(const fvad_ctx *ctx, int wait_class);
/* Per-kernel device time of the last fvad_engine_* call (HIP events on the context's stream):
 * names[i]/ms[i] for i < *n.  Enabled by fvad_ctx_enable_timing(ctx, 1). */
int fvad_ctx_enable_timing(fvad_ctx *ctx, int on);
int fvad_ctx_kernel_times(fvad_ctx *ctx, const char **names, float *ms, size_t cap, size_t *n);

/* ------------------------------------------------------------------ VAD state machine (host)
 * src/AudioPipeline/VADMachine.zig + src/structures/RollingAverage.zig, exact f64 order. */
typedef struct {
    float speech_min_freq;             /* 500   VADMachine.zig:32 */
    float speech_max_freq;             /* 2000  :33 */
    float long_term_speech_avg_sec;    /* 180   :35 */
    int32_t has_initial_long_term_avg; /* 1     :36 (?f64) */
    double initial_long_term_avg;      /* 0.005 */
    float short_term_speech_avg_sec;   /* 0.2   :38 */
    float speech_threshold_factor;     /* 10    :41 */
    float channel_vol_ratio_avg_sec;   /* 0.5   :43 */
    float channel_vol_ratio_threshold; /* 0.5   :44 */
    float min_consecutive_sec_to_open; /* 0.2   :46 */
    float max_speech_gap_sec;          /* 2     :48 */
    float min_vad_duration_sec;        /* 0.7   :50 */
} fvad_vad_config;
void fvad_vad_config_default(fvad_vad_config *c);

typedef struct {
    uint64_t sample_from, sample_to;
    float avg_channel_vol_ratio, vad_met_sec;
} fvad_speech_segment; /* VADPipeline.SpeechSegment, VADPipeline.zig:28-33 */

enum { FVAD_REC_NONE = 0, FVAD_REC_STARTED = 1, FVAD_REC_COMPLETED = 2, FVAD_REC_ABORTED = 3 };
typedef struct { int32_t recording_state; uint64_t sample_number; } fvad_vad_result; /* :18-28 */

/* smallest decision margins seen so far (the "margin audit" of SURVEY.md section 7): how close
 * any frame came to flipping `short_term > threshold` or `ratio > 0.5` */
typedef struct {
    double min_rel_threshold_margin; /* min |short_term - threshold| / threshold */
    double min_abs_ratio_margin;     /* min |channel_vol_ratio - ratio_threshold| */
    uint64_t n_frames;
} fvad_vad_audit;

typedef struct fvad_vad fvad_vad;
int fvad_vad_create(const fvad_vad_config *cfg, size_t sample_rate, size_t n_channels,
                    size_t fft_size, fvad_vad **out);                  /* VADMachine.init :75-128 */
void fvad_vad_destroy(fvad_vad *v);
/* VADMachine.run(fft_result)  VADMachine.zig:138-239 with the band volumes already summed. */
int fvad_vad_run(fvad_vad *v, uint64_t index, const float *channel_volumes, int has_ratio,
                 float volume_ratio, fvad_vad_result *out);
size_t fvad_vad_segment_count(const fvad_vad *v);
int fvad_vad_segments(const fvad_vad *v, fvad_speech_segment *out, size_t cap, size_t *n);
int fvad_vad_audit_get(const fvad_vad *v, fvad_vad_audit *out);
/* How often the long-term average's full chain (RollingAverage.zig:45-56) had to be run: this build
 * evaluates it lazily, only when a bound on the incrementally carried value cannot settle the threshold
 * comparison (results are identical either way; see host_vad.cpp). */
int fvad_vad_lazy_stats(const fvad_vad *v, uint64_t *exact_evaluations, uint64_t *lazy_pushes);
/* Many independent streams at once, bit-identical to fvad_vad_run per stream: the streams are dealt to
 * n_threads host threads (stream s -> thread s % n_threads; one thread per file is the reference's own
 * parallelism, simulator.zig:221-232) and each thread runs its streams one after the other.
 * band[s] points at [n_frames[s]][n_channels] f32, ratio[s] at [n_frames[s]].
 * first_index[s] + fft_size*k is frame k's index. */
int fvad_vad_run_many(fvad_vad *const *vads, size_t n_streams, const float *const *band,
                      const float *const *ratio, const size_t *n_frames, size_t n_channels,
                      const uint64_t *first_index, size_t fft_size, int n_threads);

/* The host stage of a whole batch in one call, straight from the engine's lane-major outputs (lane = stream *
 * n_channels + channel): per-chunk volume ratio (BufferedVolumeAnalyzer.zig:48-69), the metadata hand-overs
 * (BufferedVolumeAnalyzer.zig:33-45, BufferedDenoiser.zig:83-86,115), the sample-weighted ratio of every FFT
 * frame (BufferedFFT.zig:137-140,153), then VADMachine.run per frame on fresh machines (VADMachine.zig:138-239),
 * streams dealt to n_threads host threads.  band: lane l's n_frames sums at band + l * band_stride; chunk_rms:
 * lane l's n_chunks values at chunk_rms + l * rms_stride; chunk_size = 24000 at 48 kHz (NSNet2.zig:157-159).
 * Bit-identical to fvad_pipeline_* / fvad_vad_run on the same numbers. */
typedef struct fvad_vad_batch fvad_vad_batch;
int fvad_vad_batch_create(const fvad_vad_config *cfg, size_t sample_rate, size_t n_channels,
                          size_t fft_size, size_t n_streams, fvad_vad_batch **out);
void fvad_vad_batch_destroy(fvad_vad_batch *b);
int fvad_vad_batch_run(fvad_vad_batch *b, const float *band, size_t band_stride, size_t n_frames,
                       const float *chunk_rms, size_t rms_stride, size_t n_chunks, size_t chunk_size,
                       int n_threads);
/* The same in parts: frames [first_frame, first_frame + n_frames) of every stream, the streams' machines living on between
 * the calls, so that a host can run the VAD of the part it has while the GPU produces the next one.  first_frame = 0 starts
 * from fresh machines (fvad_vad_batch_run is this with first_frame = 0); a later part must start where the previous one
 * ended, on a chunk boundary (first_frame * fft_size a multiple of chunk_size: at 48 kHz and fft_size 1024 every 375 frames
 * = 16 chunks), and band / chunk_rms point at the part's first frame / first chunk.  Segments and audits (below) cover
 * everything run so far.  Bit-identical to one fvad_vad_batch_run over all the frames. */
int fvad_vad_batch_run_part(fvad_vad_batch *b, const float *band, size_t band_stride, size_t n_frames,
                            const float *chunk_rms, size_t rms_stride, size_t n_chunks, size_t chunk_size,
                            uint64_t first_frame, int n_threads);
size_t fvad_vad_batch_total_segments(const fvad_vad_batch *b);
/* all segments, stream after stream; offsets[s] .. offsets[s + 1] are stream s's (offsets has n_streams + 1 entries) */
int fvad_vad_batch_segments(const fvad_vad_batch *b, fvad_speech_segment *out, size_t cap,
                            size_t *offsets);
int fvad_vad_batch_audit(const fvad_vad_batch *b, size_t stream, fvad_vad_audit *out);

/* RollingAverage.zig:11-56 exposed for parity tests */
typedef struct fvad_rolling_average fvad_rolling_average;
int fvad_ra_create(size_t count, int has_initial, double initial_val, fvad_rolling_average **out);
void fvad_ra_destroy(fvad_rolling_average *ra);
double fvad_ra_push(fvad_rolling_average *ra, float sample);
int fvad_ra_last_avg(const fvad_rolling_average *ra, double *out);

/* ------------------------------------------------------------------ B1: AudioPipeline
 * (src/AudioPipeline.zig) -- what simulator.zig / main.zig hold. */
/* Recording payload == AudioBuffer (src/audio_utils/AudioBuffer.zig:16-24) as Recorder.finalize
 * builds it (Recorder.zig:131-164): ONE channel -- the quietest of the stream's channels over the
 * clip (findBestChannel, :113-129) -- covering samples [global_start_frame_number, + length).
 * Unlike the reference (callee frees, MRBRecorder.zig:9-11) the buffer belongs to the library and
 * is valid only during the callback. */
typedef struct fvad_audio_buffer {
    const float *const *channel_pcm;
    size_t n_channels, length, sample_rate;
    float duration_seconds;
    uint64_t global_start_frame_number;
} fvad_audio_buffer;
/* AudioPipeline.Callbacks (AudioPipeline.zig:14-18).  When callbacks are given, every segment the
 * state machine completes (VADPipeline.zig:215-229) schedules one original-audio clip and one
 * denoised-audio clip (AudioPipeline.zig:187-191); each is delivered as soon as its buffer holds the
 * samples up to the clip's end -- at once, or during a later push -- and a recording that restarts
 * before then replaces it, exactly like MRBRecorder.zig:76-118,160-192 on the reference's write schedule
 * (pushes written in steps of buffer_length / 2, denoised audio in 0.5 s chunks).  The denoised audio is
 * then also copied back from the GPU (1920 B per frame). */
typedef void (*fvad_recording_cb)(void *ctx, const fvad_audio_buffer *recording);
typedef struct {                              /* AudioPipeline.Callbacks, AudioPipeline.zig:14-18 */
    void *ctx;
    fvad_recording_cb on_original_recording;
    fvad_recording_cb on_denoised_recording;
} fvad_callbacks;

typedef struct {
    size_t sample_rate;                       /* AudioPipeline.Config, AudioPipeline.zig:20-26 */
    size_t n_channels;
    size_t buffer_length;                     /* 0 = sample_rate * 10 (:46); otherwise >= one 24000-sample chunk */
    int32_t skip_processing;
    size_t fft_size;                          /* VADPipeline.Config.fft_size = 1024 (:21); any even size from 4 to 16384 */
    fvad_vad_config vad_machine_config;       /* :22 */
    const fvad_vad_config *alt_vad_machine_configs; /* :24 */
    size_t n_alt_vad_machine_configs;
} fvad_pipeline_config;
void fvad_pipeline_config_default(fvad_pipeline_config *c);

typedef struct fvad_pipeline fvad_pipeline;
/* AudioPipeline.init(allocator, config, callbacks)  AudioPipeline.zig:40-102 */
int fvad_pipeline_create(fvad_ctx *ctx, const fvad_pipeline_config *cfg,
                         const fvad_callbacks *callbacks, fvad_pipeline **out);
void fvad_pipeline_destroy(fvad_pipeline *p);                          /* deinit :104-112 */
/* pushSamples(channel_pcm) -> index of the first pushed sample  AudioPipeline.zig:118-143 */
int fvad_pipeline_push_samples(fvad_pipeline *p, const float *const *channel_pcm,
                               size_t n_samples, uint64_t *first_sample_index);
uint64_t fvad_pipeline_total_write_count(const fvad_pipeline *p);      /* :114-116 */
/* pipeline.vad.vad_machine.vad_segments  (SimulationInstance.zig:221) */
size_t fvad_pipeline_segment_count(const fvad_pipeline *p);
int fvad_pipeline_segments(const fvad_pipeline *p, fvad_speech_segment *out, size_t cap,
                           size_t *n);
int fvad_pipeline_alt_segments(const fvad_pipeline *p, size_t alt_index,
                               fvad_speech_segment *out, size_t cap, size_t *n);
int fvad_pipeline_audit(const fvad_pipeline *p, fvad_vad_audit *out);
/* traces for parity tests: per-FFT-frame band sums [n][n_channels] and volume ratios [n], kept for the frames
 * processed while tracing is enabled (off by default: a live pipeline does not grow with the stream) */
int fvad_pipeline_enable_trace(fvad_pipeline *p, int on);
size_t fvad_pipeline_n_fft_frames(const fvad_pipeline *p);
int fvad_pipeline_trace(const fvad_pipeline *p, float *band_volumes, float *vol_ratio,
                        size_t cap_frames);

/* ------------------------------------------------------------------ Evaluator (host)
 * src/Evaluator.zig:90-156 + src/Evaluator/statistics.zig */
typedef struct {
    float total_positives_sec, true_positives_sec, false_positives_sec, false_negatives_sec;
    float true_positive_rate, false_negative_rate, false_discovery_rate, precision;
    float fm_index, f_score, f_score_beta;
} fvad_single_stats;                          /* statistics.SingleStats :8-37 */
typedef struct { float overall, min, max, avg; } fvad_agg_stat;          /* :39-44 */
typedef struct {
    float total_positives_sec, true_positives_sec, false_positives_sec, false_negatives_sec;
    fvad_agg_stat true_positive_rate, false_negative_rate, false_discovery_rate, precision;
    float fm_index, f_score, f_score_beta;
} fvad_aggregate_stats;                       /* statistics.AggregateStats :46-75 */
typedef struct {
    float ignore_shorter_than_sec, extrude_start, extrude_end, fill_gaps;
} fvad_stat_config;                           /* statistics.StatConfig :77-83 */
typedef struct { float from_sec, to_sec; } fvad_segment_sec;

/* SimulationInstance.storeResult's sample->second conversion (SimulationInstance.zig:237-238) */
fvad_segment_sec fvad_segment_to_sec(const fvad_speech_segment *s, size_t sample_rate);
/* Evaluator.initAndRun + statistics.fromEvaluator */
int fvad_stats_from_segments(const fvad_segment_sec *vad, size_t n_vad,
                             const fvad_segment_sec *ref, size_t n_ref,
                             const fvad_stat_config *cfg, fvad_single_stats *out);
/* statistics.aggregate(stats)  statistics.zig:116-172 -- in slice order */
int fvad_stats_aggregate(const fvad_single_stats *stats, size_t n, fvad_aggregate_stats *out);
/* ---- multi-GPU: the plan's streams are dealt round-robin to one rank (process or thread) per GPU, nothing but
 * these per-stream statistics ever crosses GPUs.  Replaces the join of the reference's one-thread-per-file
 * instances (src/simulator.zig:221-232) in front of report_generator.zig:48-68: every rank hands in the
 * SingleStats of its streams, every rank receives all n_streams of them in PLAN order (stream id = index in
 * the plan), ready for fvad_stats_aggregate -- bit-identical to a single-process run.  Transport: one
 * ncclAllGather over RCCL / xGMI (RCCL is dlopen'ed on first use), behind a header all-gather of {n_streams, local
 * status}: n_streams must be the same on every rank, and a rank whose arguments are bad still takes part, so every
 * rank returns an error instead of some of them waiting in the collective.  Bootstrap like NCCL's own: rank 0 makes
 * the 128-byte id and hands it to the other ranks by whatever channel the host has (file, socket, env). */
#define FVAD_COMM_ID_BYTES 128
typedef struct fvad_comm fvad_comm;
int fvad_comm_unique_id(uint8_t *id, size_t n_bytes);                   /* ncclGetUniqueId */
int fvad_comm_create(fvad_ctx *ctx, const uint8_t *id, size_t n_bytes, int world, int rank,
                     fvad_comm **out);                                   /* ncclCommInitRank on ctx's device */
void fvad_comm_destroy(fvad_comm *c);
int fvad_comm_world(const fvad_comm *c);
int fvad_comm_rank(const fvad_comm *c);
int fvad_stats_allgather(fvad_comm *c, const uint32_t *local_ids, const fvad_single_stats *local_stats,
                         size_t n_local, size_t n_streams, fvad_single_stats *out /* [n_streams] */);
/* formats.parseAudacitySegments / serialize  (Evaluator/formats.zig:7-56) */
int fvad_parse_audacity(const char *txt, size_t len, fvad_segment_sec *out, size_t cap,
                        size_t *n);

/* ------------------------------------------------------------------ audio file input (host)
 * Minimal RIFF/WAVE reader standing in for AudioBuffer.loadFromFile / AudioFileStream
 * (src/audio_utils/AudioBuffer.zig:26-59, AudioFileStream.zig:18-102): PCM16 or float32, any
 * channel count, returned channel-planar f32.  Free with fvad_wav_free. */
int fvad_wav_read(const char *path, float ***channel_pcm, size_t *n_channels, size_t *n_frames,
                  size_t *sample_rate);
void fvad_wav_free(float **channel_pcm, size_t n_channels);
/* The same for a PCM16 file without the conversion: channel-planar int16 for fvad_lane.pcm_i16
 * (FVAD_ERR_MODEL_FORMAT if the file is not PCM16).  Free with fvad_wav_free_i16. */
int fvad_wav_read_i16(const char *path, int16_t ***channel_pcm, size_t *n_channels, size_t *n_frames,
                      size_t *sample_rate);
void fvad_wav_free_i16(int16_t **channel_pcm, size_t n_channels);
/* AudioBuffer.saveToFile (src/audio_utils/AudioBuffer.zig:61-118) for the WAV container -- what a recording
 * callback does with its clip (main.zig saves them): planar channels -> float32 (as_pcm16 == 0) or PCM16 WAV. */
int fvad_wav_write(const char *path, const float *const *channel_pcm, size_t n_channels,
                   size_t n_frames, size_t sample_rate, int as_pcm16);

#ifdef __cplusplus
}
#endif
#endif /* FVAD_H */
