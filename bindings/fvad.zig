//! Zig binding of libfvad_hip.so (include/fvad.h, ABI 3) for recursiveGecko/Formula-VAD.
//!
//! NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no `zig` (SURVEY.md section 0.3).  What IS checked
//! mechanically is the declaration layer: tests/test_binding_zig.py parses every `extern "c" fn`, `extern struct` and
//! status constant below and compares names, arity, pointer depth, constness, scalar types, field order and struct
//! sizes / offsets with include/fvad.h (through gcc) and with the ctypes layer of formula-vad_amd/binding.py.  A function
//! of fvad.h that is not declared here fails that test unless it is on the test's short allow-list.
//!
//! It is the file a maintainer drops into the reference's `src/` and wires up in `build.zig`
//! (`exe.linkSystemLibrary("fvad_hip"); exe.addLibraryPath(...)`), written against Zig 0.11-dev like the reference
//! (README.md:74).
//!
//! Three ways to use it:
//!  (1) `GpuPipeline` replaces `AudioPipeline` wholesale in SimulationInstance.simulateVAD
//!      (src/simulator/SimulationInstance.zig:164-225) and main.zig: init(allocator, config, callbacks) /
//!      pushSamples / totalWriteCount / `pipeline.vad.vad_machine.vad_segments` / deinit, so simulator.zig itself is
//!      unchanged.  With `preload_audio = true` the whole file arrives in one pushSamples call and is one GPU batch.
//!  (2) `runBatch` hands whole streams (every file of a plan) to fvad_engine_run in ONE call -- the form the GPU wants
//!      (INTEGRATION.md) -- and `VadBatchHost` runs the host stage (volume ratio, VADMachine) over its outputs.
//!  (3) `GpuFFT` / `GpuNSNet2` replace src/FFT.zig and src/NSNet2.zig one for one, kissfft and onnxruntime
//!      underneath them (slow: one tiny launch per call; for testing).
const std = @import("std");

pub const abi_version = 3; // FVAD_ABI_VERSION
pub const comm_id_bytes = 128; // FVAD_COMM_ID_BYTES

// ------------------------------------------------------------------ status codes (the anonymous enum of fvad.h)
pub const Status = struct {
    pub const ok = 0;
    pub const err_invalid_fft_size = -1;
    pub const err_invalid_samples_length = -2;
    pub const err_invalid_window_length = -3;
    pub const err_invalid_result_length = -4;
    pub const err_invalid_bins_length = -5;
    pub const err_out_of_range = -6;
    pub const err_negative_frequency = -7;
    pub const err_invalid_input_length = -8;
    pub const err_invalid_sample_rate = -9;
    pub const err_channel_count_mismatch = -10;
    pub const err_alloc_failed = -11;
    pub const err_invalid_argument = -100;
    pub const err_no_device = -101;
    pub const err_hip = -102;
    pub const err_no_model = -103;
    pub const err_model_format = -104;
    pub const err_io = -105;
    pub const err_buffer_too_small = -106;
    pub const nn_math_f32 = 0;
    pub const nn_math_f16x3 = 1;
    pub const nn_math_bf16x3 = 2;
    pub const rec_none = 0;
    pub const rec_started = 1;
    pub const rec_completed = 2;
    pub const rec_aborted = 3;
};

// ------------------------------------------------------------------ opaque handles
pub const Ctx = opaque {}; // fvad_ctx
pub const Pipeline = opaque {}; // fvad_pipeline
pub const Fft = opaque {}; // fvad_fft
pub const NSNet2 = opaque {}; // fvad_nsnet2
pub const LaneState = opaque {}; // fvad_lane_state
pub const Vad = opaque {}; // fvad_vad
pub const VadBatch = opaque {}; // fvad_vad_batch
pub const RollingAverage = opaque {}; // fvad_rolling_average
pub const Comm = opaque {}; // fvad_comm

// ------------------------------------------------------------------ structs (field for field the typedefs of fvad.h)
pub const NSNet2Weights = extern struct { // fvad_nsnet2_weights
    n_bins: i32,
    n_fc1: i32,
    n_hidden: i32,
    n_fc2: i32,
    n_fc3: i32,
    fc1_w: ?[*]const f32,
    fc1_b: ?[*]const f32,
    gru1_w: ?[*]const f32,
    gru1_r: ?[*]const f32,
    gru1_b: ?[*]const f32,
    gru2_w: ?[*]const f32,
    gru2_r: ?[*]const f32,
    gru2_b: ?[*]const f32,
    fc2_w: ?[*]const f32,
    fc2_b: ?[*]const f32,
    fc3_w: ?[*]const f32,
    fc3_b: ?[*]const f32,
    fc4_w: ?[*]const f32,
    fc4_b: ?[*]const f32,
};

pub const Complex = extern struct { r: f32, i: f32 }; // fvad_complex == FFT.Complex (src/FFT.zig:12-14)

pub const Lane = extern struct { // fvad_lane
    pcm: ?[*]const f32 = null,
    n_samples: usize = 0,
    state: ?*LaneState = null,
    denoised: ?[*]f32 = null,
    band_sum: ?[*]f32 = null,
    band_sum_capacity: usize = 0,
    chunk_rms: ?[*]f32 = null,
    chunk_rms_capacity: usize = 0,
    fft_bins: ?[*]f32 = null,
    pcm_i16: ?[*]const i16 = null,
    denoised_i16: ?[*]i16 = null,
    spectrogram: ?[*]f32 = null,
    features: ?[*]f32 = null,
    n_chunks: usize = 0,
    n_fft_frames: usize = 0,
    first_frame_index: u64 = 0,
};

pub const EngineOpts = extern struct { // fvad_engine_opts (fill with fvad_engine_opts_default)
    on_device: i32 = 0,
    min_bin: i32 = 11,
    max_bin: i32 = 43,
    max_chunks_per_launch: i32 = 0,
    fft_size: i32 = 0,
    no_wait: i32 = 0,
    use_graph: i32 = 0,
};

pub const VadConfig = extern struct { // fvad_vad_config == VADMachine.Config (src/AudioPipeline/VADMachine.zig:30-51)
    speech_min_freq: f32 = 500,
    speech_max_freq: f32 = 2000,
    long_term_speech_avg_sec: f32 = 180,
    has_initial_long_term_avg: i32 = 1,
    initial_long_term_avg: f64 = 0.005,
    short_term_speech_avg_sec: f32 = 0.2,
    speech_threshold_factor: f32 = 10,
    channel_vol_ratio_avg_sec: f32 = 0.5,
    channel_vol_ratio_threshold: f32 = 0.5,
    min_consecutive_sec_to_open: f32 = 0.2,
    max_speech_gap_sec: f32 = 2,
    min_vad_duration_sec: f32 = 0.7,
};

pub const SpeechSegment = extern struct { // fvad_speech_segment == VADPipeline.SpeechSegment (VADPipeline.zig:28-33)
    sample_from: u64,
    sample_to: u64,
    avg_channel_vol_ratio: f32,
    vad_met_sec: f32,
};

pub const VadResult = extern struct { // fvad_vad_result == VADMachine.Result (VADMachine.zig:18-28)
    recording_state: i32,
    sample_number: u64,
};

pub const VadAudit = extern struct { // fvad_vad_audit
    min_rel_threshold_margin: f64,
    min_abs_ratio_margin: f64,
    n_frames: u64,
};

pub const AudioBuffer = extern struct { // fvad_audio_buffer == AudioBuffer (src/audio_utils/AudioBuffer.zig:16-24)
    channel_pcm: [*]const [*]const f32,
    n_channels: usize,
    length: usize,
    sample_rate: usize,
    duration_seconds: f32,
    global_start_frame_number: u64,
};

pub const RecordingCb = ?*const fn (ctx: ?*anyopaque, recording: *const AudioBuffer) callconv(.C) void; // fvad_recording_cb

pub const Callbacks = extern struct { // fvad_callbacks == AudioPipeline.Callbacks (src/AudioPipeline.zig:14-18)
    ctx: ?*anyopaque = null,
    on_original_recording: RecordingCb = null,
    on_denoised_recording: RecordingCb = null,
};

pub const PipelineConfig = extern struct { // fvad_pipeline_config == AudioPipeline.Config + VADPipeline.Config
    sample_rate: usize,
    n_channels: usize,
    buffer_length: usize = 0,
    skip_processing: i32 = 0,
    fft_size: usize = 1024,
    vad_machine_config: VadConfig = .{},
    alt_vad_machine_configs: ?[*]const VadConfig = null,
    n_alt_vad_machine_configs: usize = 0,
};

pub const SingleStats = extern struct { // fvad_single_stats == statistics.SingleStats (statistics.zig:8-37)
    total_positives_sec: f32,
    true_positives_sec: f32,
    false_positives_sec: f32,
    false_negatives_sec: f32,
    true_positive_rate: f32,
    false_negative_rate: f32,
    false_discovery_rate: f32,
    precision: f32,
    fm_index: f32,
    f_score: f32,
    f_score_beta: f32,
};

pub const AggStat = extern struct { overall: f32, min: f32, max: f32, avg: f32 }; // fvad_agg_stat (statistics.zig:39-44)

pub const AggregateStats = extern struct { // fvad_aggregate_stats == statistics.AggregateStats (statistics.zig:46-75)
    total_positives_sec: f32,
    true_positives_sec: f32,
    false_positives_sec: f32,
    false_negatives_sec: f32,
    true_positive_rate: AggStat,
    false_negative_rate: AggStat,
    false_discovery_rate: AggStat,
    precision: AggStat,
    fm_index: f32,
    f_score: f32,
    f_score_beta: f32,
};

pub const StatConfig = extern struct { // fvad_stat_config == statistics.StatConfig (statistics.zig:77-83)
    ignore_shorter_than_sec: f32,
    extrude_start: f32,
    extrude_end: f32,
    fill_gaps: f32,
};

pub const SegmentSec = extern struct { from_sec: f32, to_sec: f32 }; // fvad_segment_sec

// ------------------------------------------------------------------ functions, in the order of fvad.h
pub extern "c" fn fvad_status_name(status: c_int) [*:0]const u8;
pub extern "c" fn fvad_abi_version() c_int;

// context
pub extern "c" fn fvad_ctx_create(device: c_int, out: *?*Ctx) c_int;
pub extern "c" fn fvad_ctx_destroy(ctx: ?*Ctx) void;
pub extern "c" fn fvad_last_error(ctx: ?*const Ctx) [*:0]const u8;
pub extern "c" fn fvad_ctx_synchronize(ctx: *Ctx) c_int;
pub extern "c" fn fvad_ctx_stream(ctx: *Ctx) ?*anyopaque;
pub extern "c" fn fvad_ctx_copy_to_host(ctx: *Ctx, dst_host: *anyopaque, src_device: *const anyopaque, bytes: usize) c_int;
pub extern "c" fn fvad_host_alloc(ctx: *Ctx, bytes: usize, out: *?*anyopaque) c_int;
pub extern "c" fn fvad_host_free(ctx: *Ctx, p: ?*anyopaque) void;
pub extern "c" fn fvad_device_alloc(ctx: *Ctx, bytes: usize, out: *?*anyopaque) c_int;
pub extern "c" fn fvad_device_free(ctx: *Ctx, p: ?*anyopaque) void;
pub extern "c" fn fvad_ctx_copy_to_device(ctx: *Ctx, dst_device: *anyopaque, src_host: *const anyopaque, bytes: usize) c_int;

// NSNet2 model
pub extern "c" fn fvad_load_nsnet2_onnx(ctx: *Ctx, onnx_path: [*:0]const u8) c_int;
pub extern "c" fn fvad_load_nsnet2_weights(ctx: *Ctx, w: *const NSNet2Weights) c_int;
pub extern "c" fn fvad_load_nsnet2_synth(ctx: *Ctx, seed: u64) c_int;
pub extern "c" fn fvad_get_nsnet2_weights(ctx: *const Ctx, out: *NSNet2Weights) c_int;
pub extern "c" fn fvad_onnx_read_nsnet2(onnx_path: [*:0]const u8, out: *NSNet2Weights, owner: *?*anyopaque) c_int;
pub extern "c" fn fvad_synth_nsnet2(seed: u64, out: *NSNet2Weights, owner: *?*anyopaque) c_int;
pub extern "c" fn fvad_weights_free(owner: ?*anyopaque) void;

// B3: FFT (src/FFT.zig)
pub extern "c" fn fvad_fft_create(ctx: *Ctx, n_fft: usize, sample_rate: usize, mode_inverse: c_int, out: *?*Fft) c_int;
pub extern "c" fn fvad_fft_destroy(fft: ?*Fft) void;
pub extern "c" fn fvad_fft_forward(fft: *Fft, first: ?[*]const f32, n_first: usize, second: ?[*]const f32, n_second: usize, window: [*]const f32, n_window: usize, bins: [*]Complex, n_bins: usize) c_int;
pub extern "c" fn fvad_fft_inverse(fft: *Fft, bins: [*]const Complex, n_bins: usize, result: [*]f32, n_result: usize) c_int;
pub extern "c" fn fvad_fft_bin_count(fft: *const Fft) usize;
pub extern "c" fn fvad_fft_bin_width(fft: *const Fft) f32;
pub extern "c" fn fvad_fft_nyquist_freq(fft: *const Fft) f32;
pub extern "c" fn fvad_fft_freq_to_bin(fft: *const Fft, freq: f32, bin: *usize) c_int;
pub extern "c" fn fvad_fft_bin_to_freq(fft: *const Fft, bin: usize, freq: *f32) c_int;
pub extern "c" fn fvad_fft_forward_batch(fft: *Fft, frames: [*]const f32, n_frames: usize, window: [*]const f32, bins: ?[*]Complex, magnitudes: ?[*]f32, on_device: c_int) c_int;
pub extern "c" fn fvad_hann_window_periodic(result: [*]f32, n: usize) void;
pub extern "c" fn fvad_hann_window_symmetric(result: [*]f32, n: usize) void;
pub extern "c" fn fvad_window_norm_factor(window: [*]const f32, n: usize) f32;
pub extern "c" fn fvad_nsnet2_window(window320: [*]f32) void;

// B2: NSNet2 (src/NSNet2.zig)
pub extern "c" fn fvad_nsnet2_create(ctx: *Ctx, sample_rate: usize, out: *?*NSNet2) c_int;
pub extern "c" fn fvad_nsnet2_destroy(d: ?*NSNet2) void;
pub extern "c" fn fvad_nsnet2_chunk_size(in_sample_rate: usize) usize;
pub extern "c" fn fvad_nsnet2_denoise(d: *NSNet2, first: ?[*]const f32, n_first: usize, second: ?[*]const f32, n_second: usize, denoised_result: [*]f32, n_result: usize) c_int;

// batched engine
pub extern "c" fn fvad_lane_state_create(ctx: *Ctx, out: *?*LaneState) c_int;
pub extern "c" fn fvad_lane_state_reset(s: ?*LaneState) void;
pub extern "c" fn fvad_lane_state_destroy(s: ?*LaneState) void;
pub extern "c" fn fvad_lane_state_seek(s: *LaneState, sample_index: u64, fft_size: usize) c_int;
pub extern "c" fn fvad_engine_opts_default(o: *EngineOpts) void;
pub extern "c" fn fvad_engine_run(ctx: *Ctx, lanes: [*]Lane, n_lanes: usize, opts: ?*const EngineOpts) c_int;
pub extern "c" fn fvad_engine_enqueue_device(ctx: *Ctx, d_pcm: [*]const f32, n_lanes: usize, lane_stride: usize, n_samples: usize, d_denoised: ?[*]f32, d_band_sum: [*]f32, d_chunk_rms: ?[*]f32, opts: ?*const EngineOpts) c_int;
pub extern "c" fn fvad_engine_enqueue_device_i16(ctx: *Ctx, d_pcm16: [*]const i16, n_lanes: usize, lane_stride: usize, n_samples: usize, d_denoised16: ?[*]i16, d_band_sum: [*]f32, d_chunk_rms: ?[*]f32, opts: ?*const EngineOpts) c_int;
pub extern "c" fn fvad_nsnet2_forward(ctx: *Ctx, features: [*]const f32, n_seq: usize, T: usize, gains: [*]f32) c_int;
/// arithmetic of the NSNet2 matrix products at every batch size; returns the previous mode.  Default 0 = f32 (the ORT
/// CPU arithmetic); 1 = f16x3 emulation (22-bit operands), 2 = bf16x3 (24-bit operands, dense layers) are opt-in
pub extern "c" fn fvad_ctx_set_nn_math(ctx: *Ctx, mode: c_int) c_int;
pub extern "c" fn fvad_ctx_nn_math_effective(ctx: *const Ctx) c_int;
pub extern "c" fn fvad_ctx_last_nn_path(ctx: *const Ctx) [*:0]const u8;
pub extern "c" fn fvad_ctx_set_option(ctx: *Ctx, name: [*:0]const u8, value: ?[*:0]const u8) c_int; // e.g. "reproducible", "1"
pub extern "c" fn fvad_ctx_ws_fallbacks(ctx: *Ctx, n: *u64) c_int;
pub extern "c" fn fvad_ctx_ws2_waits(ctx: *const Ctx, wait_class: c_int) u32; // layer 1 | layer 2 << 16, 10 ns ticks
pub extern "c" fn fvad_ctx_enable_timing(ctx: *Ctx, on: c_int) c_int;
pub extern "c" fn fvad_ctx_kernel_times(ctx: *Ctx, names: [*][*:0]const u8, ms: [*]f32, cap: usize, n: *usize) c_int;

// VAD state machine (host)
pub extern "c" fn fvad_vad_config_default(c: *VadConfig) void;
pub extern "c" fn fvad_vad_create(cfg: *const VadConfig, sample_rate: usize, n_channels: usize, fft_size: usize, out: *?*Vad) c_int;
pub extern "c" fn fvad_vad_destroy(v: ?*Vad) void;
pub extern "c" fn fvad_vad_run(v: *Vad, index: u64, channel_volumes: [*]const f32, has_ratio: c_int, volume_ratio: f32, out: *VadResult) c_int;
pub extern "c" fn fvad_vad_segment_count(v: *const Vad) usize;
pub extern "c" fn fvad_vad_segments(v: *const Vad, out: [*]SpeechSegment, cap: usize, n: *usize) c_int;
pub extern "c" fn fvad_vad_audit_get(v: *const Vad, out: *VadAudit) c_int;
pub extern "c" fn fvad_vad_lazy_stats(v: *const Vad, exact_evaluations: *u64, lazy_pushes: *u64) c_int;
pub extern "c" fn fvad_vad_run_many(vads: [*]const *Vad, n_streams: usize, band: [*]const [*]const f32, ratio: [*]const [*]const f32, n_frames: [*]const usize, n_channels: usize, first_index: [*]const u64, fft_size: usize, n_threads: c_int) c_int;
pub extern "c" fn fvad_vad_batch_create(cfg: *const VadConfig, sample_rate: usize, n_channels: usize, fft_size: usize, n_streams: usize, out: *?*VadBatch) c_int;
pub extern "c" fn fvad_vad_batch_destroy(b: ?*VadBatch) void;
pub extern "c" fn fvad_vad_batch_run(b: *VadBatch, band: [*]const f32, band_stride: usize, n_frames: usize, chunk_rms: [*]const f32, rms_stride: usize, n_chunks: usize, chunk_size: usize, n_threads: c_int) c_int;
pub extern "c" fn fvad_vad_batch_run_part(b: *VadBatch, band: [*]const f32, band_stride: usize, n_frames: usize, chunk_rms: [*]const f32, rms_stride: usize, n_chunks: usize, chunk_size: usize, first_frame: u64, n_threads: c_int) c_int;
pub extern "c" fn fvad_vad_batch_total_segments(b: *const VadBatch) usize;
pub extern "c" fn fvad_vad_batch_segments(b: *const VadBatch, out: [*]SpeechSegment, cap: usize, offsets: [*]usize) c_int;
pub extern "c" fn fvad_vad_batch_audit(b: *const VadBatch, stream: usize, out: *VadAudit) c_int;
pub extern "c" fn fvad_ra_create(count: usize, has_initial: c_int, initial_val: f64, out: *?*RollingAverage) c_int;
pub extern "c" fn fvad_ra_destroy(ra: ?*RollingAverage) void;
pub extern "c" fn fvad_ra_push(ra: *RollingAverage, sample: f32) f64;
pub extern "c" fn fvad_ra_last_avg(ra: *const RollingAverage, out: *f64) c_int;

// B1: AudioPipeline (src/AudioPipeline.zig)
pub extern "c" fn fvad_pipeline_config_default(c: *PipelineConfig) void;
pub extern "c" fn fvad_pipeline_create(ctx: *Ctx, cfg: *const PipelineConfig, callbacks: ?*const Callbacks, out: *?*Pipeline) c_int;
pub extern "c" fn fvad_pipeline_destroy(p: ?*Pipeline) void;
pub extern "c" fn fvad_pipeline_push_samples(p: *Pipeline, channel_pcm: [*]const [*]const f32, n_samples: usize, first_sample_index: *u64) c_int;
pub extern "c" fn fvad_pipeline_total_write_count(p: *const Pipeline) u64;
pub extern "c" fn fvad_pipeline_segment_count(p: *const Pipeline) usize;
pub extern "c" fn fvad_pipeline_segments(p: *const Pipeline, out: [*]SpeechSegment, cap: usize, n: *usize) c_int;
pub extern "c" fn fvad_pipeline_alt_segments(p: *const Pipeline, alt_index: usize, out: [*]SpeechSegment, cap: usize, n: *usize) c_int;
pub extern "c" fn fvad_pipeline_audit(p: *const Pipeline, out: *VadAudit) c_int;
pub extern "c" fn fvad_pipeline_enable_trace(p: *Pipeline, on: c_int) c_int;
pub extern "c" fn fvad_pipeline_n_fft_frames(p: *const Pipeline) usize;
pub extern "c" fn fvad_pipeline_trace(p: *const Pipeline, band_volumes: [*]f32, vol_ratio: [*]f32, cap_frames: usize) c_int;

// Evaluator (host) and the multi-GPU statistics gather
pub extern "c" fn fvad_segment_to_sec(s: *const SpeechSegment, sample_rate: usize) SegmentSec;
pub extern "c" fn fvad_stats_from_segments(vad: [*]const SegmentSec, n_vad: usize, ref: [*]const SegmentSec, n_ref: usize, cfg: *const StatConfig, out: *SingleStats) c_int;
pub extern "c" fn fvad_stats_aggregate(stats: [*]const SingleStats, n: usize, out: *AggregateStats) c_int;
// one thread or process per GPU, stream i of the plan on rank i % world: the per-stream SingleStats of every rank,
// all-gathered over RCCL in plan order, ready for statistics.aggregate (src/Evaluator/statistics.zig:116-172) --
// replaces the join of simulator.zig:221-232's per-file threads
pub extern "c" fn fvad_comm_unique_id(id: [*]u8, n_bytes: usize) c_int; // rank 0; hand the bytes to the other ranks
pub extern "c" fn fvad_comm_create(ctx: *Ctx, id: [*]const u8, n_bytes: usize, world: c_int, rank: c_int, out: *?*Comm) c_int;
pub extern "c" fn fvad_comm_destroy(c: ?*Comm) void;
pub extern "c" fn fvad_comm_world(c: *const Comm) c_int;
pub extern "c" fn fvad_comm_rank(c: *const Comm) c_int;
pub extern "c" fn fvad_stats_allgather(c: *Comm, local_ids: [*]const u32, local_stats: [*]const SingleStats, n_local: usize, n_streams: usize, out: [*]SingleStats) c_int;
pub extern "c" fn fvad_parse_audacity(txt: [*]const u8, len: usize, out: [*]SegmentSec, cap: usize, n: *usize) c_int;

// audio file input / output (host)
pub extern "c" fn fvad_wav_read(path: [*:0]const u8, channel_pcm: *[*][*]f32, n_channels: *usize, n_frames: *usize, sample_rate: *usize) c_int;
pub extern "c" fn fvad_wav_free(channel_pcm: ?[*][*]f32, n_channels: usize) void;
pub extern "c" fn fvad_wav_read_i16(path: [*:0]const u8, channel_pcm: *[*][*]i16, n_channels: *usize, n_frames: *usize, sample_rate: *usize) c_int;
pub extern "c" fn fvad_wav_free_i16(channel_pcm: ?[*][*]i16, n_channels: usize) void;
pub extern "c" fn fvad_wav_write(path: [*:0]const u8, channel_pcm: [*]const [*]const f32, n_channels: usize, n_frames: usize, sample_rate: usize, as_pcm16: c_int) c_int;

// ================================================================== Zig-side wrappers
/// The reference's error names, recovered from the negative status codes of fvad.h.
pub const Error = error{
    InvalidFFTSize,
    InvalidSamplesLength,
    InvalidWindowLength,
    InvalidResultLength,
    InvalidBinsLength,
    OutOfRange,
    NegativeFrequency,
    InvalidInputLength,
    InvalidSampleRate,
    ChannelCountMismatch,
    OutOfMemory,
    InvalidArgument,
    NoDevice,
    NoModel,
    ModelFormat,
    IoFailure,
    BufferTooSmall,
    GpuFailure,
};

pub fn check(status: c_int) Error!void {
    return switch (status) {
        Status.ok => {},
        Status.err_invalid_fft_size => error.InvalidFFTSize,
        Status.err_invalid_samples_length => error.InvalidSamplesLength,
        Status.err_invalid_window_length => error.InvalidWindowLength,
        Status.err_invalid_result_length => error.InvalidResultLength,
        Status.err_invalid_bins_length => error.InvalidBinsLength,
        Status.err_out_of_range => error.OutOfRange,
        Status.err_negative_frequency => error.NegativeFrequency,
        Status.err_invalid_input_length => error.InvalidInputLength,
        Status.err_invalid_sample_rate => error.InvalidSampleRate,
        Status.err_channel_count_mismatch => error.ChannelCountMismatch,
        Status.err_alloc_failed => error.OutOfMemory,
        Status.err_invalid_argument => error.InvalidArgument,
        Status.err_no_device => error.NoDevice,
        Status.err_no_model => error.NoModel,
        Status.err_model_format => error.ModelFormat,
        Status.err_io => error.IoFailure,
        Status.err_buffer_too_small => error.BufferTooSmall,
        else => error.GpuFailure,
    };
}

/// One per thread like the reference's pipelines (simulator.zig:225-231): a HIP device + stream + the loaded model.
pub const Context = struct {
    handle: *Ctx,

    pub fn init(device: c_int, model_path: ?[:0]const u8) !Context {
        var ctx: ?*Ctx = null;
        try check(fvad_ctx_create(device, &ctx));
        errdefer fvad_ctx_destroy(ctx);
        // NSNet2.zig:56: the default model path
        try check(fvad_load_nsnet2_onnx(ctx.?, (model_path orelse "data/nsnet2-20ms-baseline.onnx").ptr));
        return .{ .handle = ctx.? };
    }

    pub fn deinit(self: Context) void {
        fvad_ctx_destroy(self.handle);
    }

    pub fn lastError(self: Context) [*:0]const u8 {
        return fvad_last_error(self.handle);
    }
};

fn toVadConfig(c: anytype) VadConfig { // c: VADMachine.Config (VADMachine.zig:30-51)
    return .{
        .speech_min_freq = c.speech_min_freq,
        .speech_max_freq = c.speech_max_freq,
        .long_term_speech_avg_sec = c.long_term_speech_avg_sec,
        .has_initial_long_term_avg = if (c.initial_long_term_avg != null) 1 else 0,
        .initial_long_term_avg = c.initial_long_term_avg orelse 0,
        .short_term_speech_avg_sec = c.short_term_speech_avg_sec,
        .speech_threshold_factor = c.speech_threshold_factor,
        .channel_vol_ratio_avg_sec = c.channel_vol_ratio_avg_sec,
        .channel_vol_ratio_threshold = c.channel_vol_ratio_threshold,
        .min_consecutive_sec_to_open = c.min_consecutive_sec_to_open,
        .max_speech_gap_sec = c.max_speech_gap_sec,
        .min_vad_duration_sec = c.min_vad_duration_sec,
    };
}

/// Drop-in for `AudioPipeline` (src/AudioPipeline.zig) as SimulationInstance.simulateVAD and main.zig use it.
/// `Config`, `Callbacks` and `RefAudioBuffer` are the REFERENCE's own types (this file lives in the reference's src/), so
/// call sites keep compiling: `AudioPipeline.init(allocator, config, callbacks)` becomes
/// `GpuPipeline.init(allocator, config, callbacks)`; `pipeline.vad.vad_machine.vad_segments` is an ArrayList kept up to
/// date after every pushSamples, exactly what SimulationInstance.zig:221 takes with toOwnedSlice().
pub const GpuPipeline = struct {
    const RefPipeline = @import("./AudioPipeline.zig");
    pub const Config = RefPipeline.Config;
    pub const RefCallbacks = RefPipeline.Callbacks;
    pub const RefAudioBuffer = RefPipeline.AudioBuffer;
    pub const RefSpeechSegment = RefPipeline.VADPipeline.SpeechSegment;

    /// HIP device of pipelines made by init(); one process per GPU sets it once (e.g. from LOCAL_RANK) before any init
    pub var default_device: c_int = 0;

    allocator: std.mem.Allocator,
    config: Config,
    ctx: *Ctx,
    handle: *Pipeline,
    callbacks: ?RefCallbacks,
    temp_ptrs: [][*]const f32,
    /// mirrors `pipeline.vad.vad_machine.vad_segments` (VADMachine.zig:73)
    vad: struct { vad_machine: struct { vad_segments: std.ArrayList(RefSpeechSegment) } },

    /// AudioPipeline.init(allocator, config, callbacks)  AudioPipeline.zig:40-102
    pub fn init(allocator: std.mem.Allocator, config: Config, callbacks: ?RefCallbacks) !*GpuPipeline {
        return initOnDevice(allocator, config, callbacks, default_device);
    }

    pub fn initOnDevice(allocator: std.mem.Allocator, config: Config, callbacks: ?RefCallbacks, device: c_int) !*GpuPipeline {
        var ctx: ?*Ctx = null;
        try check(fvad_ctx_create(device, &ctx));
        errdefer fvad_ctx_destroy(ctx);
        const model: [:0]const u8 = config.vad_config.denoiser_model_path orelse "data/nsnet2-20ms-baseline.onnx"; // NSNet2.zig:56
        try check(fvad_load_nsnet2_onnx(ctx.?, model.ptr));

        var alt: []VadConfig = &.{};
        if (config.vad_config.alt_vad_machine_configs) |alts| {
            alt = try allocator.alloc(VadConfig, alts.len);
            for (alts, 0..) |a, i| alt[i] = toVadConfig(a);
        }
        defer if (alt.len > 0) allocator.free(alt); // the library copies the configs in fvad_pipeline_create

        const c_cfg = PipelineConfig{
            .sample_rate = config.sample_rate,
            .n_channels = config.n_channels,
            .buffer_length = config.buffer_length orelse 0,
            .skip_processing = if (config.skip_processing) 1 else 0,
            .fft_size = config.vad_config.fft_size,
            .vad_machine_config = toVadConfig(config.vad_config.vad_machine_config),
            .alt_vad_machine_configs = if (alt.len > 0) alt.ptr else null,
            .n_alt_vad_machine_configs = alt.len,
        };

        var self = try allocator.create(GpuPipeline);
        errdefer allocator.destroy(self);
        var temp_ptrs = try allocator.alloc([*]const f32, config.n_channels);
        errdefer allocator.free(temp_ptrs);

        // the C callbacks get `self` and forward to the reference-shaped ones
        const c_cb = Callbacks{
            .ctx = self,
            .on_original_recording = &onOriginalRecording,
            .on_denoised_recording = &onDenoisedRecording,
        };
        var p: ?*Pipeline = null;
        try check(fvad_pipeline_create(ctx.?, &c_cfg, if (callbacks != null) &c_cb else null, &p));
        errdefer fvad_pipeline_destroy(p);

        self.* = .{
            .allocator = allocator,
            .config = config,
            .ctx = ctx.?,
            .handle = p.?,
            .callbacks = callbacks,
            .temp_ptrs = temp_ptrs,
            .vad = .{ .vad_machine = .{ .vad_segments = std.ArrayList(RefSpeechSegment).init(allocator) } },
        };
        return self;
    }

    /// AudioPipeline.deinit  AudioPipeline.zig:104-112
    pub fn deinit(self: *GpuPipeline) void {
        fvad_pipeline_destroy(self.handle);
        fvad_ctx_destroy(self.ctx);
        self.vad.vad_machine.vad_segments.deinit();
        self.allocator.free(self.temp_ptrs);
        self.allocator.destroy(self);
    }

    /// AudioPipeline.totalWriteCount  AudioPipeline.zig:114-116
    pub fn totalWriteCount(self: *const GpuPipeline) u64 {
        return fvad_pipeline_total_write_count(self.handle);
    }

    /// AudioPipeline.pushSamples  AudioPipeline.zig:118-143: returns the index of the first pushed sample
    pub fn pushSamples(self: *GpuPipeline, channel_pcm: []const []const f32) !u64 {
        if (channel_pcm.len != self.config.n_channels) return error.ChannelCountMismatch;
        for (channel_pcm, 0..) |ch, i| self.temp_ptrs[i] = ch.ptr;
        var first: u64 = 0;
        try check(fvad_pipeline_push_samples(self.handle, self.temp_ptrs.ptr, channel_pcm[0].len, &first));
        try self.syncSegments();
        return first;
    }

    /// appends the segments the state machine has completed since the last call to `vad.vad_machine.vad_segments`
    fn syncSegments(self: *GpuPipeline) !void {
        var list = &self.vad.vad_machine.vad_segments;
        const n = fvad_pipeline_segment_count(self.handle);
        if (n <= list.items.len) return;
        var tmp = try self.allocator.alloc(SpeechSegment, n);
        defer self.allocator.free(tmp);
        var got: usize = 0;
        try check(fvad_pipeline_segments(self.handle, tmp.ptr, n, &got));
        for (tmp[list.items.len..got]) |s| {
            try list.append(.{
                .sample_from = @intCast(s.sample_from),
                .sample_to = @intCast(s.sample_to),
                .avg_channel_vol_ratio = s.avg_channel_vol_ratio,
                .vad_met_sec = s.vad_met_sec,
            });
        }
    }

    /// the segments of alternative state-machine config `alt_index` (VADPipeline.Config.alt_vad_machine_configs)
    pub fn altSegments(self: *GpuPipeline, allocator: std.mem.Allocator, alt_index: usize) ![]SpeechSegment {
        const cap = fvad_pipeline_segment_count(self.handle) * 4 + 64;
        var out = try allocator.alloc(SpeechSegment, cap);
        errdefer allocator.free(out);
        var got: usize = 0;
        try check(fvad_pipeline_alt_segments(self.handle, alt_index, out.ptr, cap, &got));
        return allocator.realloc(out, got);
    }

    // The library's clip (one channel, valid during the call) presented as the reference's AudioBuffer; like
    // AudioPipeline.onOriginalRecording (AudioPipeline.zig:193-209) the buffer is gone when the callback returns.
    fn forward(self: *GpuPipeline, rec: *const AudioBuffer, original: bool) void {
        const cbs = self.callbacks orelse return;
        const cb = (if (original) cbs.on_original_recording else cbs.on_denoised_recording) orelse return;
        var slices: [8][]f32 = undefined;
        const n = @min(rec.n_channels, slices.len);
        for (0..n) |i| slices[i] = @constCast(rec.channel_pcm[i][0..rec.length]);
        const buf = RefAudioBuffer{
            .allocator = self.allocator,
            .n_channels = n,
            .sample_rate = rec.sample_rate,
            .channel_pcm_buf = slices[0..n],
            .length = rec.length,
            .duration_seconds = rec.duration_seconds,
            .global_start_frame_number = rec.global_start_frame_number,
        };
        cb(cbs.ctx, &buf);
    }

    fn onOriginalRecording(ctx: ?*anyopaque, rec: *const AudioBuffer) callconv(.C) void {
        const self: *GpuPipeline = @ptrCast(@alignCast(ctx.?));
        self.forward(rec, true);
    }

    fn onDenoisedRecording(ctx: ?*anyopaque, rec: *const AudioBuffer) callconv(.C) void {
        const self: *GpuPipeline = @ptrCast(@alignCast(ctx.?));
        self.forward(rec, false);
    }
};

/// Drop-in for src/FFT.zig (kissfft underneath it): same method names, arguments and errors.
pub const GpuFFT = struct {
    const SplitSlice = @import("./structures/SplitSlice.zig").SplitSlice;

    allocator: std.mem.Allocator,
    handle: *Fft,
    n_fft: usize,
    sample_rate: usize,

    /// FFT.init(allocator, n_fft, sample_rate, mode_inverse)  FFT.zig:35-76
    pub fn init(allocator: std.mem.Allocator, ctx: Context, n_fft: usize, sample_rate: usize, mode_inverse: bool) !*GpuFFT {
        var h: ?*Fft = null;
        try check(fvad_fft_create(ctx.handle, n_fft, sample_rate, if (mode_inverse) 1 else 0, &h));
        errdefer fvad_fft_destroy(h);
        var self = try allocator.create(GpuFFT);
        self.* = .{ .allocator = allocator, .handle = h.?, .n_fft = n_fft, .sample_rate = sample_rate };
        return self;
    }

    pub fn deinit(self: *GpuFFT) void { // FFT.zig:78-83
        fvad_fft_destroy(self.handle);
        self.allocator.destroy(self);
    }

    /// FFT.fft(samples, window, bins)  FFT.zig:85-113
    pub fn fft(self: *GpuFFT, samples: SplitSlice(f32), window: []const f32, bins: []Complex) !void {
        try check(fvad_fft_forward(self.handle, samples.first.ptr, samples.first.len, samples.second.ptr, samples.second.len, window.ptr, window.len, bins.ptr, bins.len));
    }

    /// FFT.invFft(bins, result)  FFT.zig:115-134 (unscaled, like kiss_fftri)
    pub fn invFft(self: *GpuFFT, bins: []const Complex, result: []f32) !void {
        try check(fvad_fft_inverse(self.handle, bins.ptr, bins.len, result.ptr, result.len));
    }

    pub fn binCount(self: GpuFFT) usize { // FFT.zig:137-139
        return fvad_fft_bin_count(self.handle);
    }

    pub fn binWidth(self: GpuFFT) f32 { // FFT.zig:142-147
        return fvad_fft_bin_width(self.handle);
    }

    pub fn nyquistFreq(self: GpuFFT) f32 { // FFT.zig:150-153
        return fvad_fft_nyquist_freq(self.handle);
    }

    pub fn freqToBin(self: GpuFFT, freq: f32) !usize { // FFT.zig:156-167: error.OutOfRange / error.NegativeFrequency
        var bin: usize = 0;
        try check(fvad_fft_freq_to_bin(self.handle, freq, &bin));
        return bin;
    }

    pub fn binToFreq(self: GpuFFT, bin_index: usize) !f32 { // FFT.zig:170-180
        var freq: f32 = 0;
        try check(fvad_fft_bin_to_freq(self.handle, bin_index, &freq));
        return freq;
    }
};

/// Drop-in for src/NSNet2.zig (onnxruntime underneath it).
pub const GpuNSNet2 = struct {
    const SplitSlice = @import("./structures/SplitSlice.zig").SplitSlice;

    handle: *NSNet2,

    /// NSNet2.init(allocator, sample_rate, model_path)  NSNet2.zig:35-142; the model comes from the context
    pub fn init(ctx: Context, sample_rate: usize) !GpuNSNet2 {
        var h: ?*NSNet2 = null;
        try check(fvad_nsnet2_create(ctx.handle, sample_rate, &h));
        return .{ .handle = h.? };
    }

    pub fn deinit(self: GpuNSNet2) void { // NSNet2.zig:144-155
        fvad_nsnet2_destroy(self.handle);
    }

    pub fn getChunkSize(in_sample_rate: usize) usize { // NSNet2.zig:157-159
        return fvad_nsnet2_chunk_size(in_sample_rate);
    }

    /// NSNet2.denoise(samples, denoised_result)  NSNet2.zig:161-237
    pub fn denoise(self: GpuNSNet2, samples: SplitSlice(f32), denoised_result: []f32) !void {
        try check(fvad_nsnet2_denoise(self.handle, samples.first.ptr, samples.first.len, samples.second.ptr, samples.second.len, denoised_result.ptr, denoised_result.len));
    }
};

/// Whole streams through the GPU in ONE call (fvad_engine_run), the form INTEGRATION.md recommends for simulator.zig with
/// preload_audio: lane = one channel of one stream, outputs per lane = chunk RMS + band sums (+ denoised audio when the
/// lane's `denoised` is set).  The caller owns every buffer of every lane.
pub fn runBatch(ctx: Context, lanes: []Lane, opts: ?EngineOpts) !void {
    var o: EngineOpts = undefined;
    if (opts) |given| {
        o = given;
    } else {
        fvad_engine_opts_default(&o);
    }
    try check(fvad_engine_run(ctx.handle, lanes.ptr, lanes.len, &o));
}

/// The host stage over runBatch's lane-major outputs: volume ratio, metadata hand-overs, VADMachine.run per frame
/// (VADMachine.zig:138-239), `n_streams` fresh machines, streams dealt to `n_threads` host threads.
pub const VadBatchHost = struct {
    handle: *VadBatch,
    n_streams: usize,

    pub fn init(cfg: VadConfig, sample_rate: usize, n_channels: usize, fft_size: usize, n_streams: usize) !VadBatchHost {
        var h: ?*VadBatch = null;
        try check(fvad_vad_batch_create(&cfg, sample_rate, n_channels, fft_size, n_streams, &h));
        return .{ .handle = h.?, .n_streams = n_streams };
    }

    pub fn deinit(self: VadBatchHost) void {
        fvad_vad_batch_destroy(self.handle);
    }

    pub fn run(self: VadBatchHost, band: []const f32, band_stride: usize, n_frames: usize, chunk_rms: []const f32, rms_stride: usize, n_chunks: usize, n_threads: c_int) !void {
        try check(fvad_vad_batch_run(self.handle, band.ptr, band_stride, n_frames, chunk_rms.ptr, rms_stride, n_chunks, fvad_nsnet2_chunk_size(48000), n_threads));
    }

    /// all segments, stream after stream; offsets[s] .. offsets[s + 1] are stream s's (offsets.len == n_streams + 1)
    /// The same in parts (frames from `first_frame` on; the machines live on between the calls): the host stage of the time
    /// slice that is back while the GPU works on the next one.  A part starts on a chunk boundary (every 375 frames at 48 kHz).
    pub fn runPart(self: VadBatchHost, band: []const f32, band_stride: usize, n_frames: usize, chunk_rms: []const f32, rms_stride: usize, n_chunks: usize, first_frame: u64, n_threads: c_int) !void {
        try check(fvad_vad_batch_run_part(self.handle, band.ptr, band_stride, n_frames, chunk_rms.ptr, rms_stride, n_chunks, fvad_nsnet2_chunk_size(48000), first_frame, n_threads));
    }

    pub fn segments(self: VadBatchHost, allocator: std.mem.Allocator, offsets: []usize) ![]SpeechSegment {
        std.debug.assert(offsets.len == self.n_streams + 1);
        const n = fvad_vad_batch_total_segments(self.handle);
        var out = try allocator.alloc(SpeechSegment, n);
        errdefer allocator.free(out);
        try check(fvad_vad_batch_segments(self.handle, out.ptr, n, offsets.ptr));
        return out;
    }
};

/// SimulationInstance.storeResult's sample -> second conversion + Evaluator.initAndRun + statistics.fromEvaluator
/// (SimulationInstance.zig:227-256, Evaluator.zig:90-156, statistics.zig:85-114) for one stream.
pub fn singleStats(vad: []const SegmentSec, ref: []const SegmentSec, cfg: StatConfig) !SingleStats {
    var out: SingleStats = undefined;
    try check(fvad_stats_from_segments(vad.ptr, vad.len, ref.ptr, ref.len, &cfg, &out));
    return out;
}

/// statistics.aggregate(stats)  statistics.zig:116-172, in slice (= plan) order
pub fn aggregate(stats: []const SingleStats) !AggregateStats {
    var out: AggregateStats = undefined;
    try check(fvad_stats_aggregate(stats.ptr, stats.len, &out));
    return out;
}
