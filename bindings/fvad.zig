//! Zig binding of libfvad_hip.so (include/fvad.h) for recursiveGecko/Formula-VAD.
//!
//! NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no `zig` (SURVEY.md section 0.3).
//! It is the file a maintainer drops into the reference's `src/` and wires up in `build.zig`
//! (`exe.linkSystemLibrary("fvad_hip"); exe.addLibraryPath(...)`), written against Zig
//! 0.11-dev like the reference (README.md:74).  Declarations mirror include/fvad.h 1:1.
//!
//! Two ways to use it:
//!  (1) GpuPipeline below replaces `AudioPipeline` wholesale in SimulationInstance.simulateVAD
//!      (src/simulator/SimulationInstance.zig:164-225): same init / pushSamples / vad_segments
//!      surface, so simulator.zig itself is unchanged.  With `preload_audio = true` the whole file
//!      arrives in one pushSamples call and is processed as one GPU batch.
//!  (2) `fvad_fft_*` / `fvad_nsnet2_*` replace the kissfft and onnxruntime calls underneath
//!      src/FFT.zig and src/NSNet2.zig one for one (slow: one tiny launch per call; for testing).
const std = @import("std");

pub const Ctx = opaque {};
pub const Pipeline = opaque {};
pub const Fft = opaque {};
pub const NSNet2 = opaque {};

pub const Complex = extern struct { r: f32, i: f32 }; // == FFT.Complex (src/FFT.zig:12-14)

pub const VadConfig = extern struct { // == VADMachine.Config (src/AudioPipeline/VADMachine.zig:30-51)
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

pub const SpeechSegment = extern struct { // == VADPipeline.SpeechSegment (VADPipeline.zig:28-33)
    sample_from: u64,
    sample_to: u64,
    avg_channel_vol_ratio: f32,
    vad_met_sec: f32,
};

pub const PipelineConfig = extern struct { // AudioPipeline.Config + VADPipeline.Config
    sample_rate: usize,
    n_channels: usize,
    buffer_length: usize = 0,
    skip_processing: i32 = 0,
    fft_size: usize = 1024,
    vad_machine_config: VadConfig = .{},
    alt_vad_machine_configs: ?[*]const VadConfig = null,
    n_alt_vad_machine_configs: usize = 0,
};

pub extern "c" fn fvad_status_name(status: c_int) [*:0]const u8;
pub extern "c" fn fvad_ctx_create(device: c_int, out: *?*Ctx) c_int;
pub extern "c" fn fvad_ctx_destroy(ctx: ?*Ctx) void;
pub extern "c" fn fvad_last_error(ctx: ?*const Ctx) [*:0]const u8;
pub extern "c" fn fvad_load_nsnet2_onnx(ctx: *Ctx, path: [*:0]const u8) c_int;
pub extern "c" fn fvad_load_nsnet2_synth(ctx: *Ctx, seed: u64) c_int;
/// arithmetic of the NSNet2 matrix products at every batch size; returns the previous mode.  The default is 0 (f32) since ABI 3
/// (ABI 2 defaulted to f16x3: callers that relied on that must now opt in)
pub extern "c" fn fvad_ctx_set_nn_math(ctx: *Ctx, mode: c_int) c_int; // 0 = f32 (default, the ORT CPU arithmetic), 1 = f16x3 emulation (22-bit operands), 2 = bf16x3 (24-bit operands, dense layers)
pub extern "c" fn fvad_ctx_nn_math_effective(ctx: *const Ctx) c_int;
pub extern "c" fn fvad_ctx_last_nn_path(ctx: *const Ctx) [*:0]const u8;
pub extern "c" fn fvad_ctx_set_option(ctx: *Ctx, name: [*:0]const u8, value: ?[*:0]const u8) c_int; // e.g. "reproducible", "1"
pub extern "c" fn fvad_ctx_ws_fallbacks(ctx: *Ctx, n: *u64) c_int;
pub extern "c" fn fvad_ctx_ws2_waits(ctx: *const Ctx, wait_class: c_int) u32; // layer 1 | layer 2 << 16, 10 ns ticks

pub extern "c" fn fvad_pipeline_create(ctx: *Ctx, cfg: *const PipelineConfig, callbacks: ?*const anyopaque, out: *?*Pipeline) c_int;
pub extern "c" fn fvad_pipeline_destroy(p: ?*Pipeline) void;
pub extern "c" fn fvad_pipeline_push_samples(p: *Pipeline, channel_pcm: [*]const [*]const f32, n_samples: usize, first_sample_index: *u64) c_int;
pub extern "c" fn fvad_pipeline_total_write_count(p: *const Pipeline) u64;
pub extern "c" fn fvad_pipeline_segment_count(p: *const Pipeline) usize;
pub extern "c" fn fvad_pipeline_segments(p: *const Pipeline, out: [*]SpeechSegment, cap: usize, n: *usize) c_int;

pub extern "c" fn fvad_fft_create(ctx: *Ctx, n_fft: usize, sample_rate: usize, mode_inverse: c_int, out: *?*Fft) c_int;
pub extern "c" fn fvad_fft_destroy(f: ?*Fft) void;
pub extern "c" fn fvad_fft_forward(f: *Fft, first: [*]const f32, n_first: usize, second: ?[*]const f32, n_second: usize, window: [*]const f32, n_window: usize, bins: [*]Complex, n_bins: usize) c_int;
pub extern "c" fn fvad_fft_inverse(f: *Fft, bins: [*]const Complex, n_bins: usize, result: [*]f32, n_result: usize) c_int;

pub extern "c" fn fvad_nsnet2_create(ctx: *Ctx, sample_rate: usize, out: *?*NSNet2) c_int;
pub extern "c" fn fvad_nsnet2_destroy(d: ?*NSNet2) void;
pub extern "c" fn fvad_nsnet2_chunk_size(in_sample_rate: usize) usize;
pub extern "c" fn fvad_nsnet2_denoise(d: *NSNet2, first: [*]const f32, n_first: usize, second: ?[*]const f32, n_second: usize, denoised: [*]f32, n_result: usize) c_int;

// ---- multi-GPU leg (one thread or process per GPU; stream i of the plan on rank i % world): the per-stream
// SingleStats of every rank, all-gathered over RCCL in plan order, ready for statistics.aggregate
// (src/Evaluator/statistics.zig:116-172) -- replaces the join of simulator.zig:221-232's per-file threads.
pub const Comm = opaque {};
pub const comm_id_bytes = 128;
pub const SingleStats = extern struct { // == statistics.SingleStats (statistics.zig:8-37)
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
pub extern "c" fn fvad_comm_unique_id(id: [*]u8, n_bytes: usize) c_int; // rank 0; hand the bytes to the other ranks
pub extern "c" fn fvad_comm_create(ctx: *Ctx, id: [*]const u8, n_bytes: usize, world: c_int, rank: c_int, out: *?*Comm) c_int;
pub extern "c" fn fvad_comm_destroy(c: ?*Comm) void;
pub extern "c" fn fvad_stats_allgather(c: *Comm, local_ids: [*]const u32, local_stats: [*]const SingleStats, n_local: usize, n_streams: usize, out: [*]SingleStats) c_int;

// ---- device memory and 16-bit transport for hosts that keep audio resident on the GPU
pub extern "c" fn fvad_device_alloc(ctx: *Ctx, bytes: usize, out: *?*anyopaque) c_int;
pub extern "c" fn fvad_device_free(ctx: *Ctx, p: ?*anyopaque) void;
pub extern "c" fn fvad_ctx_copy_to_device(ctx: *Ctx, dst_device: *anyopaque, src_host: *const anyopaque, bytes: usize) c_int;
pub extern "c" fn fvad_ctx_copy_to_host(ctx: *Ctx, dst_host: *anyopaque, src_device: *const anyopaque, bytes: usize) c_int;
pub extern "c" fn fvad_ctx_synchronize(ctx: *Ctx) c_int;
pub extern "c" fn fvad_engine_enqueue_device(ctx: *Ctx, d_pcm: [*]const f32, n_lanes: usize, lane_stride: usize, n_samples: usize, d_denoised: ?[*]f32, d_band_sum: [*]f32, d_chunk_rms: ?[*]f32, opts: ?*const anyopaque) c_int;
pub extern "c" fn fvad_engine_enqueue_device_i16(ctx: *Ctx, d_pcm16: [*]const i16, n_lanes: usize, lane_stride: usize, n_samples: usize, d_denoised16: ?[*]i16, d_band_sum: [*]f32, d_chunk_rms: ?[*]f32, opts: ?*const anyopaque) c_int;

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
    NoDevice,
    GpuFailure,
};

fn check(status: c_int) Error!void {
    return switch (status) {
        0 => {},
        -1 => error.InvalidFFTSize,
        -2 => error.InvalidSamplesLength,
        -3 => error.InvalidWindowLength,
        -4 => error.InvalidResultLength,
        -5 => error.InvalidBinsLength,
        -6 => error.OutOfRange,
        -7 => error.NegativeFrequency,
        -8 => error.InvalidInputLength,
        -9 => error.InvalidSampleRate,
        -10 => error.ChannelCountMismatch,
        -11 => error.OutOfMemory,
        -101 => error.NoDevice,
        else => error.GpuFailure,
    };
}

/// Drop-in for `AudioPipeline` as SimulationInstance.simulateVAD uses it.
pub const GpuPipeline = struct {
    allocator: std.mem.Allocator,
    ctx: *Ctx,
    handle: *Pipeline,
    temp_ptrs: [][*]const f32,

    pub fn init(allocator: std.mem.Allocator, config: PipelineConfig, model_path: ?[:0]const u8) !*GpuPipeline {
        var ctx: ?*Ctx = null;
        try check(fvad_ctx_create(0, &ctx));
        errdefer fvad_ctx_destroy(ctx);
        try check(fvad_load_nsnet2_onnx(ctx.?, (model_path orelse "data/nsnet2-20ms-baseline.onnx").ptr));
        var p: ?*Pipeline = null;
        try check(fvad_pipeline_create(ctx.?, &config, null, &p));
        errdefer fvad_pipeline_destroy(p);
        var self = try allocator.create(GpuPipeline);
        self.* = .{
            .allocator = allocator,
            .ctx = ctx.?,
            .handle = p.?,
            .temp_ptrs = try allocator.alloc([*]const f32, config.n_channels),
        };
        return self;
    }

    pub fn deinit(self: *GpuPipeline) void {
        fvad_pipeline_destroy(self.handle);
        fvad_ctx_destroy(self.ctx);
        self.allocator.free(self.temp_ptrs);
        self.allocator.destroy(self);
    }

    /// AudioPipeline.pushSamples (src/AudioPipeline.zig:118-143)
    pub fn pushSamples(self: *GpuPipeline, channel_pcm: []const []const f32) !u64 {
        for (channel_pcm, 0..) |ch, i| self.temp_ptrs[i] = ch.ptr;
        var first: u64 = 0;
        try check(fvad_pipeline_push_samples(self.handle, self.temp_ptrs.ptr, channel_pcm[0].len, &first));
        return first;
    }

    /// pipeline.vad.vad_machine.vad_segments.toOwnedSlice() (SimulationInstance.zig:221)
    pub fn vadSegments(self: *GpuPipeline, allocator: std.mem.Allocator) ![]SpeechSegment {
        const n = fvad_pipeline_segment_count(self.handle);
        var out = try allocator.alloc(SpeechSegment, n);
        var got: usize = 0;
        try check(fvad_pipeline_segments(self.handle, out.ptr, n, &got));
        return out[0..got];
    }
};
