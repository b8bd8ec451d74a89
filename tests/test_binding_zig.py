"""The C ABI's three declaration layers agree: include/fvad.h (the contract), bindings/fvad.zig (what a Zig host such as the
reference compiles against, src/AudioPipeline.zig:40-44 / src/FFT.zig:35-180 / src/NSNet2.zig:35-237 call sites) and the
ctypes table of formula-vad_amd/binding.py (what every test in this repository calls through).  No zig in this image, so the
Zig file cannot be compiled here: tests/abi_check.py parses its `extern` declarations instead and compares them with the
header mechanically -- a renamed field, a re-ordered argument or a dropped function fails here, not on a maintainer's box."""
import ctypes as C
import os
import re

import pytest

import abi_check as A
from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "fvad.h")
ZIG = os.path.join(ROOT, "bindings", "fvad.zig")
# fvad.h functions a Zig host does not need (none today: the binding declares the whole ABI)
ALLOW_UNDECLARED = ()


@pytest.fixture(scope="module")
def c_side():
    text = open(HEADER).read()
    c = A.parse_c_header(text)
    c["constants"].update(A.c_defines(text))
    return c


@pytest.fixture(scope="module")
def zig_text():
    return open(ZIG).read()


def test_parsers_see_the_whole_header(c_side, zig_text):
    # the parsers themselves: every `fvad_` prototype of the header is found (count by a plain regex over the raw text),
    # every struct, the callback type, the opaque handles
    raw = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    protos = set(re.findall(r"\b(fvad_[a-z0-9_]+)\s*\(", raw))
    assert protos == set(c_side["functions"]), protos ^ set(c_side["functions"])
    assert len(c_side["functions"]) >= 100 and len(c_side["structs"]) == 16 and len(c_side["opaque"]) == 9
    assert "fvad_recording_cb" in c_side["fnptrs"]
    assert c_side["constants"]["FVAD_ABI_VERSION"] == 3 and c_side["constants"]["FVAD_ERR_BUFFER_TOO_SMALL"] == -106
    z = A.parse_zig(zig_text)
    assert len(z["functions"]) >= 100 and len(z["structs"]) == 16


def test_zig_binding_matches_the_header(c_side, zig_text):
    bad = A.compare(c_side, A.parse_zig(zig_text), ALLOW_UNDECLARED)
    assert not bad, "\n".join(bad)


def test_zig_struct_layouts_match_gcc(c_side, zig_text):
    # sizeof / offsetof of every struct as gcc lays out the header, against the C-ABI layout of the Zig `extern struct`s
    want = A.c_layout(HEADER, c_side["structs"])
    got = A.zig_layout(A.parse_zig(zig_text)["structs"])
    assert want == got, {s: (want[s], got.get(s)) for s in want if want[s] != got.get(s)}
    assert want["fvad_lane"][0] == 128 and want["fvad_pipeline_config"][0] == 112     # (pins the checker itself)


MUTATIONS = [
    # (what, pattern, replacement, a word the finding must contain)
    ("two fields of fvad_lane swapped", "    denoised: ?[*]f32 = null,\n    band_sum: ?[*]f32 = null,", "    band_sum: ?[*]f32 = null,\n    denoised: ?[*]f32 = null,", "fvad_lane"),
    ("a field renamed", "    n_fft_frames: usize = 0,", "    n_frames: usize = 0,", "fvad_lane"),
    ("a field's type narrowed", "    sample_number: u64,", "    sample_number: u32,", "fvad_vad_result"),
    ("two arguments swapped", "fvad_pipeline_push_samples(p: *Pipeline, channel_pcm: [*]const [*]const f32, n_samples: usize,",
     "fvad_pipeline_push_samples(p: *Pipeline, n_samples: usize, channel_pcm: [*]const [*]const f32,", "fvad_pipeline_push_samples"),
    ("an argument dropped", "fvad_lane_state_seek(s: *LaneState, sample_index: u64, fft_size: usize)", "fvad_lane_state_seek(s: *LaneState, sample_index: u64)", "fvad_lane_state_seek"),
    ("a pointer level lost", "fvad_ctx_create(device: c_int, out: *?*Ctx)", "fvad_ctx_create(device: c_int, out: ?*Ctx)", "fvad_ctx_create"),
    ("constness lost", "fvad_fft_inverse(fft: *Fft, bins: [*]const Complex,", "fvad_fft_inverse(fft: *Fft, bins: [*]Complex,", "fvad_fft_inverse"),
    ("a return type changed", "pub extern \"c\" fn fvad_pipeline_total_write_count(p: *const Pipeline) u64;", "pub extern \"c\" fn fvad_pipeline_total_write_count(p: *const Pipeline) usize;", "fvad_pipeline_total_write_count"),
    ("a function dropped", "pub extern \"c\" fn fvad_engine_run(ctx: *Ctx, lanes: [*]Lane, n_lanes: usize, opts: ?*const EngineOpts) c_int;\n", "", "fvad_engine_run"),
    ("a function the header does not have", "pub extern \"c\" fn fvad_abi_version() c_int;", "pub extern \"c\" fn fvad_abi_version() c_int;\npub extern \"c\" fn fvad_made_up(x: c_int) c_int;", "fvad_made_up"),
    ("a status code changed", "pub const err_no_device = -101;", "pub const err_no_device = -111;", "FVAD_ERR_NO_DEVICE"),
    ("the callback's signature changed", "?*const fn (ctx: ?*anyopaque, recording: *const AudioBuffer) callconv(.C) void;", "?*const fn (recording: *const AudioBuffer) callconv(.C) void;", "fvad_recording_cb"),
    ("a callback slot dropped", "    on_denoised_recording: RecordingCb = null,\n", "", "fvad_callbacks"),
]


@pytest.mark.parametrize("what,old,new,word", MUTATIONS, ids=[m[0] for m in MUTATIONS])
def test_the_checker_fails_on_a_deliberately_broken_binding(c_side, zig_text, what, old, new, word):
    assert zig_text.count(old) == 1, f"mutation anchor not found: {old!r}"
    bad = A.compare(c_side, A.parse_zig(zig_text.replace(old, new)), ALLOW_UNDECLARED)
    assert bad and any(word in b for b in bad), (what, bad)


def test_a_reordered_field_also_moves_the_layout(c_side, zig_text):
    # fields of different sizes swapped: the layout check sees it even if names were ignored
    old = "    skip_processing: i32 = 0,\n    fft_size: usize = 1024,"
    new = "    fft_size: usize = 1024,\n    skip_processing: i32 = 0,"
    assert zig_text.count(old) == 1
    want = A.c_layout(HEADER, {"fvad_pipeline_config": c_side["structs"]["fvad_pipeline_config"], "fvad_vad_config": c_side["structs"]["fvad_vad_config"]})
    got = A.zig_layout(A.parse_zig(zig_text.replace(old, new))["structs"])
    assert want["fvad_pipeline_config"] != got["fvad_pipeline_config"]


def test_zig_wrappers_cover_the_reference_surface(zig_text):
    # the hand-written wrappers a reference maintainer calls: AudioPipeline's surface (AudioPipeline.zig:40,104,114,118 and
    # `pipeline.vad.vad_machine.vad_segments`, SimulationInstance.zig:221), FFT's (FFT.zig:35,78,85,115,137-180), NSNet2's
    # (NSNet2.zig:35,144,157,161), the batch form and the statistics
    for needle in ("pub fn init(allocator: std.mem.Allocator, config: Config, callbacks: ?RefCallbacks) !*GpuPipeline",
                   "pub fn initOnDevice(allocator: std.mem.Allocator, config: Config, callbacks: ?RefCallbacks, device: c_int)",
                   "pub fn pushSamples(self: *GpuPipeline, channel_pcm: []const []const f32) !u64",
                   "pub fn totalWriteCount(self: *const GpuPipeline) u64", "pub fn deinit(self: *GpuPipeline) void",
                   "vad_segments: std.ArrayList(RefSpeechSegment)",
                   "pub fn fft(self: *GpuFFT, samples: SplitSlice(f32), window: []const f32, bins: []Complex) !void",
                   "pub fn invFft(", "pub fn binCount(", "pub fn binWidth(", "pub fn nyquistFreq(", "pub fn freqToBin(", "pub fn binToFreq(",
                   "pub fn denoise(self: GpuNSNet2, samples: SplitSlice(f32), denoised_result: []f32) !void", "pub fn getChunkSize(",
                   "pub fn runBatch(ctx: Context, lanes: []Lane, opts: ?EngineOpts) !void", "pub fn aggregate(", "pub fn singleStats("):
        assert needle in zig_text, needle
    # every status code of the header that has a reference error name is mapped by check()
    for code in ("err_invalid_fft_size", "err_invalid_samples_length", "err_invalid_window_length", "err_invalid_result_length",
                 "err_invalid_bins_length", "err_out_of_range", "err_negative_frequency", "err_invalid_input_length",
                 "err_invalid_sample_rate", "err_channel_count_mismatch", "err_alloc_failed", "err_no_device"):
        assert f"Status.{code} => error." in zig_text, code


# ---------------------------------------------------------------- the ctypes layer against the same header
def _ctypes_kind(t):
    """(pointer depth, scalar size or None) of a ctypes type"""
    if t is None:
        return (0, 0)
    if t in (C.c_char_p, C.c_void_p):
        return (1, None)
    depth = 0
    while hasattr(t, "_type_") and not isinstance(t._type_, str):
        depth += 1
        t = t._type_
    if depth:
        if t in (C.c_char_p, C.c_void_p):
            depth += 1
        return (depth, None)
    return (0, C.sizeof(t))


def test_ctypes_table_matches_the_header(c_side, fv):
    sig = fv.SIGNATURES
    assert set(sig) == set(c_side["functions"]), set(sig) ^ set(c_side["functions"])
    struct_sizes = {s: v[0] for s, v in A.c_layout(HEADER, c_side["structs"]).items()}
    bad = []
    for name, (ret, params) in c_side["functions"].items():
        res, args = sig[name]
        if len(args) != len(params):
            bad.append(f"{name}: {len(params)} arguments in C, {len(args)} in ctypes")
            continue
        for i, ((pn, pt), a) in enumerate(zip(params, args)):
            depth, size = _ctypes_kind(a)
            if pt.depth == 0:
                want = A.SIZES.get(pt.base) or struct_sizes[pt.base]
                if depth != 0 or size != want:
                    bad.append(f"{name}: argument {i} ({pn}) is {pt} in C, ctypes passes depth {depth} size {size}")
            elif depth == 0:
                bad.append(f"{name}: argument {i} ({pn}) is a pointer in C, a {size}-byte scalar in ctypes")
            elif a not in (C.c_void_p, C.c_char_p) and depth != pt.depth:
                bad.append(f"{name}: argument {i} ({pn}) has pointer depth {pt.depth} in C, {depth} in ctypes")
        depth, size = _ctypes_kind(res)
        if ret.depth == 0:
            want = 0 if ret.base == "void" else (A.SIZES.get(ret.base) or struct_sizes[ret.base])
            if depth != 0 or size != want:
                bad.append(f"{name}: returns {ret} in C, ctypes depth {depth} size {size}")
        elif depth == 0:
            bad.append(f"{name}: returns a pointer in C, a scalar in ctypes")
    assert not bad, "\n".join(bad)


def test_ctypes_structs_match_the_header(c_side, fv):
    names = {"fvad_complex": fv.Complex, "fvad_nsnet2_weights": fv.Weights, "fvad_vad_config": fv.VadConfig,
             "fvad_speech_segment": fv.SpeechSegment, "fvad_vad_result": fv.VadResult, "fvad_vad_audit": fv.VadAudit,
             "fvad_lane": fv.Lane, "fvad_engine_opts": fv.EngineOpts, "fvad_audio_buffer": fv.AudioBuffer,
             "fvad_callbacks": fv.Callbacks, "fvad_pipeline_config": fv.PipelineConfig, "fvad_single_stats": fv.SingleStats,
             "fvad_agg_stat": fv.AggStat, "fvad_aggregate_stats": fv.AggregateStats, "fvad_stat_config": fv.StatConfig,
             "fvad_segment_sec": fv.SegmentSec}
    assert set(names) == set(c_side["structs"])
    layout = A.c_layout(HEADER, c_side["structs"])
    for cname, cls in names.items():
        size, offs = layout[cname]
        assert C.sizeof(cls) == size, cname
        assert [f[0] for f in cls._fields_] == [n for n, _ in c_side["structs"][cname]], cname
        for f in cls._fields_:
            assert getattr(cls, f[0]).offset == offs[f[0]], (cname, f[0])
