"""CPU tests of the oracle (oracle/liborc.so) itself: the reference's literal unit cases, numpy
cross-checks of the FFT restatement, and the internal-consistency properties the batched GPU
formulation relies on.  No GPU, no product code."""
import ctypes as C

import numpy as np
import pytest

import orc


def test_fft_matches_numpy_float64():
    # G2: impulse, DC, single-bin cosines, seeded noise, both sizes the reference uses
    rng = np.random.default_rng(1)
    for n in (320, 1024):
        cases = [np.eye(1, n, 0)[0], np.ones(n), rng.uniform(-1, 1, n)]
        for k in (1, 11, 43, 80, 160):
            cases.append(np.cos(2 * np.pi * k * np.arange(n) / n))
        for x in cases:
            x32 = x.astype(np.float32)
            X = orc.rfft(x32)
            ref = np.fft.rfft(x32.astype(np.float64))
            scale = np.abs(ref).max()
            assert np.abs(X - ref).max() <= 1e-5 * scale  # <= 1e-5 rel
            # kiss_fftri is unscaled: inverse(forward(x)) == n * x  (NSNet2.zig:323,335)
            y = orc.irfft_unscaled(X, n) / n
            assert np.abs(y - x32).max() <= 2e-6 * max(1.0, np.abs(x32).max())


def test_fft_other_radices():
    # the mixed-radix schedule (4,2,3,5,generic) for sizes the reference does not use
    rng = np.random.default_rng(2)
    for n in (6, 14, 30, 64, 100, 250):
        x = rng.uniform(-1, 1, n).astype(np.float32)
        ref = np.fft.rfft(x.astype(np.float64))
        assert np.abs(orc.rfft(x) - ref).max() <= 1e-5 * np.abs(ref).max()


def test_fft_rejects_odd_and_zero():
    assert not orc.lib().orc_fftr_alloc(0, 0)   # FFT.zig:41-43
    assert not orc.lib().orc_fftr_alloc(321, 0)


def test_fft_wrapper_errors_in_reference_order():
    L = orc.lib()
    cfg = L.orc_fftr_alloc(320, 0)
    x = np.zeros(320, np.float32)
    w = np.ones(320, np.float32)
    out = (orc.Cpx * 161)()
    assert L.orc_fft_fft(cfg, orc.fptr(x), 319, None, 0, orc.fptr(w), 320, out, 161) == -2
    assert L.orc_fft_fft(cfg, orc.fptr(x), 320, None, 0, orc.fptr(w), 319, out, 161) == -3
    assert L.orc_fft_fft(cfg, orc.fptr(x), 320, None, 0, orc.fptr(w), 320, out, 160) == -4
    # samples length is checked first (FFT.zig:91-102)
    assert L.orc_fft_fft(cfg, orc.fptr(x), 100, None, 0, orc.fptr(w), 1, out, 1) == -2
    # SplitSlice: first + second halves are concatenated
    xs = np.random.default_rng(3).uniform(-1, 1, 320).astype(np.float32)
    a, b = np.ascontiguousarray(xs[:123]), np.ascontiguousarray(xs[123:])
    out2 = (orc.Cpx * 161)()
    assert L.orc_fft_fft(cfg, orc.fptr(xs), 320, None, 0, orc.fptr(w), 320, out, 161) == 0
    assert L.orc_fft_fft(cfg, orc.fptr(a), 123, orc.fptr(b), 197, orc.fptr(w), 320, out2, 161) == 0
    assert bytes(out) == bytes(out2)
    L.orc_fftr_free(cfg)


def test_freq_to_bin_table():
    # G3: FFT.freqToBin on the 1024-point / 48 kHz transform (FFT.zig:156-167)
    L = orc.lib()
    assert L.orc_fft_freq_to_bin(1024, 48000, 500.0) == 11
    assert L.orc_fft_freq_to_bin(1024, 48000, 2000.0) == 43
    assert L.orc_fft_freq_to_bin(1024, 48000, 24000.0) == 512
    assert L.orc_fft_freq_to_bin(1024, 48000, 0.0) == 0
    assert L.orc_fft_freq_to_bin(1024, 48000, 24000.5) == -6   # OutOfRange
    assert L.orc_fft_freq_to_bin(1024, 48000, -1.0) == -7      # NegativeFrequency
    assert L.orc_fft_freq_to_bin(1024, 48000, 23.4375) == 1    # .5 rounds away from zero


def test_windows():
    # G1: both windows against float64 formulas; norm factor of the periodic Hann is 2
    w = orc.nsnet2_window()
    ref = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * np.arange(320) / 319))
    assert np.abs(w - ref).max() < 1e-6
    assert w[0] == 0.0
    wp = orc.hann_periodic(1024)
    refp = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(1024) / 1024)
    assert np.abs(wp - refp).max() < 1e-6
    nf = orc.lib().orc_window_norm_factor(orc.fptr(wp), 1024)
    assert abs(nf - 2.0) < 1e-5


def test_resample_down_up_carry():
    # G5: decimation picks every 3rd sample; upsampling is lerp with carry of the last sample
    L = orc.lib()
    x = np.arange(30, dtype=np.float32)
    out = np.zeros(10, np.float32)
    L.orc_downsample(orc.fptr(x), 30, None, 0, orc.fptr(out), 10, 3)
    assert np.array_equal(out, x[::3])
    a, b = np.ascontiguousarray(x[:7]), np.ascontiguousarray(x[7:])
    out2 = np.zeros(10, np.float32)
    L.orc_downsample(orc.fptr(a), 7, orc.fptr(b), 23, orc.fptr(out2), 10, 3)
    assert np.array_equal(out, out2)
    d = np.array([3.0, 6.0, 0.0], np.float32)
    up = np.zeros(9, np.float32)
    last = L.orc_upsample(orc.fptr(d), 3, orc.fptr(up), 9, C.c_float(0.0), 3)
    f1, f2 = np.float32(1) / np.float32(3), np.float32(2) / np.float32(3)
    exp = [0 + 3 * f1, 0 + 3 * f2, 3, 3 + 3 * f1, 3 + 3 * f2, 6, 6 - 6 * f1, 6 - 6 * f2, 0]
    assert np.allclose(up, exp, rtol=0, atol=1e-6)
    assert last == 0.0
    # two calls with the carry == one call over the concatenation
    d2 = np.random.default_rng(4).uniform(-1, 1, 20).astype(np.float32)
    whole = np.zeros(60, np.float32)
    L.orc_upsample(orc.fptr(d2), 20, orc.fptr(whole), 60, C.c_float(0.0), 3)
    p1, p2 = np.zeros(30, np.float32), np.zeros(30, np.float32)
    h1, h2 = np.ascontiguousarray(d2[:10]), np.ascontiguousarray(d2[10:])
    carry = L.orc_upsample(orc.fptr(h1), 10, orc.fptr(p1), 30, C.c_float(0.0), 3)
    L.orc_upsample(orc.fptr(h2), 10, orc.fptr(p2), 30, C.c_float(carry), 3)
    assert np.array_equal(whole, np.concatenate([p1, p2]))


def test_rolling_average_bit_exact_reference_order():
    # G6: partial fill, wrap-around and an initial value; mirrors RollingAverage.zig:11-56 in
    # float64 python with the same operation order
    L = orc.lib()
    rng = np.random.default_rng(5)
    for count, init in ((9, None), (23, None), (50, 0.005)):
        ra = L.orc_ra_create(count, 0 if init is None else 1, 0.0 if init is None else init)
        data = [0.0] * count
        written = 0
        if init is not None:
            data = [init] * count
            written = count
        widx = 0
        for _ in range(3 * count + 5):
            s = np.float32(rng.uniform(0, 0.1))
            got = L.orc_ra_push(ra, C.c_float(s))
            data[widx] = float(s)
            widx = (widx + 1) % count
            written = min(written + 1, count)
            scalar = 1.0 / written
            acc = 0.0
            for i in range(written):
                acc += data[i] * scalar
            assert got == acc  # bit-exact
        L.orc_ra_destroy(ra)


def test_segment_writer_literal_case():
    # G8: the reference's own "SegmentWriter" test, SegmentWriter.zig:130-181
    L = orc.lib()
    sw = L.orc_sw_create(10)
    first = np.array([1], np.float32)
    second = np.array([2, 3, 4], np.float32)
    w = lambda off: L.orc_sw_write(sw, orc.fptr(first), 1, orc.fptr(second), 3, off)  # noqa: E731
    assert w(0) == 4
    assert w(2) == 2
    assert w(1) == 3
    assert L.orc_sw_write_index(sw) == 9
    assert w(2) == 1
    assert w(3) == 0
    got = np.ctypeslib.as_array(L.orc_sw_data(sw), (10,))
    assert np.array_equal(got, [1, 2, 3, 4, 3, 4, 2, 3, 4, 3])
    assert L.orc_sw_is_full(sw)
    L.orc_sw_reset(sw, 5)
    assert L.orc_sw_write_index(sw) == 0 and L.orc_sw_index(sw) == 5
    L.orc_sw_destroy(sw)


def test_vad_metadata_literal_cases():
    # G8: VADMetadata.zig:70-110
    L = orc.lib()

    def run(pushes):
        m = orc.Meta()
        L.orc_meta_reset(C.byref(m))
        for kind, val, weight in pushes:
            r = orc.MetaResult()
            setattr(r, "has_" + kind, 1)
            setattr(r, "volume_" + kind, val)
            L.orc_meta_push(C.byref(m), C.byref(r), C.c_float(weight))
        return L.orc_meta_to_result(C.byref(m))

    assert abs(run([("min", 100, 1), ("min", 80, 1), ("min", 90, 1)]).volume_min - 80.0) < 1e-3
    assert abs(run([("max", 80, 1), ("max", 100, 1), ("max", 90, 1)]).volume_max - 100.0) < 1e-3
    assert abs(run([("ratio", 0.9, 1), ("ratio", 0.8, 1)]).volume_ratio - 0.85) < 1e-3
    assert abs(run([("ratio", 1.0, 1), ("ratio", 0.0, 9)]).volume_ratio - 0.1) < 1e-3
    assert run([("min", 1, 1)]).has_ratio == 0  # absent fields stay null


def test_statistics_literal_cases():
    # G8: statistics.zig:286-360 "calcFalsePositiveSec #1/#2"
    L = orc.lib()
    refs = (orc.SegSec * 2)(orc.SegSec(2, 3), orc.SegSec(4, 5))
    cfg = orc.StatConfig(0.0, 2.0, 2.0, 2.0)
    fp1 = L.orc_calc_false_positive_sec(orc.SegSec(1, 6), refs, 2, C.byref(cfg))
    fp2 = L.orc_calc_false_positive_sec(orc.SegSec(1, 10), refs, 2, C.byref(cfg))
    assert abs(fp1 - 0.0) < 1e-3
    assert abs(fp2 - 3.0) < 1e-3


def test_statistics_from_segments_and_aggregate():
    L = orc.lib()
    cfg = orc.StatConfig(0.7, 5.0, 10.0, 5.0)  # simulator.zig:127-132
    vad = (orc.SegSec * 3)(orc.SegSec(8, 20), orc.SegSec(40, 45), orc.SegSec(100, 103))
    ref = (orc.SegSec * 3)(orc.SegSec(10, 14), orc.SegSec(41, 44), orc.SegSec(60, 62))
    s = L.orc_stats_from_segments(vad, 3, ref, 3, C.byref(cfg))
    # seg 1: ref extruded to [5,24] covers it -> TP 12; seg 2: [36,54] covers -> TP 5;
    # seg 3: unmatched -> FP 3; ref 3 (2 s, unmatched) -> FN 2
    assert abs(s.true_positives_sec - 17.0) < 1e-4
    assert abs(s.false_positives_sec - 3.0) < 1e-4
    assert abs(s.false_negatives_sec - 2.0) < 1e-4
    assert abs(s.total_positives_sec - 19.0) < 1e-4  # TP is added into P (statistics.zig:92-93)
    assert abs(s.precision - 17.0 / 20.0) < 1e-6
    arr = (orc.SingleStats * 2)(s, s)
    a = L.orc_stats_aggregate(arr, 2)
    assert abs(a.total_positives_sec - 38.0) < 1e-4
    assert a.precision.min == a.precision.max == s.precision
    assert abs(a.f_score_beta - 0.7) < 1e-7


def _mk_weights(seed=11):
    rng = np.random.default_rng(seed)
    wd = {}
    for k, s in orc.WEIGHT_SHAPES(161, 400, 400, 600, 600).items():
        fan = s[-1] if len(s) == 2 else 400
        wd[k] = rng.uniform(-1, 1, s).astype(np.float32) * np.float32(1.5 / np.sqrt(fan))
    return wd


def test_nsnet2_forward_matches_float64_and_torch_gru():
    # the ONNX GRU convention (z,r,h; linear_before_reset=1) against torch.nn.GRU (r,z,n)
    import torch
    wd = _mk_weights()
    rng = np.random.default_rng(6)
    f = rng.uniform(-8, 2, (12, 161)).astype(np.float32)
    g = orc.nsnet2_forward(wd, f)
    H = 400

    def pt(M):
        z, r, h = np.split(M, 3, axis=0)
        return torch.from_numpy(np.concatenate([r, z, h], 0)).double()

    with torch.no_grad():
        x = torch.from_numpy(f).double()
        lin = lambda x, w, b: x @ torch.from_numpy(wd[w]).double().T + torch.from_numpy(wd[b]).double()  # noqa: E731
        a = lin(x, "fc1_w", "fc1_b")
        for n in ("gru1", "gru2"):
            m = torch.nn.GRU(400, 400, batch_first=True).double()
            m.weight_ih_l0.copy_(pt(wd[n + "_w"]))
            m.weight_hh_l0.copy_(pt(wd[n + "_r"]))
            m.bias_ih_l0.copy_(pt(wd[n + "_b"][:3 * H]))
            m.bias_hh_l0.copy_(pt(wd[n + "_b"][3 * H:]))
            a, _ = m(a[None])
            a = a[0]
        a = torch.relu(lin(a, "fc2_w", "fc2_b"))
        a = torch.relu(lin(a, "fc3_w", "fc3_b"))
        a = torch.sigmoid(lin(a, "fc4_w", "fc4_b"))
    assert np.abs(g - a.numpy()).max() < 2e-6
    assert g.std() > 0.02  # the synthetic net is not saturated / constant


def test_denoise_first_chunk_zero_history_and_delay():
    # 8a-S: zero history at t=0, warm-up feature rows are literal zeros (not -12), and with a
    # unit-gain network the output is the input delayed by 480 samples @48 kHz
    wd = _mk_weights()
    # force gains == 1: fc4 weights 0, bias large -> sigmoid saturates to exactly 1.0f
    wd["fc4_w"] = np.zeros_like(wd["fc4_w"])
    wd["fc4_b"] = np.full_like(wd["fc4_b"], 40.0)
    d = orc.Denoiser(wd)
    rng = np.random.default_rng(7)
    # a band-limited signal (decimation by 3 has no anti-alias filter, resample.zig:9-29)
    t = np.arange(48000) / 48000.0
    x = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1234 * t)).astype(np.float32)
    rc, y1 = d.denoise(x[:24000])
    assert rc == 0
    feats = d.features()
    assert np.all(feats[:4] == 0.0)
    assert np.all(d.gains()[4:] == 1.0)
    rc, y2 = d.denoise(x[24000:])
    y = np.concatenate([y1, y2])
    # every third output sample is a decimated-rate sample; sqrt-Hann analysis*synthesis with 50 %
    # overlap sums to ~one (the reference's window is the SYMMETRIC Hann, denominator N-1, so the
    # overlap-add ripples by ~0.5 %), i.e. the decimated input delayed by 160 samples
    dec_in = x[::3]
    dec_out = y[2::3]
    assert np.abs(dec_out[160 + 160:] - dec_in[160:-160]).max() < 5e-3
    assert np.abs(dec_out[:160]).max() < 1e-6  # first 10 ms: only the zero history
    assert d.denoise(x[:100])[0] == -8  # InvalidInputLength, NSNet2.zig:166-169


def test_pipeline_push_granularity_invariance_and_chunking():
    # AudioPipeline.pushSamples chunking (AudioPipeline.zig:118-143): results depend only on the
    # sample sequence, not on how it is cut into pushes; an incomplete trailing chunk is not
    # processed (SimulationInstance.zig:221 reads the segments without a flush)
    wd = _mk_weights()
    rng = np.random.default_rng(8)
    n = 24000 * 3 + 5000
    pcm = rng.uniform(-0.1, 0.1, (2, n)).astype(np.float32)
    a = orc.Pipeline(wd, n_channels=2, keep_denoised=True)
    assert a.push(pcm) == 0
    b = orc.Pipeline(wd, n_channels=2, keep_denoised=True)
    pos = 0
    for step in (1000, 23000, 24001, 7, 30000, 10**9):
        nxt = min(n, pos + step)
        assert b.push(pcm[:, pos:nxt]) == pos  # returns index of first pushed sample
        pos = nxt
    assert a.chunk_rms().shape == (3, 2)
    assert np.array_equal(a.band_volumes(), b.band_volumes())
    assert np.array_equal(a.denoised(), b.denoised())
    assert a.band_volumes().shape[0] == (3 * 24000) // 1024
    # frame index = absolute sample of the window start (BufferedFFT.zig:149,152)
    tr = a.vad_traces()
    assert [t[0] for t in tr] == [1024 * k for k in range(len(tr))]
    # stereo ratio: min/max of the channel RMS, carried as a sample-weighted mean
    rms = a.chunk_rms()
    ratio0 = rms[0].min() / rms[0].max()
    assert abs(a.frame_vol_ratio()[0] - ratio0) < 1e-6
    # a frame straddling chunks 0|1 (frame 23 covers samples 23552..24575)
    w0, w1 = 24000 - 23552, 24576 - 24000
    ratio1 = rms[1].min() / rms[1].max()
    mixed = (ratio0 * w0 + ratio1 * w1) / 1024
    assert abs(a.frame_vol_ratio()[23] - mixed) < 1e-6


def test_pipeline_rejects_wrong_sample_rate():
    wd = _mk_weights()
    p = orc.Pipeline(wd, sample_rate=44100)
    assert not p.h and p.err == -9  # InvalidSampleRate, VADPipeline.zig:55-58


def test_vad_machine_scenarios():
    # G7: exact integer segments for scripted band volumes (VADMachine.zig:138-325)
    L = orc.lib()
    cfg = orc.VadConfig()
    L.orc_vad_config_default(C.byref(cfg))

    def run(script):
        v = L.orc_vad_create(C.byref(cfg), 48000, 1, 1024)
        events = []
        for k, vol in enumerate(script):
            vols = np.array([vol], np.float32)
            r = L.orc_vad_run(v, 1024 * k, orc.fptr(vols), 1, C.c_float(1.0))
            if r.recording_state:
                events.append((k, r.recording_state, r.sample_number))
        n = L.orc_vad_n_segments(v)
        segs = [(L.orc_vad_segments(v)[i].sample_from, L.orc_vad_segments(v)[i].sample_to,
                 L.orc_vad_segments(v)[i].vad_met_sec) for i in range(n)]
        L.orc_vad_destroy(v)
        return events, segs

    quiet, loud = 0.001, 1.0
    # long utterance starting late: margins of 2 s on both sides
    pre, talk, post = 200, 100, 120
    ev, segs = run([quiet] * pre + [loud] * talk + [quiet] * post)
    # short-term average (9 frames) crosses 10 * long-term at the first loud frame
    start = 1024 * pre
    # the closing frame: short-term falls below threshold when enough quiet frames entered
    assert len(segs) == 1
    assert segs[0][0] == start - 96000
    assert ev[0][1] == 1 and ev[0][2] == start - 96000      # started
    assert ev[1][1] == 2 and ev[1][2] == segs[0][1]         # completed
    assert (segs[0][1] - 96000) % 1024 == 0                 # end index is a frame start
    # early utterance: the start offset clamps at 0 (VADMachine.zig:315)
    ev, segs = run([quiet] * 20 + [loud] * 100 + [quiet] * 120)
    assert segs[0][0] == 0
    # too short (< 0.7 s incl. the short-term tail): aborted, no segment
    ev, segs = run([quiet] * 200 + [loud] * 12 + [quiet] * 150)
    assert segs == [] and any(e[1] == 3 for e in ev)
    # a gap shorter than 2 s resumes the same segment
    ev, segs = run([quiet] * 200 + [loud] * 60 + [quiet] * 60 + [loud] * 60 + [quiet] * 120)
    assert len(segs) == 1
    # speech still open at end of input is dropped (no flush)
    ev, segs = run([quiet] * 200 + [loud] * 100)
    assert segs == []
    # opening cancelled when the threshold drops before 0.2 s
    ev, segs = run([quiet] * 200 + [loud] * 2 + [quiet] * 100)
    assert segs == [] and ev == []


def test_oracle_reproduces_committed_golden_vectors():
    # tests/golden/*.npz were produced by tests/golden/make_golden.py from this oracle; FFTs and the
    # dense layers are pure f32 arithmetic in a fixed order, so they reproduce bit-exactly; expf /
    # tanhf / log10f / cosf come from the host libm, so the network outputs get a few-ulp allowance
    import os
    from conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_seed7.npz"))
    assert np.array_equal(orc.nsnet2_window(), g["win320"])
    assert np.array_equal(orc.hann_periodic(1024), g["win1024"])
    assert np.array_equal(orc.rfft(g["fft320_x"] * g["win320"]), g["fft320_X"])
    assert np.array_equal(orc.rfft(g["fft1024_x"] * g["win1024"]), g["fft1024_X"])
    from conftest import load_package
    W = load_package().binding.synth_weights(7)
    gains = orc.nsnet2_forward(W, g["chunk_features"])
    assert np.abs(gains - g["chunk_gains"]).max() <= 1e-6
    p = orc.Pipeline(W, n_channels=1, keep_denoised=True)
    p.push(g["stream_pcm"][None])
    assert np.abs(p.denoised()[0] - g["stream_denoised"]).max() <= 1e-6 * np.abs(g["stream_denoised"]).max()
    assert np.allclose(p.band_volumes()[:, 0], g["stream_band"], rtol=1e-5, atol=0)
    assert np.array_equal(p.chunk_rms()[:, 0], g["stream_rms"])
