"""Simulator-harness tests: WAV input, plan parsing, per-frame metadata,
report text -- CPU; the whole plan end to end against the oracle -- GPU."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

import orc


def write_wav(path, pcm, sample_rate=48000, fmt="f32", extensible=False):
    pcm = np.asarray(pcm, np.float32)
    nch, n = pcm.shape
    inter = pcm.T.reshape(-1)
    if fmt == "f32":
        tag, bits, data = 3, 32, inter.astype("<f4").tobytes()
    else:
        tag, bits, data = 1, 16, np.clip(np.round(inter * 32768.0), -32768, 32767).astype("<i2").tobytes()
    block = nch * bits // 8
    if extensible:
        sub = struct.pack("<H", tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
        fmt_body = struct.pack("<HHIIHHHHI", 0xFFFE, nch, sample_rate, sample_rate * block, block, bits, 22, bits, 0) + sub
    else:
        fmt_body = struct.pack("<HHIIHH", tag, nch, sample_rate, sample_rate * block, block, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body
    chunks += b"LIST" + struct.pack("<I", 5) + b"junk!" + b"\x00"  # odd-sized chunk + pad byte
    chunks += b"data" + struct.pack("<I", len(data)) + data
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def test_wav_reader(fv, tmp_path):
    rng = np.random.default_rng(0)
    pcm = rng.uniform(-0.9, 0.9, (2, 1000)).astype(np.float32)
    for ext in (False, True):
        p = str(tmp_path / f"f32_{ext}.wav")
        write_wav(p, pcm, fmt="f32", extensible=ext)
        got, sr = fv.wav_read(p)
        assert sr == 48000 and np.array_equal(got, pcm)
    p = str(tmp_path / "pcm16.wav")
    write_wav(p, pcm[:1], sample_rate=44100, fmt="pcm16")
    got, sr = fv.wav_read(p)
    q = np.clip(np.round(pcm[:1] * 32768.0), -32768, 32767).astype(np.int16).astype(np.float32) / np.float32(32768)
    assert sr == 44100 and np.array_equal(got, q)          # libsndfile's normalised short -> float
    L = fv.lib()
    out, nc, nf, srr = C.POINTER(fv.c_float_p)(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    args = (C.byref(out), C.byref(nc), C.byref(nf), C.byref(srr))
    assert L.fvad_wav_read(str(tmp_path / "nope.wav").encode(), *args) == -105
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"RIFF\x00\x00\x00\x00WAVEjunk")
    assert L.fvad_wav_read(str(bad).encode(), *args) == -104
    ogg = tmp_path / "x.ogg"
    ogg.write_bytes(b"OggS" + b"\x00" * 64)
    assert L.fvad_wav_read(str(ogg).encode(), *args) == -104


def test_zig_fixed_rounding_and_report_layout(pkg, fv):
    sim = pkg.simulator
    assert sim.zig_fixed(0.25, 1) == "0.3"        # half-up on the shortest decimal, unlike C's half-even
    assert sim.zig_fixed(2.5, 0) == "3"
    assert sim.zig_fixed(np.float32(0.1), 4) == "0.1000"
    assert sim.zig_fixed(1234.5678, 1) == "1234.6"
    assert sim.zig_fixed(float("nan"), 1) == "nan"
    s = fv.stats_from_segments([(8, 20), (40, 45), (100, 103)], [(10, 14), (41, 44), (60, 62)],
                               {"ignore_shorter_than_sec": 0.7, "extrude_start": 5, "extrude_end": 10, "fill_gaps": 5})
    agg = fv.stats_aggregate([s, s])
    txt = sim.report_text(["Sainz", "Hulkenberg"], [s, s], agg)
    lines = txt.split("\n")
    hdr = next(l for l in lines if l.startswith("|") and "Name" in l)
    assert hdr == "| " + "Name".rjust(30) + " |    P |   TP |   FP |   FN |    TPR |    PPV |  FNR (!) |  FDR (!) |"
    row = next(l for l in lines if "Sainz" in l)
    assert row == "| " + "Sainz".rjust(30) + " |   19 |   17 |    3 |    2 |  89.5% |  85.0% |    10.5% |    15.0% |"
    assert len(row) == len(hdr)
    assert "Total speech duration  (P):    38.0 sec" in txt
    assert "False negatives       (FN):     4.0 sec    Min.    Avg.    Max. " in txt
    assert "True positive rate   (TPR):    89.5%  |   89.5% / 89.5% / 89.5% " in txt
    assert "F-Score (β =  0.70)" in txt
    assert txt.startswith("\n\n=> Definitions\n\nP   (Positives):")


def test_plan_loading(pkg, tmp_path):
    plan = {"instances": [{"name": "A", "audio_path": "a.wav", "ref_path": "sub/a.txt", "extra": 1}],
            "config": {"vad_config": {"vad_machine_config": {"speech_threshold_factor": 8, "initial_long_term_avg": None,
                                                               "not_a_field": 3},
                                      "alt_vad_machine_configs": [{"max_speech_gap_sec": 1.0}]},
                       "output_dir": "sim-out", "preload_audio": True, "audio_read_frame_count": 4800, "unknown": {}}}
    p = tmp_path / "plans" / "plan.json"
    p.parent.mkdir()
    p.write_text(json.dumps(plan))
    got = pkg.simulator.load_plan(str(p))
    assert got["instances"][0]["audio_path"] == str(tmp_path / "plans" / "a.wav")   # relative to the plan file
    assert got["instances"][0]["ref_path"] == str(tmp_path / "plans" / "sub" / "a.txt")
    assert got["vad_machine_config"] == {"speech_threshold_factor": 8.0, "has_initial_long_term_avg": 0}
    assert got["alt_vad_machine_configs"] == [{"max_speech_gap_sec": 1.0}]
    assert got["fft_size"] == 1024 and got["preload_audio"] and got["audio_read_frame_count"] == 4800


def test_frame_ratios_bit_exact_vs_oracle(pkg):
    # the metadata chain BufferedVolumeAnalyzer -> BufferedDenoiser -> BufferedFFT, stereo, 5 chunks
    rng = np.random.default_rng(3)
    W = pkg.binding.synth_weights(7)
    pcm = np.stack([rng.uniform(-0.2, 0.2, 24000 * 5), rng.uniform(-0.1, 0.1, 24000 * 5)]).astype(np.float32)
    p = orc.Pipeline(W, n_channels=2)
    p.push(pcm)
    got = pkg.simulator.frame_ratios(p.chunk_rms(), p.band_volumes().shape[0])
    assert np.array_equal(got, p.frame_vol_ratio())
    mono = pkg.simulator.frame_ratios(p.chunk_rms()[:, :1], 10)
    assert np.all(mono == 1.0)
    assert np.all(pkg.simulator.frame_ratios(np.zeros((2, 1), np.float32), 5) == 0.0)   # silence: max == 0 -> 0


@pytest.mark.gpu
def test_run_plan_end_to_end(pkg, fv, gpu_ctx, weights7, tmp_path):
    synth = pkg.synth
    insts = []
    expected = []
    for i, (nch, fmt) in enumerate(((1, "f32"), (2, "pcm16"))):
        pcm, labels = synth.make_stream(70.0, seed=200 + i, n_channels=nch)
        write_wav(str(tmp_path / f"s{i}.wav"), pcm, fmt=fmt)
        (tmp_path / f"s{i}.txt").write_text(synth.labels_to_audacity(labels))
        insts.append({"name": f"stream{i}", "audio_path": f"s{i}.wav", "ref_path": f"s{i}.txt"})
        decoded, _ = fv.wav_read(str(tmp_path / f"s{i}.wav"))
        ref = orc.Pipeline(weights7, n_channels=nch)
        ref.push(decoded)
        expected.append((ref.segments(), labels))
    plan = {"instances": insts, "config": {"vad_config": {}, "output_dir": "out", "preload_audio": True}}
    (tmp_path / "plan.json").write_text(json.dumps(plan))
    text, results = pkg.simulator.run_plan(str(tmp_path / "plan.json"), ctx=gpu_ctx, out=None)
    O = orc.lib()
    ocfg = orc.StatConfig(0.7, 5.0, 10.0, 5.0)
    for r, (segs_ref, labels) in zip(results, expected):
        assert [(s[0], s[1]) for s in r["segments"]] == [(s[0], s[1]) for s in segs_ref] and len(segs_ref) >= 2
        v = (orc.SegSec * len(segs_ref))(*[orc.SegSec(np.float32(s[0]) / np.float32(48000), np.float32(s[1]) / np.float32(48000)) for s in segs_ref])
        lab = fv.parse_audacity(synth.labels_to_audacity(labels))
        rf = (orc.SegSec * len(lab))(*[orc.SegSec(a, b) for a, b in lab])
        o = O.orc_stats_from_segments(v, len(segs_ref), rf, len(lab), C.byref(ocfg))
        assert r["stats"].true_positives_sec == o.true_positives_sec
        assert r["stats"].false_negatives_sec == o.false_negatives_sec and r["stats"].precision == o.precision
        assert r["audacity"].count("\n") >= len(segs_ref)
        assert r["debug_info"][0].startswith("vr:") and r["debug_info"][0].endswith("s")
    assert "stream0" in text and "stream1" in text and "=> Aggregate stats" in text
    # one context + one host thread per device entry, instances dealt round-robin: same report, same segments
    # (two contexts on device 0 stand in for two GPUs)
    text2, results2 = pkg.simulator.run_plan(str(tmp_path / "plan.json"), synth_seed=7, out=None, devices=[0, 0])
    assert text2 == text
    assert [r["segments"] for r in results2] == [r["segments"] for r in results]
    assert all(bytes(a["stats"]) == bytes(b["stats"]) for a, b in zip(results, results2))
    outs = os.listdir(tmp_path / "out")
    assert len(outs) >= 1 and sorted(os.listdir(tmp_path / "out" / outs[0])) == ["report.txt", "stream0-audacity.txt", "stream1-audacity.txt"]
