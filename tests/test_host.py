"""CPU tests of the product's host side (libfvad_hip.so loads without a GPU): the C ABI exports
every symbol include/fvad.h declares, the host logic (VAD state machine, rolling averages,
Evaluator statistics, windows, weight generation, ONNX reader) matches the oracle, GPU entry
points fail loudly without a device, and the N>1 sharding path runs over gloo with 2 ranks."""
import ctypes as C
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

import orc
from conftest import ROOT


def test_abi_exports_every_declared_symbol(fv):
    hdr = open(os.path.join(ROOT, "include", "fvad.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fvad_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fvad_recording_cb"}
    L = fv.lib()
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, f"declared in fvad.h but not exported: {missing}"
    assert declared == set(fv.SIGNATURES), (declared ^ set(fv.SIGNATURES))
    assert L.fvad_abi_version() == 3
    assert L.fvad_status_name(-8) == b"InvalidInputLength"


def test_gpu_entry_points_fail_loudly_without_device(fv):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert fv.lib().fvad_ctx_create(0, C.byref(h)) == fv.FVAD_ERR_NO_DEVICE
    with pytest.raises(fv.FvadError):
        fv.Context(0)


def test_bench_gpus_n_without_enough_gpus_is_refused_before_any_rank_starts():
    # `python bench.py --gpus 2` starts its ranks itself; on a box that does not show 2 GPUs (this container shows none)
    # the launcher refuses with exit code 2 instead of running one rank and printing n_gpus: 1
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 2 and "needs 2 GPUs" in r.stderr and r.stdout.strip() == ""
    # under a launcher that started another number of ranks than --gpus says: refused too
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=120,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 2 and "rank" in r.stderr


@pytest.mark.parametrize("unit", ["kernels_b3.hip", "kernels_h3.hip", "kernels_nn.hip", "kernels_ws.hip"])
def test_kernels_with_inline_lds_reads_compile_without_spills(unit):
    # The persistent GEMMs and the LDS-DMA recurrences read weight fragments with inline ds_read_b128 / global_load_lds
    # that the compiler knows nothing about: a spilled register there is not slow but unsafe (its in-flight read lands
    # in whatever the compiler has meanwhile put into it: DESIGN.md section 3.0b).  Every kernel of these translation units
    # must compile for gfx950 with zero spilled registers and no scratch.
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "formula-vad_amd", "csrc", unit)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    spills = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", r.stderr)]
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert names and len(names) == len(spills) == len(scratch)
    bad = [(n, sp, sc) for n, sp, sc in zip(names, spills, scratch) if sp or sc]
    assert not bad, bad


def test_windows_bit_equal_oracle(fv):
    L = fv.lib()
    w = np.zeros(320, np.float32)
    L.fvad_nsnet2_window(fv.fptr(w))
    assert np.array_equal(w, orc.nsnet2_window())
    wp = np.zeros(1024, np.float32)
    L.fvad_hann_window_periodic(fv.fptr(wp), 1024)
    assert np.array_equal(wp, orc.hann_periodic(1024))
    ws = np.zeros(77, np.float32)
    L.fvad_hann_window_symmetric(fv.fptr(ws), 77)
    assert np.array_equal(ws, orc.hann_symmetric(77))
    assert L.fvad_window_norm_factor(fv.fptr(wp), 1024) == orc.lib().orc_window_norm_factor(orc.fptr(wp), 1024)


def test_synth_weights_deterministic_and_responsive(fv, weights7):
    again = fv.synth_weights(7)
    other = fv.synth_weights(8)
    for k in fv.WEIGHT_NAMES:
        assert np.array_equal(weights7[k], again[k])
    assert not np.array_equal(weights7["fc1_w"], other["fc1_w"])
    assert weights7["gru1_w"].shape == (1200, 400) and weights7["fc4_w"].shape == (161, 600)
    # the synthetic net must discriminate: gains spread over (0,1), not all ~0.5
    rng = np.random.default_rng(0)
    f = rng.uniform(-9, 1, (54, 161)).astype(np.float32)
    g = orc.nsnet2_forward(weights7, f)
    assert g.std() > 0.15 and g.min() < 0.2 and g.max() > 0.8


def test_rolling_average_bit_exact(fv):
    L, O = fv.lib(), orc.lib()
    rng = np.random.default_rng(1)
    for count, init in ((9, None), (23, None), (8437, 0.005)):
        h = C.c_void_p()
        assert L.fvad_ra_create(count, 0 if init is None else 1, init or 0.0, C.byref(h)) == 0
        o = O.orc_ra_create(count, 0 if init is None else 1, init or 0.0)
        for _ in range(40 if count > 100 else 3 * count):
            s = C.c_float(rng.uniform(0, 0.2))
            assert L.fvad_ra_push(h, s) == O.orc_ra_push(o, s)
        a, b = C.c_double(), C.c_double()
        assert L.fvad_ra_last_avg(h, C.byref(a)) == O.orc_ra_last_avg(o, C.byref(b)) == 1
        assert a.value == b.value
        L.fvad_ra_destroy(h)
        O.orc_ra_destroy(o)


def _random_band_script(rng, n, C_):
    """band volumes that open/close the VAD several times, incl. near-threshold stretches"""
    v = np.full((n, C_), 0.002, np.float32) + rng.uniform(0, 0.002, (n, C_)).astype(np.float32)
    pos = 60
    while pos < n - 200:
        ln = int(rng.integers(3, 400))
        lvl = rng.choice([0.03, 0.051, 0.2, 1.0])
        v[pos:pos + ln] += np.float32(lvl) * rng.uniform(0.5, 1.0, (min(ln, n - pos), C_)).astype(np.float32)
        pos += ln + int(rng.integers(10, 300))
    ratio = rng.uniform(0.3, 1.0, n).astype(np.float32)
    return v, ratio


def _oracle_vad(band, ratio, overrides=None):
    O = orc.lib()
    cfg = orc.VadConfig()
    O.orc_vad_config_default(C.byref(cfg))
    for k, val in (overrides or {}).items():
        setattr(cfg, k, val)
    v = O.orc_vad_create(C.byref(cfg), 48000, band.shape[1], 1024)
    ev = []
    for k in range(band.shape[0]):
        row = np.ascontiguousarray(band[k])
        has = 0 if np.isnan(ratio[k]) else 1
        r = O.orc_vad_run(v, 1024 * k, orc.fptr(row), has, C.c_float(0.0 if not has else ratio[k]))
        ev.append((r.recording_state, r.sample_number))
    n = O.orc_vad_n_segments(v)
    p = O.orc_vad_segments(v)
    segs = [(p[i].sample_from, p[i].sample_to, p[i].avg_channel_vol_ratio, p[i].vad_met_sec) for i in range(n)]
    O.orc_vad_destroy(v)
    return ev, segs


@pytest.mark.parametrize("n_channels", [1, 2])
def test_vad_machine_identical_to_oracle(fv, n_channels):
    rng = np.random.default_rng(10 + n_channels)
    band, ratio = _random_band_script(rng, 3000, n_channels)
    ratio[5] = np.nan  # a null volume_ratio (orelse 0)
    ev_o, segs_o = _oracle_vad(band, ratio)
    m = fv.VadMachine(n_channels=n_channels)
    ev = [m.run(1024 * k, band[k], None if np.isnan(ratio[k]) else float(ratio[k])) for k in range(band.shape[0])]
    assert ev == ev_o
    assert m.segments() == segs_o and len(segs_o) >= 3
    thr_margin, ratio_margin, n = m.audit()
    assert n == band.shape[0] and thr_margin >= 0 and ratio_margin >= 0
    # non-default config: no initial long-term average, other time constants
    ov = {"has_initial_long_term_avg": 0, "long_term_speech_avg_sec": 3.0, "max_speech_gap_sec": 0.5,
          "min_vad_duration_sec": 0.3, "speech_threshold_factor": 4.0}
    ev_o, segs_o = _oracle_vad(band, ratio, ov)
    m2 = fv.VadMachine(n_channels=n_channels, overrides=ov)
    ev = [m2.run(1024 * k, band[k], None if np.isnan(ratio[k]) else float(ratio[k])) for k in range(band.shape[0])]
    assert ev == ev_o and m2.segments() == segs_o


def test_vad_lazy_long_term_average_equals_eager(fv, monkeypatch):
    # The long-term chain (8437 dependent f64 adds) is evaluated lazily: only when a rigorous bound on the
    # incrementally carried value cannot settle `short_term > threshold` or could lower the audited minimum
    # margin.  Events, segments and the audit must equal the eager evaluation bit for bit, on a long script
    # whose level wanders through four decades (so that anchor-time and current magnitudes differ) and on
    # one that hugs the threshold.
    rng = np.random.default_rng(77)
    n = 40000
    level = 10 ** (np.cumsum(rng.normal(0, 0.02, n)) % 4 - 4)
    band = (level * rng.uniform(0.5, 1.5, n)).astype(np.float32)[:, None]
    burst = np.zeros(n, bool)
    for s in rng.integers(0, n - 400, 120):
        burst[s:s + int(rng.integers(5, 300))] = True
    band[burst] *= np.float32(30)
    # frames that sit within a hair of 10 x the running level: the comparison is nearly tied
    near = rng.random(n) < 0.05
    band[near, 0] = (level[near] * 10 * (1 + rng.normal(0, 1e-7, near.sum()))).astype(np.float32)
    ratio = rng.uniform(0.3, 1.0, n).astype(np.float32)

    def run():
        m = fv.VadMachine(n_channels=1)
        ev = [m.run(1024 * k, band[k], float(ratio[k])) for k in range(n)]
        out = (ev, m.segments(), m.audit(), m.lazy_stats())
        m.close()
        return out

    ev_l, seg_l, audit_l, (exact_l, lazy_l) = run()
    monkeypatch.setenv("FVAD_VAD_EAGER", "1")
    ev_e, seg_e, audit_e, (exact_e, lazy_e) = run()
    assert ev_l == ev_e and seg_l == seg_e and len(seg_e) >= 5
    assert audit_l == audit_e
    assert lazy_e == 0 and lazy_l > 10000
    assert exact_l < lazy_l / 20, (exact_l, lazy_l)     # the chain ran for a few percent of the pushes at most


def test_vad_run_many_bit_identical_to_scalar(fv):
    # lock-step multi-stream driver (f64 re-sum vectorised across streams) == per-stream runs
    rng = np.random.default_rng(20)
    n_streams = 11  # one full group of 8 + a ragged group of 3
    bands, ratios = [], []
    for s in range(n_streams):
        b, r = _random_band_script(rng, 1500 + 37 * s, 1)
        bands.append(b)
        ratios.append(r)
    ms = [fv.VadMachine() for _ in range(n_streams)]
    fv.vad_run_many(ms, bands, ratios, n_threads=2)
    for s in range(n_streams):
        _, segs_o = _oracle_vad(bands[s], ratios[s])
        assert ms[s].segments() == segs_o
    # a second call continues from the carried state (rings partially replaced)
    more = [_random_band_script(rng, 700, 1) for _ in range(n_streams)]
    fv.vad_run_many(ms, [m[0] for m in more], [m[1] for m in more],
                    first_index=[1024 * bands[s].shape[0] for s in range(n_streams)], n_threads=3)
    for s in range(n_streams):
        _, segs_o = _oracle_vad(np.concatenate([bands[s], more[s][0]]), np.concatenate([ratios[s], more[s][1]]))
        assert ms[s].segments() == segs_o


def test_statistics_match_oracle_and_literals(fv):
    O = orc.lib()
    rng = np.random.default_rng(30)
    cfgd = {"ignore_shorter_than_sec": 0.7, "extrude_start": 5.0, "extrude_end": 10.0, "fill_gaps": 5.0}
    ocfg = orc.StatConfig(0.7, 5.0, 10.0, 5.0)
    singles_f, singles_o = [], []
    for trial in range(20):
        def segs(n):
            t = np.sort(rng.uniform(0, 600, 2 * n)).astype(np.float32)
            out = [(float(t[2 * i]), float(t[2 * i + 1])) for i in range(n)]
            rng.shuffle(out)  # initAndRun sorts by start
            return out
        vad, ref = segs(int(rng.integers(1, 12))), segs(int(rng.integers(1, 12)))
        s = fv.stats_from_segments(vad, ref, cfgd)
        ov = (orc.SegSec * len(vad))(*[orc.SegSec(a, b) for a, b in vad])
        orf = (orc.SegSec * len(ref))(*[orc.SegSec(a, b) for a, b in ref])
        o = O.orc_stats_from_segments(ov, len(vad), orf, len(ref), C.byref(ocfg))
        for name, _ in fv.SingleStats._fields_:
            a, b = getattr(s, name), getattr(o, name)
            assert a == b or (np.isnan(a) and np.isnan(b)), (trial, name, a, b)
        singles_f.append(s)
        singles_o.append(o)
    agg = fv.stats_aggregate(singles_f)
    oagg = O.orc_stats_aggregate((orc.SingleStats * 20)(*singles_o), 20)
    for name in ("total_positives_sec", "true_positives_sec", "false_positives_sec", "false_negatives_sec",
                 "fm_index", "f_score", "f_score_beta"):
        assert getattr(agg, name) == getattr(oagg, name)
    for name in ("true_positive_rate", "false_negative_rate", "false_discovery_rate", "precision"):
        for f in ("overall", "min", "max", "avg"):
            assert getattr(getattr(agg, name), f) == getattr(getattr(oagg, name), f)
    # the reference's own unit cases (statistics.zig:286-360) through the public entry point:
    # vad [1,6] vs refs [2,3],[4,5] with extrude 2/2 and fill 2 -> FP 0; vad [1,10] -> FP 3
    lit = {"extrude_start": 2.0, "extrude_end": 2.0, "fill_gaps": 2.0}
    assert abs(fv.stats_from_segments([(1, 6)], [(2, 3), (4, 5)], lit).false_positives_sec - 0.0) < 1e-3
    assert abs(fv.stats_from_segments([(1, 10)], [(2, 3), (4, 5)], lit).false_positives_sec - 3.0) < 1e-3


def test_segment_to_sec_and_audacity_parse(fv):
    L = fv.lib()
    s = fv.SpeechSegment(123456789, 223456789, 0.0, 0.0)
    r = L.fvad_segment_to_sec(C.byref(s), 48000)
    assert r.from_sec == np.float32(123456789) / np.float32(48000)  # u64 -> f32, then f32 divide
    txt = b"1.5000\t2.2500\tspeech\r\n10.0\t12.5\n\nnot a label line\n3\t4\tx\ty\n"
    out = (fv.SegmentSec * 8)()
    n = C.c_size_t()
    assert L.fvad_parse_audacity(txt, len(txt), out, 8, C.byref(n)) == 0
    assert [(out[i].from_sec, out[i].to_sec) for i in range(n.value)] == [(1.5, 2.25), (10.0, 12.5), (3.0, 4.0)]
    bad = b"abc\tdef\n"
    assert L.fvad_parse_audacity(bad, len(bad), out, 8, C.byref(n)) != 0  # parseFloat error
    from formula_vad_amd import synth
    lab = [(1.0, 2.5), (7.25, 9.0)]
    enc = synth.labels_to_audacity(lab).encode()
    assert L.fvad_parse_audacity(enc, len(enc), out, 8, C.byref(n)) == 0 and n.value == 2


# ------------------------------------------------------------------ ONNX reader
def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _vi(field, v):
    return _varint((field << 3) | 0) + _varint(v)


def _tensor(name, arr, raw=True):
    arr = np.ascontiguousarray(arr, np.float32)
    t = b"".join(_vi(1, d) for d in arr.shape) + _vi(2, 1)
    if raw:
        t += _ld(9, arr.tobytes())
    else:
        t += _ld(4, arr.tobytes())  # packed float_data
    return t + _ld(8, name.encode())


def _node(op, ins, outs, iattrs=None):
    n = b"".join(_ld(1, i.encode()) for i in ins) + b"".join(_ld(2, o.encode()) for o in outs) + _ld(4, op.encode())
    for k, v in (iattrs or {}).items():
        n += _ld(5, _ld(1, k.encode()) + _vi(3, v) + _vi(20, 2))
    return n


def _make_onnx(wd, gemm=False, h0=None, const_nodes=False):
    """hand-encode the NSNet2-baseline graph (MatMul+Add / GRU with linear_before_reset) as protobuf.
    h0: None = GRU nodes without initial_h; "computed" = a zeros tensor built by shape ops (what PyTorch
    exports for nn.GRU without h0); "zeros" / "nonzero" = an initializer; "input" = a graph input."""
    inits, nodes, init_names = [], [], []

    def add_init(name, arr, raw=True):
        inits.append(_tensor(name, arr, raw))
        init_names.append(name)

    def dense(i, x, y, w, b, act=None):
        if gemm:
            add_init(f"w{i}", wd[w])  # [out][in], transB=1
            add_init(f"b{i}", wd[b], raw=False)
            nodes.append(_node("Gemm", [x, f"w{i}", f"b{i}"], [f"d{i}"], {"transB": 1}))
        else:
            add_init(f"w{i}", wd[w].T)  # MatMul B operand is [in][out]
            add_init(f"b{i}", wd[b])
            nodes.append(_node("MatMul", [x, f"w{i}"], [f"m{i}"]))
            nodes.append(_node("Add", [f"b{i}", f"m{i}"], [f"d{i}"]))
        if act:
            nodes.append(_node(act, [f"d{i}"], [y]))
        return y if act else f"d{i}"

    x = dense(1, "input", None, "fc1_w", "fc1_b")
    nodes.append(_node("Transpose", [x], ["t0"]))
    cur = "t0"
    for g in (1, 2):
        add_init(f"W{g}", wd[f"gru{g}_w"][None])
        add_init(f"R{g}", wd[f"gru{g}_r"][None])
        add_init(f"B{g}", wd[f"gru{g}_b"][None])
        gru_in = [cur, f"W{g}", f"R{g}", f"B{g}"]
        if h0 == "computed":
            nodes.append(_node("Shape", [cur], [f"shp{g}"]))
            nodes.append(_node("ConstantOfShape", [f"shp{g}"], [f"h0_{g}"]))
            gru_in += ["", f"h0_{g}"]
        elif h0 in ("zeros", "nonzero"):
            add_init(f"h0_{g}", np.full((1, 1, 400), 0.0 if h0 == "zeros" else 0.25, np.float32))
            gru_in += ["", f"h0_{g}"]
        elif h0 == "input":
            gru_in += ["", f"h0_{g}"]
        nodes.append(_node("GRU", gru_in, [f"y{g}", f"h{g}"],
                           {"hidden_size": 400, "linear_before_reset": 1}))
        nodes.append(_node("Squeeze", [f"y{g}"], [f"s{g}"]))
        cur = f"s{g}"
    nodes.append(_node("Transpose", [cur], ["t1"]))
    x = dense(2, "t1", "r2", "fc2_w", "fc2_b", "Relu")
    x = dense(3, x, "r3", "fc3_w", "fc3_b", "Relu")
    dense(4, x, "output", "fc4_w", "fc4_b", "Sigmoid")
    if const_nodes:  # every weight as a Constant node (AttributeProto{name="value", t=tensor, type=TENSOR})
        consts = []
        for name, t in zip(init_names, inits):
            consts.append(_ld(2, name.encode()) + _ld(4, b"Constant") + _ld(5, _ld(1, b"value") + _ld(5, t) + _vi(20, 4)))
        nodes = consts + nodes
        inits = []
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"nsnet2") + b"".join(_ld(5, t) for t in inits)
    graph += _ld(11, _ld(1, b"input"))                       # GraphProto.input: ValueInfoProto{name}
    if h0 == "input":
        graph += _ld(11, _ld(1, b"h0_1")) + _ld(11, _ld(1, b"h0_2"))
    return _vi(1, 7) + _ld(2, b"pytorch") + _ld(7, graph)


@pytest.mark.parametrize("gemm", [False, True])
def test_onnx_reader_roundtrip(fv, weights7, tmp_path, gemm):
    path = tmp_path / "nsnet2-20ms-baseline.onnx"
    path.write_bytes(_make_onnx(weights7, gemm))
    got = fv.read_onnx(str(path))
    for k in fv.WEIGHT_NAMES:
        assert np.array_equal(got[k], weights7[k]), k


@pytest.mark.parametrize("dims", [(400, 400, 600, 600), (96, 72, 200, 136), (64, 48, 64, 64)], ids=lambda d: "x".join(map(str, d)))
def test_onnx_reader_on_a_file_written_by_pytorchs_exporter(fv, tmp_path, dims):
    # the reader against a graph a real exporter wrote (tests/torch_export.py: torch.onnx.export of an nn.Module of
    # the NSNet2 architecture): MatMul + Add with transposed initializers, GRU W / R / B in ONNX gate order, PyTorch's
    # node and initializer names; square layers (64 x 64) cannot be told apart by shape.  The weights it returns,
    # run through the oracle, must reproduce torch's own forward pass.
    from torch_export import export_in_subprocess
    path = str(tmp_path / "exported.onnx")
    x, y = export_in_subprocess(path, dims, seed=5)
    w = fv.read_onnx(path)
    assert w["fc1_w"].shape == (dims[0], 161) and w["gru1_w"].shape == (3 * dims[1], dims[0])
    assert w["gru2_r"].shape == (3 * dims[1], dims[1]) and w["gru1_b"].shape == (6 * dims[1],)
    assert w["fc3_w"].shape == (dims[3], dims[2]) and w["fc4_w"].shape == (161, dims[3])
    g = orc.nsnet2_forward(w, x[0])
    assert np.abs(g - y[0]).max() <= 2e-6, np.abs(g - y[0]).max()
    assert y.min() < 0.2 and y.max() > 0.8             # the module is not stuck around 0.5


def test_onnx_reader_weights_as_constant_nodes(fv, weights7, tmp_path):
    path = tmp_path / "const.onnx"
    path.write_bytes(_make_onnx(weights7, const_nodes=True))
    got = fv.read_onnx(str(path))
    for k in fv.WEIGHT_NAMES:
        assert np.array_equal(got[k], weights7[k]), k


@pytest.mark.parametrize("h0,ok", [("computed", True), ("zeros", True), ("nonzero", False), ("input", False)])
def test_onnx_reader_initial_state_variants(fv, weights7, tmp_path, h0, ok):
    # the reference feeds the session one tensor and no state (NSNet2.zig:57-58): a constant zero
    # initial_h (as PyTorch exports it) is the same model; a non-zero or externally fed state is not
    path = tmp_path / f"h0_{h0}.onnx"
    path.write_bytes(_make_onnx(weights7, h0=h0))
    if ok:
        got = fv.read_onnx(str(path))
        for k in fv.WEIGHT_NAMES:
            assert np.array_equal(got[k], weights7[k]), k
    else:
        L = fv.lib()
        w, owner = fv.Weights(), C.c_void_p()
        assert L.fvad_onnx_read_nsnet2(str(path).encode(), C.byref(w), C.byref(owner)) == -104


def test_onnx_reader_errors(fv, weights7, tmp_path):
    L = fv.lib()
    w, owner = fv.Weights(), C.c_void_p()
    assert L.fvad_onnx_read_nsnet2(str(tmp_path / "missing.onnx").encode(), C.byref(w), C.byref(owner)) == -105
    p = tmp_path / "garbage.onnx"
    p.write_bytes(b"\x00\x01\x02not a protobuf")
    assert L.fvad_onnx_read_nsnet2(str(p).encode(), C.byref(w), C.byref(owner)) == -104
    blob = _make_onnx(weights7)
    p2 = tmp_path / "trunc.onnx"
    p2.write_bytes(blob[: len(blob) // 2])
    assert L.fvad_onnx_read_nsnet2(str(p2).encode(), C.byref(w), C.byref(owner)) == -104


# ------------------------------------------------------------------ sharding (N > 1 path)
def test_round_robin_partition(pkg):
    sh = pkg.shard
    parts = [sh.streams_for_rank(21, r, 8) for r in range(8)]
    assert [len(p) for p in parts] == [3, 3, 3, 3, 3, 2, 2, 2]
    assert sorted(sum(parts, [])) == list(range(21))


_WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package()
import torch.distributed as dist
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{PORT}", rank=RANK, world_size=2)
n_streams = 5
ids = pkg.shard.streams_for_rank(n_streams, RANK, 2)
stats = [np.arange(11, dtype=np.float32) + 100 * i for i in ids]
allst = pkg.shard.gather_stats(ids, stats, n_streams, dist=dist)
assert allst.shape == (5, 11)
for i in range(n_streams):
    assert np.array_equal(allst[i], np.arange(11, dtype=np.float32) + 100 * i), (RANK, i)
dist.barrier()
dist.destroy_process_group()
print("rank", RANK, "ok")
"""


def test_gather_stats_two_ranks_gloo(tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        code = f"ROOT={ROOT!r}\nPORT={port}\nRANK={rank}\n" + _WORKER
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()
        assert f"rank {rank} ok" in out.decode()


_WORKER_CFG4 = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(PORT), RANK=str(RANK), WORLD_SIZE=str(WORLD))
dist.init_process_group("gloo")
pkg = load_package()
fv = pkg.binding
n_streams = 21
# every rank can build every stream's statistics (seeded); it contributes only its own
rng = np.random.default_rng(5)
cfg = {"ignore_shorter_than_sec": 0.7, "extrude_start": 5.0, "extrude_end": 10.0, "fill_gaps": 5.0}
every = []
for i in range(n_streams):
    t = np.sort(rng.uniform(0, 7200, 80)).astype(np.float32)
    ref = [(float(t[2 * k]), float(t[2 * k + 1])) for k in range(40)]
    vad = [(a + float(rng.normal(0, 0.3)), b + float(rng.normal(0, 0.5))) for a, b in ref if rng.uniform() > 0.1]
    every.append(fv.stats_from_segments(vad, ref, cfg))
ids = pkg.shard.streams_for_rank(n_streams, RANK, WORLD)
assert len(ids) == (3 if RANK < 5 else 2)                       # 3,3,3,3,3,2,2,2
allst = pkg.shard.gather_stats(ids, [fv.single_stats_to_array(every[i]) for i in ids], n_streams, dist=dist)
got = fv.stats_aggregate([fv.array_to_single_stats(a) for a in allst])
want = fv.stats_aggregate(every)                                # the single-process aggregate, plan order
assert bytes(got) == bytes(want), RANK
assert all(np.array_equal(allst[i], fv.single_stats_to_array(every[i])) for i in range(n_streams))
dist.barrier()
dist.destroy_process_group()
print("rank", RANK, "ok")
"""


def test_cfg4_uneven_21_stream_gather_equals_single_process_aggregate():
    # BASELINE config 4's deal: 21 streams over 8 ranks = 3,3,3,3,3,2,2,2.  The gathered, plan-ordered statistics
    # must aggregate to the same bytes as statistics.aggregate over all 21 in one process (statistics.zig:116-172:
    # in-order f32 sums)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 8
    procs = []
    for rank in range(world):
        code = f"ROOT={ROOT!r}\nPORT={port}\nRANK={rank}\nWORLD={world}\n" + _WORKER_CFG4
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()
        assert f"rank {rank} ok" in out.decode()


def test_wav_read_i16_keeps_the_samples(fv, tmp_path):
    # the 16-bit transport's file side: PCM16 samples come back untouched and planar, and decode to what
    # fvad_wav_read returns; a float file is refused
    import struct
    rng = np.random.default_rng(2)
    s16 = rng.integers(-32768, 32768, (2, 1000)).astype(np.int16)
    path = str(tmp_path / "a.wav")
    data = np.ascontiguousarray(s16.T).tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, 48000, 48000 * 4, 4, 16)
                + b"data" + struct.pack("<I", len(data)) + data)
    got, sr = fv.wav_read_i16(path)
    assert sr == 48000 and got.dtype == np.int16 and np.array_equal(got, s16)
    f32, _ = fv.wav_read(path)
    assert np.array_equal(f32, s16.astype(np.float32) * np.float32(1.0 / 32768.0))
    with open(path, "wb") as f:
        d = np.zeros(8, np.float32).tobytes()
        f.write(b"RIFF" + struct.pack("<I", 36 + len(d)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 1, 48000, 48000 * 4, 4, 32)
                + b"data" + struct.pack("<I", len(d)) + d)
    with pytest.raises(fv.FvadError):
        fv.wav_read_i16(path)


def test_split_stream_ranges(pkg):
    # time-split sharding (BASELINE config 5): contiguous balanced chunk ranges, each rank's job starts two
    # chunks early and runs one chunk late, clipped to the stream
    sh = pkg.shard
    assert sh.split_stream(65, 3) == [(0, 22), (22, 44), (44, 65)]
    assert sh.split_stream(7, 8) == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 7)]
    for n, w in ((72000, 8), (10, 3), (1, 1)):
        r = sh.split_stream(n, w)
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert max(c1 - c0 for c0, c1 in r) - min(c1 - c0 for c0, c1 in r) <= 1
    assert sh.time_split_job(65, 0, 22) == (0, 23) and sh.time_split_job(65, 22, 44) == (20, 45)
    assert sh.time_split_job(65, 44, 65) == (42, 65) and sh.time_split_job(65, 1, 5) == (0, 6)


def test_new_abi_argument_errors_without_a_device(fv):
    # entry points added in ABI version 2 reject bad arguments before touching a device
    L = fv.lib()
    assert L.fvad_comm_unique_id(None, 128) == fv.FVAD_ERR_INVALID_ARGUMENT
    buf = (C.c_uint8 * 64)()
    assert L.fvad_comm_unique_id(buf, 64) == fv.FVAD_ERR_INVALID_ARGUMENT           # needs 128 bytes
    h = C.c_void_p()
    big = (C.c_uint8 * 128)()
    assert L.fvad_comm_create(None, big, 128, 1, 0, C.byref(h)) == fv.FVAD_ERR_INVALID_ARGUMENT
    assert L.fvad_stats_allgather(None, None, None, 0, 1, None) == fv.FVAD_ERR_INVALID_ARGUMENT
    assert L.fvad_comm_world(None) == 0 and L.fvad_comm_rank(None) == -1
    assert L.fvad_lane_state_seek(None, 0, 0) == fv.FVAD_ERR_INVALID_ARGUMENT
    assert L.fvad_device_alloc(None, 16, C.byref(h)) == fv.FVAD_ERR_INVALID_ARGUMENT
    assert L.fvad_pipeline_enable_trace(None, 1) == fv.FVAD_ERR_INVALID_ARGUMENT
    assert L.fvad_engine_enqueue_device_i16(None, None, 1, 8, 24000, None, None, None, None) == fv.FVAD_ERR_INVALID_ARGUMENT


def test_vad_batch_equals_oracle_pipeline(fv, pkg):
    # fvad_vad_batch: per-chunk ratio -> metadata hand-overs -> per-frame ratio -> state machines, for several
    # streams at once from lane-major band sums / chunk RMS.  Fed with the oracle pipeline's own band sums and RMS
    # it must give the oracle's segments, mono and stereo
    W = fv.synth_weights(7)
    for nch, seeds in ((1, (40, 44)), (2, (41,))):
        bands, rmss, want = [], [], []
        for seed in seeds:
            pcm, _ = pkg.synth.make_stream(80.0, seed=seed, n_channels=nch)
            p = orc.Pipeline(W, n_channels=nch)
            p.push(pcm)
            bands.append(p.band_volumes().T.copy())      # [channel][frame]
            rmss.append(p.chunk_rms().T.copy())
            want.append([(s[0], s[1], s[2], s[3]) for s in p.segments()])
        b = fv.VadBatch(len(seeds), n_channels=nch)
        got = b.run(np.ascontiguousarray(np.concatenate(bands)), np.ascontiguousarray(np.concatenate(rmss)), n_threads=2)
        assert sum(len(w) for w in want) >= 2
        for g, w in zip(got, want):
            assert [(x[0], x[1]) for x in g] == [(x[0], x[1]) for x in w]
            assert all(abs(x[2] - y[2]) <= 1e-6 and x[3] == y[3] for x, y in zip(g, w))
        assert b.audit(0)[2] == bands[0].shape[1]
        b.close()
    # a frame beyond the chunks that carry its ratio is refused
    b = fv.VadBatch(1)
    with pytest.raises(fv.FvadError):
        b.run(np.zeros((1, 100), np.float32), np.ones((1, 1), np.float32))
    b.close()


def test_vad_batch_in_parts_equals_one_run(fv):
    # fvad_vad_batch_run_part: the machines live on between the parts, so a host can run the VAD of what it has while the GPU
    # produces the rest.  Random part boundaries on the common grid of chunks and frames (every 375 frames = 16 chunks), mono and
    # stereo, speech bursts that straddle boundaries: segments and the margin audit equal one run over everything, bit for bit
    rng = np.random.default_rng(12)
    for nch in (1, 2):
        n_streams, n_units = 3, 40                                   # 40 x 16 chunks = 320 s
        n_frames, n_chunks = n_units * 375, n_units * 16
        band = (0.004 + 0.002 * rng.random((n_streams * nch, n_frames))).astype(np.float32)
        for s in range(n_streams * nch):                             # bursts of "speech": the band sum jumps for 1..6 s
            t = 200
            while t < n_frames - 400:
                d = int(rng.integers(47, 280))
                band[s, t:t + d] += np.float32(0.05) * rng.random(d).astype(np.float32) + np.float32(0.03)
                t += d + int(rng.integers(150, 900))
        rms = (0.01 + 0.05 * rng.random((n_streams * nch, n_chunks))).astype(np.float32)
        whole = fv.VadBatch(n_streams, n_channels=nch)
        want = whole.run(band, rms, n_threads=2)
        assert sum(len(w) for w in want) >= 3
        for trial in range(3):
            cuts = sorted(set(int(x) for x in rng.integers(1, n_units, 5)))
            edges = [0] + cuts + [n_units]
            parts = fv.VadBatch(n_streams, n_channels=nch)
            got = None
            for u0, u1 in zip(edges, edges[1:]):
                got = parts.run_part(band[:, u0 * 375: u1 * 375], rms[:, u0 * 16: u1 * 16], u0 * 375, n_threads=1 + trial)
            assert got == want, (nch, edges)
            for s in range(n_streams):
                assert parts.audit(s) == whole.audit(s)
            # a part that does not follow the previous one, or that starts off the chunk grid, is refused
            with pytest.raises(fv.FvadError):
                parts.run_part(band[:, 375:750], rms[:, 16:32], 375)
            parts.close()
        off = fv.VadBatch(n_streams, n_channels=nch)
        off.run_part(band[:, :100], rms[:, :5], 0)
        with pytest.raises(fv.FvadError):
            off.run_part(band[:, 100:200], rms[:, 4:9], 100)         # 100 frames do not end on a chunk boundary
        with pytest.raises(fv.FvadError):
            off.run_part(band[:, :375], rms[:, :15], 0)              # a frame without its chunk's ratio
        off.close()
        # ... and run() after parts starts from fresh machines again
        again = fv.VadBatch(n_streams, n_channels=nch)
        again.run_part(band[:, : 375 * 7], rms[:, : 16 * 7], 0)
        assert again.run(band, rms) == want
        again.close()
        whole.close()


def test_wav_write_round_trips(fv, tmp_path):
    # AudioBuffer.saveToFile for WAV: float32 is lossless; PCM16 is lrintf(clip(x) * 32767) and comes back as s / 32768
    rng = np.random.default_rng(4)
    x = rng.uniform(-1.2, 1.2, (2, 777)).astype(np.float32)
    p32, p16 = str(tmp_path / "f32.wav"), str(tmp_path / "i16.wav")
    fv.wav_write(p32, x, 16000)
    got, sr = fv.wav_read(p32)
    assert sr == 16000 and np.array_equal(got, x)
    fv.wav_write(p16, x, 48000, pcm16=True)
    s16, sr = fv.wav_read_i16(p16)
    assert sr == 48000 and np.array_equal(s16, np.rint(np.clip(x, -1, 1) * np.float32(32767.0)).astype(np.int16))
    fv.wav_write(p16, np.zeros((1, 0), np.float32))               # an empty clip is a valid file
    assert fv.wav_read(p16)[0].shape == (1, 0)
    with pytest.raises(fv.FvadError):
        fv.wav_write(str(tmp_path / "no" / "dir.wav"), x)


def test_golden_vad_stream_segments(fv):
    # committed band volumes of a 120 s synthetic stream -> the exact segment list
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_vad_seed40.npz"))
    m = fv.VadMachine()
    for k in range(g["band"].shape[0]):
        m.run(1024 * k, g["band"][k], float(g["ratio"][k]))
    segs = m.segments()
    assert [(s[0], s[1]) for s in segs] == [tuple(int(v) for v in r) for r in g["segments"]]
    assert np.array_equal(np.array([s[2] for s in segs], np.float32), g["seg_ratio"])
    assert np.array_equal(np.array([s[3] for s in segs], np.float32), g["seg_met"])
    assert len(segs) == 9


# ------------------------------------------------------------------ the one numeric fixture the reference holds for a16 / f3
# /root/reference/README.md:28-61: the performance report of the reference's own 21-stream run (per-stream rows as
# printed: whole seconds, rates to 0.1 %) and its aggregate block.  It pins statistics.aggregate's formulas
# (statistics.zig:116-182: in-order sums, min / avg / max of the four rates, overall rates, f_score(0.7), fm_index) and
# report_generator.zig:21-116's layout at printed precision.  (P, TP, FP, FN; TPR, PPV, FNR, FDR as printed.)
README_ROWS = [
    ("2023 Monaco FP1 - Perez", 1137, 1135, 5, 2, 99.8, 99.6, 0.2, 0.4),
    ("2023 Miami Race - Sargeant", 1092, 1075, 6, 17, 98.4, 99.4, 1.6, 0.6),
    ("2023 Miami Race - Gasly", 1447, 1362, 23, 86, 94.1, 98.3, 5.9, 1.7),
    ("2023 Miami Race - Perez", 1025, 996, 5, 29, 97.2, 99.5, 2.8, 0.5),
    ("2023 Miami Race - Leclerc", 1222, 1222, 0, 0, 100.0, 100.0, 0.0, 0.0),
    ("2023 Miami Race - De Vries", 952, 940, 0, 12, 98.7, 100.0, 1.3, 0.0),
    ("2023 Miami Race - Zhou", 1082, 1070, 11, 12, 98.9, 99.0, 1.1, 1.0),
    ("2023 Miami Race - Magnussen", 1028, 1020, 6, 7, 99.3, 99.5, 0.7, 0.5),
    ("2023 Miami Race - Russell", 1435, 1398, 8, 37, 97.4, 99.4, 2.6, 0.6),
    ("2023 Miami Race - Norris", 513, 512, 0, 1, 99.8, 100.0, 0.2, 0.0),
    ("2023 Miami Race - Stroll", 1114, 1108, 0, 6, 99.5, 100.0, 0.5, 0.0),
    ("2023 Miami Race - Tsunoda", 671, 664, 0, 6, 99.1, 100.0, 0.9, 0.0),
    ("2023 Miami Race - Verstappen", 1049, 1039, 0, 10, 99.0, 100.0, 1.0, 0.0),
    ("2023 Miami Race - Sainz", 1447, 1436, 8, 11, 99.3, 99.4, 0.7, 0.6),
    ("2023 Miami Race - Albon", 561, 547, 0, 14, 97.5, 100.0, 2.5, 0.0),
    ("2023 Miami Race - Hulkenberg", 617, 617, 18, 0, 100.0, 97.2, 0.0, 2.8),
    ("2023 Miami Race - Ocon", 597, 594, 14, 3, 99.5, 97.7, 0.5, 2.3),
    ("2023 Miami Race - Hamilton", 1261, 1233, 10, 28, 97.8, 99.2, 2.2, 0.8),
    ("2023 Miami Race - Alonso", 1172, 1154, 0, 18, 98.4, 100.0, 1.6, 0.0),
    ("2023 Miami Race - Bottas", 575, 573, 0, 2, 99.6, 100.0, 0.4, 0.0),
    ("2023 Miami Race - Piastri", 822, 782, 0, 40, 95.1, 100.0, 4.9, 0.0),
]
README_AGGREGATE = """
=> Aggregate stats 

Total speech duration  (P): 20822.3 sec
True positives        (TP): 20480.1 sec
False positives       (FP):   113.3 sec
False negatives       (FN):   342.2 sec    Min.    Avg.    Max. 
True positive rate   (TPR):    98.4%  |   94.1% / 98.5% /100.0% 
Precision            (PPV):    99.4%  |   97.2% / 99.4% /100.0% 
False negative rate  (FNR):     1.6%  |    0.0% /  1.5% /  5.9% 
False discovery rate (FDR):     0.6%  |    0.0% /  0.6% /  2.8% 
F-Score (β =  0.70)       :    99.1% 
Fowlkes-Mallows index     :    98.9% 
"""


def test_readme_aggregate_block(fv, pkg):
    f32 = np.float32
    singles = []
    for name, P, TP, FP, FN, tpr, ppv, fnr, fdr in README_ROWS:
        s = fv.SingleStats()
        s.total_positives_sec, s.true_positives_sec, s.false_positives_sec, s.false_negatives_sec = P, TP, FP, FN
        # the rates as printed (to 0.1 %: closer to the run's own values than anything recomputed from seconds that were
        # rounded to whole numbers -- 664 / 671 is 98.96 % where the run had 99.1 %)
        s.true_positive_rate, s.precision, s.false_negative_rate, s.false_discovery_rate = tpr / 100, ppv / 100, fnr / 100, fdr / 100
        s.f_score_beta = 0.7
        # statistics.fromEvaluator's formulas (statistics.zig:105-112) on the printed seconds agree with them within that rounding
        for got, want in ((f32(TP) / f32(P), tpr), (f32(TP) / (f32(TP) + f32(FP)), ppv), (f32(FN) / f32(P), fnr), (f32(FP) / (f32(FP) + f32(TP)), fdr)):
            assert abs(got * 100 - want) <= 0.2, (name, got * 100, want)
        singles.append(s)
    agg = fv.stats_aggregate(singles)
    sim = pkg.simulator
    txt = sim.report_text([r[0] for r in README_ROWS], singles, agg)
    # the per-stream table, byte for byte (report_generator.zig:21-27's row format)
    want_rows = ["| {} | {:>4d} | {:>4d} | {:>4d} | {:>4d} | {:>5.1f}% | {:>5.1f}% | {:>7.1f}% | {:>7.1f}% |".format(r[0].rjust(30), *r[1:])
                 for r in README_ROWS]
    got_rows = [l for l in txt.split("\n") if l.startswith("| ") and "2023" in l]
    assert len(got_rows) == 21
    assert got_rows == want_rows
    got_block = txt[txt.index("\n=> Aggregate stats"):]
    got_lines, want_lines = got_block.split("\n"), README_AGGREGATE.split("\n")
    assert len(got_lines) == len(want_lines)
    # sums of 21 rows rounded to whole seconds: within 21 x 0.5 s of the printed sums, in the printed layout
    for i, key in ((3, "total_positives_sec"), (4, "true_positives_sec"), (5, "false_positives_sec"), (6, "false_negatives_sec")):
        g, w = got_lines[i], want_lines[i]
        assert g[:28] == w[:28] and g[35:] == w[35:] and len(g) == len(w), (g, w)
        assert abs(float(g[28:35]) - float(w[28:35])) <= 10.5, (g, w)
    # the rate lines, F-score and Fowlkes-Mallows: byte for byte (overall rates, min / avg / max over the streams)
    for i in (0, 1, 2, 7, 8, 9, 10, 11, 12, 13):
        assert got_lines[i] == want_lines[i], (i, got_lines[i], want_lines[i])
    # the formulas behind the block (statistics.zig:116-182)
    assert abs(agg.f_score * 100 - 99.1) < 0.05 and abs(agg.fm_index * 100 - 98.9) < 0.05
    assert agg.true_positive_rate.min == min(s.true_positive_rate for s in singles)
    assert agg.precision.max == 1.0 and agg.false_discovery_rate.min == 0.0
    # the oracle's aggregate (the CPU restatement the GPU tests are checked against) is pinned by the same block
    O = orc.lib()
    osingles = (orc.SingleStats * 21)()
    for o, s in zip(osingles, singles):
        for name, _ in fv.SingleStats._fields_:
            setattr(o, name, getattr(s, name))
    oagg = O.orc_stats_aggregate(osingles, 21)
    for name in ("total_positives_sec", "true_positives_sec", "false_positives_sec", "false_negatives_sec", "fm_index", "f_score", "f_score_beta"):
        assert getattr(agg, name) == getattr(oagg, name), name
    for name in ("true_positive_rate", "false_negative_rate", "false_discovery_rate", "precision"):
        for f in ("overall", "min", "max", "avg"):
            assert getattr(getattr(agg, name), f) == getattr(getattr(oagg, name), f), (name, f)
