"""Batch-evaluation harness reproducing the observable behaviour of the reference's `simulator`
executable (src/simulator.zig, src/simulator/SimulationInstance.zig, src/simulator/report_generator.zig)
on top of the C ABI -- SURVEY.md section 8 f1/f3.  Plumbing only (JSON, paths, text); audio decode,
the pipeline, the Evaluator and the statistics all live behind include/fvad.h.

    python -m ... simulator.py -i plan.json           (or: run_plan(path))

Plan schema = the reference's (simulator.zig:41-76, tmp/plan.example.json), unknown fields ignored
(simulator.zig:152-154); audio/ref paths are relative to the plan file (simulator.zig:146,
SimulationInstance.zig:101-104).  Differences, by design:
  * audio files are WAV (PCM16 / float32), not OGG: there is no libsndfile here;
  * instances are not run one-thread-each (simulator.zig:221-232): all channels of all instances on a GPU
    form ONE batch, then the per-instance VAD state machines run on the host; with --devices the
    instances are dealt round-robin to one context + one host thread per GPU;
  * `preload_audio` only changes how samples are pushed in the reference, not the result.
"""
import argparse
import json
import os
import sys
import time
from decimal import ROUND_HALF_UP, Decimal

import numpy as np

from . import binding as fv

VAD_FIELDS = ("speech_min_freq", "speech_max_freq", "long_term_speech_avg_sec", "initial_long_term_avg",
              "short_term_speech_avg_sec", "speech_threshold_factor", "channel_vol_ratio_avg_sec",
              "channel_vol_ratio_threshold", "min_consecutive_sec_to_open", "max_speech_gap_sec",
              "min_vad_duration_sec")


def vad_overrides(cfg_json):
    """VADMachine.Config fields of the plan (VADMachine.zig:30-51) -> binding overrides"""
    out = {}
    for k, v in (cfg_json or {}).items():
        if k == "initial_long_term_avg":
            if v is None:
                out["has_initial_long_term_avg"] = 0
            else:
                out["has_initial_long_term_avg"] = 1
                out["initial_long_term_avg"] = float(v)
        elif k in VAD_FIELDS:
            out[k] = float(v)
    return out


def load_plan(path):
    """-> dict(instances=[{name, audio_path, ref_path}], config={...}, base_path)"""
    with open(path) as f:
        plan = json.load(f)
    base = os.path.dirname(path) or "."
    cfg = plan.get("config", {}) or {}
    vad_cfg = cfg.get("vad_config", {}) or {}
    return {
        "base_path": base,
        "instances": [{"name": i["name"],
                       "audio_path": os.path.normpath(os.path.join(base, i["audio_path"])),
                       "ref_path": os.path.normpath(os.path.join(base, i["ref_path"]))}
                      for i in plan["instances"]],
        "fft_size": int(vad_cfg.get("fft_size", 1024)),
        "vad_machine_config": vad_overrides(vad_cfg.get("vad_machine_config")),
        "alt_vad_machine_configs": [vad_overrides(a) for a in (vad_cfg.get("alt_vad_machine_configs") or [])],
        "denoiser_model_path": vad_cfg.get("denoiser_model_path"),
        "output_dir": cfg.get("output_dir"),
        "preload_audio": bool(cfg.get("preload_audio", False)),
        "audio_read_frame_count": int(cfg.get("audio_read_frame_count", 48000)),
    }


# ------------------------------------------------------------------ Zig-style number formatting
def zig_fixed(x, precision):
    """std.fmt `{d:.N}`: shortest round-trip decimal of the value (as f64), rounded half-up"""
    x = float(x)
    if x != x:
        return "nan"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    q = Decimal(1).scaleb(-precision)
    return str(Decimal(repr(x)).quantize(q, rounding=ROUND_HALF_UP))


def _f(x, width, precision):
    return zig_fixed(x, precision).rjust(width)


DEFINITIONS = (  # report_generator.zig:10-19
    "P   (Positives):                            Total duration of real speech segments (from reference labels)\n"
    "TP  (True positives):                       Duration of correctly detected speech segments\n"
    "FP  (False positives):                      Duration of incorrectly detected speech segments\n"
    "FN  (False negatives):                      Duration of missed speech segments\n"
    "TPR (True positive rate, sensitivity):      Probability that VAD detects a real speech segment. = TP / P \n"
    "PPV (Precision, Positive predictive value): Probability that detected speech segment is true.   = TP / (TP + FP) \n"
    "FNR (False negative rate, miss rate):       Probability that VAD misses a speech segment.       = FN / P \n"
    "FDR (False discovery rate):                 Probability that detected speech segment is false.  = FP / (TP + FP) ")


def report_text(names, stats, agg):
    """report_generator.bufPrintSimulationReport (report_generator.zig:29-116)"""
    out = ["\n\n=> Definitions\n\n" + DEFINITIONS, "\n\n=> Performance Report\n\n"]
    hdr = ("Name", "P", "TP", "FP", "FN", "TPR", "PPV", "FNR (!)", "FDR (!)")
    widths = (30, 4, 4, 4, 4, 6, 6, 8, 8)
    out.append("| " + " | ".join(h.rjust(w) for h, w in zip(hdr, widths)) + " |\n")
    out.append("| " + " | ".join("-" * w for w in widths) + " |\n")
    for name, s in zip(names, stats):
        out.append("| {} | {} | {} | {} | {} | {}% | {}% | {}% | {}% |\n".format(
            name.rjust(30), _f(s.total_positives_sec, 4, 0), _f(s.true_positives_sec, 4, 0),
            _f(s.false_positives_sec, 4, 0), _f(s.false_negatives_sec, 4, 0),
            _f(np.float32(s.true_positive_rate) * np.float32(100), 5, 1),
            _f(np.float32(s.precision) * np.float32(100), 5, 1),
            _f(np.float32(s.false_negative_rate) * np.float32(100), 7, 1),
            _f(np.float32(s.false_discovery_rate) * np.float32(100), 7, 1)))
    out.append("\n=> Aggregate stats \n\n")
    out.append("Total speech duration  (P): {} sec\n".format(_f(agg.total_positives_sec, 7, 1)))
    out.append("True positives        (TP): {} sec\n".format(_f(agg.true_positives_sec, 7, 1)))
    out.append("False positives       (FP): {} sec\n".format(_f(agg.false_positives_sec, 7, 1)))
    out.append("False negatives       (FN): {} sec".format(_f(agg.false_negatives_sec, 7, 1)))
    out.append("    Min.    Avg.    Max. \n")
    for label, a in (("True positive rate   (TPR)", agg.true_positive_rate), ("Precision            (PPV)", agg.precision),
                     ("False negative rate  (FNR)", agg.false_negative_rate), ("False discovery rate (FDR)", agg.false_discovery_rate)):
        p = lambda v: _f(np.float32(v) * np.float32(100), 5, 1)  # noqa: E731
        out.append("{}:   {}%  |  {}% /{}% /{}% \n".format(label, p(a.overall), p(a.min), p(a.avg), p(a.max)))
    out.append("F-Score (β = {})       :   {}% \n".format(_f(agg.f_score_beta, 5, 2), _f(np.float32(agg.f_score) * np.float32(100), 5, 1)))
    out.append("Fowlkes-Mallows index     :   {}% \n".format(_f(np.float32(agg.fm_index) * np.float32(100), 5, 1)))
    return "".join(out)


def audacity_txt(vad_secs, debug_infos, ref_secs, cfg):
    """formats.serializeEvaluatorToAudacityTxt (formats.zig:38-56): VAD segments (sorted by start) with
    their comment, then the reference segments nothing overlapped, labelled "missed".
    (The overlap test of every VAD segment against every reference is one float32 matrix expression: as a Python double loop
    it was 0.44 s per two-hour instance -- 9 s for config 4's plan, ten times its GPU time.)"""
    order = sorted(range(len(vad_secs)), key=lambda i: vad_secs[i][0])
    refs = sorted(ref_secs, key=lambda r: r[0])
    vs = np.array([vad_secs[i] for i in order], np.float32).reshape(-1, 2)
    rs = np.array(refs, np.float32).reshape(-1, 2)
    # overlap = f32(min(ends)) - f32(max(starts)) > 0, as SpeechSegment.overlap computes it (SpeechSegment.zig:22-57)
    ov = (np.minimum(vs[:, 1, None], rs[None, :, 1]) - np.maximum(vs[:, 0, None], rs[None, :, 0])) > 0
    v_matched = ov.any(axis=1) if rs.shape[0] else np.zeros(vs.shape[0], bool)
    r_matched = ov.any(axis=0) if vs.shape[0] else np.zeros(rs.shape[0], bool)
    lines = []
    for j, i in enumerate(order):
        v = vad_secs[i]
        comment = debug_infos[i] if v_matched[j] else "UNMATCHED " + debug_infos[i]
        lines.append("{}\t{}\t{}\n".format(zig_fixed(v[0], 4), zig_fixed(v[1], 4), comment))
    for k, r in enumerate(refs):
        if not r_matched[k]:
            lines.append("{}\t{}\t{}\n".format(zig_fixed(r[0], 4), zig_fixed(r[1], 4), "missed"))
    return "".join(lines)


def _make_ctx(plan, device, synth_seed):
    ctx = fv.Context(device)
    if plan["denoiser_model_path"]:
        ctx.load_onnx(os.path.join(plan["base_path"], plan["denoiser_model_path"]))
    elif os.path.exists("data/nsnet2-20ms-baseline.onnx"):   # NSNet2.zig:56 default
        ctx.load_onnx("data/nsnet2-20ms-baseline.onnx")
    else:
        if synth_seed is None:
            ctx.close()
            raise FileNotFoundError("no denoiser_model_path in the plan and no data/nsnet2-20ms-baseline.onnx "
                                    "(pass synth_seed to run on random-init weights)")
        ctx.load_synth(synth_seed)
    return ctx


def _run_instances(ctx, plan, audio):
    """The path for a set of instances on one context: ONE GPU batch over all their channels, then the
    per-instance VAD state machines on the host.  Returns [(segments, audit)] in the order given."""
    if not audio:
        return []
    lanes = [pcm[c] for pcm in audio for c in range(pcm.shape[0])]
    vm = plan["vad_machine_config"]
    bin_w = np.float32(48000) / np.float32(plan["fft_size"])  # band edges: FFT.freqToBin (FFT.zig:156-167)
    min_bin = int(np.round(np.float32(vm.get("speech_min_freq", 500.0)) / bin_w))
    max_bin = int(np.round(np.float32(vm.get("speech_max_freq", 2000.0)) / bin_w))
    res = ctx.engine_run(lanes, min_bin=min_bin, max_bin=max_bin, fft_size=plan["fft_size"])
    # host stage: the library's batched form (frame metadata + VAD state machine), one call per instance (instances differ
    # in length and channel count) -- the instances side by side on host threads, like the reference's thread per file
    # (simulator.zig:221-232): a two-hour stream's state machine is 36 ms on one core, 21 of them one after the other were
    # as long as the GPU's part of the plan
    from concurrent.futures import ThreadPoolExecutor
    first = np.cumsum([0] + [pcm.shape[0] for pcm in audio])

    def host_stage(i):
        C_ = audio[i].shape[0]
        r = res[first[i]:first[i] + C_]
        band = np.ascontiguousarray(np.stack([x["band_sum"] for x in r]))      # [channel][frame]
        rms = np.ascontiguousarray(np.stack([x["chunk_rms"] for x in r]))
        vb = fv.VadBatch(1, n_channels=C_, fft_size=plan["fft_size"], overrides=vm)
        try:
            segs = vb.run(band, rms)[0] if band.shape[1] else []
            return segs, vb.audit(0)
        finally:
            vb.close()

    n_workers = max(1, min(len(audio), os.cpu_count() or 1, 16))
    if n_workers == 1:
        return [host_stage(i) for i in range(len(audio))]
    with ThreadPoolExecutor(max_workers=n_workers) as pool:
        return list(pool.map(host_stage, range(len(audio))))


def run_plan(plan_path, ctx=None, synth_seed=None, out=sys.stdout, devices=None):
    """Runs a whole plan; returns (report_text, per_instance_results).

    devices: list of HIP device indices -- one context and one host thread per entry, instance i on
    devices[i % len(devices)] (the reference's unit of parallelism is the instance: one thread per file,
    simulator.zig:221-232; here one thread per GPU with that GPU's instances as one batch).  The report is built
    from the per-instance statistics in PLAN order whatever the split (report_generator.zig:48-68,
    statistics.zig:116-172), so it is identical for every device list."""
    import threading
    plan = load_plan(plan_path)
    own_ctx = ctx is None
    if own_ctx:
        ctxs = [_make_ctx(plan, d, synth_seed) for d in (devices or [0])]
    else:
        ctxs = [ctx]
    def read_instance(inst):
        try:    # PCM16 files stay 16-bit all the way to the GPU (half the PCIe / HBM bytes; converted by the kernel
            pcm, sr = fv.wav_read_i16(inst["audio_path"])   # that reads them, bit-identical to converting first)
        except fv.FvadError:
            pcm, sr = fv.wav_read(inst["audio_path"])
        if sr != 48000:
            raise fv.FvadError(-9, f"{inst['name']}: sample rate {sr}")   # VADPipeline.zig:55-58
        with open(inst["ref_path"], "rb") as f:
            return pcm, fv.parse_audacity(f.read())

    # the files side by side (the reference opens them on a thread each, SimulationInstance.zig:144-152): reading and
    # de-interleaving a two-hour file is most of a plan's wall time once the GPU does the rest
    from concurrent.futures import ThreadPoolExecutor
    n_readers = max(1, min(len(plan["instances"]), os.cpu_count() or 1, 8))
    if n_readers == 1:
        loaded = [read_instance(i) for i in plan["instances"]]
    else:
        with ThreadPoolExecutor(max_workers=n_readers) as pool:
            loaded = list(pool.map(read_instance, plan["instances"]))
    audio = [a for a, _ in loaded]
    refs = [r for _, r in loaded]
    t0 = time.perf_counter()
    n_ctx = len(ctxs)
    parts = [[i for i in range(len(audio)) if i % n_ctx == d] for d in range(n_ctx)]
    done = [None] * n_ctx
    errs = []

    def work(d):
        try:
            done[d] = _run_instances(ctxs[d], plan, [audio[i] for i in parts[d]])
        except Exception as e:  # re-raised below, in the caller's thread
            errs.append(e)

    if n_ctx == 1:
        work(0)
    else:
        th = [threading.Thread(target=work, args=(d,)) for d in range(n_ctx)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    if errs:
        raise errs[0]
    per_inst = [None] * len(audio)
    for d in range(n_ctx):
        for i, r in zip(parts[d], done[d]):
            per_inst[i] = r
    elapsed = time.perf_counter() - t0
    vm = plan["vad_machine_config"]
    # --- Evaluator + statistics (simulator.zig:127-132)
    stat_cfg = {"ignore_shorter_than_sec": float(np.float32(vm.get("min_vad_duration_sec", 0.7))),
                "extrude_start": 5.0, "extrude_end": 10.0, "fill_gaps": 5.0}
    names, stats, results = [], [], []
    for inst, (segs, audit), ref in zip(plan["instances"], per_inst, refs):
        secs = [(float(np.float32(s[0]) / np.float32(48000)), float(np.float32(s[1]) / np.float32(48000))) for s in segs]
        infos = ["vr:{} vad:{}s".format(zig_fixed(s[2], 2), zig_fixed(s[3], 1)) for s in segs]  # SimulationInstance.zig:240-244
        st = fv.stats_from_segments(secs, ref, stat_cfg)
        names.append(inst["name"])
        stats.append(st)
        results.append({"name": inst["name"], "segments": segs, "segments_sec": secs, "debug_info": infos,
                        "stats": st, "audacity": audacity_txt(secs, infos, ref, stat_cfg), "audit": audit})
    agg = fv.stats_aggregate(stats)
    text = report_text(names, stats, agg)
    if plan["output_dir"]:
        out_dir = os.path.join(plan["base_path"], plan["output_dir"], str(int(time.time())))  # simulator.zig:157-168
        os.makedirs(out_dir, exist_ok=True)
        for r in results:
            with open(os.path.join(out_dir, f"{r['name']}-audacity.txt"), "w") as f:
                f.write(r["audacity"])
        with open(os.path.join(out_dir, "report.txt"), "w") as f:
            f.write(text)
    if out is not None:
        out.write(text)
        audio_s = sum(p.shape[1] for p in audio) / 48000.0
        out.write(f"\n[{audio_s:.0f} s of audio in {elapsed:.2f} s = {audio_s / elapsed:.0f}x realtime]\n")
    if own_ctx:
        for c in ctxs:
            c.close()
    return text, results


def frame_ratios(chunk_rms, n_frames, fft_size=1024, chunk=24000):
    """Per-frame volume_ratio exactly as the metadata flows through the three buffered stages
    (BufferedVolumeAnalyzer.zig:33-45 -> BufferedDenoiser.zig:83-86,115 -> BufferedFFT.zig:137-140,153):
    all f32, weights are sample counts."""
    rms = np.asarray(chunk_rms, np.float32)
    vmin = np.minimum(np.float32(1), rms.min(axis=1))
    vmax = np.maximum(np.float32(0), rms.max(axis=1))
    ratio = np.where(vmax == 0, np.float32(0), vmin / np.where(vmax == 0, np.float32(1), vmax)).astype(np.float32)
    w = np.float32(chunk)
    r2 = ((ratio * w) / w).astype(np.float32)     # analyzer stage
    r2 = ((r2 * w) / w).astype(np.float32)        # denoiser stage
    out = np.empty(n_frames, np.float32)
    for k in range(n_frames):
        lo, hi = k * fft_size, (k + 1) * fft_size
        c0, c1 = lo // chunk, (hi - 1) // chunk
        if c0 == c1:
            n = np.float32(fft_size)
            out[k] = (np.float32(0) + r2[c0] * n) / n
        else:
            n0 = np.float32((c0 + 1) * chunk - lo)
            n1 = np.float32(hi - c1 * chunk)
            s = np.float32(0) + r2[c0] * n0
            s = s + r2[c1] * n1
            out[k] = s / (n0 + n1)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description="Formula-VAD simulator harness on MI355X")
    ap.add_argument("-i", "--input", required=True, help="Simulation plan (path to JSON)")  # simulator.zig:78-82
    ap.add_argument("--synth-seed", type=int, default=None, help="use random-init NSNet2 weights")
    ap.add_argument("--devices", default="0", help="comma-separated HIP devices: one context + one thread each, "
                                                   "instances dealt round-robin (e.g. 0,1,2,3,4,5,6,7)")
    a = ap.parse_args(argv)
    run_plan(a.input, synth_seed=a.synth_seed, devices=[int(d) for d in a.devices.split(",") if d != ""])


if __name__ == "__main__":
    main()
