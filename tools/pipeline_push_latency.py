"""Latency of AudioPipeline.pushSamples-sized pushes through fvad_pipeline (the live-daemon use of the
reference, main.zig): 1 s pushes of a stereo stream, recorder callbacks on.  python tools/pipeline_push_latency.py"""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("formula_vad_amd", os.path.join(ROOT, "formula-vad_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "formula-vad_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["formula_vad_amd"] = pkg; spec.loader.exec_module(pkg)
fv = pkg.binding
ctx = fv.Context(0)
ctx.load_synth(7)
for n_ch, push_s, record in ((2, 1.0, False), (1, 1.0, True), (1, 0.5, False), (2, 5.0, False)):
    pcm, _ = pkg.synth.make_stream(120.0, seed=40, n_channels=n_ch)
    p = fv.AudioPipeline(ctx, n_channels=n_ch, record=record)
    n = int(push_s * 48000)
    lat = []
    for o in range(0, pcm.shape[1] - n + 1, n):
        t0 = time.perf_counter()
        p.push_samples(pcm[:, o:o + n])
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.array(lat[2:])
    print(f"push {push_s:.1f} s x {n_ch} channel(s), recorder {'on' if record else 'off'}: median {np.median(lat):.2f} ms, "
          f"p95 {np.percentile(lat, 95):.2f} ms, {push_s * 1e3 / np.median(lat):.0f}x realtime, "
          f"{len(p.segments())} segments, {len(p.recordings['denoised'])} clips")
    p.close()
