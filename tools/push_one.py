"""One AudioPipeline of one channel pushed 0.5 s at a time, 400 pushes (for rocprofv3 --hip-trace --stats: where the host
time of a push goes).  python tools/push_one.py"""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("formula_vad_amd", os.path.join(ROOT, "formula-vad_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "formula-vad_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["formula_vad_amd"] = pkg; spec.loader.exec_module(pkg)
fv = pkg.binding
ctx = fv.Context(0); ctx.load_synth(7)
pcm, _ = pkg.synth.make_stream(210.0, seed=40, n_channels=1)
p = fv.AudioPipeline(ctx, n_channels=1, record=False)
n = 24000
lat = []
for o in range(0, 410 * n, n):
    t0 = time.perf_counter()
    p.push_samples(pcm[:, o:o + n])
    lat.append((time.perf_counter() - t0) * 1e3)
lat = np.array(lat[10:])
print(f"0.5 s mono pushes: median {np.median(lat):.3f} ms, mean {lat.mean():.3f}, p95 {np.percentile(lat, 95):.3f}")
