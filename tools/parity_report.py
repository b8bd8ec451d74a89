"""Measured GPU-vs-oracle errors of the current build (the numbers quoted in DESIGN.md section 4).
Run on the GPU box: python tools/parity_report.py"""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("formula_vad_amd", os.path.join(ROOT, "formula-vad_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "formula-vad_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["formula_vad_amd"] = pkg; spec.loader.exec_module(pkg)
import orc  # the checker
fv = pkg.binding
ctx = fv.Context(0)
ctx.load_synth(7)
W = ctx.weights()
rng = np.random.default_rng(0)

# NSNet2 graph, small-batch path and large-batch path
for n_seq in (8, 2100):
    f = rng.uniform(-11, 2, (n_seq, 54, 161)).astype(np.float32)
    g = ctx.nsnet2_forward(f)
    pick = list(range(min(n_seq, 8))) if n_seq < 100 else [0, 191, 192, 1023, 2047, 2099]
    ref = np.stack([orc.nsnet2_forward(W, f[i]) for i in pick])
    err = np.abs(g[pick] - ref)
    print(f"gains, {n_seq} sequences: max abs {err.max():.2e}, max rel (floor 1e-2) {(err / np.maximum(np.abs(ref), 1e-2)).max():.2e}")

# full path on a 30 s stereo stream
pcm, _ = pkg.synth.make_stream(30.0, seed=3, n_channels=2)
n = (pcm.shape[1] // 24000) * 24000
out = ctx.engine_run([pcm[0][:n].copy(), pcm[1][:n].copy()], want_denoised=True)
p = orc.Pipeline(W, n_channels=2, keep_denoised=True)
p.push(pcm[:, :n])
den_ref, band_ref, rms_ref = p.denoised(), p.band_volumes(), p.chunk_rms()
for c in range(2):
    d = out[c]["denoised"]
    print(f"channel {c}: denoised max err {np.abs(d - den_ref[c]).max() / np.abs(den_ref[c]).max():.2e} of peak, "
          f"rel L2 {np.linalg.norm(d - den_ref[c]) / np.linalg.norm(den_ref[c]):.2e}; "
          f"band sums max rel {np.abs(out[c]['band_sum'] - band_ref[:, c]).max() / np.abs(band_ref[:, c]).max():.2e} of max, "
          f"worst per-frame rel {(np.abs(out[c]['band_sum'] - band_ref[:, c]) / band_ref[:, c]).max():.2e}; "
          f"chunk RMS max rel {(np.abs(out[c]['chunk_rms'] - rms_ref[:, c]) / rms_ref[:, c]).max():.2e}")
