"""BASELINE config 2 (FFT isolation): device-resident rate of fvad_fft_forward_batch (window + rFFT-320 + |X|) at
1024 and 2^20 frames, magnitudes only and bins + magnitudes, with a digest of the outputs (two builds on one box: FVAD_LIB_PATH).
python tools/fft_batch_rate.py"""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("formula_vad_amd", os.path.join(ROOT, "formula-vad_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "formula-vad_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["formula_vad_amd"] = pkg; spec.loader.exec_module(pkg)
fv = pkg.binding
L = fv.lib()
ctx = fv.Context(0)
f = fv.FFT(ctx, 320, 16000)
win = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * np.arange(320) / 319)).astype(np.float32)
d_win = ctx.device_alloc(320 * 4); ctx.to_device(d_win, win)
rng = np.random.default_rng(1)
for n in (1024, 1 << 20):
    x = rng.uniform(-1, 1, (n, 320)).astype(np.float32)
    d_x = ctx.device_alloc(x.nbytes); ctx.to_device(d_x, x)
    d_mag = ctx.device_alloc(n * 161 * 4)
    d_bins = ctx.device_alloc(n * 161 * 8)
    for what, b, m, nbytes in (("mag", None, d_mag, 1924), ("bins+mag", d_bins, d_mag, 1924 + 1288)):
        reps = 200 if n == 1024 else 20
        L.fvad_fft_forward_batch(f.h, d_x, n, d_win, b, m, 1)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            L.fvad_fft_forward_batch(f.h, d_x, n, d_win, b, m, 1)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        import hashlib
        h_m = np.empty((n, 161), np.float32); ctx.to_host(h_m, m)
        dig = hashlib.sha256(h_m.tobytes())
        if b is not None:
            h_b = np.empty((n, 161, 2), np.float32); ctx.to_host(h_b, b)
            dig.update(h_b.tobytes())
        print(f"n={n} {what}: {dt*1e6:.1f} us/launch, {n*nbytes/dt/1e9:.0f} GB/s = {n*nbytes/dt/8e12:.3f} of 8 TB/s   digest {dig.hexdigest()[:16]}", flush=True)
    for d in (d_x, d_mag, d_bins):
        ctx.device_free(d)
