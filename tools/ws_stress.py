"""Stress of the weight-stationary recurrence's in-launch hand-off (kernels_ws.hip): the same small batches over and
over while a second context keeps the chip busy with large launches (uneven load); every repetition must give the
bit-identical result (a stale or torn read of another workgroup's h tile would not).  python tools/ws_stress.py [reps]"""
import importlib.util, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("formula_vad_amd", os.path.join(ROOT, "formula-vad_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "formula-vad_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["formula_vad_amd"] = pkg; spec.loader.exec_module(pkg)
fv = pkg.binding
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = fv.Context(0); ctx.load_synth(7)
bg = fv.Context(0); bg.load_synth(7)
rng = np.random.default_rng(0)
stop = False
def background():
    f = rng.uniform(-11, 2, (4096, 54, 161)).astype(np.float32)
    while not stop:
        bg.nsnet2_forward(f)
th = threading.Thread(target=background); th.start()
bad = 0
try:
    for n_seq in (1, 16, 82, 160, 330, 1000):
        f = np.random.default_rng(n_seq).uniform(-11, 2, (n_seq, 54, 161)).astype(np.float32)
        ref = ctx.nsnet2_forward(f)
        t0 = time.perf_counter()
        for r in range(reps):
            g = ctx.nsnet2_forward(f)
            if not np.array_equal(g, ref):
                bad += 1
                print(f"n_seq={n_seq} rep {r}: {np.count_nonzero(g != ref)} values differ, max {np.abs(g - ref).max():.3e}", flush=True)
        print(f"n_seq={n_seq}: {reps} repetitions, {(time.perf_counter() - t0) / reps * 1e3:.2f} ms each under load", flush=True)
finally:
    stop = True
    th.join()
print("mismatches:", bad, " fallback passes:", ctx.ws_fallbacks())
sys.exit(1 if bad else 0)
