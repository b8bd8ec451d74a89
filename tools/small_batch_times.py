"""Per-kernel times of the small-batch path (BASELINE config 3: 82 chunks; and a single-chunk push).
Run on the GPU box: python tools/small_batch_times.py"""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("formula_vad_amd", os.path.join(ROOT, "formula-vad_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "formula-vad_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["formula_vad_amd"] = pkg; spec.loader.exec_module(pkg)
fv = pkg.binding
ctx = fv.Context(0)
ctx.load_synth(7)
CHUNK = 24000
for lanes, chunks in ((2, 41), (1, 1), (2, 1), (8, 2), (16, 16), (64, 16)):
    pcm = [pkg.synth.make_stream(chunks * 0.5 + 0.1, seed=30 + i)[0][0][: chunks * CHUNK].copy() for i in range(lanes)]
    ctx.engine_run(pcm)
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ctx.engine_run(pcm)
    dt = (time.perf_counter() - t0) / reps
    kt = ctx.kernel_times()
    ctx.enable_timing(False)
    print(f"lanes={lanes} chunks/lane={chunks}: {dt*1e3:.2f} ms wall per call (incl. H2D/D2H);",
          {k: round(v / reps, 3) for k, v in kt.items()}, "sum %.2f" % (sum(kt.values()) / reps))
