"""BASELINE config 3's literal batch (82 chunks, device-resident) in a loop, for rocprofv3 --kernel-trace --stats:
true kernel durations against the wall time of a call (the difference is launch gaps).  python tools/cfg3_loop.py [graph]"""
import os, sys, time
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
L = fv.lib()
ctx = fv.Context(0); ctx.load_synth(7)
CH = 24000
pcm = np.stack([pkg.synth.make_stream(20.5, seed=30 + i)[0][0][: 41 * CH] for i in range(2)])
d = ctx.device_alloc(pcm.nbytes); ctx.to_device(d, pcm)
band = ctx.device_alloc(2 * (41 * CH // 1024) * 4); rms = ctx.device_alloc(2 * 41 * 4)
use_graph = len(sys.argv) > 1 and sys.argv[1] == "graph"
for it in range(204):
    if it == 4:
        ctx.synchronize(); t0 = time.perf_counter()
    ctx.enqueue_device(d, 2, 41 * CH, 41 * CH, None, band, rms, use_graph=use_graph)
ctx.synchronize()
print(f"{'graph' if use_graph else 'direct'}: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per call", ctx.last_nn_path())
