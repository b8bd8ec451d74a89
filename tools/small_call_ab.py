"""Time per call of small device-resident batches (calls queued back to back, as bench.py's cfg3 extra), with a digest of everything
a call returns: the program to run under two builds of the library on one box (FVAD_LIB_PATH selects the build).
python tools/small_call_ab.py [lanes x chunks ...]   (default: 2x41 1x1 1x8 2x32)"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
ctx = fv.Context(0); ctx.load_synth(7)
CH = 24000
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(2, 41), (1, 1), (1, 8), (2, 32)]
for lanes, chunks in shapes:
    pcm = np.stack([pkg.synth.make_stream(chunks / 2 + 0.5, seed=30 + i)[0][0][: chunks * CH] for i in range(lanes)])
    d = ctx.device_alloc(pcm.nbytes); ctx.to_device(d, pcm)
    nfr = max(chunks * CH // 1024, 1)
    den = ctx.device_alloc(pcm.nbytes); band = ctx.device_alloc(lanes * nfr * 4); rms = ctx.device_alloc(lanes * chunks * 4)

    def run(n):
        for _ in range(n):
            ctx.enqueue_device(d, lanes, chunks * CH, chunks * CH, den, band, rms, no_wait=True)
        ctx.synchronize()

    best = 1e9
    for rep in range(5):
        run(20)
        t0 = time.perf_counter(); run(200); best = min(best, (time.perf_counter() - t0) / 200)
    h = hashlib.sha256()
    for arr, dev in ((np.empty((lanes, chunks * CH), np.float32), den), (np.empty((lanes, nfr), np.float32), band), (np.empty((lanes, chunks), np.float32), rms)):
        h.update(ctx.to_host(arr, dev).tobytes())
    print(f"{lanes} x {chunks} chunks: {best * 1e3:.4f} ms per call  digest {h.hexdigest()[:12]}  ({ctx.last_nn_path()})", flush=True)
    for x in (d, den, band, rms):
        ctx.device_free(x)
