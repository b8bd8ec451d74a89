"""ms per call of the device-resident path at launch sizes between the batch curve's points (chunks = lanes x chunks per lane):
which sizes fill the chip badly.  python tools/batch_sizes.py [max_chunks=N ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = bench.load_package(); fv = pkg.binding
ctx = fv.Context(0); ctx.load_synth(7)
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1)
    ctx.set_option(k, v)
host = [pkg.synth.make_stream(64.5, seed=30 + i)[0][0][: 128 * 24000] for i in range(2)]
pts = tuple((128, c) for c in (8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 128, 160, 192, 224, 256))
r = bench.batch_curve(fv, ctx, host, points=pts, budget_s=0.25)
for p in r["points"]:
    print(f"{p['chunks']:6d} chunks: {p['ms']:8.3f} ms  {p['ms'] * 1e3 / p['chunks']:6.3f} us/chunk  {p['frames_per_s'] / 1e6:6.2f} M frames/s   {p['nn_path']}", flush=True)
