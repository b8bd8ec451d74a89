"""Throughput against batch size (bench.py's `extra.batch_curve`) on its own: python tools/batch_curve.py [option=value ...]
(context options, e.g. gru_kernel=v5w0, to compare kernel selections at a point)."""
import os, sys, json
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
r = bench.batch_curve(fv, ctx, host)
for p in r["points"]:
    print(f"{p['chunks']:6d} chunks ({p['lanes']:3d} lanes): {p['ms']:9.3f} ms  {p['frames_per_s'] / 1e6:7.2f} M frames/s   {p['nn_path']}", flush=True)
print(json.dumps({k: v for k, v in r.items() if k != "points"}))
