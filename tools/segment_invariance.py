"""Default kernel selection: do the SEGMENTS of a stream depend on what it was batched with?  N streams of S seconds are run alone, all
together and in random groups (different launch sizes, hence different kernel families and launch plans); every stream's segment list
must be the same in every composition, and the smallest decision margin seen is printed.  python tools/segment_invariance.py [N=8] [S=1200] [seed=1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
secs = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
ctx = fv.Context(0); ctx.load_synth(7)
streams = []
for i in range(n):
    base, _ = pkg.synth.make_stream(secs + 0.5, seed=700 + i)
    streams.append(base[0][: secs * 48000].copy())


def run(idx):
    res = ctx.engine_run([streams[i] for i in idx])
    out = {}
    for i, r in zip(idx, res):
        vb = fv.VadBatch(1)
        segs = vb.run(np.ascontiguousarray(r["band_sum"][None, :]), np.ascontiguousarray(r["chunk_rms"][None, :]))[0]
        out[i] = ([(s[0], s[1]) for s in segs], vb.audit(0), ctx.last_nn_path())
        vb.close()
    return out


ref = run(list(range(n)))
print(f"all {n} together: {sum(len(v[0]) for v in ref.values())} segments [{ref[0][2]}]")
bad = 0
worst = min(v[1][0] for v in ref.values())
comps = [[i] for i in range(n)] + [list(rng.permutation(n)[: int(rng.integers(2, n))]) for _ in range(6)]
for comp in comps:
    got = run([int(i) for i in comp])
    for i, (segs, audit, path) in got.items():
        worst = min(worst, audit[0])
        if segs != ref[i][0]:
            bad += 1
            print(f"stream {i} in composition {comp}: segments differ [{path}]")
print(f"{len(comps)} compositions, streams whose segments differ from the all-together run: {bad}; smallest relative threshold margin seen {worst:.3e}")
sys.exit(1 if bad else 0)
