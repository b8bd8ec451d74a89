"""Randomised shapes through the two entry points whose kernel selection depends on the batch (python tools/fuzz_shapes.py
[cases] [seed], on the GPU box):
  * fvad_nsnet2_forward with random (n_seq, T), odd T included: three sequences per case against the oracle, and the
    first sequences recomputed in a batch of another size -- within a kernel family the bits must not move;
  * fvad_engine_run with random ragged lanes (1..40 chunks, f32 or PCM16), one-shot against a random
    max_chunks_per_launch split and against two pushes through lane states: bit for bit.
Exit status 1 on the first mismatch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
import orc
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = fv.Context(0); ctx.load_synth(7)
W = ctx.weights()
bad = 0

def family(path):
    return "large" if ("panel_gemm3" in path or "rec3" in path) else "small"

t0 = time.time()
for c in range(cases):
    n = int(rng.choice([1, 2, 3, 5, 16, 17, 31, 32, 33, 64, 80, 81, 82, 96, 97, 130, 200, 384, 385, 700, int(rng.integers(1, 1200))]))
    T = int(rng.choice([1, 2, 3, 5, 7, 11, 54, 54, 54, int(rng.integers(1, 80))]))
    f = rng.uniform(-11, 2, (n, T, 161)).astype(np.float32)
    g = ctx.nsnet2_forward(f)
    path = ctx.last_nn_path()
    for i in rng.choice(n, size=min(3, n), replace=False):
        ref = orc.nsnet2_forward(W, f[i])
        err = float((np.abs(g[i] - ref) / np.maximum(np.abs(ref), 1e-2)).max())
        if not err <= 1e-4:
            bad += 1; print(f"nsnet2 n={n} T={T} seq {i}: rel err {err:.2e} ({path})", flush=True)
    k = int(rng.integers(1, n + 1))
    g2 = ctx.nsnet2_forward(f[:k])
    path2 = ctx.last_nn_path()
    if path == path2:
        if not np.array_equal(g2, g[:k]):
            bad += 1; print(f"nsnet2 n={n} T={T}: first {k} sequences alone differ ({path} | {path2})", flush=True)
    elif np.abs(g2 - g[:k]).max() > 2e-5:
        bad += 1; print(f"nsnet2 n={n} T={T}: first {k} sequences alone differ by {np.abs(g2 - g[:k]).max():.2e} ({path} | {path2})", flush=True)
print(f"nsnet2: {cases} cases, {time.time() - t0:.0f} s, mismatches so far {bad}", flush=True)

def same(a, b, what):
    global bad
    for u, v in zip(a, b):
        for key in ("denoised", "band_sum", "chunk_rms"):
            if not np.array_equal(u[key], v[key]):
                bad += 1; print(f"engine {what}: {key} differs", flush=True); return

t0 = time.time()
for c in range(cases):
    n_l = int(rng.integers(1, 7))
    # default mode: every launch of up to 1536 chunks runs the pipelined recurrence with one accumulation order (the input
    # projections in a GEMM in front or in the kernel: the same bits); the `reproducible` cases (every third) pin the
    # large-batch family
    lens = [int(rng.integers(1, 41)) for _ in range(n_l)]
    i16 = bool(rng.integers(0, 2))
    streams = []
    for i, nc in enumerate(lens):
        x = pkg.synth.make_stream(nc * 0.5 + 0.5, seed=int(rng.integers(1, 1 << 30)))[0][0][: nc * 24000]
        streams.append(np.round(x * 32767).astype(np.int16) if i16 else x.copy())
    # the small-batch family (the default at these sizes) and the large-batch one ("reproducible") in turn: within
    # either, a chunk's bits do not depend on how lanes and chunks are split over launches
    with ctx.options(**({"reproducible": "1"} if c % 3 == 2 else {})):
        whole = ctx.engine_run(streams, want_denoised=True)
        mc = int(rng.integers(1, sum(lens) + 1))
        split = ctx.engine_run(streams, want_denoised=True, max_chunks_per_launch=mc)
        same(whole, split, f"lanes {lens} i16={i16} max_chunks {mc}")
        if min(lens) >= 2:
            cut = [int(rng.integers(1, nc)) for nc in lens]
            sts = [ctx.lane_state() for _ in streams]
            a = ctx.engine_run([x[: k * 24000] for x, k in zip(streams, cut)], want_denoised=True, states=sts)
            b = ctx.engine_run([x[k * 24000:] for x, k in zip(streams, cut)], want_denoised=True, states=sts)
            for st in sts:
                fv.lib().fvad_lane_state_destroy(st)
            two = [{key: np.concatenate([u[key], v[key]]) for key in ("denoised", "band_sum", "chunk_rms")} for u, v in zip(a, b)]
            same(whole, two, f"lanes {lens} i16={i16} two pushes at {cut}")
print(f"engine: {cases} cases, {time.time() - t0:.0f} s", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
