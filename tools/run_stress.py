"""fvad_engine_run's host-buffer pipeline under random shapes: ragged lanes (f32 or PCM16, denoised audio back or not), random
lane-group schedules (run_groups), random numbers of copy threads -- every call compared bit for bit (`reproducible`) with the
single-group, single-launch-plan reference of the same lanes.  python tools/run_stress.py [cases=40] [seed=1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = fv.Context(0); ctx.load_synth(7)
ctx.set_option("reproducible", "1")
base, _ = pkg.synth.make_stream(120.0, seed=77)
base = base[0]
SCHEDS = ["4,4,4,4", "1,3,4,8", "1,3,4,4,3,1", "2,2,4,4,4", "8,8", "1,1,1,1,1,1,10", "16", "5,5,6", "2,4,4,4,2"]
bad = 0
for case in range(cases):
    n_l = int(rng.integers(8, 40))
    pcm16 = bool(rng.integers(0, 2))
    want_den = bool(rng.integers(0, 2))
    lanes = []
    for i in range(n_l):
        n = int(rng.integers(40, 240)) * 24000 + int(rng.integers(0, 24000))
        x = np.roll(base, 4801 * int(rng.integers(0, 1000)))[:n].copy()
        lanes.append(np.clip(np.rint(x * 32768.0), -32768, 32767).astype(np.int16) if pcm16 else x)
    total = sum(x.nbytes for x in lanes)
    ctx.set_option("no_pipeline", "1")
    ref = ctx.engine_run(lanes, want_denoised=want_den)
    ctx.set_option("no_pipeline", None)
    sched = SCHEDS[int(rng.integers(0, len(SCHEDS)))]
    threads = int(rng.choice([1, 2, 3, 8, 16]))
    ctx.set_option("run_groups", sched); ctx.set_option("copy_threads", str(threads))
    out = ctx.engine_run(lanes, want_denoised=want_den)
    ctx.set_option("run_groups", None); ctx.set_option("copy_threads", None)
    ok = True
    for a, b in zip(out, ref):
        ok &= np.array_equal(a["band_sum"], b["band_sum"]) and np.array_equal(a["chunk_rms"], b["chunk_rms"])
        if want_den:
            ok &= np.array_equal(a["denoised"], b["denoised"])
    bad += not ok
    print(f"case {case}: {n_l} lanes, {total >> 20} MB, {'PCM16' if pcm16 else 'f32'}, denoised {want_den}, groups {sched}, {threads} copy threads: {'same bits' if ok else 'DIFFERENT'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
