"""The spin kernels next to ANOTHER PROCESS on the same GPU: a child process keeps the chip busy with large-batch NSNet2
passes (its workgroups occupy CUs the weight-stationary launch needs: that launch cannot be co-resident) while this process
makes N one-chunk pushes through fvad_engine_run.  Reports the pushes' latency (p50 / p99 / max), how many passes took the
fallback, and whether any push's bits differ from the idle reference (the weight-stationary result and its fallback's are the
two legal outcomes).  python tools/ws_two_process.py [pushes] [child_batch]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    pkg = load_package(); fv = pkg.binding
    ctx = fv.Context(0); ctx.load_synth(7)
    n = int(sys.argv[2])
    f = np.random.default_rng(1).uniform(-11, 2, (n, 54, 161)).astype(np.float32)
    t_end = time.time() + float(sys.argv[3])
    k = 0
    print("CHILD_READY", flush=True)
    while time.time() < t_end:
        ctx.nsnet2_forward(f); k += 1
    print(f"child: {k} passes of {n} sequences", flush=True)
    sys.exit(0)

pushes = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
child_n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
pkg = load_package(); fv = pkg.binding
ctx = fv.Context(0); ctx.load_synth(7)
x = pkg.synth.make_stream(1.0, seed=77)[0][0][:24000].copy()

def push():
    t0 = time.perf_counter()
    o = ctx.engine_run([x], want_denoised=True)[0]
    return time.perf_counter() - t0, o

_, ref = push()
with ctx.options(ws_spin_ticks="0"):
    _, ref_fb = push()                       # the fallback's result (gru_lat arithmetic): the other legal outcome
fb0 = ctx.ws_fallbacks()
idle = np.array([push()[0] for _ in range(200)])
print(f"idle:   p50 {np.median(idle) * 1e3:.3f} ms  p99 {np.percentile(idle, 99) * 1e3:.3f} ms  max {idle.max() * 1e3:.3f} ms", flush=True)
child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(child_n), "60"], stdout=subprocess.PIPE, text=True)
assert "CHILD_READY" in child.stdout.readline()
time.sleep(1.0)
lat, diff = [], 0
for _ in range(pushes):
    dt, o = push()
    lat.append(dt)
    same = all(np.array_equal(o[k], ref[k]) for k in ("denoised", "band_sum", "chunk_rms"))
    same_fb = all(np.array_equal(o[k], ref_fb[k]) for k in ("denoised", "band_sum", "chunk_rms"))
    diff += 0 if (same or same_fb) else 1
lat = np.array(lat)
child.terminate()
print(child.stdout.read().strip())
print(f"shared: p50 {np.median(lat) * 1e3:.3f} ms  p99 {np.percentile(lat, 99) * 1e3:.3f} ms  max {lat.max() * 1e3:.3f} ms  "
      f"fallback passes {ctx.ws_fallbacks() - fb0} of {pushes}  pushes with other bits than the two legal results: {diff}", flush=True)
sys.exit(1 if diff else 0)
