"""Per-kernel device times (HIP events around every launch) of one device-resident launch sequence of lanes x chunks_per_lane chunks,
and a digest of the denoised audio.  python tools/launch_times.py [lanes=64] [chunks_per_lane=16] [option=value ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cpl = int(sys.argv[2]) if len(sys.argv) > 2 else 16
CH = 24000
ctx = fv.Context(0); ctx.load_synth(7)
for kv in sys.argv[3:]:
    k, v = kv.split("=", 1)
    ctx.set_option(k, v)
rng = np.random.default_rng(5)
pcm = (rng.standard_normal((lanes, cpl * CH), dtype=np.float32) * 0.1)
n_samp = cpl * CH
d_pcm = ctx.device_alloc(pcm.nbytes); ctx.to_device(d_pcm, pcm)
d_den = ctx.device_alloc(pcm.nbytes)
d_band = ctx.device_alloc(lanes * (n_samp // 1024) * 4); d_rms = ctx.device_alloc(lanes * cpl * 4)
ctx.enqueue_device(d_pcm, lanes, n_samp, n_samp, d_den, d_band, d_rms)
ctx.synchronize()
ctx.enable_timing(True)
reps = 5
for _ in range(reps):
    ctx.enqueue_device(d_pcm, lanes, n_samp, n_samp, d_den, d_band, d_rms)
kt = ctx.kernel_times()
ctx.enable_timing(False)
print(f"{lanes * cpl} chunks [{ctx.last_nn_path()}]")
for k, v in kt.items():
    print(f"   {k:28s} {v / reps * 1e3:9.1f} us")
print(f"   {'sum':28s} {sum(kt.values()) / reps * 1e3:9.1f} us")
import hashlib
den = np.empty_like(pcm); ctx.to_host(den, d_den)
print("   digest of the denoised audio:", hashlib.sha256(den.tobytes()).hexdigest()[:16])
