"""K1 / K3 / K4 (stft320_logpow, istft320_ola_up3, fft1024_bandsum) device times of one device-resident launch sequence, and the
bits of everything the pipeline returns as one digest (to compare two builds of the spectral kernels on the same box).
python tools/spectral_times.py [lanes=128] [chunks_per_lane=128]"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cpl = int(sys.argv[2]) if len(sys.argv) > 2 else 128
CH = 24000
ctx = fv.Context(0); ctx.load_synth(7)
rng = np.random.default_rng(5)
one = (rng.standard_normal(lanes * cpl * CH // 8, dtype=np.float32) * 0.1)
pcm = np.tile(one, 8).reshape(lanes, cpl * CH)
pcm *= np.linspace(0.5, 1.5, lanes, dtype=np.float32)[:, None]
n_samp = cpl * CH
n_fr = n_samp // 1024
d_pcm = ctx.device_alloc(pcm.nbytes); ctx.to_device(d_pcm, pcm)
d_den = ctx.device_alloc(pcm.nbytes)
d_band = ctx.device_alloc(lanes * n_fr * 4); d_rms = ctx.device_alloc(lanes * cpl * 4)
ctx.enqueue_device(d_pcm, lanes, n_samp, n_samp, d_den, d_band, d_rms)
ctx.synchronize()
ctx.enable_timing(True)
reps = 3
for _ in range(reps):
    ctx.enqueue_device(d_pcm, lanes, n_samp, n_samp, d_den, d_band, d_rms)
kt = ctx.kernel_times()
ctx.enable_timing(False)
spectral = {k: v / reps for k, v in kt.items() if k in ("stft320_logpow", "istft320_ola_up3", "fft1024_bandsum")}
scale = 49152 / (lanes * cpl)
print(f"{lanes * cpl} chunks: " + ", ".join(f"{k} {v:.3f} ms" for k, v in spectral.items())
      + f" | sum {sum(spectral.values()):.3f} ms (x {scale:.2f} = {sum(spectral.values()) * scale:.2f} ms at 49152 chunks)  [{ctx.last_nn_path()}]")
den = np.empty_like(pcm); band = np.empty((lanes, n_fr), np.float32); rms = np.empty((lanes, cpl), np.float32)
ctx.to_host(den, d_den); ctx.to_host(band, d_band); ctx.to_host(rms, d_rms)
h = hashlib.sha256()
for a in (den, band, rms):
    h.update(a.tobytes())
print("digest of denoised audio + band sums + chunk RMS:", h.hexdigest()[:16], " finite:", bool(np.isfinite(den).all() and np.isfinite(band).all()))
