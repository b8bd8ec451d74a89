"""Regenerates the committed golden vectors from the CPU oracle (oracle/liborc.so) and the
library's seeded synthetic weights / audio generator.  The reference holds no golden vector for
this path and cannot be run here, so these are ORACLE outputs -- parity
unpinned against the reference itself; they pin the oracle and the GPU path against regressions
and against each other.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402
from conftest import load_package  # noqa: E402

pkg = load_package()
fv = pkg.binding
W = fv.synth_weights(7)

rng = np.random.default_rng(1)
win320 = orc.nsnet2_window()
win1024 = orc.hann_periodic(1024)
x320 = rng.uniform(-1, 1, 320).astype(np.float32)
x1024 = rng.uniform(-1, 1, 1024).astype(np.float32)

pcm, _ = pkg.synth.make_stream(1.0, seed=77)
stream = pcm[0][:48000].copy()
p = orc.Pipeline(W, n_channels=1, keep_denoised=True)
p.push(stream[None])
d = orc.Denoiser(W)
d.denoise(stream[:24000])
d.denoise(stream[24000:])

np.savez_compressed(
    os.path.join(HERE, "golden_seed7.npz"),
    win320=win320, win1024=win1024,
    fft320_x=x320, fft320_X=orc.rfft(x320 * win320),
    fft1024_x=x1024, fft1024_X=orc.rfft(x1024 * win1024),
    chunk_features=d.features(), chunk_gains=d.gains(),
    stream_pcm=stream, stream_denoised=p.denoised()[0], stream_band=p.band_volumes()[:, 0],
    stream_rms=p.chunk_rms()[:, 0])

# a longer stream: VAD inputs and the exact segment list (host state machine regression)
pcm, labels = pkg.synth.make_stream(120.0, seed=40)
q = orc.Pipeline(W, n_channels=1)
q.push(pcm)
segs = np.array([(s[0], s[1]) for s in q.segments()], dtype=np.uint64)
np.savez_compressed(
    os.path.join(HERE, "golden_vad_seed40.npz"),
    band=q.band_volumes(), ratio=q.frame_vol_ratio(), segments=segs,
    seg_ratio=np.array([s[2] for s in q.segments()], np.float32),
    seg_met=np.array([s[3] for s in q.segments()], np.float32),
    labels=np.array(labels, np.float64))
print("segments:", segs.tolist())
