"""Seeded synthetic 48 kHz test audio: a low-passed noise floor ("engine")
plus harmonic bursts in the 500-2000 Hz speech band, with the burst schedule as reference labels.
Input generation only -- nothing here is on the measured path."""
import numpy as np

SAMPLE_RATE = 48000


def _one_pole_lp(x, fc, sr=SAMPLE_RATE):
    # y[n] = a*x[n] + (1-a)*y[n-1], evaluated blockwise with an FFT-free cumulative trick would
    # change rounding; a plain recursive filter via scipy keeps it simple and deterministic.
    from scipy.signal import lfilter
    a = 1.0 - np.exp(-2.0 * np.pi * fc / sr)
    return lfilter([a], [1.0, -(1.0 - a)], x)


def make_stream(seconds, seed, n_channels=1, speech=True, noise_sigma=0.02, peak=0.3):
    """Returns (pcm [n_channels][n] float32, labels [(from_sec, to_sec)])"""
    rng = np.random.default_rng(0xF0A05EED + int(seed))
    n = int(round(seconds * SAMPLE_RATE))
    t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
    sig = np.zeros(n, dtype=np.float64)
    labels = []
    if speech:
        pos = rng.uniform(1.0, 4.0)
        while pos < seconds - 1.0:
            dur = rng.uniform(1.0, 6.0)
            end = min(pos + dur, seconds - 0.5)
            i0, i1 = int(pos * SAMPLE_RATE), int(end * SAMPLE_RATE)
            f0 = rng.uniform(110.0, 220.0)
            tt = t[i0:i1] - t[i0]
            burst = np.zeros(i1 - i0)
            for k in range(4, 9):  # harmonics inside 500-2000 Hz
                burst += np.sin(2 * np.pi * k * f0 * tt + rng.uniform(0, 2 * np.pi)) / 5.0
            am = 0.6 + 0.4 * np.sin(2 * np.pi * 4.0 * tt)  # 4 Hz syllabic modulation
            ramp = np.minimum(1.0, np.minimum(tt, tt[-1] - tt) / 0.02)
            sig[i0:i1] += peak * burst * am * ramp
            labels.append((float(pos), float(end)))
            pos = end + rng.uniform(3.0, 20.0)
    out = np.zeros((n_channels, n), dtype=np.float32)
    for c in range(n_channels):
        noise = _one_pole_lp(rng.normal(0.0, noise_sigma, n), 300.0)
        gain = 1.0 if c == 0 else rng.uniform(0.6, 0.9)  # channels differ a little (vol ratio)
        out[c] = np.clip(gain * sig + noise, -1.0, 1.0).astype(np.float32)
    return out, labels


def labels_to_audacity(labels):
    """formats.zig:47 line format: from TAB to TAB text"""
    return "".join(f"{a:.4f}\t{b:.4f}\tspeech\n" for a, b in labels)
