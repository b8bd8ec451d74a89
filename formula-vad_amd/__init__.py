"""formula-vad_amd: MI355X (gfx950) implementation of Formula-VAD's spectral front end + NSNet2 +
VAD decision path behind the reference's AudioPipeline / NSNet2 / FFT interface.

The product is csrc/ (HIP kernels + C++ host code) built into libfvad_hip.so with the C ABI of
include/fvad.h; this package is the thin Python plumbing over it (ctypes binding, synthetic input
generator, multi-GPU stream sharding).  The directory name is not a Python identifier; load it
with `importlib` as `formula_vad_amd` (tests/conftest.py, bench.py and __graft_entry__.py do).
"""
from . import binding, synth, shard, simulator  # noqa: F401

__all__ = ["binding", "synth", "shard", "simulator"]
