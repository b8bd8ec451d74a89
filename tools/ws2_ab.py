"""A/B of the pipelined recurrence's options on ONE box (boxes differ by several per cent): python tools/ws2_ab.py [variant ...]
times fvad_nsnet2_forward's recurrence (+ the input-projection GEMM where one runs) at 82 sequences, 64 and 1 for each value
of the context option ws2_variant given (default: 0 1024).  Bits: 1024 layer 1's input projection from a GEMM in front
(round 3's form), 16 groups of 13 + 25 where 25 + 25 fit, 8 the 8-wavefront kernel;
with the diagnostics build (FVAD_LIB_PATH=formula-vad_amd/libfvad_hip_diag.so) also 256 no input rows, 512 no input
projection (timing only: wrong results)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
ctx = fv.Context(0); ctx.load_synth(7)
variants = [int(a) for a in sys.argv[1:]] or [0, 1024]
rng = np.random.default_rng(0)
REPS = 30
for n in (82, 64, 1):
    f = rng.uniform(-11, 2, (n, 54, 161)).astype(np.float32)
    base = None
    for rnd in range(2):                      # two rounds: drift of the box shows as a difference between them
        for v in variants:
            with ctx.options(ws2_variant=str(v)):
                g = ctx.nsnet2_forward(f)
                ctx.enable_timing(True)
                for _ in range(REPS):
                    ctx.nsnet2_forward(f)
                kt = ctx.kernel_times()
                ctx.enable_timing(False)
                path = ctx.last_nn_path()
            rec = sum(x for k, x in kt.items() if "rec" in k) / REPS
            gem = kt.get("gru1_in_gemm_fc1folded", 0.0) / REPS
            if base is None:
                base = g
            same = "same bits as first" if np.array_equal(g, base) else f"max |d| vs first {np.abs(g - base).max():.1e}"
            print(f"n={n:3d} round {rnd} variant {v:5d}: recurrence {rec * 1e3:7.1f} us + gi1 GEMM {gem * 1e3:5.1f} us = {(rec + gem) * 1e3:7.1f} us "
                  f"({rec * 1e3 / 55:.2f} us/step)  fallbacks {ctx.ws_fallbacks()}  {same}  [{path.split('+')[-1].strip()[:40]}]", flush=True)
