"""Where a row tile of gru_ws2m_kernel goes: the recurrence at 256 and 1024 sequences as it is and as timing-only variants
(diagnostics build: FVAD_LIB_PATH=formula-vad_amd/libfvad_hip_diag.so): 4096 no fetch behind a step's first row tile, 8192 no
gate math, 16384 no products.  python tools/ws2m_ab.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
ctx = fv.Context(0); ctx.load_synth(7)
rng = np.random.default_rng(0)
for n in (256, 1024):
    f = rng.uniform(-11, 2, (n, 54, 161)).astype(np.float32)
    for v in ([int(a) for a in sys.argv[1:]] or (0, 4096, 8192, 16384, 4096 + 16384, 4096 + 8192 + 16384)):
        with ctx.options(ws2_variant=str(v)):
            ctx.nsnet2_forward(f)
            ctx.enable_timing(True)
            for _ in range(10):
                ctx.nsnet2_forward(f)
            kt = ctx.kernel_times()
            ctx.enable_timing(False)
        rec = sum(x for k, x in kt.items() if "rec" in k) / 10
        print(f"n={n:5d} variant {v:6d}: recurrence {rec * 1e3:8.1f} us = {rec * 1e6 / 55 * 2.1:8.0f} clocks per step at 2.1 GHz", flush=True)
