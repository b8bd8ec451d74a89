"""Where a step of the pipelined two-layer recurrence goes: gru_ws2k_kernel (one row tile per group: K split over 16
wavefronts) and gru_ws2_kernel (8 wavefronts; ws2_variant bit 8 forces it) timed as they are and as timing-only variants
(context option ws2_variant; the variants give WRONG results) at BASELINE config 3's 82 chunks and at one chunk,
next to gru_ws (two launches + layer 2's input projection).  Run on the GPU box: python tools/ws2_variants.py
Needs the diagnostics build: make -C formula-vad_amd/csrc diag, then FVAD_LIB_PATH=formula-vad_amd/libfvad_hip_diag.so python tools/ws2_variants.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
ctx = fv.Context(0); ctx.load_synth(7)
rng = np.random.default_rng(0)
for n in (82, 64, 1):
    f = rng.uniform(-11, 2, (n, 54, 161)).astype(np.float32)
    for name, opts in (("gru_ws2 (16 waves, K split)", {}), ("  no input projection in layer 2", {"ws2_variant": 1}), ("  no row-major h2 store", {"ws2_variant": 2}),
                       ("  layer 1 alone", {"ws2_variant": 4}),
                       ("  layer 2: h1 fetched, no W_ih product", {"ws2_variant": 32}),
                       ("  13 + 25 workgroups per group (the shape of six row tiles)", {"ws2_variant": 16}),
                       ("gru_ws2 (8 waves)", {"ws2_variant": 8}), ("  no input projection in layer 2", {"ws2_variant": 9}),
                       ("  layer 1 alone", {"ws2_variant": 12}), ("gru_ws (2 launches + GEMM)", {"gru_kernel": "v5w0"})):
        with ctx.options(**opts):
            ctx.nsnet2_forward(f)
            ctx.enable_timing(True)
            for _ in range(5):
                ctx.nsnet2_forward(f)
            kt = ctx.kernel_times()
            ctx.enable_timing(False)
        rec = sum(v for k, v in kt.items() if "rec" in k or k == "gru2_in_gemm") / 5
        print(f"n={n:3d} {name:62s} recurrences {rec * 1e3:7.1f} us  ({rec * 1e3 / 55:.2f} us per pipelined step)", flush=True)
