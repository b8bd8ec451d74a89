"""Accuracy and per-kernel time of the f16x3 matrix path (fvad_ctx_set_nn_math f16x3, kernels_h3.hip) against the f32
MFMA path, float64 numpy and the oracle, on fvad_nsnet2_forward with a large batch.
  python tools/h3_check.py [n_seq]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
import orc
from test_gpu import _nsnet2_float64

def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
    pkg = load_package(); fv = pkg.binding
    w = fv.synth_weights(7)
    rng = np.random.default_rng(21)
    base = rng.uniform(-11, 2, (6, 54, 161)).astype(np.float32)
    g64 = np.stack([_nsnet2_float64(w, s) for s in base])
    g_orc = np.stack([orc.nsnet2_forward(w, s) for s in base])
    f = np.tile(base, (n_seq // 6, 1, 1))
    out = {}
    for math in ("f32", "f16x3", "bf16x3"):
        ctx = fv.Context(0); ctx.load_synth(7); ctx.set_nn_math(math)
        g = ctx.nsnet2_forward(f)
        ctx.enable_timing(True)
        g = ctx.nsnet2_forward(f)
        times = ctx.kernel_times()
        ctx.enable_timing(False)
        e = np.abs(g[-6:] - g64)
        first = np.abs(g[:6] - g64)
        out[math] = g
        print(f"{math:6s} max|g - f64| last6 {e.max():.3e} first6 {first.max():.3e}  rms {np.sqrt((e**2).mean()):.3e}   vs oracle {np.abs(g[-6:] - g_orc).max():.3e}")
        print("       ", "  ".join(f"{k}={v:.3f}" for k, v in times.items()))
        ctx.close()
    print(f"oracle max|g - f64| {np.abs(g_orc - g64).max():.3e};  f16x3 vs f32 path {np.abs(out['f16x3'] - out['f32']).max():.3e};  "
          f"bf16x3 vs f32 path {np.abs(out['bf16x3'] - out['f32']).max():.3e}")

main()
