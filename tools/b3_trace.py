"""Diagnostic: name every NSNet2 stage of the bf16x3 path on stderr (option trace_kernels) for a small batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
ctx = fv.Context(0); ctx.load_synth(7)
ctx.set_nn_math(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
ctx.set_option("trace_kernels", "1")
f = np.random.default_rng(0).uniform(-11, 2, (int(sys.argv[2]) if len(sys.argv) > 2 else 20, 54, 161)).astype(np.float32)
g = ctx.nsnet2_forward(f)
print("path", ctx.last_nn_path(), "finite", bool(np.all(np.isfinite(g))), g.min(), g.max(), flush=True)
