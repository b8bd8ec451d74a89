import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
rng = np.random.default_rng(1)
f = rng.uniform(-11, 2, (n_seq, 54, 161)).astype(np.float32)
ctx = fv.Context(0); ctx.load_synth(7)
g = ctx.nsnet2_forward(f)
ctx.enable_timing(True)
for i in range(3):
    g = ctx.nsnet2_forward(f)
    print({k: round(v, 3) for k, v in ctx.kernel_times().items()}, flush=True)
