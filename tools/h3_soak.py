"""Determinism soak of the f16x3 kernels: the same large batch through fvad_nsnet2_forward N times must give the
same bits every time (a race on an LDS slab or a missed wait would show as a difference sooner or later), and the
same sequence placed in every 16-sequence group of the batch must come out identical.
  python tools/h3_soak.py [n_seq] [repeats]"""
import os, sys, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package

def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    fv = load_package().binding
    rng = np.random.default_rng(5)
    base = rng.uniform(-11, 2, (192, 54, 161)).astype(np.float32)
    f = np.tile(base, (n_seq // 192, 1, 1))
    ctx = fv.Context(0); ctx.load_synth(7)
    first = None
    for r in range(reps):
        g = ctx.nsnet2_forward(f)
        h = hashlib.sha256(g.tobytes()).hexdigest()[:16]
        same_groups = bool(np.array_equal(g.reshape(-1, 192, 54, 161), np.broadcast_to(g[:192], (n_seq // 192, 192, 54, 161))))
        print(f"run {r}: sha {h} groups identical {same_groups}", flush=True)
        if first is None:
            first = h
        if h != first or not same_groups:
            print("MISMATCH"); sys.exit(1)
    print("ok")

main()
