"""Diagnostic: hipGraph (fvad_engine_opts.use_graph) replay against direct launches through model reloads, interleaved direct calls and
host-buffer calls; prints which outputs differ at each step (tests/test_gpu.py asserts the same sequence)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package()
fv = pkg.binding
L = fv.lib()
ctx = fv.Context(0)
ctx.load_synth(7)
dev = torch.device("cuda", 0)
import ctypes as C
GRAPH = False
def run(d_pcm, n_samples, bufs=None):
    n_l = d_pcm.shape[0]
    n_ch = n_samples // 24000
    if bufs is None:
        bufs = (torch.zeros((n_l, n_ch * 24000 // 1024), dtype=torch.float32, device=dev),
                torch.zeros((n_l, n_ch), dtype=torch.float32, device=dev),
                torch.zeros((n_l, n_ch * 24000), dtype=torch.float32, device=dev))
    band, rms, den = bufs
    opts = fv.EngineOpts()
    L.fvad_engine_opts_default(C.byref(opts))
    opts.use_graph = 1 if GRAPH else 0
    fv.check(L.fvad_engine_enqueue_device(ctx.h, d_pcm.data_ptr(), n_l, d_pcm.stride(0), n_samples,
                                          den.data_ptr(), band.data_ptr(), rms.data_ptr(), C.byref(opts)), "enqueue", ctx.h)
    ctx.synchronize()
    return band.cpu().numpy(), rms.cpu().numpy(), den.cpu().numpy()
def cmp(tag, a, b):
    bad = np.argwhere(a[2] != b[2])
    where = f"first bad (lane, sample) {tuple(bad[0])} last {tuple(bad[-1])} n {len(bad)}" if len(bad) else ""
    print(tag, [bool(np.array_equal(u, v)) for u, v in zip(a, b)], where, flush=True)
a, _ = pkg.synth.make_stream(8.0, seed=71)
xa = torch.from_numpy(np.stack([np.roll(a[0], 997 * i) for i in range(6)])[:, : 16 * 24000].copy()).to(dev)
ref_a = run(xa, 16 * 24000)
ref_short = run(xa, 4 * 24000)
bufs = (torch.zeros((6, 16 * 24000 // 1024), dtype=torch.float32, device=dev),
        torch.zeros((6, 16), dtype=torch.float32, device=dev),
        torch.zeros((6, 16 * 24000), dtype=torch.float32, device=dev))
x = xa.clone()
GRAPH = True
cmp("capture", run(x, 16*24000, bufs), ref_a)
cmp("replay", run(x, 16*24000, bufs), ref_a)
cmp("other shape", run(xa, 4*24000), ref_short)
cmp("cached again", run(x, 16*24000, bufs), ref_a)
ctx.load_synth(8)
g8 = run(x, 16*24000, bufs)
cmp("seed8 replay", run(x, 16*24000, bufs), g8)
GRAPH = False
r8 = run(xa, 16*24000)
cmp("seed8 direct", r8, g8)
GRAPH = True
cmp("seed8 replay after direct", run(x, 16*24000, bufs), g8)
cmp("seed8 replay again", run(x, 16*24000, bufs), g8)
many = [a[0][: 2 * 24000].copy() for _ in range(40)]
ctx.engine_run(many)
cmp("seed8 replay after host call", run(x, 16*24000, bufs), g8)
