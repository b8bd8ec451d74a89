"""Timeline of a step of gru_ws2k_kernel (context option ws2_variant = 64): shader-clock stamps of the first layer-1 and
the first layer-2 workgroup of group 0, averaged over the steps of one pass at BASELINE config 3's 82 chunks.
python tools/ws2_trace.py   (on the GPU box)
Needs the diagnostics build: make -C formula-vad_amd/csrc diag, then FVAD_LIB_PATH=formula-vad_amd/libfvad_hip_diag.so python tools/ws2_trace.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
L = fv.lib()
ctx = fv.Context(0); ctx.load_synth(7)
f = np.random.default_rng(0).uniform(-11, 2, (82, 54, 161)).astype(np.float32)
VAR = 64 + (int(sys.argv[1]) if len(sys.argv) > 1 else 0)
print("ws2_variant", VAR)
with ctx.options(ws2_variant=VAR):
    for _ in range(3):
        ctx.nsnet2_forward(f)
    buf = (C.c_uint32 * 2000)()
    L.fvad_debug_ws_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
    rc = L.fvad_debug_ws_trace(ctx.h, buf, 2000)
    assert rc == 0, rc
tr = np.array(buf, dtype=np.uint32).astype(np.int64)
names = {0: "poll start", 1: "flags seen", 2: "operand in LDS", 3: "W_ih chains done (wave 0)", 4: "product start", 5: "product done",
         6: "gates start", 7: "h stored", 8: "drained / flag", 9: "next h1: poll start", 10: "next h1: flags seen", 11: "next h1: in LDS"}
for layer, label in ((0, "layer 1, workgroup 0"), (1, "layer 2, workgroup 13")):
    t = tr[layer * 1000: layer * 1000 + 54 * 12].reshape(54, 12)
    steps = range(10, 50)
    drained = t[:, 8]
    period = np.mean([(drained[s] - drained[s - 1]) & 0xFFFFFFFF for s in steps])
    print(f"{label}: step period {period:.0f} clocks")
    for k in sorted(names):
        if layer == 0 and k in (3, 9, 10, 11):
            continue
        d = np.mean([((t[s, k] - t[s - 1, 8]) & 0xFFFFFFFF) for s in steps])
        print(f"   {names[k]:28s} {d:8.0f} clocks after the previous step's flag")
