"""GRU recurrence time per layer for mid-size batches, per kernel shape (to calibrate nn_dispatch.cpp's cost
model).  Run on the GPU box: python tools/gru_crossover.py"""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("formula_vad_amd", os.path.join(ROOT, "formula-vad_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "formula-vad_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["formula_vad_amd"] = pkg; spec.loader.exec_module(pkg)
fv = pkg.binding
ctx = fv.Context(0)
ctx.load_synth(7)
rng = np.random.default_rng(0)
sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 12288, 16384, 24576, 32768]
for n in sizes:
    f = rng.uniform(-11, 2, (n, 54, 161)).astype(np.float32)
    row = []
    for k in ("v6w0", "v5w0", "v4w8", "v3w4", "v3w8", "v3w12", ""):
        ctx.set_option("gru_kernel", k or None)
        try:
            ctx.nsnet2_forward(f)
            ctx.enable_timing(True)
            ctx.nsnet2_forward(f)
            kt = ctx.kernel_times()
            ctx.enable_timing(False)
            row.append(f"{k or 'auto'}={kt.get('gru1_rec', kt.get('gru12_rec_pipelined', float('nan'))):.2f}/{sum(v for a, v in kt.items()):.2f}")
        except Exception as e:
            row.append(f"{k}=ERR")
    print(f"n={n}: gru1 ms / all kernels ms:", "  ".join(row), flush=True)
