"""gru_ws2k: the delay of a step's first flag poll (context option ws2_waits = layer 1 | layer 2 << 16, in 10 ns ticks) against the time of the
pipelined recurrence at 1 and 82 chunks, then what the context option ws2_calibrate picks on this box.  python tools/ws2_delay.py [n ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
ctx = fv.Context(0); ctx.load_synth(7)
rng = np.random.default_rng(0)
for n in [int(a) for a in sys.argv[1:]]:
    f = rng.uniform(-11, 2, (n, 54, 161)).astype(np.float32)
    ctx.nsnet2_forward(f)
    ctx.enable_timing(True)
    for _ in range(6):
        ctx.nsnet2_forward(f)
    kt = ctx.kernel_times()
    ctx.enable_timing(False)
    print(f"n={n:3d} built-in waits (the default): {sum(v for k, v in kt.items() if 'rec' in k) / 6 * 1e3:6.1f} us", flush=True)
    fine = os.environ.get("WS2_DELAY_FINE")
    d1s = (0, 20, 30, 40, 50, 60, 70, 80) if not fine else ((28, 32, 36, 40, 44, 48) if n > 80 else (40, 45, 50, 55, 60, 65))
    d2s = (0, 20, 30, 40, 50, 60, 70, 80, 100) if not fine else ((48, 54, 60, 66, 72, 78) if n > 80 else (40, 45, 50, 55, 60, 65))
    for d1 in d1s:
        row = []
        for d2 in d2s:
            with ctx.options(ws2_waits=(4 * d1) | ((4 * d2) << 16)):
                ctx.nsnet2_forward(f)
                ctx.enable_timing(True)
                for _ in range(6):
                    ctx.nsnet2_forward(f)
                kt = ctx.kernel_times()
                ctx.enable_timing(False)
            rec = sum(v for k, v in kt.items() if "rec" in k) / 6
            row.append(f"{rec * 1e3:6.1f}")
        print(f"n={n:3d} layer-1 wait {d1 * 0.04:4.2f} us | layer-2 waits " + ", ".join(f"{d * 0.04:.2f}" for d in d2s) + " us: " + " ".join(row), flush=True)

import time
t0 = time.perf_counter()
ctx.set_option("ws2_calibrate", 1)
print(f"ws2_calibrate took {(time.perf_counter() - t0) * 1e3:.0f} ms; waits in effect (layer 1, layer 2; 10 ns ticks) by class: "
      + ", ".join(f"{c}: {ctx.ws2_waits(c)}" for c in (1, 2, 3)), flush=True)
def rec_us(f, reps=20):
    ctx.nsnet2_forward(f)
    ctx.enable_timing(True)
    for _ in range(reps):
        ctx.nsnet2_forward(f)
    kt = ctx.kernel_times()
    ctx.enable_timing(False)
    return sum(v for k, v in kt.items() if "rec" in k) / reps * 1e3


for n, cls in ((1, 1), (82, 3)):   # the table's entry against the measured one, alternating, in one process
    f = rng.uniform(-11, 2, (n, 54, 161)).astype(np.float32)
    cal = ctx.ws2_waits(cls)
    ctx.set_option("ws2_calibrate", 0)
    tab = ctx.ws2_waits(cls)
    for rnd in range(3):
        row = []
        for name, w in (("table", tab), ("calibrated", cal)):
            with ctx.options(ws2_waits=w[0] | (w[1] << 16)):
                row.append(f"{name} {w}: {rec_us(f):6.1f} us")
        print(f"n={n:3d} round {rnd}: " + "   ".join(row), flush=True)
    ctx.set_option("ws2_calibrate", 1)
