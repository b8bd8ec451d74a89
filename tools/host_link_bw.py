"""Raw host<->device copy rates on the GPU box (pinned / pageable, both directions) and single-thread memcpy:
the ceilings for the PCIe-inclusive figures in bench.py extras.  python tools/host_link_bw.py"""
import torch, time
n = 256 << 20
d = torch.empty(n, dtype=torch.uint8, device="cuda")
hp = torch.empty(n, dtype=torch.uint8).pin_memory()
hq = torch.ones(n, dtype=torch.uint8)
for name, h in (("pinned", hp), ("pageable", hq)):
    for direction in ("h2d", "d2h"):
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if direction == "h2d": d.copy_(h, non_blocking=True)
            else: h.copy_(d, non_blocking=True)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(name, direction, "%.1f GB/s" % (n / dt / 1e9))
import numpy as np
a = np.ones(n, np.uint8); b = np.empty(n, np.uint8); b[:] = 0
t0 = time.perf_counter(); b[:] = a; dt = time.perf_counter() - t0
print("single-thread memcpy %.1f GB/s" % (n / dt / 1e9))
