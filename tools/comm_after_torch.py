"""Does the library's RCCL gather work in a process that has imported torch first (bench.py's situation: torch brings
its own copies of the ROCm runtime and of RCCL)?  One-rank communicator through the C ABI.  python tools/comm_after_torch.py"""
import os, sys
import torch
torch.cuda.set_device(0)
x = torch.zeros(4, device="cuda")            # torch's HIP runtime is initialised
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
ctx = fv.Context(0)
comm = fv.Comm(ctx, fv.comm_unique_id(), 1, 0)
st = fv.SingleStats(); st.total_positives_sec = 3.0
out = comm.allgather_stats([0], [st], 1)
print("COMM_AFTER_TORCH_OK", out[0].total_positives_sec, [l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l or "hsa-runtime" in l][:6])
