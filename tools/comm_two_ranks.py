"""Rehearsal of the native RCCL statistics gather with several ranks (normally one rank per GPU; on a one-GPU box
RCCL refuses two ranks on one device, which this script reports).  python tools/comm_two_ranks.py [world]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
import torch
ndev = torch.cuda.device_count()
ctx = fv.Context(RANK % max(ndev, 1))
if RANK == 0:
    open(IDFILE + ".tmp", "wb").write(fv.comm_unique_id()); os.rename(IDFILE + ".tmp", IDFILE)
while not os.path.exists(IDFILE):
    time.sleep(0.05)
uid = open(IDFILE, "rb").read()
comm = fv.Comm(ctx, uid, WORLD, RANK)
n_streams = 21
ids = pkg.shard.streams_for_rank(n_streams, RANK, WORLD)
stats = []
for i in ids:
    st = fv.SingleStats()
    for j, (name, _) in enumerate(fv.SingleStats._fields_):
        setattr(st, name, float(100 * i + j))
    stats.append(st)
out = comm.allgather_stats(ids, stats, n_streams)
for i, o in enumerate(out):
    assert o.total_positives_sec == 100.0 * i and o.f_score_beta == 100.0 * i + 10, (RANK, i)
comm.close(); ctx.close()
print("rank", RANK, "ok", flush=True)
'''
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
idfile = os.path.join(tempfile.mkdtemp(), "uid")
procs = [subprocess.Popen([sys.executable, "-c", f"ROOT={ROOT!r}\nRANK={r}\nWORLD={world}\nIDFILE={idfile!r}\n" + WORKER],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
rc = 0
for r, p in enumerate(procs):
    try:
        out, _ = p.communicate(timeout=120)
    except subprocess.TimeoutExpired:
        p.kill(); out, _ = p.communicate(); out += b"\n[timeout]"
    print(f"--- rank {r} rc={p.returncode}\n" + out.decode()[-1500:])
    rc |= p.returncode or 0
sys.exit(1 if rc else 0)
