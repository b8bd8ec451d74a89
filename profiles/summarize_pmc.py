"""Per-kernel HBM traffic from the rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE
runs of `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra`; usage:
summarize_pmc.py <tag> <chunks_per_launch>, e.g. `r01b 49152`).
Counter units are KiB.  Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on
gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads, WRITE_SIZE is exact
for 16 B/lane stores -> hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE.  Access shapes here are float4
per lane in 64-byte row segments, which the guide marks uncalibrated: treat as an estimate."""
import collections
import csv
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
out = {}
for name, key in (("fetch_size", "fetch_kib"), ("write_size", "write_kib")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(here, f"{tag}_pmc_{name}.csv"))):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "rocclr" in k:
            continue
        out.setdefault(k, {})[key] = sum(v) / len(v)
# request counts by size, when the round's passes include them (tools/profile_round.sh rdreq / wrreq): bytes = sum of
# count x size, no correction factor -- what settles which of "FETCH_SIZE" and "2 x FETCH_SIZE" is right for a kernel's own
# access shape (round 5: every kernel of this library issues 128-byte read requests almost exclusively, FETCH_SIZE tallies
# them at 64 bytes, and 2 x FETCH_SIZE + WRITE_SIZE equals the request-sized total to three digits)
for name, cols in (("rdreq", ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")),
                   ("wrreq", ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"))):
    path = os.path.join(here, f"{tag}_pmc_{name}.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "rocclr" in k or k not in out:
            continue
        m = {c: sum(x) / len(x) for c, x in v.items()}
        if name == "rdreq":
            out[k]["read_requests"] = {"total": m.get(cols[0], 0.0), "32B": m.get(cols[1], 0.0), "64B": m.get(cols[2], 0.0), "128B": m.get(cols[3], 0.0)}
            out[k]["hbm_read_bytes_by_request_size"] = 32 * m.get(cols[1], 0.0) + 64 * m.get(cols[2], 0.0) + 128 * m.get(cols[3], 0.0)
        else:
            w, w64 = m.get(cols[0], 0.0), m.get(cols[1], 0.0)
            out[k]["hbm_write_bytes_by_request_size"] = 64 * w64 + 32 * (w - w64)
for k, v in out.items():
    v["hbm_bytes_fetch_x2_plus_write"] = (2 * v.get("fetch_kib", 0) + v.get("write_kib", 0)) * 1024
    v["hbm_bytes_per_launch_raw_counters"] = (v.get("fetch_kib", 0) + v.get("write_kib", 0)) * 1024
    if "hbm_read_bytes_by_request_size" in v:
        v["hbm_bytes_by_request_size"] = v["hbm_read_bytes_by_request_size"] + v.get("hbm_write_bytes_by_request_size", v.get("write_kib", 0) * 1024)
        v["hbm_bytes_per_launch"] = v["hbm_bytes_by_request_size"]
        v["hbm_bytes_basis"] = "request counts x request sizes (TCC_EA0_RDREQ_32B/64B/128B, WRREQ_64B): no correction factor"
    else:
        v["hbm_bytes_per_launch"] = v["hbm_bytes_fetch_x2_plus_write"]
        v["hbm_bytes_basis"] = "2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction; confirmed against request sizes in round 5)"
# MFMA busy fraction and clock from the third pass, when present (GRBM_GUI_ACTIVE is summed over the 8
# XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs)
mf = os.path.join(here, f"{tag}_pmc_mfma_busy.csv")
if os.path.exists(mf):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(mf)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in agg.items():
        if "rocclr" in k or k not in out:
            continue
        mean = lambda x: sum(x) / len(x)
        gui = mean(v["GRBM_GUI_ACTIVE"]) / 8
        out[k]["clock_GHz"] = gui / mean(v["ns"])
        out[k]["mfma_busy_frac"] = mean(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / (gui * 1024)
# provenance: bench.py prints these next to roofline.traffic and flags a profile whose kernel sources differ from
# the build it is running (arguments 3 / 4 override: the commit and digest of the build the passes were run on)
import datetime
import hashlib
import subprocess


def _code_only(text):
    """the source without // comments, indentation and blank lines: a comment edit is not a kernel change"""
    out = []
    for line in text.splitlines():
        i = line.find("//")
        if i >= 0 and line[:i].count('"') % 2 == 0:
            line = line[:i]
        line = line.strip()
        if line:
            out.append(line)
    return "\n".join(out).encode()


def _digest(): # bench.py's kernel_source_digest
    h = hashlib.sha256()
    d = os.path.join(here, "..", "formula-vad_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(_code_only(open(os.path.join(d, name), "r", errors="replace").read()))
    return h.hexdigest()[:16]


try:
    commit = sys.argv[3] if len(sys.argv) > 3 else subprocess.check_output(["git", "-C", here, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = None
json.dump({"chunks_per_launch": chunks, "recorded_at": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%M:%SZ"),
           "source_commit": commit, "kernel_source_digest": sys.argv[4] if len(sys.argv) > 4 else _digest(), "kernels": out},
          open(os.path.join(here, f"{tag}_pmc_summary.json"), "w"), indent=1)
for k, v in out.items():
    print(f"{k:45s} {v['hbm_bytes_per_launch'] / 1e9:7.2f} GB/launch ({'request-sized' if 'hbm_bytes_by_request_size' in v else '2xFETCH+WRITE'}), 2xFETCH+WRITE {v['hbm_bytes_fetch_x2_plus_write'] / 1e9:7.2f}, raw FETCH+WRITE {v['hbm_bytes_per_launch_raw_counters'] / 1e9:7.2f}, "
          f"mfma busy {v.get('mfma_busy_frac', 0):.3f}, clock {v.get('clock_GHz', 0):.2f} GHz")
