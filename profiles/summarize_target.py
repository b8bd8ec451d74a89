"""Per-kernel summary of one profiling target of tools/profile_round.sh (cfg2, cfg3, calib):
    python3 profiles/summarize_target.py <tag> <target>        reads profiles/<tag>_<target>_kernel_stats.csv and
                                                                 profiles/<tag>_<target>_pmc_*.csv, writes <tag>_<target>_summary.json
Per kernel: calls and mean duration from the --stats run; the mean of every counter of the --pmc runs; and, derived,
  hbm_read_bytes_by_request_size  = 32 RDREQ_32B + 64 RDREQ_64B + 128 RDREQ_128B   (request counts x their sizes: no correction)
  hbm_write_bytes_by_request_size = 64 WRREQ_64B + 32 (WRREQ - WRREQ_64B)
  fetch_size_bytes (FETCH_SIZE x 1024) and its ratio to the request-sized read bytes -- the factor the guide's "double FETCH_SIZE"
  rule stands for, measured on this kernel's own access shape instead of assumed;
  valu_busy = SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES-normalised wave cycles (quad-cycles, the guide's cycle-constants table),
  issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, mfma_busy_frac and clock as in summarize_pmc.py."""
import collections
import csv
import datetime
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
tag, target = sys.argv[1], sys.argv[2]
prefix = os.path.join(here, f"{tag}_{target}")


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


out = {}
stats = prefix + "_kernel_stats.csv"
if os.path.exists(stats):
    for r in csv.DictReader(open(stats)):
        k = short(r["Name"])
        if "rocclr" in k:
            continue
        out.setdefault(k, {}).update(calls=int(r["Calls"]), mean_us=float(r["AverageNs"]) / 1e3, min_us=float(r["MinNs"]) / 1e3,
                                     max_us=float(r["MaxNs"]) / 1e3)
for name in ("fetch_size", "write_size", "rdreq", "wrreq", "mfma_busy", "sq"):
    path = f"{prefix}_pmc_{name}.csv"
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if "rocclr" in k:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if name == "mfma_busy":
            agg[k]["_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in agg.items():
        d = out.setdefault(k, {}).setdefault("counters", {})
        for c, xs in v.items():
            d[c] = sum(xs) / len(xs)
for k, v in out.items():
    c = v.get("counters", {})
    if "TCC_EA0_RDREQ_sum" in c:
        r32, r64, r128 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        v["hbm_read_bytes_by_request_size"] = 32 * r32 + 64 * r64 + 128 * r128
        v["read_requests"] = {"total": c["TCC_EA0_RDREQ_sum"], "32B": r32, "64B": r64, "128B": r128}
    if "TCC_EA0_WRREQ_sum" in c:
        w, w64 = c["TCC_EA0_WRREQ_sum"], c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        v["hbm_write_bytes_by_request_size"] = 64 * w64 + 32 * (w - w64)
    if "FETCH_SIZE" in c:
        v["fetch_size_bytes"] = c["FETCH_SIZE"] * 1024
        if v.get("hbm_read_bytes_by_request_size"):
            v["fetch_size_over_request_sized_bytes"] = v["fetch_size_bytes"] / v["hbm_read_bytes_by_request_size"]
    if "WRITE_SIZE" in c:
        v["write_size_bytes"] = c["WRITE_SIZE"] * 1024
    if "GRBM_GUI_ACTIVE" in c and c.get("_ns"):
        gui = c["GRBM_GUI_ACTIVE"] / 8
        v["clock_GHz"] = gui / c["_ns"]
        v["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024)
    if c.get("SQ_WAVE_CYCLES"):
        v["issue_stall_frac_of_wave_cycles"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        v["waiting_frac_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        v["valu_active_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_BUSY_CYCLES"):
        v["valu_active_per_busy_cycle"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_BUSY_CYCLES"]
    if "mean_us" in v and v.get("hbm_read_bytes_by_request_size") is not None:
        tot = v["hbm_read_bytes_by_request_size"] + v.get("hbm_write_bytes_by_request_size", v.get("write_size_bytes", 0.0))
        v["hbm_GBps_by_request_size"] = tot / (v["mean_us"] * 1e-6) / 1e9
json.dump({"tag": tag, "target": target, "recorded_at": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%M:%SZ"), "kernels": out},
          open(prefix + "_summary.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("mean_us", 0) * kv[1].get("calls", 1)):
    print(f"{k[:58]:58s} calls {v.get('calls', 0):5d}  {v.get('mean_us', 0):9.1f} us  read {v.get('hbm_read_bytes_by_request_size', 0) / 1e6:10.2f} MB  "
          f"write {v.get('hbm_write_bytes_by_request_size', 0) / 1e6:9.2f} MB  FETCH/req {v.get('fetch_size_over_request_sized_bytes', 0):.3f}  "
          f"mfma {v.get('mfma_busy_frac', 0):.3f}  valu/wave {v.get('valu_active_frac_of_wave_cycles', 0):.3f}  stall {v.get('issue_stall_frac_of_wave_cycles', 0):.3f}")
