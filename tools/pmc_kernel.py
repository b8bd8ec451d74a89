"""Summarise a rocprofv3 --pmc CSV per kernel: mean counter values per dispatch.
usage: python tools/pmc_kernel.py <counter_collection.csv> [kernel-substring]"""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "End_Timestamp" in r:
        agg[k]["_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    print(k)
    for c, xs in sorted(v.items()):
        print(f"   {c:32s} mean {sum(xs)/len(xs):16.1f}  max {max(xs):16.1f}  n {len(xs)}")
