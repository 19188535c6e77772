#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc pass: python tools/pmc_per_kernel.py pmc_counter_collection.csv
One line per kernel (short name): dispatches, summed duration, every counter summed over the kernel's dispatches."""
import collections
import csv
import json
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(int)
seen = set()
meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("mhap::", "")
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (name, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        dur[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        meta[name] = {"vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"), "lds": r.get("LDS_Block_Size"), "scratch": r.get("Scratch_Size"), "wg": r.get("Workgroup_Size")}
disp = collections.Counter(n for n, _ in seen)
for name in sorted(dur, key=lambda n: -dur[n]):
    print(json.dumps({"kernel": name, "dispatches": disp[name], "ms": round(dur[name] / 1e6, 4), **meta[name], "counters": {k: float("%.4g" % v) for k, v in sorted(acc[name].items())}}))
