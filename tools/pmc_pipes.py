#!/usr/bin/env python3
"""Per-kernel pipe counters of a C2 step, per launch, from rocprofv3 --pmc passes (tools/profile_round.sh collects them in two passes of
eight counters): what bench.py's `kernel_ceilings` are computed from (every kernel gets a stated ceiling).

  python tools/pmc_pipes.py pass1/pmc_counter_collection.csv pass2/pmc_counter_collection.csv > profiles/r05_pmc_pipes.json

Counter semantics on gfx950 as observed in these passes (rocprofv3 sums a counter over its instances): GRBM_GUI_ACTIVE is summed over the
8 XCDs (cycles); SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT are cycles summed over the 256 CUs; SQ_INSTS_* are wave-level instructions;
SQ_WAVE_CYCLES / SQ_BUSY_CYCLES count quad-cycles.  Derived per kernel:
  lds_pipe_frac     = SQ_LDS_IDX_ACTIVE / (256 * GRBM_GUI_ACTIVE / 8)        share of the kernel's cycles a CU's LDS pipe is busy
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE               share of those cycles lost to bank conflicts
The VALU side needs the kernel's live time (bench.py): SQ_INSTS_VALU per launch / time against the 1.05e12 wave-instructions/s the chip
sustains on full-rate VALU work at any occupancy (tools/issue_probe.hip, profiles/r05_issue_probe.txt: 6.7e13 lane-ops/s)."""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    n = re.sub(r"\(.*", "", name).replace("void ", "").replace("mhap::", "")
    return n.split("<")[0] if not n.startswith("minhash_kernel<") else ("minhash_kernel_weighted" if re.match(r"minhash_kernel<\d+, \w+, (true|1)", n) else "minhash_kernel")


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    meta = {}
    for path in sys.argv[1:]:
        seen = set()
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k.startswith("__amd") or "at::" in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (path, k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                disp[(path, k)].add(r["Dispatch_Id"])
                dur[(path, k)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                meta[k] = {"vgpr": int(r.get("VGPR_Count") or 0), "sgpr": int(r.get("SGPR_Count") or 0), "scratch_bytes_per_lane": int(r.get("Scratch_Size") or 0),
                           "lds_static_bytes": int(r.get("LDS_Block_Size") or 0), "workgroup": int(r.get("Workgroup_Size") or 0)}
    out = {}
    for k in sorted(acc):
        n = max(len(v) for (p, kk), v in disp.items() if kk == k)
        ms = max(dur[(p, kk)] / len(disp[(p, kk)]) for (p, kk) in disp if kk == k)
        c = {name: v / n for name, v in acc[k].items()}
        e = dict(meta[k], launches=n, ms_under_pmc=round(ms, 4), per_launch={name: float("%.5g" % v) for name, v in sorted(c.items())})
        gui = c.get("GRBM_GUI_ACTIVE")
        if gui and "SQ_LDS_IDX_ACTIVE" in c:
            e["lds_pipe_frac"] = round(c["SQ_LDS_IDX_ACTIVE"] / (256.0 * gui / 8.0), 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        out[k] = e
    from mhap_amd import build as mbuild
    json.dump({"source": "rocprofv3 --pmc, two passes of eight counters, bench.py --steps 1 --warmup 0 (C2: 100k x 10kb)",
               "source_digest": mbuild.source_digest(), "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
