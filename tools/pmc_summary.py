#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.

  python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv \
                              gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv > profiles/r02_pmc_traffic.json

Units and gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB;
FETCH_SIZE counts 128-B requests at 64 B for wide coalesced streaming reads, so it is reported both raw and doubled
(`fetch_bytes_x2`); WRITE_SIZE is uncalibrated and reported raw.  bench.py picks `hbm_bytes_per_launch` =
fetch_bytes_x2 + write_bytes for kernels whose reads are wide coalesced streams, raw fetch otherwise (field `fetch_rule`).
"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WIDE_STREAM = {"candidate_kernel", "hash_kmers_kernel", "ordered_kernel"}   # dwordx2/x4 coalesced row reads


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("mhap::", "")
    base = n.split("<")[0]
    if base == "minhash_w1_kernel":               # the weight-1 strands' kernel is what "minhash_kernel" means in bench.py's roofline
        return "minhash_kernel"
    if base == "minhash_kernel" and "<" in n:      # <U, BITSLICED, WEIGHTED, PROF>: the weight-1 launch and the weighted launch are different kernels
        args = [a.strip() for a in n.split("<", 1)[1].rstrip(">").split(",")]
        if len(args) >= 3 and args[2] in ("true", "1"):
            base += "_weighted"
    return base


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for path in sys.argv[1:]:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, cs in sorted(acc.items()):
        if k.startswith("__amd") or "at::" in k:
            continue
        f = cs.get("FETCH_SIZE", [])
        w = cs.get("WRITE_SIZE", [])
        e = {"launches": max(len(f), len(w))}
        if f:
            e["fetch_bytes_raw"] = sum(f) / len(f) * 1024
            e["fetch_bytes_x2"] = 2 * e["fetch_bytes_raw"]
        if w:
            e["write_bytes"] = sum(w) / len(w) * 1024
        rule = "x2 (wide coalesced stream)" if k in WIDE_STREAM else "raw (narrow/gather accesses: uncalibrated)"
        e["fetch_rule"] = rule
        fb = e.get("fetch_bytes_x2" if k in WIDE_STREAM else "fetch_bytes_raw", 0.0)
        e["hbm_bytes_per_launch"] = fb + e.get("write_bytes", 0.0)
        out[k] = e
    # bench.py's kernel-time slots that are several kernels: bytes of one pass through all of them
    groups = {"index_build_kernel": ("index_tile_kernel", "index_offsets_kernel", "index_bins_kernel"),
              "index_query_kernel": ("index_query_kernel", "index_query_dense_kernel")}
    for gname, members in groups.items():
        passes = max([out[m]["launches"] for m in members[-1:] if m in out] or [0]) if gname == "index_build_kernel" else \
            max([out[m]["launches"] for m in members[:1] if m in out] or [0])
        if not passes:
            continue
        tot = {"fetch_bytes_raw": 0.0, "write_bytes": 0.0}
        for m in members:
            if m in out:
                for fkey in tot:
                    tot[fkey] += out[m].get(fkey, 0.0) * out[m]["launches"]
        if gname == "index_query_kernel" and gname in out:
            out["index_query_tier1_kernel"] = dict(out[gname])
        e = {"launches": passes, "fetch_bytes_raw": tot["fetch_bytes_raw"] / passes, "fetch_bytes_x2": 2 * tot["fetch_bytes_raw"] / passes,
             "write_bytes": tot["write_bytes"] / passes, "fetch_rule": "raw (narrow/gather accesses: uncalibrated); sum over " + " + ".join(members)}
        e["hbm_bytes_per_launch"] = e["fetch_bytes_raw"] + e["write_bytes"]
        out[gname] = e
    from mhap_amd import build as mbuild      # the byte counts belong to the kernels of exactly these sources (bench.py checks the stamp)
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0, 100k x 10kb",
               "source_digest": mbuild.source_digest(), "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
