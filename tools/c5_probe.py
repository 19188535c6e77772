"""BASELINE configs[4] at one rank's size, probed piecewise (run on the GPU box): index N reads of the c5 generator (planted repeat family +
-f filter), then search only the first Q forward queries against the FULL index — per-query costs of the candidate tiers and of the
second stage at 1.25 M entries without forming the whole job's record set.
  python tools/c5_probe.py [reads=625000] [queries=40000] [divergence=0.01] [config=c5rank]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mhap_amd
from mhap_amd import MinHashSearch, workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 625000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
div = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
cfgname = sys.argv[4] if len(sys.argv) > 4 else "c5rank"
cfg = W.CONFIGS[cfgname]
el, sp, _ = cfg["repeats"]
t = time.time()
fa = mhap_amd.synth_reads(n, cfg["length"], seed=cfg["seed"], repeats=(el, sp, div))
t_gen = time.time() - t
tmp = tempfile.mkdtemp(prefix="c5probe_")
fpath = os.path.join(tmp, "kmers.txt")
t = time.time()
nlines = W.write_filter_file(fa, fpath, max_reads=2000)
flt = mhap_amd.FrequencyCounts.from_file(fpath, filter_cutoff=1e-5, repeat_weight=0.9)
t_flt = time.time() - t
p = W.params_for(cfgname, device=0)
out = {"reads": n, "queries": q, "divergence": div, "gen_s": round(t_gen, 2), "filter_lines": nlines, "filter_kept": int(len(flt.hashes)), "filter_s": round(t_flt, 2)}
with MinHashSearch(p, kmer_filter=flt) as ms:
    ms.stage(fa)
    for it in range(2):
        ms.clear(); ms.reset_kernel_times()
        t0 = time.perf_counter(); ms.add_staged(); ms.synchronize(); t1 = time.perf_counter()
        ms.prepare_index(); ms.synchronize(); t2 = time.perf_counter()
        recs = ms.find_matches(0, 2 * q)     # entries [0, 2q): q forward queries
        t3 = time.perf_counter()
    kt = ms.kernel_times(); st = ms.stats()
    out.update({"sketch_ms": round((t1 - t0) * 1e3, 1), "index_build_wall_ms": round((t2 - t1) * 1e3, 1), "search_ms": round((t3 - t2) * 1e3, 1),
                "records": int(len(recs)), "kernel_ms": {k: round(v["ms"], 2) for k, v in kt.items() if v["ms"] > 0},
                "launches": {k: v["launches"] for k, v in kt.items() if v["ms"] > 0},
                "stats": {k: int(v) for k, v in st.items()}})
    c = st["candidates_compared"]
    out["candidates_per_query"] = round(c / max(q, 1), 1)
    out["records_per_query"] = round(len(recs) / max(q, 1), 2)
    out["ns_per_candidate_second_stage"] = round(kt["overlap"]["ms"] * 1e6 / max(c, 1), 2)
    out["us_per_query_candidate_stage"] = round(kt["index_query"]["ms"] * 1e3 / max(q, 1), 2)
print(json.dumps(out))
