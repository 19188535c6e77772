"""What ONE rank of an N-GPU run does, measured on one MI355X (communication excluded): sketch + index its 1/N of the reads
(mhap_index_add_staged), then search the forward query rows of ALL ranks against that shard (mhap_find_matches_device with the
toSelf id rules — exactly what mhap_dist_find_matches_self runs after its all-gather).  The other ranks' rows are sketched
beforehand, untimed, straight into the buffers an all-gather would have filled (DESIGN.md §5's memory plan, allocated for real:
`hbm_gb` in the output).  Prints the per-phase wall times: the compute side of DESIGN.md §5's model.
  python tools/emulate_rank.py [N=8] [config=c2|c4|c5|c5rank|...] [iterations=4] [count|keep]"""
import json, os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mhap_amd
from mhap_amd import MinHashSearch, workloads as W
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfgname = sys.argv[2] if len(sys.argv) > 2 else "c2"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cfg = W.CONFIGS[cfgname]
n_total, L, H, S = cfg["reads"], cfg["length"], cfg["hashes"], 1536
p = W.params_for(cfgname, device=0)
flt = None
if cfg["filter"]:      # the -f file every rank loads: k-mer counts of reads 0, stride, 2 stride, ... of the data set (bench.py does the same)
    stride = max(1, n_total // 2000)
    head = W.config_reads(cfgname, shard=0, nshards=stride)
    fpath = os.path.join(tempfile.mkdtemp(prefix="emurank_"), "kmers.txt")
    W.write_filter_file(head, fpath, max_reads=2000)
    flt = mhap_amd.FrequencyCounts.from_file(fpath, filter_cutoff=1e-5, repeat_weight=0.9)
dev = torch.device("cuda", 0)
ms = MinHashSearch(p, kmer_filter=flt)
counts = [len(range(r, n_total, world)) for r in range(world)]
tot = sum(counts)
# the gathered forward rows of all ranks, rank after rank (what the exchange leaves in HBM)
g_mh = torch.empty((tot, H), dtype=torch.int32, device=dev)
g_od = torch.empty((tot, S, 2), dtype=torch.int32, device=dev)
g_mt = torch.empty((tot, 4), dtype=torch.int32, device=dev)
all_ids = np.empty(tot, dtype=np.int64)
nmax = max(counts)
mh = torch.empty((2 * nmax, H), dtype=torch.int32, device=dev); od = torch.empty((2 * nmax, S, 2), dtype=torch.int32, device=dev)
mt = torch.empty((2 * nmax, 4), dtype=torch.int32, device=dev)
off = 0
fa0 = None
for r in range(world):
    fa = W.config_reads(cfgname, shard=r, nshards=world)
    n = len(fa)
    ms.stage(fa); ms.sketch_staged_device(mh.data_ptr(), od.data_ptr(), mt.data_ptr()); ms.synchronize()
    g_mh[off:off + n].copy_(mh[0:2 * n:2]); g_od[off:off + n].copy_(od[0:2 * n:2]); g_mt[off:off + n].copy_(mt[0:2 * n:2])
    all_ids[off:off + n] = fa.ids
    off += n
    if r == 0:
        fa0 = fa
    else:
        del fa
del mh, od, mt
torch.cuda.synchronize()
ms.stage(fa0)
res = {}
runs = []
for it in range(iters):
    ms.clear(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms.add_staged(); ms.synchronize()
    t1 = time.perf_counter()
    # (10^8 records and more per rank — all of configs[4] — are counted, not kept: the mirror's array of them would be what is measured)
    count_only = (sys.argv[4] == "count") if len(sys.argv) > 4 else n_total >= 2_000_000
    r = ms.find_matches_device(g_mh.data_ptr(), g_od.data_ptr(), g_mt.data_ptr(), all_ids, to_self=True, count_only=count_only)
    nrec = int(r) if count_only else len(r)
    t2 = time.perf_counter()
    kt = ms.kernel_times(); ms.reset_kernel_times(); st = ms.stats()
    free, total = torch.cuda.mem_get_info()
    runs.append({"sketch_and_index_ms": (t1 - t0) * 1e3, "search_all_queries_ms": (t2 - t1) * 1e3, "rank_step_ms_without_comm": (t2 - t0) * 1e3,
                 "kernel_ms": {k: v["ms"] for k, v in kt.items() if v["ms"] > 0}})
    res = {"world": world, "config": cfgname, "reads_per_rank": len(fa0), "queries_all_ranks": tot, "records_this_rank": nrec,
           "records_kept_by_the_caller": not count_only,
           "candidates_this_rank": int(st["candidates_compared"]), "slow_pairs": int(st["slow_pairs"]),
           "gathered_row_bytes": int(sum(x.numel() * 4 for x in (g_mh, g_od, g_mt))),
           "hbm_gb": {"in_use": round((total - free) / 2**30, 1), "total": round(total / 2**30, 1)}}
# round 6: the MEDIAN over the iterations after the first (one iteration's search wall moved by 0.3 ms between two runs of this script on one
# box: what the last iteration happened to be was what got printed), the fastest beside it
use = runs[1:] if len(runs) > 1 else runs
med = lambda xs: float(np.median(np.array(xs)))
for key in ("sketch_and_index_ms", "search_all_queries_ms", "rank_step_ms_without_comm"):
    res[key] = round(med([r[key] for r in use]), 2)
res["kernel_ms"] = {k: round(med([r["kernel_ms"].get(k, 0.0) for r in use]), 3) for k in use[0]["kernel_ms"]}
res["fastest_iteration"] = {key: round(min(r[key] for r in use), 2) for key in ("sketch_and_index_ms", "search_all_queries_ms", "rank_step_ms_without_comm")}
res["iterations"] = len(use)
print(json.dumps(res))
