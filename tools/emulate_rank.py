"""What ONE rank of an N-GPU run does, measured on one MI355X (communication excluded): sketch + index its 1/N of the reads
(mhap_index_add_staged), then search the forward query rows of ALL ranks against that shard (mhap_find_matches_device with the
toSelf id rules — exactly what mhap_dist_find_matches_self runs after its all-gather).  The other ranks' rows are sketched
beforehand, untimed, as if they had arrived over xGMI.  Prints the per-phase wall times: the compute side of DESIGN.md §5's model.
  python tools/emulate_rank.py [N=8] [config=c2]"""
import os, sys, time, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mhap_amd
from mhap_amd import MhapParams, MinHashSearch, workloads as W
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfgname = sys.argv[2] if len(sys.argv) > 2 else "c2"
cfg = W.CONFIGS[cfgname]
n_total, L, H, S = cfg["reads"], cfg["length"], cfg["hashes"], 1536
p = MhapParams(num_hashes=H, ordered_sketch_size=S, device=0)
dev = torch.device("cuda", 0)
ms = MinHashSearch(p)
rows, ids = [], []
for r in range(world):          # every rank's forward rows (rank 0's own tables are recomputed inside the timed region)
    fa = W.config_reads(cfgname, shard=r, nshards=world)
    n = len(fa)
    mh = torch.zeros((2 * n, H), dtype=torch.int32, device=dev); od = torch.zeros((2 * n, S, 2), dtype=torch.int32, device=dev)
    mt = torch.zeros((2 * n, 4), dtype=torch.int32, device=dev)
    ms.stage(fa); ms.sketch_staged_device(mh.data_ptr(), od.data_ptr(), mt.data_ptr()); ms.synchronize()
    rows.append((mh[0::2].contiguous(), od[0::2].contiguous(), mt[0::2].contiguous())); ids.append(fa.ids.copy())
    if r == 0:
        fa0 = fa
g = tuple(torch.cat([b[i] for b in rows], 0) for i in range(3))
all_ids = np.concatenate(ids)
del rows
torch.cuda.synchronize()
ms.stage(fa0)
res = {}
for it in range(4):
    ms.clear(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms.add_staged(); ms.synchronize()
    t1 = time.perf_counter()
    nrec = len(ms.find_matches_device(g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), all_ids, to_self=True))
    t2 = time.perf_counter()
    kt = ms.kernel_times(); ms.reset_kernel_times()
    res = {"world": world, "config": cfgname, "reads_per_rank": len(fa0), "sketch_and_index_ms": round((t1 - t0) * 1e3, 2),
           "search_all_queries_ms": round((t2 - t1) * 1e3, 2), "rank_step_ms_without_comm": round((t2 - t0) * 1e3, 2), "records_this_rank": nrec,
           "kernel_ms": {k: round(v["ms"], 3) for k, v in kt.items() if v["ms"] > 0},
           "gathered_row_bytes": int(sum(x.numel() * 4 for x in g))}
print(json.dumps(res))
