"""What ONE rank of an N-GPU run does, measured on one MI355X (communication excluded): sketch its 1/N of the reads, index them,
then search the N forward-query bundles against that index (the bundles of the other ranks are sketched beforehand, untimed, as if
they had arrived over xGMI).  Prints the per-phase wall times next to the 1-GPU step: the compute side of DESIGN.md §5's model.
  python tools/emulate_rank.py [N=8] [config=c2]"""
import os, sys, time, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mhap_amd
from mhap_amd import MhapParams, MinHashSearch, workloads as W, distributed as md
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfgname = sys.argv[2] if len(sys.argv) > 2 else "c2"
cfg = W.CONFIGS[cfgname]
n_total, L, H, S = cfg["reads"], cfg["length"], cfg["hashes"], 1536
p = MhapParams(num_hashes=H, ordered_sketch_size=S, device=0)
dev = torch.device("cuda", 0)
n_pad = md.shard_size(n_total, world)
bundles = []
ms = MinHashSearch(p)
for r in range(world):          # every rank's tables (rank 0's are recomputed inside the timed region)
    fa = md.pad_shard(W.config_reads(cfgname, shard=r, nshards=world), n_total, world)
    mh = torch.zeros((2 * n_pad, H), dtype=torch.int32, device=dev); od = torch.zeros((2 * n_pad, S, 2), dtype=torch.int32, device=dev)
    mt = torch.zeros((2 * n_pad, 4), dtype=torch.int32, device=dev)
    ms.stage(fa); ms.sketch_staged_device(mh.data_ptr(), od.data_ptr(), mt.data_ptr()); ms.synchronize()
    bundles.append((md.forward_rows(mh), md.forward_rows(od), md.forward_rows(mt)))
    if r == 0:
        loc = (mh, od, mt); fa0 = fa
lids, lfwd = md.local_entry_ids(n_total, world, 0)
ms.stage(fa0)
res = {}
for it in range(3):
    ms.clear(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms.sketch_staged_device(loc[0].data_ptr(), loc[1].data_ptr(), loc[2].data_ptr()); ms.synchronize()
    t1 = time.perf_counter()
    ms.set_device_index(lids, lfwd, loc[0].data_ptr(), loc[1].data_ptr(), loc[2].data_ptr()); ms.prepare_index()
    t2 = time.perf_counter()
    nrec = 0
    for t in range(world):
        origin = (0 - t) % world
        b = bundles[origin]
        nrec += len(ms.find_matches_device(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), md.bundle_ids(n_total, world, origin), to_self=True))
    t3 = time.perf_counter()
    # the default exchange: every rank's forward rows gathered, ONE search call
    g = tuple(torch.cat([b[i] for b in bundles], 0) for i in range(3))
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    nrec_g = len(ms.find_matches_device(g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), md.all_bundle_ids(n_total, world), to_self=True))
    t5 = time.perf_counter()
    assert nrec_g == nrec
    res = {"world": world, "config": cfgname, "reads_per_rank": n_pad, "sketch_ms": round((t1 - t0) * 1e3, 2), "index_ms": round((t2 - t1) * 1e3, 2),
           "search_ring_bundles_ms": round((t3 - t2) * 1e3, 2), "search_gathered_ms": round((t5 - t4) * 1e3, 2),
           "rank_step_ms_without_comm": round((t2 - t0 + t5 - t4) * 1e3, 2), "rank_step_ms_ring_without_comm": round((t3 - t0) * 1e3, 2), "records_this_rank": nrec,
           "bundle_bytes": int(sum(x.numel() * 4 for x in bundles[0]))}
print(json.dumps(res))
