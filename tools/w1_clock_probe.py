"""Does the weight-1 MinHash launch of ONE RANK OF EIGHT (25 000 strands, ~10.6 ms) run slower than an eighth of the full launch because of
what the chip did just before it?  The same add (mhap_index_add_staged of rank 0's reads), timed by its kernels' events, after
  idle   : 100 ms of host sleep (the card drops to its idle clocks)
  loop   : the step loop of tools/emulate_rank.py (add, search, add, ...) — what the rank model measures
  b2b    : a second add right behind a first one (clear in between): the chip has just run 10 ms at its power limit
  hot    : behind 3 adds in a row
With MHAP_MINHASH_PROF=1 the kernel prints the mean shader clock it held.   python tools/w1_clock_probe.py [N=8] [config=c2]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mhap_amd
from mhap_amd import MinHashSearch, workloads as W
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfgname = sys.argv[2] if len(sys.argv) > 2 else "c2"
p = W.params_for(cfgname, device=0)
ms = MinHashSearch(p)
fa = W.config_reads(cfgname, shard=0, nshards=world)
ms.stage(fa)
def add():
    ms.clear(); ms.reset_kernel_times()
    t0 = time.perf_counter(); ms.add_staged(); ms.synchronize(); t1 = time.perf_counter()
    kt = ms.kernel_times()
    return round((t1 - t0) * 1e3, 3), round(kt["minhash"]["ms"], 3)
for _ in range(3): add()
out = {}
for mode in ("idle", "b2b", "hot", "idle", "b2b", "hot"):
    r = []
    for rep in range(4):
        if mode == "idle": time.sleep(0.1)
        elif mode == "b2b": time.sleep(0.1); add()
        elif mode == "hot": time.sleep(0.1); add(); add(); add()
        print("== %s" % mode, file=sys.stderr, flush=True)
        r.append(add())
    out.setdefault(mode, []).extend(r)
print(json.dumps({"world": world, "config": cfgname, "strands": 2 * len(fa), "add_wall_ms, minhash_kernel_ms": out}))
