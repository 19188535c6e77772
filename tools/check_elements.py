"""Independent count of the "table elements processed" statistic (MinHashSearch.java:173): for every query (forward strand) and
MinHash slot, the stored entries with the same value — from the exported MinHash rows with numpy, against the library's number.
    python tools/check_elements.py [c5slice|c2|...] [reads]"""
import sys, os, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhap_amd
from mhap_amd import MhapParams, MinHashSearch, workloads as W

name = sys.argv[1] if len(sys.argv) > 1 else "c5slice"
cfg = W.CONFIGS[name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["reads"]
fa = W.config_reads(name, shard=0, nshards=1, reads=n, length=cfg["length"], error_rate=0.15)
flt = None
if cfg["filter"]:
    path = os.path.join(tempfile.mkdtemp(), "kmers.txt")
    W.write_filter_file(fa, path, max_reads=2000)
    flt = mhap_amd.FrequencyCounts.from_file(path, filter_cutoff=1e-5, repeat_weight=0.9)
p = MhapParams(num_hashes=cfg["hashes"])
with MinHashSearch(p, kmer_filter=flt) as ms:
    ms.add_data(fa)
    recs = ms.find_matches()
    st = ms.stats()
    want = 0
    ne = ms.size()
    CH = 20000
    cols = []
    status, is_fwd = [], []
    for f in range(0, ne, CH):
        sk = ms.export(f, min(CH, ne - f))
        cols.append(sk["minhash"].copy()); status.append(sk["status"].copy()); is_fwd.append(sk["is_fwd"].copy())
    mh = np.concatenate(cols); status = np.concatenate(status); is_fwd = np.concatenate(is_fwd)
    stored = status == 0
    query = stored & (is_fwd != 0)
    for s in range(mh.shape[1]):
        vals, inv, cnt = np.unique(mh[stored, s], return_inverse=True, return_counts=True)
        qv = mh[query, s]
        pos = np.searchsorted(vals, qv)
        want += int(cnt[pos].sum())          # every query value is a stored value (its own entry)
print({"config": name, "reads": n, "entries": int(ne), "records": len(recs), "table_elements": int(st["table_elements"]),
       "numpy_count": want, "equal": int(st["table_elements"]) == want, "splits": int(st["index_splits"])})
