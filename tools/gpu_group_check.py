"""GPU check of the in-library multi-GPU path on a one-GPU box: N ranks on device 0 (peer transport) and one RCCL rank must
reproduce the single-handle records."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mhap_amd
from mhap_amd import MhapParams, MinHashSearch, MinHashSearchGroup

def sha(recs):
    return hashlib.sha256("\n".join(sorted(mhap_amd.records_to_lines(recs))).encode()).hexdigest()[:16]

n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 4000, 5000
fa = mhap_amd.synth_reads(n, L, seed=77, error_rate=0.12)
q = mhap_amd.synth_reads(500, L, seed=77, error_rate=0.12, shard=1, nshards=3)   # queries drawn from the same genome
q = mhap_amd.FastaData(q.bases, q.offsets, q.lengths, np.arange(len(q), dtype=np.int64) + n + 1)
p = MhapParams(num_hashes=256, ordered_sketch_size=768, device=0)
with MinHashSearch(p) as ms:
    ms.add_data(fa)
    ref = ms.find_matches()
    refq = ms.find_matches_stream(q)
print("single", len(ref), sha(ref), "q", len(refq), sha(refq), flush=True)
ok = True
for N in (1, 2, 3, 5):
    with MinHashSearchGroup(p, n=N, devices=[0] * N) as g:
        t = time.perf_counter()
        g.add_data(fa)
        r = g.find_matches()
        dt = time.perf_counter() - t
        rq = g.find_matches_stream(q)
        st = g.stats()
    good = sha(r) == sha(ref) and sha(rq) == sha(refq)
    ok &= good
    print("group", N, len(r), sha(r), "q", len(rq), sha(rq), "ok" if good else "MISMATCH", f"{dt*1e3:.1f} ms", st["matches_found"], flush=True)
# two batches: the deal continues across calls
with MinHashSearchGroup(p, n=3, devices=[0, 0, 0]) as g:
    h = n // 2 + 1
    g.add_data(fa.subset(np.arange(h))); g.add_data(fa.subset(np.arange(h, n)))
    r = g.find_matches()
    good = sha(r) == sha(ref); ok &= good
    print("group 3, two batches", len(r), "ok" if good else "MISMATCH", flush=True)
# RCCL, one rank
try:
    uid = MinHashSearch.dist_unique_id()
    with MinHashSearch(p) as ms:
        ms.dist_init(0, 1, uid)
        ms.add_data(fa)
        r = ms.dist_find_matches()
        rq = ms.dist_find_matches_stream(q)
        print("rccl 1 rank", len(r), sha(r), "ok" if (sha(r) == sha(ref) and sha(rq) == sha(refq)) else "MISMATCH", ms.dist_last_timing(), flush=True)
        ok &= sha(r) == sha(ref) and sha(rq) == sha(refq)
except Exception as e:
    print("rccl 1 rank FAILED:", e); ok = False
if os.environ.get("MHAP_GROUP_TRANSPORT_TEST"):
    os.environ["MHAP_GROUP_TRANSPORT"] = "rccl"
    try:
        with MinHashSearchGroup(p, n=1, devices=[0]) as g:
            g.add_data(fa); r = g.find_matches()
            print("group rccl n=1", len(r), "ok" if sha(r) == sha(ref) else "MISMATCH"); ok &= sha(r) == sha(ref)
    except Exception as e:
        print("group rccl FAILED:", e); ok = False
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
