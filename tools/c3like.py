"""C3-like read-length mix (log-normal, median ~8 kb, tail to 50 kb: reads beyond 24.6 kb take the materialised-hash path) on the GPU:
kernel times of the add + self search, and parity against the oracle on a subset when asked (MHAP_C3LIKE_PARITY=1)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mhap_amd
from mhap_amd import MhapParams, MinHashSearch, FastaData
N = int(os.environ.get("MHAP_C3LIKE_READS", "30000"))
LMAX = 50000
fa = mhap_amd.synth_reads(N, LMAX, seed=23, error_rate=0.15, coverage=30.0 * LMAX / 9000)
rng = np.random.default_rng(5)
L = np.clip(rng.lognormal(9.0, 0.55, N).astype(np.int32), 500, LMAX)
fa.lengths[:] = L
print("reads", N, "median", int(np.median(L)), "max", int(L.max()), "reads > 24591:", int((L > 24591).sum()), "Gbases", round(L.sum() / 1e9, 3))
p = MhapParams()
with MinHashSearch(p) as ms:
    for it in range(2):
        ms.clear(); ms.stage(fa); ms.reset_kernel_times()
        t = time.perf_counter(); ms.add_staged(); ms.synchronize(); dt = time.perf_counter() - t
        t = time.perf_counter(); recs = ms.find_matches(); ds = time.perf_counter() - t
    kt = ms.kernel_times()
print("add_staged %.1f ms, search %.1f ms, records %d" % (dt * 1e3, ds * 1e3, len(recs)), {k: round(v["ms"], 2) for k, v in kt.items() if v["ms"] > 0})
if os.environ.get("MHAP_C3LIKE_PARITY"):
    import oracle_lib as O
    sub = fa.subset(np.arange(0, min(N, 1500)))
    with MinHashSearch(p) as ms:
        ms.add_data(sub); got = sorted(mhap_amd.records_to_lines(ms.find_matches()))
    want = O.record_lines(O.run_self(sub, nthreads=16)["records"])
    print("parity on", len(sub), "reads:", got == want, len(got))
