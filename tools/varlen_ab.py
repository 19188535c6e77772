import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mhap_amd
from mhap_amd import MhapParams, MinHashSearch
fa = mhap_amd.synth_reads(40000, 12000, seed=11, error_rate=0.15)
rng = np.random.default_rng(3)
# log-normal-ish read lengths between 1.5 kb and 12 kb, a few long ones at the END of the file (worst case for in-order hand-out)
L = np.clip(rng.lognormal(8.3, 0.5, len(fa)).astype(np.int32), 1500, 12000)
L[-200:] = 12000
fa.lengths[:] = L
p = MhapParams()
with MinHashSearch(p) as ms:
    for it in range(3):
        ms.clear(); ms.stage(fa); ms.reset_kernel_times()
        t = time.perf_counter(); ms.add_staged(); ms.synchronize(); dt = time.perf_counter() - t
    kt = ms.kernel_times()
print(os.environ.get("MHAP_NO_LENGTH_ORDER", "ordered"), "add_staged %.1f ms" % (dt * 1e3), {k: round(v["ms"], 2) for k, v in kt.items() if v["ms"] > 0})
