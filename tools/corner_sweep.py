#!/usr/bin/env python3
"""Scale corners of the flag space on the GPU box: every scenario in a process of its own (a fault or a hang ends the scenario, not the sweep),
one line each — ok / FAILED, wall time of add and search, records, kernel ms.  No oracle at these sizes: the sweep looks for crashes, hangs
and performance cliffs (round 5 found a memory fault at --num-hashes >= 2 752 and a 185x cliff at >= 1 024 this way); parity at the same flags
is the fuzz sweep's job (tests/fuzz_parity.py, FUZZ_WIDE=1).
  python tools/corner_sweep.py            # all scenarios
  python tools/corner_sweep.py NAME ...   # some"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCEN = {
    # name: (reads, length, params, extra)
    "short_1M_x_300": (1_000_000, 300, dict(), {}),
    "S_8192": (20_000, 10_000, dict(ordered_sketch_size=8192), {}),
    "S_1": (20_000, 10_000, dict(ordered_sketch_size=1), {}),
    "H_2048": (20_000, 10_000, dict(num_hashes=2048), {}),
    "H_8192_small": (1_000, 10_000, dict(num_hashes=8192), {}),
    "H_1": (20_000, 10_000, dict(num_hashes=1, num_min_matches=1), {}),
    "k12_k2_8": (30_000, 10_000, dict(kmer_size=12, ordered_kmer_size=8), {}),
    "k21": (30_000, 10_000, dict(kmer_size=21), {}),
    "min_matches_1": (20_000, 10_000, dict(num_min_matches=1), {}),
    "threshold_0": (20_000, 10_000, dict(threshold=0.0), {}),
    "reads_with_N": (20_000, 10_000, dict(), {"n_frac": 0.5}),
    "long_200_x_1M": (200, 1_000_000, dict(), {}),
    "long_2000_x_100k": (2_000, 100_000, dict(), {}),
    "query_mode": (20_000, 10_000, dict(), {"query": 20_000}),
    "min_store_length": (30_000, 10_000, dict(min_store_length=12_000), {}),
    "max_shift_neg": (20_000, 10_000, dict(max_shift=-0.5), {}),
}

CHILD = r'''
import sys, json, time
import numpy as np
sys.path.insert(0, %r)
import mhap_amd
from mhap_amd import MhapParams, MinHashSearch
reads, length, params, extra = json.loads(sys.argv[1])
fa = mhap_amd.synth_reads(reads, length, seed=11, error_rate=0.12)
if extra.get("n_frac"):
    rng = np.random.default_rng(3)
    pick = rng.random(len(fa)) < extra["n_frac"]
    for i in np.nonzero(pick)[0]:
        o = int(fa.offsets[i]); L = int(fa.lengths[i])
        fa.bases[o + L // 2: o + L // 2 + 3] = ord("N")
out = {}
with MinHashSearch(MhapParams(**params)) as ms:
    t0 = time.perf_counter(); ms.add_data(fa); ms.synchronize(); out["add_s"] = round(time.perf_counter() - t0, 3)
    t0 = time.perf_counter()
    if extra.get("query"):
        q = mhap_amd.synth_reads(extra["query"], length, seed=11, error_rate=0.12, shard=1, nshards=2)
        r = ms.find_matches_stream(q)
    else:
        r = ms.find_matches()
    out["search_s"] = round(time.perf_counter() - t0, 3)
    out["records"] = int(len(r))
    kt = ms.kernel_times(); st = ms.stats()
    out["kernel_ms"] = {k: round(v["ms"], 1) for k, v in kt.items() if v["ms"] > 0}
    out["candidates"] = int(st["candidates_compared"]); out["slow_pairs"] = int(st["slow_pairs"])
print("RESULT " + json.dumps(out))
''' % ROOT


def main():
    names = sys.argv[1:] or list(SCEN)
    bad = 0
    for name in names:
        reads, length, params, extra = SCEN[name]
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, "-c", CHILD, json.dumps([reads, length, params, extra])], capture_output=True, text=True, timeout=float(os.environ.get("SWEEP_TIMEOUT", "600")))
            res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
            if r.returncode == 0 and res:
                print(f"ok     {name:20s} {reads} x {length} {params} {extra}  {res[-1][7:]}", flush=True)
            else:
                bad += 1
                tail = (r.stderr or r.stdout).strip().splitlines()[-3:]
                print(f"FAILED {name:20s} {reads} x {length} {params} {extra}  rc {r.returncode} after {time.time() - t0:.0f} s: {' | '.join(t[:160] for t in tail)}", flush=True)
        except subprocess.TimeoutExpired:
            bad += 1
            print(f"FAILED {name:20s} {reads} x {length} {params} {extra}  TIMEOUT after {time.time() - t0:.0f} s", flush=True)
    print("failed scenarios:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
