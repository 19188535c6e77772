"""Randomised end-to-end parity sweep (run on the GPU box): random flags and read mixes, sorted record lines vs the oracle."""
import os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # (test infrastructure: lives under tests/ because it loads the oracle)
import mhap_amd
import oracle_lib as O
from mhap_amd import FastaData, MhapParams, MinHashSearch

def mutate(rnd, s, rate):
    out = []
    for ch in s:
        r = rnd.random()
        if r < rate / 3: continue
        if r < 2 * rate / 3: out.append(rnd.choice("ACGT")); continue
        out.append(ch)
        if r < rate: out.append(rnd.choice("ACGT"))
    return "".join(out)

def main(iters, seed0):
    bad = 0
    for it in range(iters):
        rnd = random.Random(seed0 + it)
        G = "".join(rnd.choice("ACGT") for _ in range(rnd.randrange(6000, 20000)))
        if rnd.random() < 0.5:   # plant a repeat family
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.randrange(20, 400)))
            g = list(G)
            for _ in range(rnd.randrange(2, 12)):
                p = rnd.randrange(0, len(G) - len(unit)); g[p:p + len(unit)] = list(mutate(rnd, unit, 0.02))[:len(unit)]
            G = "".join(g)
        seqs = []
        err = rnd.choice([0.0, 0.02, 0.06, 0.12])
        for _ in range(rnd.randrange(40, 220)):
            L = rnd.randrange(30, 4000); o = rnd.randrange(0, max(1, len(G) - L))
            s = mutate(rnd, G[o:o + L], err)
            if rnd.random() < 0.5: s = O.rc(s)
            if rnd.random() < 0.03: s = s[:len(s) // 2] + "N" * rnd.randrange(1, 5) + s[len(s) // 2:]
            if s: seqs.append(s)
        if rnd.random() < 0.15:   # a few long reads: more k-mers than one LDS table / the register state of the weight kernel holds
            for _ in range(rnd.randrange(1, 4)):
                Lg = rnd.randrange(13000, 40000)
                seqs.append(mutate(rnd, (G * (Lg // len(G) + 2))[rnd.randrange(0, len(G)):][:Lg], 0.05))
        fa = FastaData.from_strings(seqs)
        kw = dict(kmer_size=rnd.choice([16, 16, 16, 12, 14, 18, 21]), num_hashes=rnd.choice([16, 64, 128, 200, 512, 13, 100, 257]),
                  ordered_kmer_size=rnd.choice([12, 12, 8, 10, 13]), ordered_sketch_size=rnd.choice([32, 100, 300, 512, 1536]),
                  num_min_matches=rnd.choice([1, 2, 3, 5]), threshold=rnd.choice([0.0, 0.5, 0.78, 0.9]), max_shift=rnd.choice([0.05, 0.2, 0.4]),
                  min_store_length=rnd.choice([0, 0, 500, 2000]), min_olap_length=rnd.choice([0, 50, 116, 500]))
        kw["repeat_weight"] = rnd.choice([0.9, 0.9, 0.9, 0.5, 1.0, -1.0])
        if os.environ.get("FUZZ_WIDE"):   # the corners beyond the everyday flags (a stream of its own: the default draws stay what they were)
            rw = random.Random(seed0 * 7919 + it)
            if rw.random() < 0.6: kw["num_hashes"] = rw.choice([700, 768, 769, 1024, 1500, 2047, 2752, 3000, 4096])
            if rw.random() < 0.3: kw["ordered_sketch_size"] = rw.choice([1, 2, 2048, 2049, 4096, 8192])
            if rw.random() < 0.2: kw["num_min_matches"] = rw.choice([7, 30, 200])
        p = MhapParams(**kw)
        flt = oflt = None
        if rnd.random() < 0.4:   # a -f filter over the reads' own k-mers: tf-idf weights, optionally the --supress-noise whitelist
            k = p.kmer_size
            counts = {}
            for sq in seqs[::2]:
                for i in range(0, max(0, len(sq) - k + 1), rnd.choice([1, 2, 5])):
                    km = sq[i:i + k]
                    if "N" not in km: counts[km] = counts.get(km, 0) + 1
            items = sorted(counts.items(), key=lambda kv: (-kv[1], kv[0]))[:rnd.choice([20, 300, 3000])]
            if items:
                total = float(sum(c for _, c in items))
                hashes = np.array([int(O.kmer_hashes64(km, k, True)[0]) for km, _ in items], dtype=np.int64)
                fracs = np.array([c / total for _, c in items])
                cutoff = float(np.quantile(fracs, rnd.choice([0.0, 0.5, 0.9])))
                rw = p.repeat_weight
                offset = rw if 0.0 <= rw < 1.0 else 0.0
                rng, no_tf, mode = rnd.choice([3.0, 3.0, 1.0, 7.5]), rnd.random() < 0.3, rnd.choice([0, 0, 1, 2])
                flt = mhap_amd.FrequencyCounts(hashes, fracs, cutoff, offset, rng, no_tf, supress_noise=mode, size_bloom=len(hashes))
                oflt = O.Filter(hashes, fracs, cutoff, offset, rng, no_tf, remove_unique=mode, whitelist=hashes, size_bloom=len(hashes))
                kw["filter"] = (len(items), round(cutoff, 6), rng, no_tf, mode)
        want = O.run_self(fa, k=p.kmer_size, H=p.num_hashes, k2=p.ordered_kmer_size, S=p.ordered_sketch_size, nthreads=8,
                          num_min_matches=p.num_min_matches, min_store_length=p.min_store_length, min_olap_length=p.min_olap_length,
                          threshold=p.threshold, max_shift=p.max_shift, repeat_weight=p.repeat_weight, flt=oflt, cap=1 << 22)
        with MinHashSearch(p, kmer_filter=flt) as ms:
            ms.add_data(fa)
            got = sorted(mhap_amd.records_to_lines(ms.find_matches()))
            st = ms.stats()
        ok = got == O.record_lines(want["records"])
        bad += not ok
        print(("ok  " if ok else "FAIL"), it, len(seqs), "reads err", err, kw, "records", len(got), "slow", st["slow_pairs"], flush=True)
    print("failures:", bad)
    return bad

if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1000) else 0)
