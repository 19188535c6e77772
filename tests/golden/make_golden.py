"""Regenerates tests/golden/small_reads.{fasta,json}.

The fixture is ORACLE-DERIVED, not JVM-verified: the reference (Java) cannot run in the build image.
`verify_against_jar.sh` in this directory re-derives the records with a real mhap.jar when a JVM exists.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as O  # noqa: E402
from mhap_amd import FastaData  # noqa: E402

PARAMS = {"k": 16, "H": 64, "k2": 12, "S": 256, "min_olap_length": 116}


def main():
    rnd = random.Random(20260928)
    genome = "".join(rnd.choice("ACGT") for _ in range(5000))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    lines = []
    for i in range(30):
        L = rnd.randint(900, 1400)
        st = rnd.randint(0, len(genome) - L)
        s = list(genome[st:st + L])
        for j in range(L):
            r = rnd.random()
            if r < 0.01:
                s[j] = rnd.choice("ACGT")
            elif r < 0.015:
                s[j] = ""
            elif r < 0.02:
                s[j] = s[j] + rnd.choice("ACGT")
        s = "".join(s)
        if rnd.random() < 0.5:
            s = "".join(comp[c] for c in reversed(s))
        if i == 3:
            s = s[:80]                          # shorter than --min-olap-length: skipped, still consumes an id
        if i == 5:
            s = s[:300] + "NNNNRYNN" + s[300:]  # IUPAC codes hashed as-is
        if i == 7:
            s = s.lower()                       # upper-cased by the reader
        lines.append(f">read{i} some description")
        w = rnd.choice([60, 70, 100000])
        for o in range(0, len(s), w):
            lines.append(s[o:o + w])
    path = os.path.join(HERE, "small_reads.fasta")
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    fa = FastaData.from_file(path)
    p = PARAMS
    res = O.run_self(fa, k=p["k"], H=p["H"], k2=p["k2"], S=p["S"], min_olap_length=p["min_olap_length"], nthreads=2, want_minhash=True)
    out = {"note": "oracle-derived, not JVM-verified", "params": p, "n_reads": len(fa),
           "sorted_records": O.record_lines(res["records"]), "minhash_first4": res["minhash"][:4].tolist(),
           "status": res["status"].tolist()}

    # ---- -q mode: queries with reads below --min-olap-length (they take an id but do not advance the id offset of later files,
    # MhapMain.java:462,537) against the index above, toSelf = false, expectation from the oracle's primitives ----
    qlines, qseqs = [], []
    for i in range(12):
        L = rnd.randint(700, 1200)
        st = rnd.randint(0, len(genome) - L)
        s = genome[st:st + L]
        if rnd.random() < 0.5:
            s = "".join(comp[c] for c in reversed(s))
        if i in (0, 6):
            s = s[:90]
        qlines += [f">query{i}", s]
        qseqs.append(s)
    with open(os.path.join(HERE, "small_queries.fasta"), "w") as fh:
        fh.write("\n".join(qlines) + "\n")
    n_sketched = int((res["status"][0::2] == 0).sum())             # = seqStreamer.getNumberProcessed()/2
    ent = []
    for i in range(len(fa)):
        sq = fa.sequence(i)
        if len(sq) < p["min_olap_length"]:
            continue
        for fwd, t in ((1, sq), (0, O.rc(sq))):
            ent.append((int(fa.ids[i]), fwd, len(sq), O.minhash(t, p["k"], p["H"])[1], O.ordered(t, p["k2"], p["S"])))
    qrecs = []
    for qi, sq in enumerate(qseqs):
        if len(sq) < p["min_olap_length"]:
            continue
        qid = n_sketched + qi + 1
        qmh = O.minhash(sq, p["k"], p["H"])[1]
        _, qo, qlen = O.ordered(sq, p["k2"], p["S"])
        for mid, fwd, L, mh, (_, mo, mlen) in ent:
            if int((qmh == mh).sum()) < 3:
                continue
            r = O.overlap(qo, qlen, mo, mlen)
            if r["score"] >= 0.78:
                b1, b2 = (r["b1"], r["b2"]) if fwd else (L - r["b2"] - 1, L - r["b1"] - 1)
                qrecs.append(O.format_record({"from_id": qid, "to_id": mid, "score": r["score"], "raw": r["raw"], "a1": r["a1"], "a2": r["a2"],
                                              "alen": len(sq), "b1": b1, "b2": b2, "blen": L, "to_rc": 0 if fwd else 1}))
    out["query_records_no_self"] = sorted(qrecs)

    # ---- -f filter file (canonical 16-mers of the index reads with their fractions) + --supress-noise ----
    counts = {}
    for i in range(len(fa)):
        sq = fa.sequence(i)
        for j in range(0, len(sq) - 15, 2):
            km = sq[j:j + 16]
            if set(km) <= set("ACGT"):
                r = "".join(comp[c] for c in reversed(km))
                km = min(km, r)
                counts[km] = counts.get(km, 0) + 1
    total = float(sum(counts.values()))
    items = sorted(counts.items(), key=lambda kv: (-kv[1], kv[0]))[:1500]
    with open(os.path.join(HERE, "small_kmers.txt"), "w") as fh:
        fh.write(f"{len(items)} {len(items)}\n")
        for km, c in items:
            fh.write(f"{km}\t{c / total:.10f}\n")
    import mhap_amd
    cutoff = 2.0e-4
    out["filter_threshold"] = cutoff
    for key, mode in (("filter_records", 0), ("supress_noise_1_records", 1), ("supress_noise_2_records", 2)):
        fc = mhap_amd.FrequencyCounts.from_file(os.path.join(HERE, "small_kmers.txt"), filter_cutoff=cutoff, repeat_weight=0.9, supress_noise=mode)
        of = O.Filter(fc.hashes, fc.fractions, cutoff, 0.9, 3.0, False, remove_unique=mode, whitelist=fc.whitelist, size_bloom=fc.size_bloom)
        rr = O.run_self(fa, k=p["k"], H=p["H"], k2=p["k2"], S=p["S"], min_olap_length=p["min_olap_length"], nthreads=2, flt=of)
        out[key] = O.record_lines(rr["records"])
    with open(os.path.join(HERE, "small_reads.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print(len(fa), "reads,", len(out["sorted_records"]), "records;", len(out["query_records_no_self"]), "query records;",
          [len(out[k]) for k in ("filter_records", "supress_noise_1_records", "supress_noise_2_records")])


if __name__ == "__main__":
    main()
