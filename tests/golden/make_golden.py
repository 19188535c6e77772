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
    with open(os.path.join(HERE, "small_reads.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print(len(fa), "reads,", len(out["sorted_records"]), "records")


if __name__ == "__main__":
    main()
