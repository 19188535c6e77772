"""Synthetic workloads of BASELINE.json's configs and the `-f` k-mer filter file that goes with config 5.

Measurement/test tooling only: the generator itself is C (mhap_synth_reads_repeats in host_util.cpp); this module names the
configurations, writes FASTA files and counts k-mers for the filter file the way a k-mer counter (meryl/jellyfish) would.
"""
import os

import numpy as np

from .api import FastaData, MhapParams, synth_reads, synth_reads_from_genome

SEED = 0x4D484150

# name -> (reads, read length, params, generator seed index, planted repeats (element, spacing, divergence) or None, uses -f)
CONFIGS = {
    # BASELINE configs[0]: 1k x 5 kb, k=16, --num-hashes 256 (the reference's own CPU-runnable case)
    "c1": dict(reads=1000, length=5000, hashes=256, seed=SEED ^ 1, repeats=None, filter=False,
               label="1000 synthetic PacBio-style reads x 5000 bp, --num-hashes 256 (BASELINE configs[0])"),
    # BASELINE configs[1]: the configuration the metric is quoted on
    "c2": dict(reads=100000, length=10000, hashes=512, seed=SEED ^ 2, repeats=None, filter=False,
               label="100000 synthetic PacBio-style reads x 10000 bp (BASELINE configs[1])"),
    # BASELINE configs[3] read shape (15 kb, 512 hashes) on one GPU's share of the 1M-read job
    "c4slice": dict(reads=60000, length=15000, hashes=512, seed=SEED ^ 4, repeats=None, filter=False,
                    label="60000 synthetic reads x 15000 bp: single-GPU slice of BASELINE configs[3] (1M x 15 kb over 8 GPUs)"),
    # BASELINE configs[3] in full (its reference run is an 8-GPU job; the tables, the index and the scratch of the whole job fit one MI355X)
    "c4": dict(reads=1000000, length=15000, hashes=512, seed=SEED ^ 4, repeats=None, filter=False,
               label="1000000 synthetic reads x 15000 bp (BASELINE configs[3] in full, on one GPU)"),
    # BASELINE configs[4] read shape (12 kb) + planted repeat family + -f filter file + --filter-threshold
    "c5slice": dict(reads=40000, length=12000, hashes=512, seed=SEED ^ 5, repeats=(300, 3000, 0.01), filter=True,
                    label="40000 synthetic reads x 12000 bp with a planted 300-bp repeat family (one copy per 3 kb, 1% divergence), "
                          "-f k-mer filter file, --filter-threshold 1e-5: single-GPU slice of BASELINE configs[4]"),
    # One rank's share of BASELINE configs[4] by SIZE (5M reads over 8 GPUs = 625 000 reads x 12 kb) as a self-overlap job on one GPU:
    # 1.25 M index entries.  The planted family is the slice's (300 bp, one copy per 3 kb: Alu-like density) at 5 % divergence from the
    # consensus instead of 1 %: candidate pairs and records of a one-consensus family grow with the SQUARE of the read count (a k-mer
    # above --filter-threshold 1e-5 is in 12 % of all 12-kb reads by definition), and at 1 % this size has 3 000 candidates and 250
    # records per read — 2 x 10^9 pairs and 1.6 x 10^8 records a step (10^10 records for the 8-GPU job), no job anyone runs; at 5 %
    # (AluY/AluS-like copies) it is 340 candidates and 46 records per read, still forty times the true overlaps, with 326 k-mers over the
    # filter's cutoff.  Both were measured at this size: profiles/r04_c5_probe.txt
    "c5rank": dict(reads=625000, length=12000, hashes=512, seed=SEED ^ 5, repeats=(300, 3000, 0.05), filter=True,
                   label="625000 synthetic reads x 12000 bp with a planted 300-bp repeat family (one copy per 3 kb, 5% divergence), "
                         "-f k-mer filter file, --filter-threshold 1e-5: one rank's share (by size) of BASELINE configs[4], self-overlap on one GPU"),
    # BASELINE configs[4] itself (5M reads): not a single-GPU bench config (tables + index + scratch of 10M strands exceed 288 GB); it
    # exists so that tools/emulate_rank.py can deal it over 8 ranks and run ONE rank's step against the gathered rows of all of them
    "c5": dict(reads=5000000, length=12000, hashes=512, seed=SEED ^ 5, repeats=(300, 3000, 0.05), filter=True,
               label="5000000 synthetic reads x 12000 bp with the planted repeat family (5% divergence), -f filter (BASELINE configs[4]; 8-GPU job)"),
}


def config_reads(name, shard=0, nshards=1, reads=None, length=None, error_rate=0.15):
    c = CONFIGS[name]
    n = reads or c["reads"]
    L = length or c["length"]
    return synth_reads(n, L, seed=c["seed"], error_rate=error_rate, shard=shard, nshards=nshards, repeats=c["repeats"])


def _revcomp_codes(x):
    return (3 - x[::-1]).astype(np.uint8)


def ecoli_like_genome(seed=SEED ^ 3, size=4_600_000):
    """A genome with the REPEAT STRUCTURE of E. coli K-12 (the genome behind BASELINE configs[2], whose real reads are in neither
    tree): 4.6 Mbp of random sequence with seven copies of a ~5-kb rRNA-operon-like element at 99 % identity to their consensus (five
    on one strand, two on the other, as rrnA-H sit around the origin) and twelve insertion-sequence copies of 0.8-1.3 kb in three
    families at 99.5 % (IS1-, IS5-, IS3-like), either strand.  Returns (codes uint8[size], list of (name, start, length, strand))."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size, dtype=np.uint8)
    placed = []

    def plant(name, elem, copies, div):
        for c in range(copies):
            x = elem.copy()
            m = rng.random(len(x)) < div
            x[m] = (x[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
            strand = int(rng.integers(0, 2)) if name != "rrn" else (0 if c < 5 else 1)
            if strand:
                x = _revcomp_codes(x)
            for _ in range(1000):                                # a position clear of every earlier copy (by 20 kb)
                at = int(rng.integers(0, size - len(x)))
                if all(at + len(x) + 20000 <= a or a + n + 20000 <= at for _, a, n, _ in placed):
                    break
            g[at:at + len(x)] = x
            placed.append((name, at, len(x), strand))

    plant("rrn", rng.integers(0, 4, 5000, dtype=np.uint8), 7, 0.01)
    plant("IS1", rng.integers(0, 4, 800, dtype=np.uint8), 5, 0.005)
    plant("IS5", rng.integers(0, 4, 1200, dtype=np.uint8), 4, 0.005)
    plant("IS3", rng.integers(0, 4, 1300, dtype=np.uint8), 3, 0.005)
    # forty tandem repeats (REP / microsatellite-like: a unit of 3-60 bp repeated over 100-500 bp): the k-mers of one READ that occur
    # more than once — what the default tf weighting (no -f) gives a weight > 1 (J/sketch/MinHashSketch.java:98-128)
    for t in range(40):
        unit = rng.integers(0, 4, int(rng.integers(3, 61)), dtype=np.uint8)
        total = int(rng.integers(100, 501))
        x = np.tile(unit, total // len(unit) + 1)[:total]
        for _ in range(1000):
            at = int(rng.integers(0, size - total))
            if all(at + total + 2000 <= a or a + n + 2000 <= at for _, a, n, _ in placed):
                break
        g[at:at + total] = x
        placed.append(("tandem", at, total, 0))
    return g, placed


def ecoli_like_reads(n=90000, seed=SEED ^ 3, lmax=45000, error_rate=0.15, short_every=997, n_runs_every=50):
    """Stand-in for BASELINE configs[2] (E. coli PacBio P6-C4, ~90 k reads) with real repeat structure: reads of a log-normal length
    mix (median ~8 kb, tail to 45 kb: reads past 24 591 bases take the materialised-hash path) drawn from ecoli_like_genome at 15 %
    error, a few reads under --min-olap-length, a run of N in 2 % of the reads (raw-byte strands).  Default flags meet weights > 1
    WITHOUT -f here (a read across an operon or an IS copy repeats no k-mer, but its MinHash buckets are shared by every read of every
    copy: long buckets, seven-fold candidate sets, duplicated-hash groups in the join).  (reads, genome, placed copies)"""
    g, placed = ecoli_like_genome(seed)
    rng = np.random.default_rng(seed + 1)
    L = np.clip(rng.lognormal(9.0, 0.55, n).astype(np.int32), 80, lmax)
    if short_every:
        L[::short_every] = rng.integers(10, 116, len(L[::short_every]))
    fa = synth_reads_from_genome(g, L, seed=seed, error_rate=error_rate)
    if n_runs_every:
        for i in rng.choice(n, max(1, n // n_runs_every), replace=False):
            w = int(rng.integers(5, 200))
            o = int(fa.offsets[i]) + int(rng.integers(0, max(1, int(L[i]) - w)))
            fa.bases[o:o + min(w, int(L[i]))] = ord("N")
    return fa, g, placed


def write_fasta(fasta, path, prefix="r"):
    """One record per read, header `>r<id>`, the whole sequence on one line."""
    with open(path, "wb") as fh:
        for i in range(len(fasta)):
            o, n = int(fasta.offsets[i]), int(fasta.lengths[i])
            fh.write(b">" + prefix.encode() + str(int(fasta.ids[i])).encode() + b"\n")
            fh.write(fasta.bases[o:o + n].tobytes())
            fh.write(b"\n")


_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def count_kmers(fasta, k=16, canonical=True, max_reads=None):
    """(k-mer values as uint32/uint64 with A<C<G<T = 0..3, first base most significant; counts; total windows).
    canonical=True counts a k-mer together with its reverse complement under the lexicographically smaller string — what
    the reference's filter loader assumes (J/sketch/FrequencyCounts.java:169 hashes the line's k-mer canonicalised)."""
    assert k <= 32
    n = len(fasta) if max_reads is None else min(len(fasta), max_reads)
    vals = []
    for i in range(n):
        o, L = int(fasta.offsets[i]), int(fasta.lengths[i])
        if L < k:
            continue
        codes = _CODE[fasta.bases[o:o + L]]
        ok = codes != 255
        c = codes.astype(np.uint64) & 3
        nw = L - k + 1
        v = np.zeros(nw, dtype=np.uint64)
        r = np.zeros(nw, dtype=np.uint64)
        good = np.ones(nw, dtype=bool)
        for j in range(k):
            v = (v << np.uint64(2)) | c[j:j + nw]
            r = r | ((np.uint64(3) - c[j:j + nw]) << np.uint64(2 * j))
            good &= ok[j:j + nw]
        if canonical:
            v = np.minimum(v, r)
        vals.append(v[good])
    allv = np.concatenate(vals) if vals else np.zeros(0, np.uint64)
    if k <= 16:
        allv = allv.astype(np.uint32)
    u, cnt = np.unique(allv, return_counts=True)
    return u, cnt, int(allv.shape[0])


def kmer_string(v, k):
    return "".join("ACGT"[(int(v) >> (2 * (k - 1 - j))) & 3] for j in range(k))


def write_filter_file(fasta, path, k=16, min_fraction=2.5e-6, canonical=True, max_reads=4000):
    """The `-f` file: first line "<distinct k-mers> <lines>", then `kmer<TAB>fraction` in descending order of fraction
    (J/sketch/FrequencyCounts.java:102-104,158-184; docs/source/quickstart.rst).  Counts come from the first max_reads reads
    (a k-mer counter's output on a sample; fractions are ratios, so a sample estimates them), lines below min_fraction are
    not written (the loader drops everything under --filter-threshold anyway)."""
    u, cnt, total = count_kmers(fasta, k, canonical, max_reads)
    frac = cnt.astype(np.float64) / max(total, 1)
    keep = np.nonzero(frac >= min_fraction)[0]
    order = keep[np.lexsort((u[keep], -cnt[keep]))]
    with open(path, "w") as fh:
        fh.write(f"{len(u)} {len(order)}\n")
        for i in order:
            fh.write(f"{kmer_string(u[i], k)}\t{frac[i]:.10e}\n")
    return len(order)


def params_for(name, device=-1, **over):
    c = CONFIGS[name]
    kw = dict(kmer_size=16, num_hashes=c["hashes"], ordered_kmer_size=12, ordered_sketch_size=1536, device=device)
    kw.update(over)
    return MhapParams(**kw)


def c3_fasta_path():
    """BASELINE configs[2] (E. coli PacBio P6-C4 reads) is not in either tree: supplied on the box through MHAP_C3_FASTA."""
    p = os.environ.get("MHAP_C3_FASTA")
    return p if p and os.path.exists(p) else None
