"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same inputs — bit-exact."""
import json
import os
import random

import numpy as np
import pytest

import oracle_lib as O
import mhap_amd
from mhap_amd import FastaData, MhapParams, MinHashSearch
from mhap_amd import workloads as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _rand_seq(rnd, n, alphabet="ACGT"):
    return "".join(rnd.choice(alphabet) for _ in range(n))


def _assert_sketch_parity(fa, p, flt=None, oflt=None):
    with MinHashSearch(p, kmer_filter=flt) as ms:
        sk = ms.sketch(fa)
    for i in range(len(fa)):
        seq = fa.sequence(i)
        for strand, s in ((0, seq), (1, O.rc(seq))):
            e = 2 * i + strand
            if len(seq) < p.min_olap_length:
                assert sk["status"][e] == 2, (i, strand)
                continue
            rc1, mh = O.minhash(s, p.kmer_size, p.num_hashes, p.repeat_weight, oflt)
            rc2, od, _ = O.ordered(s, p.ordered_kmer_size, p.ordered_sketch_size)
            if strand == 1 and sk["status"][e - 1] != 0:
                assert sk["status"][e] != 0        # forward failed -> read dropped
                continue
            if rc1 or rc2:
                assert sk["status"][e] == 1, (i, strand)
                continue
            assert sk["status"][e] == 0, (i, strand)
            assert sk["minhash"][e].tolist() == mh.tolist(), ("minhash", i, strand)
            n = sk["ordered_size"][e]
            assert n == od.shape[0], ("ordered size", i, strand)
            assert sk["ordered"][e, :n].tolist() == od.tolist(), ("ordered", i, strand)


def test_sketch_parity_golden_fasta():
    fa = FastaData.from_file(os.path.join(GOLD, "small_reads.fasta"))
    _assert_sketch_parity(fa, MhapParams(num_hashes=64, ordered_sketch_size=256))
    _assert_sketch_parity(fa, MhapParams())   # defaults: H=512, S=1536 > n for these reads


def test_sketch_parity_edge_cases():
    rnd = random.Random(17)
    seqs = [
        _rand_seq(rnd, 5000),
        _rand_seq(rnd, 13),                    # >= k2, < k  -> ZeroNGrams from MinHash
        _rand_seq(rnd, 16),                    # exactly one k-mer
        _rand_seq(rnd, 11),                    # < k2
        _rand_seq(rnd, 1025), _rand_seq(rnd, 1039), _rand_seq(rnd, 1040), _rand_seq(rnd, 2063),   # hash tile edges
        "A" * 3000,                            # one k-mer with weight 2985; all ordered hashes equal (ties by position)
        ("ACGTTGCA" * 400),                    # few distinct k-mers with large tf weights
        _rand_seq(rnd, 2000) * 2,              # every k-mer twice
        _rand_seq(rnd, 3000, "ACGTN"),         # non-ACGT chars: raw-byte path + rc translate table
        _rand_seq(rnd, 700, "ACGTRYKMBDHVSWN-*"),
        _rand_seq(rnd, 30000),
        _rand_seq(rnd, 1),
    ]
    fa = FastaData.from_strings(seqs)
    _assert_sketch_parity(fa, MhapParams(num_hashes=128, ordered_sketch_size=1536, min_olap_length=0))
    _assert_sketch_parity(fa, MhapParams(num_hashes=48, ordered_sketch_size=100, min_olap_length=116))
    _assert_sketch_parity(fa, MhapParams(num_hashes=33, ordered_sketch_size=64, min_olap_length=0, repeat_weight=-1.0))


def test_sketch_parity_other_kmer_sizes():
    rnd = random.Random(23)
    fa = FastaData.from_strings([_rand_seq(rnd, rnd.randint(200, 3000)) for _ in range(12)] + ["ACGT" * 200])
    for k, k2 in ((15, 13), (21, 14), (9, 7), (24, 17)):
        _assert_sketch_parity(fa, MhapParams(kmer_size=k, ordered_kmer_size=k2, num_hashes=64, ordered_sketch_size=300, min_olap_length=50))


def _self_lines(fa, p, flt=None):
    with MinHashSearch(p, kmer_filter=flt) as ms:
        ms.add_data(fa)
        recs = ms.find_matches()
        st = ms.stats()
    return sorted(mhap_amd.records_to_lines(recs)), st


def test_self_overlap_golden_records():
    g = json.load(open(os.path.join(GOLD, "small_reads.json")))
    fa = FastaData.from_file(os.path.join(GOLD, "small_reads.fasta"))
    pp = g["params"]
    p = MhapParams(kmer_size=pp["k"], num_hashes=pp["H"], ordered_kmer_size=pp["k2"], ordered_sketch_size=pp["S"],
                   min_olap_length=pp["min_olap_length"])
    lines, st = _self_lines(fa, p)
    assert lines == g["sorted_records"]
    assert st["matches_found"] == len(g["sorted_records"])


@pytest.mark.parametrize("err,H,S", [(0.15, 256, 1536), (0.05, 512, 1536), (0.08, 100, 400)])
def test_self_overlap_matches_oracle_config1(err, H, S):
    """BASELINE configs[0] shape: 1k x 5kb reads, k=16, --num-hashes 256 (plus two variants)."""
    n = 1000 if H == 256 else 300
    fa = mhap_amd.synth_reads(n, 5000, seed=0x4D484150 ^ 1, error_rate=err)
    p = MhapParams(num_hashes=H, ordered_sketch_size=S)
    want = O.run_self(fa, H=H, S=S, nthreads=8)
    lines, st = _self_lines(fa, p)
    assert len(want["records"]) > 100
    assert lines == O.record_lines(want["records"])
    assert st["candidates_compared"] == want["compared"]
    assert st["strands_indexed"] == want["strands"]


def test_self_overlap_filters_and_thresholds():
    fa = mhap_amd.synth_reads(200, 3000, seed=77, error_rate=0.06)
    # mix lengths so --min-store-length splits the reads into short and long
    idx = np.arange(200)
    fa.lengths[idx % 3 == 0] = 1500
    for kw in (dict(min_store_length=2000), dict(num_min_matches=1, threshold=0.0), dict(threshold=0.95, max_shift=0.05),
               dict(num_min_matches=6, max_shift=0.5)):
        p = MhapParams(num_hashes=128, ordered_sketch_size=512, **kw)
        want = O.run_self(fa, H=128, S=512, nthreads=8, num_min_matches=p.num_min_matches, min_store_length=p.min_store_length,
                          threshold=p.threshold, max_shift=p.max_shift)
        lines, _ = _self_lines(fa, p)
        assert lines == O.record_lines(want["records"]), kw


def test_self_overlap_with_repeat_filter():
    """tf-idf weighted MinHash (-f file, J/sketch/FrequencyCounts.java) incl. the repeat-weight < 0 (v1.0) mode."""
    rnd = random.Random(5)
    fa = mhap_amd.synth_reads(150, 3000, seed=31, error_rate=0.05)
    # plant a repeat so that some k-mers are frequent, then list the most frequent k-mers as the filter
    rep = _rand_seq(rnd, 300)
    seqs = [fa.sequence(i) for i in range(len(fa))]
    seqs = [s[:500] + rep + s[800:] if i % 2 == 0 else s for i, s in enumerate(seqs)]
    fa = FastaData.from_strings(seqs)
    counts = {}
    for s in seqs:
        for i in range(len(s) - 15):
            counts[s[i:i + 16]] = counts.get(s[i:i + 16], 0) + 1
    total = sum(counts.values())
    top = sorted(counts.items(), key=lambda kv: -kv[1])[:400]
    kmers = [k for k, _ in top]
    fracs = np.array([c / total for _, c in top])
    hashes = np.array([int(O.kmer_hashes64(k, 16, True)[0]) for k in kmers], dtype=np.int64)   # canonical (doRC default)
    for rw, no_tf in ((0.9, False), (0.5, True), (-1.0, False), (1.0, False)):
        cutoff = 1e-5
        offset = rw if 0.0 <= rw < 1.0 else 0.0
        flt = mhap_amd.FrequencyCounts(hashes, fracs, cutoff, offset, 3.0, no_tf)
        oflt = O.Filter(hashes, fracs, cutoff, offset, 3.0, no_tf)
        p = MhapParams(num_hashes=128, ordered_sketch_size=512, repeat_weight=rw)
        _assert_sketch_parity(FastaData.from_strings(seqs[:6]), p, flt, oflt)
        want = O.run_self(fa, H=128, S=512, nthreads=8, repeat_weight=rw, flt=oflt)
        lines, _ = _self_lines(fa, p, flt)
        assert lines == O.record_lines(want["records"]), (rw, no_tf)


def test_supress_noise_whitelist():
    """--supress-noise 1|2 (FrequencyCounts removeUnique): the Bloom-filter whitelist of the filter file's k-mers.  Mode 1 drops
    every k-mer that is not in the file (a read without any listed k-mer has no sketch), mode 2 gives them idf 1."""
    rnd = random.Random(15)
    fa = mhap_amd.synth_reads(90, 2500, seed=44, error_rate=0.04)
    seqs = [fa.sequence(i) for i in range(len(fa))] + [_rand_seq(rnd, 800)]      # the last read shares nothing with the whitelist
    counts = {}
    for s in seqs[:-1]:
        for i in range(0, len(s) - 15, 3):                                         # a third of the k-mers is "in the file"
            counts[s[i:i + 16]] = counts.get(s[i:i + 16], 0) + 1
    total = sum(counts.values())
    items = sorted(counts.items(), key=lambda kv: (-kv[1], kv[0]))
    kmers = [k for k, _ in items]
    fracs = np.array([c / total for _, c in items])
    hashes = np.array([int(O.kmer_hashes64(k, 16, True)[0]) for k in kmers], dtype=np.int64)
    fa2 = FastaData.from_strings(seqs)
    for mode, rw in ((1, 0.9), (2, 0.9), (1, 1.0), (1, -1.0)):
        offset = rw if 0.0 <= rw < 1.0 else 0.0
        flt = mhap_amd.FrequencyCounts(hashes, fracs, 1e-4, offset, 3.0, False, supress_noise=mode, size_bloom=len(hashes))
        oflt = O.Filter(hashes, fracs, 1e-4, offset, 3.0, False, remove_unique=mode, whitelist=hashes, size_bloom=len(hashes))
        p = MhapParams(num_hashes=128, ordered_sketch_size=512, repeat_weight=rw)
        _assert_sketch_parity(FastaData.from_strings(seqs[:4] + seqs[-1:]), p, flt, oflt)
        want = O.run_self(fa2, H=128, S=512, nthreads=8, repeat_weight=rw, flt=oflt)
        got, _ = _self_lines(fa2, p, flt)
        assert got == O.record_lines(want["records"]) and len(got) > 20, (mode, rw)
        if mode == 1:
            assert want["status"][2 * (len(seqs) - 1)] == 1      # ZeroNGramsFoundException: nothing of that read is whitelisted


def test_index_vs_stream_mode():
    """-s index -q queries (toSelf=false): brute-force expectation built from oracle primitives."""
    base = mhap_amd.synth_reads(60, 2500, seed=91, error_rate=0.05)
    index = base.subset(np.arange(0, 40))
    index.ids[:] = np.arange(1, 41)
    queries = base.subset(np.arange(30, 60))
    queries.ids[:] = np.arange(41, 71)           # id offset = #index reads (MhapMain.java:462,527)
    p = MhapParams(num_hashes=64, ordered_sketch_size=400)
    with MinHashSearch(p) as ms:
        ms.add_data(index)
        recs = ms.find_matches_stream(queries)
    got = sorted(mhap_amd.records_to_lines(recs))
    ent = []
    for i in range(len(index)):
        s = index.sequence(i)
        for fwd, t in ((1, s), (0, O.rc(s))):
            ent.append((int(index.ids[i]), fwd, len(s), O.minhash(t, 16, 64)[1], O.ordered(t, 12, 400)))
    want = []
    for qi in range(len(queries)):
        s = queries.sequence(qi)
        qmh = O.minhash(s, 16, 64)[1]
        _, qo, qlen = O.ordered(s, 12, 400)
        for mid, fwd, L, mh, (_, mo, mlen) in ent:
            if int((qmh == mh).sum()) < 3:
                continue
            r = O.overlap(qo, qlen, mo, mlen)
            if r["score"] >= 0.78:
                b1, b2 = (r["b1"], r["b2"]) if fwd else (L - r["b2"] - 1, L - r["b1"] - 1)
                want.append(O.format_record({"from_id": int(queries.ids[qi]), "to_id": mid, "score": r["score"], "raw": r["raw"],
                                             "a1": r["a1"], "a2": r["a2"], "alen": len(s), "b1": b1, "b2": b2, "blen": L,
                                             "to_rc": 0 if fwd else 1}))
    assert len(want) > 20
    assert got == sorted(want)


def test_index_export_roundtrip_and_query_sharding():
    """Sketch tables survive export -> add_sketches (the `.dat` path) and sharded query ranges add up."""
    fa = mhap_amd.synth_reads(120, 3000, seed=55, error_rate=0.05)
    p = MhapParams(num_hashes=64, ordered_sketch_size=300)
    with MinHashSearch(p) as ms:
        ms.add_data(fa)
        full = sorted(mhap_amd.records_to_lines(ms.find_matches()))
        n = ms.size()
        parts = []
        for lo in range(0, n, 50):
            parts += mhap_amd.records_to_lines(ms.find_matches(lo, 50))
        assert sorted(parts) == full
        tables = ms.export()
    keep = tables["status"] == 0
    sk = {k: v[keep] for k, v in tables.items() if k != "status"}
    with MinHashSearch(p) as ms2:
        ms2.add_sketches(sk)
        ms2.prepare_index()                 # explicit inverted-index build (mhap_index_prepare); the searches below reuse it
        again = sorted(mhap_amd.records_to_lines(ms2.find_matches()))
        again2 = sorted(mhap_amd.records_to_lines(ms2.find_matches(0, 60))) + sorted(mhap_amd.records_to_lines(ms2.find_matches(60, -1)))
        kt = ms2.kernel_times()
    assert again == full and len(full) > 50 and sorted(again2) == full
    assert kt["index_build"]["launches"] == 1 and kt["index_query"]["launches"] == 3


def test_search_with_query_sketches_already_in_device_memory():
    """mhap_find_matches_device — what one rank of a sharded search runs after its all-gather, and what tools/emulate_rank.py times: the
    forward rows of every read, sketched into caller-owned device buffers, against the index of the same reads under the toSelf id
    rules = the self overlap; with no sink the records are counted, not delivered."""
    import torch
    fa = mhap_amd.synth_reads(300, 3000, seed=77, error_rate=0.06)
    p = MhapParams(num_hashes=128, ordered_sketch_size=512, device=0)
    n, H, S = len(fa), 128, 512
    dev = torch.device("cuda", 0)
    mh = torch.empty((2 * n, H), dtype=torch.int32, device=dev); od = torch.empty((2 * n, S, 2), dtype=torch.int32, device=dev)
    mt = torch.empty((2 * n, 4), dtype=torch.int32, device=dev)
    with MinHashSearch(p) as ms:
        ms.stage(fa); ms.sketch_staged_device(mh.data_ptr(), od.data_ptr(), mt.data_ptr()); ms.synchronize()
        q_mh, q_od, q_mt = mh[0::2].contiguous(), od[0::2].contiguous(), mt[0::2].contiguous()
        ms.add_staged()
        want = sorted(mhap_amd.records_to_lines(ms.find_matches()))
        got = sorted(mhap_amd.records_to_lines(ms.find_matches_device(q_mh.data_ptr(), q_od.data_ptr(), q_mt.data_ptr(), fa.ids, to_self=True)))
        cnt = ms.find_matches_device(q_mh.data_ptr(), q_od.data_ptr(), q_mt.data_ptr(), fa.ids, to_self=True, count_only=True)
    assert len(want) > 100 and got == want and cnt == len(want)


def test_ordered_kernel_in_two_launches_and_the_host_made_query_list(monkeypatch):
    """Round 6: (a) long weight-1 launches run the ordered kernel in two parts, one in front of the MinHash launch (with the weighted strands'
    launch beside it) and one behind it — forced here on a small data set with repeats (so that there ARE weighted strands), for every split
    incl. an odd one and with the weighted launch waiting; sketches and records must not notice.  (b) the search of device-resident rows
    with its query list made on the host, as rounds 1-5 did (MHAP_QUERY_LIST_HOST=1), gives the device-made list's records — with rows
    that were not sketched (too short) among the queries."""
    import torch
    fa0 = mhap_amd.synth_reads(260, 4000, seed=4242, error_rate=0.05, repeats=(300, 1500, 0.01))
    short = mhap_amd.synth_reads(6, 100, seed=5, error_rate=0.05)     # below the minimum overlap length (116): status 2, no query
    bases = np.concatenate([fa0.bases, short.bases]); lengths = np.concatenate([fa0.lengths, short.lengths])
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    fa = mhap_amd.FastaData(bases, offsets, lengths.astype(np.int32), np.arange(len(lengths), dtype=np.int64) + 1)
    p = MhapParams(num_hashes=128, ordered_sketch_size=512, device=0)
    want = O.record_lines(O.run_self(fa, H=128, S=512, nthreads=8)["records"])
    assert len(want) > 100
    n, H, S = len(fa), 128, 512
    dev = torch.device("cuda", 0)
    ref = None
    for env in ({"MHAP_ORDERED_SPLIT": "0"}, {"MHAP_ORDERED_SPLIT": "55"}, {"MHAP_ORDERED_SPLIT": "33"}, {"MHAP_ORDERED_SPLIT": "100"},
                {"MHAP_ORDERED_SPLIT": "55", "MHAP_ORDERED_NOWAIT": "0"}, {"MHAP_ORDERED_FIRST": "1"}):
        for k in ("MHAP_ORDERED_SPLIT", "MHAP_ORDERED_NOWAIT", "MHAP_ORDERED_FIRST"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        mh = torch.empty((2 * n, H), dtype=torch.int32, device=dev); od = torch.full((2 * n, S, 2), -7, dtype=torch.int32, device=dev)
        mt = torch.empty((2 * n, 4), dtype=torch.int32, device=dev)
        with MinHashSearch(p) as ms:
            ms.stage(fa); ms.sketch_staged_device(mh.data_ptr(), od.data_ptr(), mt.data_ptr()); ms.synchronize()
            tabs = (mh.cpu().numpy().copy(), od.cpu().numpy().copy(), mt.cpu().numpy().copy())
            ms.add_staged()
            assert sorted(mhap_amd.records_to_lines(ms.find_matches())) == want, env
            if ref is None:
                ref = tabs
                q = (mh[0::2].contiguous(), od[0::2].contiguous(), mt[0::2].contiguous())
                assert int((q[2][:, 3] != 0).sum()) >= 6                      # the short reads' rows are among the queries
                dev_list = sorted(mhap_amd.records_to_lines(ms.find_matches_device(q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), fa.ids, to_self=True)))
                st0 = ms.stats()["queries_searched"]
                monkeypatch.setenv("MHAP_QUERY_LIST_HOST", "1")
                host_list = sorted(mhap_amd.records_to_lines(ms.find_matches_device(q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), fa.ids, to_self=True)))
                st1 = ms.stats()["queries_searched"]
                monkeypatch.delenv("MHAP_QUERY_LIST_HOST")
                assert dev_list == want and host_list == want
                assert st1 - st0 == n - 6 and st0 > 0                        # both paths count the sketched rows only
            else:
                for a, b, name in zip(ref, tabs, ("minhash", "ordered", "meta")):
                    sized = tabs[2][:, 0]                                    # ordered rows are defined up to their size (meta word 0)
                    if name == "ordered":
                        for e in range(2 * n):
                            assert np.array_equal(a[e, :sized[e]], b[e, :sized[e]]), (env, e)
                    else:
                        assert np.array_equal(a, b), (env, name)


def test_group_of_ranks_on_one_device_matches_the_oracle(monkeypatch):
    """The multi-GPU path inside the library (mhap_group_*: reads dealt round-robin, one index shard per rank, forward query rows
    gathered by peer copies, every rank scoring all queries against its shard under the toSelf id rules) with 2, 3 and 4 ranks
    sharing this box's one GPU: self overlap, -q mode (toSelf = false), a ragged data set (short and unsketchable reads, fewer
    reads than ranks) and incremental adds all give the oracle's records."""
    from mhap_amd import MinHashSearchGroup
    fa = mhap_amd.synth_reads(700, 3000, seed=202, error_rate=0.07)
    p = MhapParams(num_hashes=128, ordered_sketch_size=512, device=0)
    want = O.record_lines(O.run_self(fa, H=128, S=512, nthreads=8)["records"])
    assert len(want) > 300
    q = mhap_amd.synth_reads(700, 3000, seed=202, error_rate=0.07, shard=3, nshards=10)
    q = mhap_amd.FastaData(q.bases, q.offsets, q.lengths, np.arange(len(q), dtype=np.int64) + 701)
    with MinHashSearch(p) as ms:
        ms.add_data(fa)
        want_q = sorted(mhap_amd.records_to_lines(ms.find_matches_stream(q)))
    assert len(want_q) > 50
    for n in (2, 3, 4):
        with MinHashSearchGroup(p, n=n, devices=[0] * n) as g:
            g.add_data(fa)
            assert sorted(mhap_amd.records_to_lines(g.find_matches())) == want, n
            assert sorted(mhap_amd.records_to_lines(g.find_matches_stream(q))) == want_q, n
            st = g.stats()
            assert st["strands_indexed"] == 2 * len(fa) and st["matches_found"] == len(want) + len(want_q)
            # the same data set again through the cleared group, in three uneven batches (the deal continues across calls)
            g.clear()
            for lo, hi in ((0, 5), (5, 333), (333, len(fa))):
                g.add_data(fa.subset(np.arange(lo, hi)))
            assert sorted(mhap_amd.records_to_lines(g.find_matches())) == want, n
            if n == 3:   # the second stage's early "below the threshold" on gathered query rows (their histograms are made after the gate)
                monkeypatch.setenv("MHAP_OVERLAP_PRUNE", "1")
                assert sorted(mhap_amd.records_to_lines(g.find_matches())) == want
                assert sorted(mhap_amd.records_to_lines(g.find_matches_stream(q))) == want_q
                monkeypatch.delenv("MHAP_OVERLAP_PRUNE")
    # ragged: reads below --min-olap-length, reads shorter than k, and fewer sketchable reads than ranks
    seqs = [fa.sequence(i) for i in range(6)] + ["ACGT" * 20, "ACGTACGTAC", fa.sequence(6)[:130]]
    small = mhap_amd.FastaData.from_strings(seqs)
    want_s = O.record_lines(O.run_self(small, H=128, S=512, nthreads=2)["records"])
    with MinHashSearchGroup(p, n=4, devices=[0, 0, 0, 0]) as g:
        g.add_data(small)
        assert sorted(mhap_amd.records_to_lines(g.find_matches())) == want_s
    one = mhap_amd.FastaData.from_strings([fa.sequence(0)])
    with MinHashSearchGroup(p, n=3, devices=[0, 0, 0]) as g:      # two ranks hold nothing at all
        g.add_data(one)
        assert len(g.find_matches()) == 0


def test_rccl_rank_of_one_and_errors_of_the_sharded_search():
    """mhap_dist_init over RCCL with a world of one rank (all a one-GPU box can form) runs the same pack / gather / search code as
    N ranks; the collective entry points fail loudly on a handle that is no rank, and on an index that is not made of sketched reads."""
    fa = mhap_amd.synth_reads(400, 2500, seed=203, error_rate=0.06)
    p = MhapParams(num_hashes=64, ordered_sketch_size=256, device=0)
    want = O.record_lines(O.run_self(fa, H=64, S=256, nthreads=8)["records"])
    with MinHashSearch(p) as ms:
        with pytest.raises(mhap_amd.MhapError, match="not a rank"):
            ms.dist_find_matches()
        ms.dist_init(0, 1, MinHashSearch.dist_unique_id())
        ms.add_data(fa)
        assert sorted(mhap_amd.records_to_lines(ms.dist_find_matches())) == want and len(want) > 100
        tm = ms.dist_last_timing()
        assert tm["total_ms"] > 0 and tm["gather_small_ms"] >= 0
        # the transport's own view of the rank (what bench.py prints per rank) and the exchange by itself
        view = ms.dist_info()
        assert view["comm_count"] == 1 and view["comm_user_rank"] == 0 and view["comm_device"] == view["handle_device"] == 0
        assert view["rccl_version"] and view["pci_bus_id"] and ":" in view["pci_bus_id"], view
        assert ms.dist_selftest(1 << 20) > 0.0 and ms.dist_selftest(3) > 0.0
        # a LOCAL error in a collective search (the sink says stop) leaves the communicator usable: the next search is complete
        # (round 4's guard tore the RCCL communicator down on every non-OK return — ADVICE r04)
        import ctypes as C
        from mhap_amd import api
        cb = api._SINK(lambda recs, n, user: 1)
        rc = ms._lib.mhap_dist_find_matches_self(ms._h, cb, None)
        assert rc != 0
        assert sorted(mhap_amd.records_to_lines(ms.dist_find_matches())) == want
        assert ms.dist_info()["comm_count"] == 1
        tables = ms.export()
        ms.clear()
        keep = np.arange(len(tables["ids"])) % 2 == 0       # forward entries only: not pairs any more
        ms.add_sketches({k: v[keep] for k, v in tables.items() if k != "status"})
        with pytest.raises(mhap_amd.MhapError, match="pairs"):
            ms.dist_find_matches()


def test_record_sink_can_abort_a_search_and_the_handle_survives(monkeypatch):
    """The sink's return value ends the search (include/mhap_hip.h: non-zero aborts): with 128-query chunks the post stage runs on the
    library's worker thread, one chunk behind the kernels — the abort must surface as MHAP_E_STATE without a hang, whichever chunk's
    sink call asks for it, and the same handle must search correctly afterwards.  Also with the tails inline (MHAP_SEARCH_PIPELINE=0)."""
    import ctypes as C
    from mhap_amd import api
    fa = mhap_amd.synth_reads(500, 2500, seed=77, error_rate=0.05)
    p = MhapParams(num_hashes=64, ordered_sketch_size=256)
    monkeypatch.setenv("MHAP_QUERY_CHUNK", "128")
    for pipeline in ("1", "0"):
        monkeypatch.setenv("MHAP_SEARCH_PIPELINE", pipeline)
        with MinHashSearch(p) as ms:
            ms.add_data(fa)
            want = _sorted_records(ms.find_matches())
            assert len(want) > 300
            for stop_at in (1, 2, 3):
                calls = {"n": 0, "recs": 0}

                def sink(recs, n, user):
                    calls["n"] += 1
                    calls["recs"] += n
                    return 1 if calls["n"] == stop_at else 0

                cb = api._SINK(sink)
                rc = ms._lib.mhap_find_matches_self(ms._h, C.c_int64(0), C.c_int64(-1), cb, None)
                assert rc == -4 and b"sink aborted" in ms._lib.mhap_last_error(ms._h), (pipeline, stop_at, rc)
                assert calls["n"] == stop_at and 0 < calls["recs"] < len(want)
                assert np.array_equal(_sorted_records(ms.find_matches()), want)
    monkeypatch.delenv("MHAP_SEARCH_PIPELINE")


def test_eager_exchange_gathers_during_the_add(monkeypatch):
    """mhap_dist_set_eager on the ranks of a group (three ranks sharing this box's GPU, peer copies): the add that fills the empty
    index gathers every rank's forward rows while it computes (ordered rows behind the ordered kernel, which runs first; MinHash rows,
    meta and ids behind the MinHash kernel) and the search finds them in place — same records as the oracle; a second search reuses
    them; an index filled by two adds, a group with an empty rank, and a one-rank RCCL job all fall back or take part correctly."""
    from mhap_amd import MinHashSearchGroup
    fa = mhap_amd.synth_reads(600, 3000, seed=211, error_rate=0.07)
    p = MhapParams(num_hashes=128, ordered_sketch_size=512, device=0)
    want = O.record_lines(O.run_self(fa, H=128, S=512, nthreads=8)["records"])
    assert len(want) > 200
    for force_peer in ("0", "1"):
        monkeypatch.setenv("MHAP_GROUP_FORCE_PEER", force_peer)
        with MinHashSearchGroup(p, n=3, devices=[0, 0, 0]) as g:
            for r in range(3):
                g.rank(r).dist_set_eager(True)
            g.add_data(fa)
            assert sorted(mhap_amd.records_to_lines(g.find_matches())) == want
            assert [g.rank(r).dist_eager_searches() for r in range(3)] == [1, 1, 1]
            assert sorted(mhap_amd.records_to_lines(g.find_matches())) == want            # the gathered rows are still those of the index
            assert g.rank(0).dist_eager_searches() == 2
            g.clear()                                                                     # two adds: the second cannot gather eagerly
            g.add_data(fa.subset(np.arange(0, 250)))
            g.add_data(fa.subset(np.arange(250, len(fa))))
            assert sorted(mhap_amd.records_to_lines(g.find_matches())) == want
            assert g.rank(0).dist_eager_searches() == 2
            g.clear()
            g.add_data(fa)                                                                # ... and eager again
            assert sorted(mhap_amd.records_to_lines(g.find_matches())) == want
            assert g.rank(1).dist_eager_searches() == 3
            g.clear()
            two = fa.subset(np.arange(2))                                                 # rank 2 gets no read: everybody falls back
            g.add_data(two)
            assert len(g.find_matches()) == len(O.run_self(two, H=128, S=512, nthreads=2)["records"])
    monkeypatch.delenv("MHAP_GROUP_FORCE_PEER")
    with MinHashSearch(p) as ms:                                                          # one RCCL rank
        ms.dist_init(0, 1, MinHashSearch.dist_unique_id())
        ms.dist_set_eager(True)
        ms.add_data(fa)
        assert sorted(mhap_amd.records_to_lines(ms.dist_find_matches())) == want and ms.dist_eager_searches() == 1
        ms.clear()
        monkeypatch.setenv("MHAP_BATCH_BASES", "400000")                                  # several launch groups: not eligible
        ms.add_data(fa)
        monkeypatch.delenv("MHAP_BATCH_BASES")
        assert sorted(mhap_amd.records_to_lines(ms.dist_find_matches())) == want and ms.dist_eager_searches() == 1


def test_group_survives_a_failing_rank_and_reports_its_reason(monkeypatch):
    """One rank of a group of three fails BEFORE it reaches the exchange (its shard is not made of sketched reads): the other ranks,
    already at the hub's barrier, are released instead of waiting for ever; the group reports the failing rank's reason, not a
    victim's "another rank failed"; and once the shards are valid again the same group searches correctly (the hub is reset when all
    rank threads have returned).  Run twice: plain device copies, and with the peer-copy call forced between ranks that share the
    device (MHAP_GROUP_FORCE_PEER=1: the branch a multi-GPU node takes)."""
    fa = mhap_amd.synth_reads(300, 2500, seed=207, error_rate=0.06)
    p = MhapParams(num_hashes=64, ordered_sketch_size=256, device=0)
    want = O.record_lines(O.run_self(fa, H=64, S=256, nthreads=8)["records"])
    for force in ("0", "1"):
        monkeypatch.setenv("MHAP_GROUP_FORCE_PEER", force)
        with mhap_amd.MinHashSearchGroup(p, n=3, devices=[0, 0, 0]) as g:
            g.add_data(fa)
            assert sorted(mhap_amd.records_to_lines(g.find_matches())) == want and len(want) > 50
            r1 = g.rank(1)
            tables = r1.export()
            r1.clear()
            keep = np.arange(len(tables["ids"])) % 2 == 0       # forward entries only: rank 1's shard is not pairs any more
            r1.add_sketches({k: v[keep] for k, v in tables.items() if k != "status"})
            with pytest.raises(mhap_amd.MhapError, match="rank 1: .*pairs"):
                g.find_matches()
            g.clear()
            g.add_data(fa)
            assert sorted(mhap_amd.records_to_lines(g.find_matches())) == want
    monkeypatch.delenv("MHAP_GROUP_FORCE_PEER")


def _awkward_fasta(path, fa, width=70):
    """The reads of `fa` as a FASTA file that exercises the parser: wrapped lines, lower case, CRLF, an IUPAC read, reads below
    --min-olap-length and below k, empty records, names with white space and commas (ids count non-empty records only)."""
    import random
    rnd = random.Random(4)
    with open(path, "w", newline="") as fh:
        for i in range(len(fa)):
            s = fa.sequence(i)
            if i % 11 == 3:
                s = s.lower()
            if i % 17 == 5:
                s = s[:300] + "NRYKM" + s[305:]               # raw-byte read (IUPAC codes survive, FastaData.java:194)
            eol = "\r\n" if i % 5 == 2 else "\n"
            fh.write(f">read{i},extra words{eol}")
            if i % 3 == 0:
                fh.write(s + eol)
            else:
                w = width + rnd.randrange(20)
                fh.write(eol.join(s[j:j + w] for j in range(0, len(s), w)) + eol)
            if i % 13 == 7:
                fh.write(">empty_record" + eol)                # no sequence: skipped, takes no id
            if i % 19 == 9:
                fh.write(">tiny" + eol + "ACGTAC" + eol)       # shorter than k: takes an id, is never sketched
            if i % 23 == 11:
                fh.write(">short" + eol + s[:100] + eol)       # below --min-olap-length 116


def test_streamed_ingest_matches_the_copying_path_and_the_oracle(tmp_path, monkeypatch):
    """mhap_index_add_scan (mapped file -> groups packed by host threads while the GPU sketches the previous group, inverted index
    extended in place group after group) must give what FastaData.from_file + add_data gives, and both the oracle's records: wrapped
    and one-line records, lower case, CRLF, IUPAC reads, short and empty records; also with the file gzipped, as -q queries, and
    over a group of ranks."""
    import gzip
    from mhap_amd import FastaScan, MinHashSearchGroup
    fa = mhap_amd.synth_reads(360, 3000, seed=77, error_rate=0.06)
    path = str(tmp_path / "awkward.fasta")
    _awkward_fasta(path, fa)
    whole = FastaData.from_file(path)
    p = MhapParams(num_hashes=128, ordered_sketch_size=512)
    want = O.record_lines(O.run_self(whole, H=128, S=512, nthreads=8)["records"])
    assert len(want) > 100
    with FastaScan(path) as sc:
        ids, lens, names = sc.info()
        assert ids.tolist() == whole.ids.tolist() and lens.tolist() == whole.lengths.tolist()
        assert names[0] == "read0" and "empty_record" not in names and names.count("tiny") > 3
        for group_bases in ("150000", "40000000"):             # many small groups / one group
            monkeypatch.setenv("MHAP_INGEST_GROUP_BASES", group_bases)
            with MinHashSearch(p) as ms:
                ms.add_scan(sc)
                assert ms.size() == 2 * len(whole)
                got = sorted(mhap_amd.records_to_lines(ms.find_matches()))
                kt = ms.kernel_times()
                assert got == want, group_bases
                # the inverted index was filled group by group while the reads were sketched: the search did not rebuild it
                assert kt["index_build"]["launches"] <= (len(whole) * 3000) // int(group_bases) + 6
                again = sorted(mhap_amd.records_to_lines(ms.find_matches_scan(sc)))      # the same file as -q queries (toSelf = false)
            with MinHashSearch(p) as ms2:
                ms2.add_data(whole)
                assert sorted(mhap_amd.records_to_lines(ms2.find_matches_stream(whole))) == again
        monkeypatch.setenv("MHAP_INGEST_GROUP_BASES", "300000")
        with MinHashSearchGroup(p, n=3, devices=[0, 0, 0]) as g:
            g.add_scan(sc)
            assert sorted(mhap_amd.records_to_lines(g.find_matches())) == want
    gz = str(tmp_path / "awkward.fasta.gz")
    with open(path, "rb") as src, gzip.open(gz, "wb") as dst:
        dst.write(src.read())
    with FastaScan(gz) as sc, MinHashSearch(p) as ms:
        ms.add_scan(sc)
        assert sorted(mhap_amd.records_to_lines(ms.find_matches())) == want
    # adds in batches: with mhap_index_reserve the eagerly built index is extended in place, without it the first search rebuilds it
    for reserve in (True, False):
        with MinHashSearch(p) as ms:
            if reserve:
                ms.reserve(len(whole))
            for lo in range(0, len(whole), 97):
                ms.add_data(whole.subset(np.arange(lo, min(len(whole), lo + 97))))
            assert sorted(mhap_amd.records_to_lines(ms.find_matches())) == want, reserve


def test_batching_and_chunking_do_not_change_results(monkeypatch):
    fa = mhap_amd.synth_reads(300, 2000, seed=8, error_rate=0.05)
    p = MhapParams(num_hashes=64, ordered_sketch_size=300)
    ref, _ = _self_lines(fa, p)
    monkeypatch.setenv("MHAP_BATCH_BASES", "50000")
    monkeypatch.setenv("MHAP_QUERY_CHUNK", "128")
    small, _ = _self_lines(fa, p)
    monkeypatch.setenv("MHAP_NO_TRIANGULAR", "1")
    rect, _ = _self_lines(fa, p)
    assert small == ref and rect == ref and len(ref) > 100


def test_full_size_properties_config2_slice():
    """Size-independent properties at BASELINE configs[1] read shape (10 kb, H=512, S=1536) on a slice of reads:
    rc(rc(x)) sketches equal x's, every record obeys id/coordinate invariants, and a read vs its own copy is found."""
    fa = mhap_amd.synth_reads(2000, 10000, seed=0x4D484150 ^ 2, error_rate=0.15)
    p = MhapParams()
    with MinHashSearch(p) as ms:
        sk = ms.sketch(fa.subset(np.arange(8)))
        rcfa = FastaData.from_strings([O.rc(fa.sequence(i)) for i in range(8)])
        sk2 = ms.sketch(rcfa)
        for i in range(8):   # strand symmetry: sketch(rc(x)).fwd == sketch(x).rc and vice versa
            assert np.array_equal(sk["minhash"][2 * i + 1], sk2["minhash"][2 * i])
            assert np.array_equal(sk["ordered"][2 * i], sk2["ordered"][2 * i + 1])
        ms.add_data(fa)
        recs = ms.find_matches()
    assert len(recs) > 100
    assert np.all(recs["to_id"] < recs["from_id"])                       # MinHashSearch.java:215-219 with minStore 0
    assert np.all((recs["score"] >= 0.78) & (recs["a1"] >= 0) & (recs["a2"] <= 10000 - 11) & (recs["a1"] <= recs["a2"]))
    assert np.all((recs["b1"] >= -1) & (recs["b2"] <= 10000) & (recs["alen"] == 10000) & (recs["blen"] == 10000))
    pairs = set(zip(recs["from_id"].tolist(), recs["to_id"].tolist(), recs["to_rc"].tolist()))
    assert len(pairs) == len(recs)                                       # one record per (query, entry)
    # oracle cross-check on a 150-read prefix (every pair among them must agree exactly)
    sub = fa.subset(np.arange(150))
    want = O.record_lines(O.run_self(sub, nthreads=8)["records"])
    m = (recs["from_id"] <= 150) & (recs["to_id"] <= 150)
    assert sorted(mhap_amd.records_to_lines(recs[m])) == want


def _sorted_records(recs):
    names = ["from_id", "to_id", "to_rc", "score", "raw", "a1", "a2", "alen", "b1", "b2", "blen"]
    a = np.zeros(len(recs), dtype=[(n, recs.dtype[n]) for n in names])
    for n in names:
        a[n] = recs[n]
    return np.sort(a, order=names)


def test_index_line_table_every_path_against_the_plain_index(monkeypatch):
    """Round 6: the LINE table of the first query tier (one 64-byte line per lookup: index_lines_kernel) — lines of every filling (sparse,
    the default 3.5-7 postings, packed so that many lines spill into their partner line or send the lookup to ends / items), long buckets
    (a value thousands of entries share: 'see the buckets'), a query with more such slots than the tier keeps (handed to the next tier),
    with the index self-check on (every stored posting must be found through its line) — records, statistics and the oracle all agree
    with the search through bucket bounds + postings."""
    rnd = random.Random(77)
    fa1 = mhap_amd.synth_reads(1500, 3000, seed=77, error_rate=0.10)
    rep = _rand_seq(rnd, 400)                                               # a repeat element in 300 more reads: long buckets
    extra = [_rand_seq(rnd, 900) + rep + _rand_seq(rnd, 700) for _ in range(300)]
    fa = FastaData.from_strings([fa1.sequence(i) for i in range(len(fa1))] + extra)
    p = MhapParams(num_hashes=128, ordered_sketch_size=500)
    want = O.record_lines(O.run_self(fa, H=128, S=500, nthreads=16, cap=1 << 22)["records"])
    monkeypatch.setenv("MHAP_DEBUG_INDEX", "1")
    seen = {}
    for tag, env in (("plain", {"MHAP_INDEX_LINES": "0"}), ("default", {}), ("sparse", {"MHAP_INDEX_LINES": "1", "MHAP_INDEX_LINE_LOAD": "1"}),
                     ("packed", {"MHAP_INDEX_LINES": "1", "MHAP_INDEX_LINE_LOAD": "14"})):
        for k in ("MHAP_INDEX_LINES", "MHAP_INDEX_LINE_LOAD"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        lines, st = _self_lines(fa, p)
        assert lines == want, tag
        seen[tag] = (st["table_elements"], st["candidates_compared"], st["matches_found"])
    assert len(set(seen.values())) == 1 and len(want) > 3000, seen
    # -q style: the same index, queries from other tables (nothing to skip as "the query's own strand")
    for k in ("MHAP_INDEX_LINES", "MHAP_INDEX_LINE_LOAD"):
        monkeypatch.delenv(k, raising=False)
    q = mhap_amd.synth_reads(200, 3000, seed=77, error_rate=0.12)
    got = {}
    for tag, env in (("plain", "0"), ("lines", "1")):
        monkeypatch.setenv("MHAP_INDEX_LINES", env)
        with MinHashSearch(p) as ms:
            ms.add_data(fa)
            got[tag] = sorted(mhap_amd.records_to_lines(ms.find_matches_stream(q)))
    assert got["plain"] == got["lines"] and len(got["lines"]) > 50


def test_candidate_paths_agree_and_overflow_fallback(monkeypatch):
    """Inverted-index candidates (default) == brute-force all-pairs candidates == oracle, including values shared by thousands
    of entries (run cap + overflow lists) and queries whose hit set overflows the per-query LDS count table (3072 distinct
    entries: split into hash-partition passes)."""
    rnd = random.Random(41)
    base = _rand_seq(rnd, 220)
    seqs = [base] * 3300 + [_rand_seq(rnd, 220) for _ in range(20)]
    fa = FastaData.from_strings(seqs)
    p = MhapParams(num_hashes=16, ordered_sketch_size=32, min_olap_length=50)
    want = O.run_self(fa, H=16, S=32, min_olap_length=50, nthreads=8, cap=1 << 23)
    assert len(want["records"]) > 5_000_000
    want_sorted = _sorted_records(want["records"])
    for tiers, splits in (("2", False), ("1", True)):
        # default: a hit set that outgrows the first tier's table is counted by the dense second tier (no split at 3320 distinct hits);
        # first tier alone: it is split into hash-partition passes
        monkeypatch.setenv("MHAP_INDEX_TIERS", tiers)
        with MinHashSearch(p) as ms:
            ms.add_data(fa)
            got = ms.find_matches()
            st = ms.stats()
        assert (st["index_splits"] > 0) == splits and st["slot_compares"] == 0   # nothing went to the brute-force kernel
        assert np.array_equal(_sorted_records(got), want_sorted)
    monkeypatch.delenv("MHAP_INDEX_TIERS")
    small = mhap_amd.synth_reads(400, 2500, seed=12, error_rate=0.05)
    p2 = MhapParams(num_hashes=96, ordered_sketch_size=300)
    a, sa = _self_lines(small, p2)
    monkeypatch.setenv("MHAP_CANDIDATES", "bruteforce")
    b, sb = _self_lines(small, p2)
    assert a == b and len(a) > 100
    assert sa["table_elements"] > 0 and sa["slot_compares"] == 0 and sb["slot_compares"] > 0
    assert sa["candidates_compared"] == sb["candidates_compared"]


def test_index_query_tiers_on_huge_hit_sets(monkeypatch):
    """14 000 crafted entries that all share the values of two MinHash slots (2 hits < numMinMatches: counted, never a candidate)
    plus 40 identical entries: every query's hit set holds 14 000 distinct entries.  The first tier hands such a query to the
    second, whose dense counters (one per stored entry) take it in one pass; with the second tier switched off
    (MHAP_INDEX_TIERS=1) the first splits the hit set into hash-partition passes.  Candidates and records equal the all-pairs
    path's either way and the processed-elements statistic equals its closed form."""
    fa = mhap_amd.synth_reads(1, 1500, seed=3, error_rate=0.0)
    p = MhapParams(num_hashes=16, ordered_sketch_size=32, min_olap_length=50)
    with MinHashSearch(p) as ms:
        ms.add_data(fa)
        base = ms.export()
    n, ncopy = 14000, 40
    rng = np.random.default_rng(5)
    mh = rng.integers(-2**31, 2**31, size=(n, 16), dtype=np.int64).astype(np.int32)
    mh[:, 0] = base["minhash"][0, 0]
    mh[:, 1] = base["minhash"][0, 1]
    mh[:ncopy] = base["minhash"][0]
    for s in range(2, 16):   # the random slots must not collide by accident (the closed form below assumes it)
        assert len(np.unique(mh[ncopy:, s])) == n - ncopy and base["minhash"][0, s] not in mh[ncopy:, s]
    sk = {"ids": np.arange(n, dtype=np.int64), "is_fwd": np.ones(n, np.uint8), "seq_length": np.repeat(base["seq_length"][:1], n),
          "minhash": mh, "ordered": np.repeat(base["ordered"][:1], n, axis=0), "ordered_size": np.repeat(base["ordered_size"][:1], n),
          "ordered_seqlen": np.repeat(base["ordered_seqlen"][:1], n)}
    out = {}
    for mode in ("index", "first-tier-only", "middle-tier", "bruteforce"):
        if mode == "bruteforce":
            monkeypatch.setenv("MHAP_CANDIDATES", "bruteforce")
        if mode == "first-tier-only":
            monkeypatch.setenv("MHAP_INDEX_TIERS", "1")
        if mode == "middle-tier":                  # the tier between the two that a large index gets (its 8192-entry table hands these 14 000-entry hit sets on to the dense counters)
            monkeypatch.setenv("MHAP_INDEX_MID", "1")
        with MinHashSearch(p) as ms:
            ms.add_sketches(sk)
            out[mode] = (_sorted_records(ms.find_matches()), ms.stats())
        monkeypatch.delenv("MHAP_INDEX_TIERS", raising=False)
        monkeypatch.delenv("MHAP_INDEX_MID", raising=False)
    monkeypatch.delenv("MHAP_CANDIDATES")
    (ri, si), (r1, s1), (rm, sm), (rb, sb) = out["index"], out["first-tier-only"], out["middle-tier"], out["bruteforce"]
    assert np.array_equal(ri, rb) and np.array_equal(r1, rb) and np.array_equal(rm, rb) and len(ri) == ncopy * (ncopy - 1) // 2
    assert sm["index_splits"] == 0 and sm["table_elements"] == si["table_elements"] and sm["candidates_compared"] == si["candidates_compared"]
    assert si["index_splits"] == 0 and s1["index_splits"] > 0 and si["slot_compares"] == 0 and sb["slot_compares"] > 0
    assert si["candidates_compared"] == s1["candidates_compared"] == sb["candidates_compared"] == ncopy * (ncopy - 1) // 2
    want_elements = 2 * n * n + 14 * (ncopy * ncopy + (n - ncopy))
    assert si["table_elements"] == want_elements and s1["table_elements"] == want_elements


def test_num_min_matches_up_to_the_slot_count_in_every_tier(monkeypatch):
    """--num-min-matches far beyond the default: the first tier's packed words give the hit count the bits the entry index leaves
    (a count must reach numMinMatches plus one add in flight per lane without wrapping), the dense tier (MHAP_INDEX_DENSE=1: every
    query) counts in 16 bits.  Near-identical copies of one read share most of their 512 MinHash values; records, compared pairs
    and processed elements equal the oracle's at numMinMatches = 3, 150 and 250 in both."""
    rnd = random.Random(9)
    base = _rand_seq(rnd, 3000)
    seqs = [_rand_seq(rnd, 3000) for _ in range(40)]
    for c in range(12):
        s = list(base)
        for _ in range(c):                         # copy c differs from the original in c bases
            s[rnd.randrange(3000)] = rnd.choice("ACGT")
        seqs.append("".join(s))
    fa = FastaData.from_strings(seqs)
    for nmm in (3, 150, 250):
        p = MhapParams(num_hashes=512, ordered_sketch_size=512, num_min_matches=nmm)
        want = O.run_self(fa, H=512, S=512, num_min_matches=nmm, nthreads=8)
        for dense in (False, True):
            if dense:
                monkeypatch.setenv("MHAP_INDEX_DENSE", "1")
            with MinHashSearch(p) as ms:
                ms.add_data(fa)
                got = ms.find_matches()
                st = ms.stats()
            monkeypatch.delenv("MHAP_INDEX_DENSE", raising=False)
            assert np.array_equal(_sorted_records(got), _sorted_records(want["records"])), (nmm, dense)
            assert st["table_elements"] == want["elements"] and st["candidates_compared"] == want["compared"], (nmm, dense)
            assert len(got) >= 30 if nmm < 250 else len(got) > 0, (nmm, dense, len(got))


def test_dense_second_tier_ranges_and_independent_element_count(monkeypatch):
    """A repeat-rich sample (the C5 slice's generator: a 300-bp repeat family planted every 3 kb) with more than 65 536 stored
    entries through every form of the dense second tier: the compact kernel (4-bit thermometers, 131 072 entries per pass: one pass
    here), the same with 8 192 entries per pass on an index whose long buckets are class-ordered (a pass streams only its part of a
    bucket, found by bisection) and on an unordered one (every pass streams every bucket), and the 16-bit-counter kernel (three ranges
    of 32 768).  Records equal the first-tier-only path's (hash-partition passes) and the middle tier's, the index finds every posting
    where a lookup expects it and long buckets in class order (MHAP_DEBUG_INDEX self-check), and the processed-elements statistic
    equals an independent count from the exported MinHash rows in every mode."""
    from mhap_amd import workloads as W
    import tempfile, os
    n = 34000
    fa = W.config_reads("c5slice", shard=0, nshards=1, reads=n, length=2000, error_rate=0.15)
    path = os.path.join(tempfile.mkdtemp(), "kmers.txt")
    W.write_filter_file(fa, path, max_reads=2000)
    flt = mhap_amd.FrequencyCounts.from_file(path, filter_cutoff=1e-5, repeat_weight=0.9)
    p = MhapParams(num_hashes=64, ordered_sketch_size=256)
    monkeypatch.setenv("MHAP_DEBUG_INDEX", "1")
    modes = {
        "tiers": {},
        "grouped-multipass": {"MHAP_INDEX_GROUP": "1", "MHAP_INDEX_GROUP_T": "8", "MHAP_INDEX_CLASS_LOG": "12", "MHAP_DENSE_RANGE_LOG": "13"},
        "ungrouped-multipass": {"MHAP_INDEX_GROUP": "0", "MHAP_DENSE_RANGE_LOG": "13"},
        "grouped-all-dense": {"MHAP_INDEX_GROUP": "1", "MHAP_INDEX_GROUP_T": "3", "MHAP_INDEX_CLASS_LOG": "10", "MHAP_DENSE_RANGE_LOG": "11", "MHAP_INDEX_DENSE": "1"},
        "counters": {"MHAP_DENSE_TIER": "counters"},
        "first-tier-only": {"MHAP_INDEX_TIERS": "1"},
        "middle-tier": {"MHAP_INDEX_MID": "1"},     # first tier -> 8192-entry table -> dense counters for what outgrows that too
    }
    out = {}
    for mode, env in modes.items():
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with MinHashSearch(p, kmer_filter=flt) as ms:
            ms.add_data(fa)
            recs = _sorted_records(ms.find_matches())
            st = ms.stats()
            if mode == "tiers":
                assert ms.size() == 2 * n > 65536
                sk = ms.export()
        for k in env:
            monkeypatch.delenv(k)
        out[mode] = (recs, st)
    stored = sk["status"] == 0
    query = stored & (sk["is_fwd"] != 0)
    want = 0
    for s in range(sk["minhash"].shape[1]):
        vals, cnt = np.unique(sk["minhash"][stored, s], return_counts=True)
        want += int(cnt[np.searchsorted(vals, sk["minhash"][query, s])].sum())
    ra, sa = out["tiers"]
    assert len(ra) > 1000
    for mode, (r, st) in out.items():
        assert np.array_equal(ra, r), mode
        assert st["table_elements"] == want, (mode, st["table_elements"], want)
    for mode in ("grouped-multipass", "ungrouped-multipass", "grouped-all-dense", "counters"):
        assert out[mode][1]["index_splits"] > 0, mode          # further entry ranges were needed for some query


def test_inverted_index_with_a_shared_repeat(monkeypatch):
    """5 200 reads that all carry the same 2 kb repeat at H = 512 (the repeat's k-mers win most MinHash slots, so thousands of
    entries share the value of a slot): insertion stays O(1) per posting (overflow lists), large hit sets go to the second query tier, records
    equal the oracle's and the index build time stays bounded.  A second run makes the 10 400 entries look like a million to the
    dense tier: buckets of thousands of postings class-ordered in classes of 256 entries, 21 passes of 512 entries, more long buckets
    per query than the bounds table has rows for (those are streamed whole by every pass)."""
    rnd = random.Random(77)
    rep = _rand_seq(rnd, 2000)
    seqs = []
    for i in range(5200):
        s = list(rep)
        for _ in range(60):                       # 3 % point differences per copy
            s[rnd.randrange(2000)] = rnd.choice("ACGT")
        seqs.append(_rand_seq(rnd, 300) + "".join(s) + _rand_seq(rnd, 300))
    fa = FastaData.from_strings(seqs)
    p = MhapParams(num_hashes=512, ordered_sketch_size=512, threshold=0.96)
    want = O.run_self(fa, H=512, S=512, threshold=0.96, nthreads=16, cap=1 << 24)
    with MinHashSearch(p) as ms:
        ms.add_data(fa)
        got = ms.find_matches()
        st, kt = ms.stats(), ms.kernel_times()
    assert np.array_equal(_sorted_records(got), _sorted_records(want["records"]))
    assert st["table_elements"] == want["elements"] and st["candidates_compared"] == want["compared"]
    assert st["index_splits"] == 0      # hit sets of 10 400 entries: over the first tier's table, within the second tier's
    assert kt["index_build"]["ms"] < 200.0, kt["index_build"]      # 5.3 M postings; quadratic runs took seconds
    for k, v in {"MHAP_INDEX_GROUP": "1", "MHAP_INDEX_GROUP_T": "64", "MHAP_INDEX_CLASS_LOG": "8", "MHAP_DENSE_RANGE_LOG": "9", "MHAP_DEBUG_INDEX": "1"}.items():
        monkeypatch.setenv(k, v)
    with MinHashSearch(p) as ms:
        ms.add_data(fa)
        got2 = ms.find_matches()
        st2 = ms.stats()
    assert np.array_equal(_sorted_records(got2), _sorted_records(want["records"]))
    assert st2["table_elements"] == want["elements"] and st2["candidates_compared"] == want["compared"] and st2["index_splits"] > 0


def test_minhash_large_num_hashes_and_short_strands():
    """--num-hashes large enough that one wave's slot table fills most of the LDS (fewer waves per workgroup), plus
    strands on both sides of the 512-k-mer switch between per-chain rows and bit-sliced rows."""
    rnd = random.Random(77)
    fa = FastaData.from_strings([_rand_seq(rnd, n) for n in (150, 520, 527, 528, 600, 2100, 2200, 4200, 9000)])
    _assert_sketch_parity(fa, MhapParams(num_hashes=4000, ordered_sketch_size=64, min_olap_length=100))
    _assert_sketch_parity(fa, MhapParams(num_hashes=8192, ordered_sketch_size=64, min_olap_length=100))


def test_long_reads_partitioned_kmer_weights():
    """Reads with more k-mers than the LDS weight table holds (24 576): hash-partitioned passes; >65 534: wider positions."""
    rnd = random.Random(123)
    a = _rand_seq(rnd, 40000)
    seqs = [_rand_seq(rnd, 26000), _rand_seq(rnd, 70000), _rand_seq(rnd, 131500), a + a[:15000] + a[5000:9000], "ACGTTGCAAT" * 3000]
    fa = FastaData.from_strings(seqs)
    _assert_sketch_parity(fa, MhapParams(num_hashes=32, ordered_sketch_size=128))


def _mutate(rnd, s, rate):
    out = []
    for ch in s:
        r = rnd.random()
        if r < rate / 3:
            continue                               # deletion
        if r < 2 * rate / 3:
            out.append(rnd.choice("ACGT"))         # substitution
            continue
        out.append(ch)
        if r < rate:
            out.append(rnd.choice("ACGT"))         # insertion
    return "".join(out)


def test_overlap_join_groups_and_lane_fallback(monkeypatch):
    """Second stage on repeat-rich reads: sketches with duplicated ordered-k-mer hashes (replayed as groups by the
    wave-per-pair kernel), near-identical reads (more joined k-mers than it keeps) and low-complexity reads (groups longer
    than it keeps) that go to the per-lane merge.  Default path == MHAP_OVERLAP=lane == oracle."""
    rnd = random.Random(2024)
    unit = _rand_seq(rnd, 30)
    parts = []
    for _ in range(40):                               # interspersed repeats of one unit, lightly diverged, with unique spacers
        parts.append(_mutate(rnd, unit, 0.01))
        parts.append(_rand_seq(rnd, rnd.randrange(400, 900)))
    genome = "".join(parts)
    seqs = []
    for _ in range(260):
        L = rnd.randrange(1500, 3500)
        o = rnd.randrange(0, len(genome) - L)
        s = _mutate(rnd, genome[o:o + L], rnd.choice([0.03, 0.05, 0.07]))
        seqs.append(O.rc(s) if rnd.random() < 0.5 else s)
    seqs += [genome[1000:4000]] * 3                   # identical reads: every ordered k-mer joins
    seqs += [("ACGTTGCA" * 300)[i:i + 2200] for i in range(4)] + ["ACGGT" * 500, "A" * 1800, "A" * 2100]
    fa = FastaData.from_strings(seqs)
    for kw in (dict(), dict(num_min_matches=1, threshold=0.0, max_shift=0.4)):
        p = MhapParams(num_hashes=128, ordered_sketch_size=600, **kw)
        want = O.run_self(fa, H=128, S=600, nthreads=8, num_min_matches=p.num_min_matches, threshold=p.threshold, max_shift=p.max_shift)
        monkeypatch.delenv("MHAP_OVERLAP", raising=False)
        a, sa = _self_lines(fa, p)
        monkeypatch.setenv("MHAP_OVERLAP", "lane")
        b, sb = _self_lines(fa, p)
        monkeypatch.delenv("MHAP_OVERLAP", raising=False)
        assert a == O.record_lines(want["records"]), kw
        assert b == a
        # the early "below the threshold" from position histograms, forced on (by default only candidate-rich chunks use it) and off
        for prune in ("1", "0"):
            monkeypatch.setenv("MHAP_OVERLAP_PRUNE", prune)
            c, sc = _self_lines(fa, p)
            assert c == a and sc["candidates_compared"] == sa["candidates_compared"], (kw, prune)
        monkeypatch.delenv("MHAP_OVERLAP_PRUNE")
        assert 0 < sa["slow_pairs"] < sa["candidates_compared"], sa
        assert sb["slow_pairs"] == 0 and sb["candidates_compared"] == sa["candidates_compared"]
        print("slow pairs", sa["slow_pairs"], "of", sa["candidates_compared"])
        assert len(a) > 1000


def test_fused_and_separate_kmer_hashing_agree(monkeypatch):
    """k=16/k2=12 packed strands are hashed inside the weight kernel (block-mix tables); raw-byte strands of the same batch
    and MHAP_FUSED_HASH=0 use hash_kmers_kernel.  All three give the oracle's sketches."""
    rnd = random.Random(99)
    seqs = [_rand_seq(rnd, rnd.randrange(20, 9000)) for _ in range(60)]
    seqs += [_rand_seq(rnd, 4000, "ACGTN"), _rand_seq(rnd, 12303), _rand_seq(rnd, 12304), _rand_seq(rnd, 16), _rand_seq(rnd, 27),
             "ACGT" * 1000, _rand_seq(rnd, 2500).lower()]
    fa = FastaData.from_strings(seqs)
    p = MhapParams(num_hashes=64, ordered_sketch_size=256, min_olap_length=0)
    _assert_sketch_parity(fa, p)
    with MinHashSearch(p) as ms:
        a = ms.sketch(fa)
    monkeypatch.setenv("MHAP_FUSED_HASH", "0")
    with MinHashSearch(p) as ms:
        b = ms.sketch(fa)
    for key in ("minhash", "ordered", "ordered_size", "status"):
        assert np.array_equal(a[key], b[key]), key


def test_perchain_minhash_switch_agrees(monkeypatch):
    """MHAP_MINHASH=perchain (every row in 64-bit registers, both launches) gives the bit-sliced rows' sketches — weight-1 strands
    and strands with repeated k-mers (second launch, own stream) alike."""
    rnd = random.Random(123)
    unit = _rand_seq(rnd, 150)
    seqs = [_rand_seq(rnd, rnd.randrange(300, 7000)) for _ in range(40)] + [unit * 12 + _rand_seq(rnd, 2500) for _ in range(6)]
    fa = FastaData.from_strings(seqs)
    p = MhapParams(num_hashes=128, ordered_sketch_size=256)
    _assert_sketch_parity(fa, p)
    with MinHashSearch(p) as ms:
        a = ms.sketch(fa)
    monkeypatch.setenv("MHAP_MINHASH", "perchain")
    with MinHashSearch(p) as ms:
        b = ms.sketch(fa)
    for key in ("minhash", "ordered", "ordered_size", "status"):
        assert np.array_equal(a[key], b[key]), key


def test_index_table_reuse_and_incremental_adds():
    """The inverted index of a fresh add is built while its reads are sketched and reused by later searches; an index that
    grows by a second add is rebuilt at search time.  Both give the one-shot result."""
    fa = mhap_amd.synth_reads(900, 2500, seed=31, error_rate=0.06)
    p = MhapParams(num_hashes=128, ordered_sketch_size=400)
    want, _ = _self_lines(fa, p)
    assert len(want) > 500
    with MinHashSearch(p) as ms:
        ms.add_data(fa)
        a = sorted(mhap_amd.records_to_lines(ms.find_matches()))
        b = sorted(mhap_amd.records_to_lines(ms.find_matches()))          # second search: table reused
        kt = ms.kernel_times()
    assert a == want and b == want
    assert kt["index_build"]["launches"] in (1, 2) and kt["index_query"]["launches"] == 2
    half = len(fa) // 2
    f1 = FastaData.from_strings([fa.sequence(i) for i in range(half)])
    f2 = FastaData.from_strings([fa.sequence(i) for i in range(half, len(fa))], id_offset=half)
    with MinHashSearch(p) as ms:
        ms.add_data(f1)
        first = sorted(mhap_amd.records_to_lines(ms.find_matches()))      # uses the eagerly built table of the first half
        ms.add_data(f2)
        c = sorted(mhap_amd.records_to_lines(ms.find_matches()))          # entry set changed: rebuilt
    assert c == want and 0 < len(first) < len(want)


def test_self_search_with_ids_that_do_not_rise_with_the_entries():
    """The self search skips tiles by id order only while the ids rise with the entries — looked at once per generation of the index, when the
    ids arrive (under the add's kernels since round 5), not in every search.  Ids descending, ids shuffled, and an index whose second add brings
    ids BELOW the first's (the first generation's answer must not outlive it): records against the oracle (J/impl/MinHashSearch.java:204-214,
    the `m.id < q.id` rule of the self search)."""
    fa = mhap_amd.synth_reads(260, 3000, seed=515, error_rate=0.1)
    p = MhapParams(num_hashes=256, ordered_sketch_size=600)
    for kind in ("descending", "shuffled"):
        ids = fa.ids[::-1].copy() if kind == "descending" else np.random.default_rng(5).permutation(fa.ids)
        fb = FastaData(fa.bases, fa.offsets, fa.lengths, ids)
        want = O.record_lines(O.run_self(fb, H=256, S=600, nthreads=8)["records"])
        got, _ = _self_lines(fb, p)
        assert got == want and len(want) > 50, kind
    half = len(fa) // 2
    hi, lo = fa.subset(np.arange(half, len(fa))), fa.subset(np.arange(half))   # the higher ids first
    both = FastaData(np.concatenate([hi.bases, lo.bases]), np.concatenate([hi.offsets, lo.offsets + len(hi.bases)]),
                     np.concatenate([hi.lengths, lo.lengths]), np.concatenate([hi.ids, lo.ids]))
    want = O.record_lines(O.run_self(both, H=256, S=600, nthreads=8)["records"])
    with MinHashSearch(p) as ms:
        ms.add_data(hi)
        first = ms.find_matches()                                          # (ids rising: tiles skipped)
        ms.add_data(lo)
        got = sorted(mhap_amd.records_to_lines(ms.find_matches()))
    assert got == want and 0 < len(first) < len(want)


def test_growing_batches_keep_the_read_back_buffer():
    """A small add followed by a much larger one re-allocates the pinned staging buffer; the (separate) pinned bounce buffer of
    the meta/record read-backs must survive it (round-1 use-after-free), and a destroyed handle frees both."""
    small = mhap_amd.synth_reads(40, 1500, seed=77, error_rate=0.05)
    big = mhap_amd.synth_reads(700, 2500, seed=31, error_rate=0.06)
    p = MhapParams(num_hashes=128, ordered_sketch_size=400)
    want, _ = _self_lines(big, p)
    for _ in range(2):
        with MinHashSearch(p) as ms:
            ms.add_data(small)
            assert len(ms.find_matches()) > 0
            ms.clear()
            ms.add_data(big)                       # staging grows past 1.125x the first batch
            got = sorted(mhap_amd.records_to_lines(ms.find_matches()))
            assert got == want
            q = ms.find_matches_stream(small)      # and a third, smaller staging round
            assert q.shape[0] >= 0


@pytest.mark.timeout(2400)
def test_full_config2_against_oracle():
    """BASELINE configs[1] at FULL size (100 000 reads x 10 kb, H = 512): every record of the GPU run against the CPU oracle's run of
    the same reads (a few minutes of host time on the container's CPU quota), and against the fingerprint bench.py prints."""
    import hashlib
    from mhap_amd import workloads as W
    fa = W.config_reads("c2")
    assert len(fa) == 100000
    got, st = _self_lines(fa, MhapParams())
    sha = hashlib.sha256("\n".join(got).encode()).hexdigest()
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        threads = max(1, int(round(float(q) / float(per)))) if q != "max" else len(os.sched_getaffinity(0))
    except Exception:
        threads = len(os.sched_getaffinity(0))
    want = O.run_self(fa, nthreads=threads, cap=1 << 22)
    assert got == O.record_lines(want["records"])
    assert st["candidates_compared"] == want["compared"] and st["table_elements"] == want["elements"]
    assert len(got) == 41915 and sha.startswith("8f75366010aaae7d")      # the fingerprint of the round-1 and round-2 bench lines


@pytest.mark.timeout(2400)
def test_full_config4_on_one_gpu_properties_and_subset_parity():
    """BASELINE configs[3] at FULL size — 1 000 000 reads x 15 kb, 2 M strands, H = 512, S = 1536 (quoted as an 8-GPU job; the
    tables, the index and the scratch of the whole job fit one MI355X) — through the streaming-free C ABI path: the size-independent
    properties of every record, the record count and checksum the benchmark prints for this configuration, and full parity with
    the CPU oracle on the pairs among a 3 000-read subset (without -f a pair's record depends on its two reads only)."""
    import hashlib
    from mhap_amd import workloads as W
    fa = W.config_reads("c4")
    assert len(fa) == 1000000
    with MinHashSearch(MhapParams()) as ms:
        ms.add_data(fa)
        recs = ms.find_matches()
        st = ms.stats()
    assert st["strands_indexed"] == 2000000 and st["queries_searched"] == 1000000
    assert len(recs) == 313605                                              # profiles/r02_bench_c4.json, profiles/r03_e2e_probe.txt
    assert np.all(recs["to_id"] < recs["from_id"]) and np.all((recs["from_id"] <= 1000000) & (recs["to_id"] >= 1))
    assert np.all((recs["score"] >= 0.78) & (recs["score"] <= 1.0) & (recs["raw"] >= 3))
    assert np.all((recs["a1"] >= 0) & (recs["a1"] <= recs["a2"]) & (recs["a2"] <= 15000 - 11) & (recs["alen"] == 15000) & (recs["blen"] == 15000))
    assert np.all((recs["b1"] >= -1) & (recs["b2"] <= 15000) & (recs["b1"] <= recs["b2"]))
    pairs = np.unique(np.stack([recs["from_id"], recs["to_id"], recs["to_rc"].astype(np.int64)], 1), axis=0)
    assert len(pairs) == len(recs)                                          # one record per (query, stored strand)
    lines = mhap_amd.records_to_lines(recs)
    csum = 0
    for ln in lines:
        csum = (csum + int.from_bytes(hashlib.sha256(ln.encode()).digest()[:8], "little")) & ((1 << 64) - 1)
    assert len(set(lines)) == len(lines)
    # pairs among reads 1..3000 exactly as the oracle scores them
    nsub = 3000
    want = O.record_lines(O.run_self(fa.subset(np.arange(nsub)), nthreads=16, cap=1 << 20)["records"])
    m = (recs["from_id"] <= nsub) & (recs["to_id"] <= nsub)
    assert sorted(mhap_amd.records_to_lines(recs[m])) == want and len(want) >= 1
    print(f"c4 full: {len(recs)} records, checksum {csum:016x}, {len(want)} of them among the first {nsub} reads")


def test_full_config5_rank_properties_and_subset_parity(tmp_path):
    """One rank's share of BASELINE configs[4] at its true size — 625 000 reads x 12 kb, 1.25 M index entries (ten ranges of the
    dense query tier), planted repeat family, generated -f file, --filter-threshold 1e-5 — as a self-overlap job on one GPU: the
    size-independent properties of every record, and full parity with the CPU oracle, under the SAME filter, on the pairs among a
    2 000-read subset (a pair's record depends on its two reads and the filter only)."""
    from mhap_amd import workloads as W
    fa = W.config_reads("c5rank")
    n = len(fa)
    assert n == 625000
    ffile = tmp_path / "kmers.txt"
    W.write_filter_file(fa, str(ffile), max_reads=2000)
    flt = mhap_amd.FrequencyCounts.from_file(str(ffile), filter_cutoff=1e-5, repeat_weight=0.9)
    assert (flt.fractions >= 1e-5).sum() > 100
    with MinHashSearch(MhapParams(), kmer_filter=flt) as ms:
        ms.add_data(fa)
        recs = ms.find_matches()
        st, kt = ms.stats(), ms.kernel_times()
    assert st["strands_indexed"] == 2 * n and st["queries_searched"] == n
    assert len(recs) > 5000000 and st["candidates_compared"] > 5 * len(recs) // 2
    assert np.all(recs["to_id"] < recs["from_id"]) and np.all((recs["from_id"] <= n) & (recs["to_id"] >= 1))
    assert np.all((recs["score"] >= 0.78) & (recs["score"] <= 1.0) & (recs["raw"] >= 3))
    assert np.all((recs["a1"] >= 0) & (recs["a1"] <= recs["a2"]) & (recs["a2"] <= 12000 - 11) & (recs["alen"] == 12000) & (recs["blen"] == 12000))
    assert np.all((recs["b1"] >= -1) & (recs["b2"] <= 12000) & (recs["b1"] <= recs["b2"]))
    key = (recs["from_id"].astype(np.int64) << 21) | (recs["to_id"].astype(np.int64) << 1) | recs["to_rc"].astype(np.int64)
    assert len(np.unique(key)) == len(recs)                                 # one record per (query, stored strand)
    # the dense tier covered the index in ranges, each long bucket streamed once (the query stage is a fraction of the sketch time)
    # (round 3's dense tier took 14 us per query here: 8.7 s, five times the MinHash kernel; a cold run redoes the first chunk once its
    #  candidates outgrow the initial buffer, which doubles that chunk's query time)
    assert st["index_splits"] > 0
    print(f"c5rank: index_query {kt['index_query']['ms']:.0f} ms against minhash {kt['minhash']['ms']:.0f} ms "
          f"(ratio {kt['index_query']['ms'] / max(kt['minhash']['ms'], 1e-9):.2f}; a figure, not an assertion: timing does not belong in the parity gate)")
    nsub = 2000
    oflt = O.Filter(flt.hashes, flt.fractions, 1e-5, 0.9, 3.0, False)
    want = O.record_lines(O.run_self(fa.subset(np.arange(nsub)), nthreads=16, flt=oflt, cap=1 << 22)["records"])
    m = (recs["from_id"] <= nsub) & (recs["to_id"] <= nsub)
    assert sorted(mhap_amd.records_to_lines(recs[m])) == want and len(want) >= 20
    print(f"c5rank: {len(recs)} records from {st['candidates_compared']} candidates ({st['slow_pairs']} through the per-lane kernel), "
          f"{len(want)} among the first {nsub} reads; kernel ms " + ", ".join(f"{k}={v['ms']:.0f}" for k, v in kt.items() if v["ms"] > 0))


def _check_c3_records(recs, L, ids_ok=True):
    """Properties every record of a default-flag self search must have (J/impl/MinHashSearch.java:215-241, MatchResult.java:46-65)."""
    assert np.all(recs["to_id"] < recs["from_id"]) and np.all((recs["score"] >= 0.78) & (recs["score"] <= 1.0) & (recs["raw"] >= 3))
    assert np.all(recs["alen"] == L[recs["from_id"] - 1]) and np.all(recs["blen"] == L[recs["to_id"] - 1])
    assert np.all((recs["a1"] >= 0) & (recs["a1"] <= recs["a2"]) & (recs["a2"] <= recs["alen"] - 11) & (recs["b1"] >= -1) & (recs["b2"] <= recs["blen"]))
    key = (recs["from_id"].astype(np.int64) << 32) | (recs["to_id"].astype(np.int64) << 1) | recs["to_rc"].astype(np.int64)
    assert len(np.unique(key)) == len(recs)                                 # one record per (query, stored strand)


def _subset_parity(fa, recs, idx, nthreads=16):
    """GPU records among the reads `idx` of the full run == the oracle run on those reads alone (sketches and pair decisions do not
    depend on the other reads without -f; the ids of a subset are the reads' own)."""
    idx = np.unique(np.asarray(idx, dtype=np.int64))
    sub = fa.subset(idx)
    want = O.record_lines(O.run_self(sub, nthreads=nthreads, cap=1 << 22)["records"])
    inset = np.zeros(int(fa.ids.max()) + 2, dtype=bool)
    inset[sub.ids] = True
    m = inset[recs["from_id"]] & inset[recs["to_id"]]
    got = sorted(mhap_amd.records_to_lines(recs[m]))
    assert got == want
    return len(want)


def test_config5_one_percent_family_subset_parity(tmp_path):
    """The HARDER repeat structure of BASELINE configs[4]: the planted 300-bp family at 1 % divergence from its consensus (the c5rank / c5
    configurations use 5 %, workloads.py says why; the 1 % family yields twelve times the candidates per read at a rank's size and was only
    ever probed for time — VERDICT r05).  100 000 reads x 12 kb, generated -f file, --filter-threshold 1e-5: properties of every record, and
    parity with the oracle under the same filter on the pairs among the first 1 500 reads."""
    from mhap_amd import workloads as W
    cfg = W.CONFIGS["c5rank"]
    el, sp, _ = cfg["repeats"]
    n = 100000
    fa = mhap_amd.synth_reads(n, cfg["length"], seed=cfg["seed"], repeats=(el, sp, 0.01))
    ffile = tmp_path / "kmers.txt"
    W.write_filter_file(fa, str(ffile), max_reads=2000)
    flt = mhap_amd.FrequencyCounts.from_file(str(ffile), filter_cutoff=1e-5, repeat_weight=0.9)
    assert (flt.fractions >= 1e-5).sum() > 100
    with MinHashSearch(MhapParams(), kmer_filter=flt) as ms:
        ms.add_data(fa)
        recs = ms.find_matches()
        st, kt = ms.stats(), ms.kernel_times()
    assert st["strands_indexed"] == 2 * n and st["queries_searched"] == n and len(recs) > 1000000
    assert np.all(recs["to_id"] < recs["from_id"]) and np.all((recs["score"] >= 0.78) & (recs["score"] <= 1.0) & (recs["raw"] >= 3))
    assert np.all((recs["a1"] >= 0) & (recs["a1"] <= recs["a2"]) & (recs["a2"] <= 12000 - 11) & (recs["alen"] == 12000) & (recs["blen"] == 12000))
    key = (recs["from_id"].astype(np.int64) << 21) | (recs["to_id"].astype(np.int64) << 1) | recs["to_rc"].astype(np.int64)
    assert len(np.unique(key)) == len(recs)
    nsub = 1500
    oflt = O.Filter(flt.hashes, flt.fractions, 1e-5, 0.9, 3.0, False)
    want = O.record_lines(O.run_self(fa.subset(np.arange(nsub)), nthreads=16, flt=oflt, cap=1 << 22)["records"])
    m = (recs["from_id"] <= nsub) & (recs["to_id"] <= nsub)
    assert sorted(mhap_amd.records_to_lines(recs[m])) == want and len(want) >= 20
    print(f"c5, 1 % family, {n} reads: {len(recs)} records from {st['candidates_compared']} candidates ({st['candidates_compared'] / n:.0f} per read, "
          f"{st['slow_pairs']} through the per-lane kernel), {len(want)} among the first {nsub} reads; kernel ms " + ", ".join(f"{k}={v['ms']:.0f}" for k, v in kt.items() if v["ms"] > 0))


def test_config3_ecoli_shaped_reads_at_scale_subset_parity():
    """Stand-in for BASELINE configs[2] (real E. coli P6-C4 reads: the file is in neither tree) at its scale AND with its repeat structure
    (round 6; the round-5 stand-in had a repeat-free genome): 90 000 reads of a log-normal length mix (median 8 kb, tail to 45 kb: reads
    past 24 591 bases take the materialised-hash path, short ones fall under --min-olap-length, runs of N make raw-byte strands) drawn
    from a 4.6-Mbp genome with seven ~5-kb operon copies at 99 %, twelve IS copies of 0.8-1.3 kb in three families and forty tandem
    repeats (k-mers of weight > 1 under the default tf weighting, no -f) — workloads.ecoli_like_reads.  Default flags.  Properties of
    every record; full parity with the oracle on the pairs among the first 1 500 reads AND among the 500 reads with the most records
    (the reads across operon / IS copies: seven-fold candidate sets, long buckets, duplicated-hash groups in the join)."""
    fa, genome, placed = W.ecoli_like_reads(90000)
    L = fa.lengths
    assert (L > 24591).sum() > 300 and (L < 116).sum() >= 1 and len(placed) >= 19 + 40
    with MinHashSearch(MhapParams()) as ms:
        ms.add_data(fa)
        recs = ms.find_matches()
        st = ms.stats(); kt = ms.kernel_times()
    assert st["queries_searched"] == int((L >= 116).sum()) and len(recs) > 100000
    _check_c3_records(recs, L)
    per_read = np.bincount(recs["from_id"], minlength=len(fa) + 2) + np.bincount(recs["to_id"], minlength=len(fa) + 2)
    busiest = np.argsort(-per_read)[:500] - 1                              # ids are 1-based
    # the repeat copies are really met: the busiest reads have several times the records of a read at unique 30x coverage
    assert per_read[busiest + 1].min() > 2 * np.median(per_read[per_read > 0])
    n1 = _subset_parity(fa, recs, np.arange(1500))
    n2 = _subset_parity(fa, recs, busiest)
    assert n1 >= 5 and n2 >= 2000
    print(f"c3 stand-in (E. coli-shaped): {len(recs)} records from {st['candidates_compared']} candidates ({st['slow_pairs']} through the per-lane kernel); "
          f"parity on {n1} records among the first 1500 reads and {n2} among the 500 busiest reads; {int((L > 24591).sum())} reads on the materialised-hash path; "
          "kernel ms " + ", ".join(f"{k}={v['ms']:.0f}" for k, v in kt.items() if v["ms"] > 0))


def test_config3_real_reads_when_supplied(tmp_path):
    """BASELINE configs[2]: real E. coli PacBio P6-C4 reads, default parameters.  The file is in neither tree; a box that has it names it
    in MHAP_C3_FASTA and this test is the parity gate that goes with `bench.py --config c3`: properties of every record, oracle parity
    on the pairs among the first 1 500 reads and among the 500 busiest reads, and the native driver on the same file — identical
    records, MHAP's stderr contract (J/main/MhapMain.java:463-476,576)."""
    path = W.c3_fasta_path()
    if path is None:
        pytest.skip("C3 skipped: MHAP_C3_FASTA is not set / not readable")
    import subprocess
    fa = FastaData.from_file(path)
    L = fa.lengths
    with MinHashSearch(MhapParams()) as ms:
        ms.add_data(fa)
        recs = ms.find_matches()
        st = ms.stats()
    assert st["queries_searched"] == int((L >= 116).sum()) and len(recs) > 0
    _check_c3_records(recs, L)
    per_read = np.bincount(recs["from_id"], minlength=len(fa) + 2) + np.bincount(recs["to_id"], minlength=len(fa) + 2)
    n1 = _subset_parity(fa, recs, np.arange(min(1500, len(fa))))
    n2 = _subset_parity(fa, recs, np.argsort(-per_read)[:500] - 1 if len(fa) > 500 else np.arange(len(fa)))
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mhap_amd", "lib", "mhap-hip")
    r = subprocess.run([cli, "-s", path], capture_output=True, text=True, timeout=3600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = sorted(l for l in r.stdout.split("\n") if l)
    assert lines == sorted(mhap_amd.records_to_lines(recs))
    nsk = 2 * int((L >= 116).sum())
    assert f"Processed {nsk} unique sequences (fwd and rev)." in r.stderr and "Time (s) to read and hash from file:" in r.stderr
    assert "Time (s) to score and output to self:" in r.stderr and f"Total matches found: {len(recs)}" in r.stderr
    print(f"c3 (real reads, {len(fa)} reads): {len(recs)} records; parity on {n1} + {n2} records of the two subsets; native driver identical")


def test_config4_read_shape_slice():
    """BASELINE configs[3]/[4] read shapes (15 kb and 12 kb reads, H=512, S=1536: more than 12288 k-mers per strand takes the
    24-k-mers-per-lane weight kernel, 8 bit-sliced MinHash rows) on a slice of reads: full record parity with the oracle."""
    for L, n, seed in ((15000, 260, 4), (12000, 300, 5)):
        fa = mhap_amd.synth_reads(n, L, seed=0x4D484150 ^ seed, error_rate=0.15)
        p = MhapParams()
        want = O.run_self(fa, nthreads=8)
        got, st = _self_lines(fa, p)
        assert got == O.record_lines(want["records"]), L
        assert len(got) > 20 and st["slow_pairs"] == 0


def test_config5_shape_filter_and_repeats(tmp_path, monkeypatch):
    """BASELINE configs[4] shape: 12 kb reads with a planted repeat family, a generated -f k-mer filter file and
    --filter-threshold 1e-5 (tf-idf weights: 3 for k-mers absent from the filter, 1..3 for the popular ones, times the
    multiplicity) — the weighted k-mers run on the bit-sliced MinHash rows.  Sketch and record parity with the oracle."""
    from mhap_amd import workloads as W
    fa = W.config_reads("c5slice", reads=420)
    ffile = tmp_path / "kmers.txt"
    nlines = W.write_filter_file(fa, str(ffile), max_reads=420, min_fraction=5e-6)
    assert nlines > 50
    flt = mhap_amd.FrequencyCounts.from_file(str(ffile), filter_cutoff=1e-5, repeat_weight=0.9)
    assert (flt.fractions >= 1e-5).sum() > 20
    oflt = O.Filter(flt.hashes, flt.fractions, 1e-5, 0.9, 3.0, False)
    p = MhapParams()
    _assert_sketch_parity(fa.subset(range(6)), p, flt, oflt)
    want = O.run_self(fa, nthreads=8, flt=oflt)
    got, st = _self_lines(fa, p, flt)
    assert got == O.record_lines(want["records"]) and len(got) > 50
    monkeypatch.setenv("MHAP_OVERLAP_PRUNE", "1")            # second stage with the early "below the threshold" forced on
    got_p, st_p = _self_lines(fa, p, flt)
    monkeypatch.delenv("MHAP_OVERLAP_PRUNE")
    assert got_p == got and st_p["candidates_compared"] == st["candidates_compared"]
    # no filter, same reads: plain tf weights (repeated k-mers inside a read take the weight-2.. classes)
    want2 = O.run_self(fa, nthreads=8)
    got2, _ = _self_lines(fa, p)
    assert got2 == O.record_lines(want2["records"])


def test_weight_classes_and_large_weights():
    """Reads built to hit every weight path of the MinHash kernel: tandem repeats (multiplicities 2..40 -> classes 2..6 and the
    per-chain rows for larger weights), a read that is one k-mer repeated (a single distinct k-mer of weight n), tf-idf with a
    large --repeat-idf-scale (uniform weight 12 on the bit-sliced rows, past the fine jump tables), and short strands."""
    rnd = random.Random(11)
    unit = _rand_seq(rnd, 40)
    seqs = []
    for reps in (2, 3, 5, 7, 12, 40):
        seqs.append(_rand_seq(rnd, 1500) + unit * reps + _rand_seq(rnd, 1200))
    seqs.append("ACGT" * 500)                      # 4 distinct k-mers, weight ~ 500 each
    seqs.append("A" * 900 + _rand_seq(rnd, 2500))   # one k-mer of weight 885 next to unique ones
    seqs.append(_rand_seq(rnd, 130))               # shorter than one per-chain row
    seqs.append(_rand_seq(rnd, 700))
    base = _rand_seq(rnd, 6000)
    seqs += [base[i:i + 3500] for i in (0, 500, 1000, 1500, 2000, 2500)]
    fa = FastaData.from_strings(seqs)
    for H in (64, 512):
        p = MhapParams(num_hashes=H, ordered_sketch_size=600)
        _assert_sketch_parity(fa, p)
    # tf-idf: every k-mer outside a (tiny) filter gets weight round(12.0) = 12
    kmers = [unit[i:i + 16] for i in range(0, 20)]
    hashes = np.array([int(O.kmer_hashes64(k, 16, True)[0]) for k in kmers], dtype=np.int64)
    fracs = np.linspace(2e-3, 1e-4, len(kmers))
    flt = mhap_amd.FrequencyCounts(hashes, fracs, 1e-5, 0.9, 12.0, True)
    oflt = O.Filter(hashes, fracs, 1e-5, 0.9, 12.0, True)
    p = MhapParams(num_hashes=256, ordered_sketch_size=600)
    _assert_sketch_parity(fa, p, flt, oflt)
    want = O.run_self(fa, H=256, S=600, nthreads=8, flt=oflt)
    got, _ = _self_lines(fa, p, flt)
    assert got == O.record_lines(want["records"])


@pytest.mark.parametrize("split", ["1", "0"])
def test_few_weighted_strands_go_round_the_waves_of_a_workgroup(split, monkeypatch):
    """A launch of a few strands with repeated k-mers (a small job: C1 has six) gives every strand a WORKGROUP whose four waves take the
    strand's rows in turn — bit-sliced rows of 2 048 k-mers and per-chain rows alike — and merge their minima in LDS (SPLIT in
    minhash_kernel); MHAP_MINHASH_SPLIT=0 keeps one wave per strand.  Strands of one to seven rows with tandem repeats of several
    multiplicities, against the oracle (J/sketch/MinHashSketch.java:66-131), both ways."""
    monkeypatch.setenv("MHAP_MINHASH_SPLIT", split)
    rnd = random.Random(77)
    seqs = []
    for n, reps in ((1800, 2), (4300, 2), (6200, 3), (9000, 2), (14000, 5), (14000, 1), (2500, 9), (7000, 33)):
        unit = _rand_seq(rnd, 45)
        body = _rand_seq(rnd, n)
        cut = n // 3
        seqs.append(body[:cut] + unit * reps + body[cut:] + (unit if reps > 1 else ""))
    seqs.append(_rand_seq(rnd, 5000))      # (weight-1 strands: the other launch)
    seqs.append(_rand_seq(rnd, 300))
    fa = FastaData.from_strings(seqs)
    for H in (64, 512, 1100):
        _assert_sketch_parity(fa, MhapParams(num_hashes=H, ordered_sketch_size=600))


def test_random_flag_and_read_mixes():
    """A few draws of tests/fuzz_parity.py (random flags incl. k != 16 and odd k2, repeat families, N runs): 0 mismatches."""
    import fuzz_parity
    assert fuzz_parity.main(8, 777) == 0


def test_minhash_queue_overflow_redo_path_against_the_oracle(tmp_path):
    """The weight-1 MinHash kernel defers a row's candidates to a per-wave queue; a row that overflows it ("never seen" with 2 047 entries)
    is redone one k-mer at a time (sketch_kernels.hip, minhash_w1_kernel: `if (!ok)`).  That path was exact by inspection only (VERDICT
    r04 2d).  The variant build `qcap64` (mhap_amd/build.py VARIANTS: -DMH_QCAP=64 -> 31 entries per wave; built by
    __graft_entry__.build()) overflows on every full row, so here the redo runs for whole strands (more strands than resident waves)
    AND for the row items of the launch's tail (merge buffer + finish kernel), on strands of one to three rows, and must reproduce
    a2 (J/sketch/MinHashSketch.java:130-154) bit for bit.  The library path is fixed at import, hence the child process."""
    import subprocess
    import sys
    from mhap_amd import build as B
    lib = B.variant_path("qcap64")
    assert os.path.exists(lib), f"{lib} missing: __graft_entry__.build() builds it (python -m mhap_amd.build --variants)"
    rnd = random.Random(2105)
    seqs = [_rand_seq(rnd, rnd.choice((900, 2063, 2100, 3000, 4111, 4200, 5300))) for _ in range(2300)]   # 4 600 strands > 4 096 resident waves
    fasta = tmp_path / "reads.fasta"
    with open(fasta, "w") as fh:
        for i, s in enumerate(seqs):
            fh.write(f">r{i}\n{s}\n")
    out = tmp_path / "mh.npy"
    child = ("import sys, numpy as np, mhap_amd\n"
             "from mhap_amd import FastaData, MhapParams, MinHashSearch\n"
             "assert mhap_amd.api._LIB_PATH.endswith('libmhaphip_qcap64.so'), mhap_amd.api._LIB_PATH\n"
             "fa = FastaData.from_file(sys.argv[1])\n"
             "with MinHashSearch(MhapParams(num_hashes=int(sys.argv[3]), ordered_sketch_size=64)) as ms:\n"
             "    sk = ms.sketch(fa)\n"
             "    kt = ms.kernel_times()\n"
             "assert kt['minhash']['launches'] > 0\n"
             "np.save(sys.argv[2], sk['minhash']); np.save(sys.argv[2] + '.status.npy', sk['status'])\n")
    env = dict(os.environ, MHAP_LIB_PATH=lib, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    from concurrent.futures import ThreadPoolExecutor
    # H = 512: every full row overflows on its own count.  H = 16 / 13: a first row queues about 2 x H entries — the count often FITS the 31
    # entries and only the re-queued rests of multi-chain masks run over: the case that was dropped silently until round 5 (found by the
    # fuzz sweep on this variant; with the shipped 2 047 entries it is a first row at --num-hashes 1024)
    for H in (512, 16, 13):
        r = subprocess.run([sys.executable, "-c", child, str(fasta), str(out), str(H)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        mh, st = np.load(out), np.load(str(out) + ".status.npy")
        assert (st == 0).all()

        def differs(i):   # (the oracle call releases the GIL: ctypes)
            return sum(int(mh[2 * i + strand].tolist() != O.minhash(s, 16, H)[1].tolist()) for strand, s in ((0, seqs[i]), (1, O.rc(seqs[i]))))
        with ThreadPoolExecutor(16) as ex:
            bad = sum(ex.map(differs, range(len(seqs))))
        assert bad == 0, f"H = {H}: {bad} of {2 * len(seqs)} MinHash rows differ from the oracle on the queue-overflow path"


def test_minhash_at_1024_hashes_where_a_first_row_fills_the_queue():
    """--num-hashes 1024 with the shipped library: a strand's first row queues about two candidates per slot — 2 048 entries against the
    queue's 2 047 — so some rows overflow on their count (exact redo) and some fit with their re-queued rests running over (redo since round 5,
    silently dropped before).  MinHash rows against the oracle."""
    rnd = random.Random(1024)
    seqs = [_rand_seq(rnd, rnd.choice((2063, 2100, 2600, 4200))) for _ in range(260)]
    fa = FastaData.from_strings(seqs)
    with MinHashSearch(MhapParams(num_hashes=1024, ordered_sketch_size=64)) as ms:
        sk = ms.sketch(fa)
    assert (sk["status"] == 0).all()
    from concurrent.futures import ThreadPoolExecutor

    def differs(i):
        return sum(int(sk["minhash"][2 * i + strand].tolist() != O.minhash(s, 16, 1024)[1].tolist()) for strand, s in ((0, seqs[i]), (1, O.rc(seqs[i]))))
    with ThreadPoolExecutor(16) as ex:
        bad = sum(ex.map(differs, range(len(seqs))))
    assert bad == 0, f"{bad} of {2 * len(seqs)} MinHash rows differ from the oracle at --num-hashes 1024"


@pytest.mark.parametrize("H", [1024, 2048, 3000])
def test_self_overlap_at_large_num_hashes(H):
    """--num-hashes beyond the 768 the first query tier keeps its first look for, up into the range where its indexed-vector write ran past the
    vectors (a memory fault from 2 752 on until round 5) and where a strand's first row fills the default candidate queue: the whole path —
    sketches, index, every query handed to the dense tier, second stage — against the oracle (J/impl/MinHashSearch.java:150-251)."""
    fa = mhap_amd.synth_reads(160, 3000, seed=4242 + H, error_rate=0.08)
    p = MhapParams(num_hashes=H, ordered_sketch_size=512)
    want = O.record_lines(O.run_self(fa, H=H, S=512, nthreads=16)["records"])
    got, _ = _self_lines(fa, p)
    assert got == want and len(want) > 100, (len(got), len(want))


@pytest.mark.parametrize("err", [0.0, 0.01, 0.04, 0.08])
def test_low_error_reads_take_the_wide_join_passes(err, monkeypatch):
    """Reads better than the 15 %-error ones MHAP was built for: two overlapping reads then join hundreds of their bottom 12-mers (all of
    them at error 0), more than the 128 the join kernel keeps per pair.  Such pairs go through the same kernel compiled with room for 512
    and 1 536 (search_kernels_wide.hip / _wide2.hip) instead of the per-lane merge; records against the oracle
    (J/sketch/BottomOverlapSketch.java:592-630), with the passes on (default), with one, and off — and nearly no pair is left for the lane kernel."""
    fa = mhap_amd.synth_reads(260, 5000, seed=900 + int(err * 100), error_rate=err)
    p = MhapParams()
    want = O.record_lines(O.run_self(fa, nthreads=16)["records"])
    assert len(want) > 200
    for wide in (None, "1", "0"):
        if wide is None:
            monkeypatch.delenv("MHAP_JOIN_WIDE", raising=False)
        else:
            monkeypatch.setenv("MHAP_JOIN_WIDE", wide)
        with MinHashSearch(p) as ms:
            ms.add_data(fa)
            got = sorted(mhap_amd.records_to_lines(ms.find_matches()))
            st = ms.stats()
        assert got == want, (err, wide)
        if wide is None and err <= 0.04:
            assert st["slow_pairs"] * 20 <= st["candidates_compared"], (err, st["slow_pairs"], st["candidates_compared"])
        if wide == "0" and err <= 0.01:
            assert st["slow_pairs"] * 2 >= st["candidates_compared"]      # (without the passes most pairs are the lane kernel's: the test sees them)
    monkeypatch.delenv("MHAP_JOIN_WIDE", raising=False)
