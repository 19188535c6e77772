"""CPU tests of the product's host logic: the C-ABI library loads and exports every symbol the header declares,
the kernels' __host__ __device__ arithmetic (hash windows, second-stage lane) equals the oracle, and the reference's
IO conventions (FASTA ids, record text format) hold.  No compute entry point is called here (no GPU)."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

import oracle_lib as O
import mhap_amd
from mhap_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    lib = mhap_amd.load_library()
    hdr = open(os.path.join(ROOT, "include", "mhap_hip.h")).read()
    declared = set(re.findall(r"\b(mhap_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mhap_record_sink"}
    assert declared, "no declarations parsed"
    assert declared == set(api.EXPORTED_SYMBOLS)
    for name in sorted(declared):
        assert getattr(lib, name) is not None


def test_struct_layouts_match_header():
    assert C.sizeof(api._Params) == 8 * 4 + 3 * 8
    assert api.RECORD_DTYPE.itemsize == 64
    assert O.ORC_RECORD.itemsize == 64


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mhap_amd.MhapError):
        mhap_amd.MinHashSearch(mhap_amd.MhapParams())


def _hash_windows(seq, k, k2):
    lib = mhap_amd.load_library()
    s = seq.encode("latin-1")
    o64 = np.zeros(max(len(s) - k + 1, 1), dtype=np.int64)
    o32 = np.zeros(max(len(s) - k2 + 1, 1), dtype=np.int32)
    assert lib.mhap_selftest_hash_windows(s, C.c_int32(len(s)), C.c_int32(k), C.c_int32(k2), api._ptr(o64), api._ptr(o32)) == 0
    return o64[:max(len(s) - k + 1, 0)], o32[:max(len(s) - k2 + 1, 0)]


@pytest.mark.parametrize("k,k2", [(16, 12), (16, 14), (15, 13), (21, 11), (8, 5), (24, 12), (13, 16), (1, 1), (7, 3)])
def test_kernel_hash_arithmetic_matches_oracle(k, k2):
    rnd = random.Random(k * 100 + k2)
    seq = "".join(rnd.choice("ACGTNRY") for _ in range(211))
    h64, h32 = _hash_windows(seq, k, k2)
    assert h64.tolist() == O.kmer_hashes64(seq, k).tolist()
    assert h32.tolist() == O.kmer_hashes32(seq, k2).tolist()


def test_bit_transpose_used_by_bitsliced_minhash():
    lib = mhap_amd.load_library()
    rnd = np.random.default_rng(3)
    a = rnd.integers(0, 2**32, size=32, dtype=np.uint64).astype(np.uint32)
    b = a.copy()
    assert lib.mhap_selftest_transpose32(api._ptr(b)) == 0
    for r in range(32):
        for c in range(32):
            assert (int(b[r]) >> c) & 1 == (int(a[c]) >> r) & 1


def test_xorshift_jump_tables_match_stepping():
    lib = mhap_amd.load_library()
    M = (1 << 64) - 1

    def step(x):
        x ^= (x << 21) & M
        x ^= x >> 35
        x ^= (x << 4) & M
        return x
    rnd = random.Random(5)
    out = C.c_uint64()
    for n in (0, 1, 15, 16, 17, 63, 64, 65, 96, 97, 128, 192, 193, 300, 511, 512, 513, 1000, 1536, 1537, 3071, 16384):   # past 96 steps: coarse + fine table
        key = rnd.getrandbits(64)
        x = key
        for _ in range(n):
            x = step(x)
        assert lib.mhap_selftest_xorshift_jump(C.c_uint64(key), C.c_int32(n), C.byref(out)) == 0
        assert out.value == x, n


def test_xorshift_unjump_recovers_the_key():
    """The weight-1 MinHash kernel keeps only a slot's minimal chain value and turns it back into the winning k-mer key
    (inverse jump tables + up to 3 forward steps): unjump(step^n(key), n) == key for every slot count it can meet."""
    lib = mhap_amd.load_library()
    M = (1 << 64) - 1

    def step(x):
        x ^= (x << 21) & M
        x ^= x >> 35
        x ^= (x << 4) & M
        return x
    rnd = random.Random(6)
    out = C.c_uint64()
    for n in list(range(1, 14)) + [63, 64, 65, 255, 256, 257, 510, 511, 512, 513, 768, 1023, 8192]:
        for key in (rnd.getrandbits(64), 1, M, 0):
            x = key
            for _ in range(n):
                x = step(x)
            assert lib.mhap_selftest_xorshift_unjump(C.c_uint64(x), C.c_int32(n), C.byref(out)) == 0
            assert out.value == key, (n, key)


def test_filter_kmer_hash_is_canonical_when_rc():
    lib = mhap_amd.load_library()
    out = C.c_int64()
    for kmer in ("ACGTACGTACGTACGT", "TTTTTTTTTTTTTTTT", "GATTACAGATTACAGA", "ACGTN"):
        for do_rc in (0, 1):
            assert lib.mhap_hash_kmer(kmer.encode(), C.c_int32(len(kmer)), C.c_int32(do_rc), C.byref(out)) == 0
            assert out.value == int(O.kmer_hashes64(kmer, len(kmer), bool(do_rc))[0])


def _lane(A, lenA, B, lenB, max_shift=0.2, stride=1):
    lib = mhap_amd.load_library()
    A = np.ascontiguousarray(A, dtype=np.int32)
    B = np.ascontiguousarray(B, dtype=np.int32)
    out = np.zeros(8, dtype=np.int32)
    assert lib.mhap_selftest_overlap_lane(api._ptr(A), C.c_int32(A.shape[0]), C.c_int32(lenA), api._ptr(B), C.c_int32(B.shape[0]),
                                          C.c_int32(lenB), C.c_double(max_shift), C.c_int32(stride), api._ptr(out)) == 0
    return dict(zip(["empty", "raw", "a1", "a2", "b1", "b2", "inter", "k"], out.tolist()))


def test_second_stage_lane_matches_oracle():
    rnd = random.Random(99)
    nonempty = 0
    for trial in range(40):
        la, lb = rnd.randint(300, 3000), rnd.randint(300, 3000)
        g = "".join(rnd.choice("ACGT") for _ in range(max(la, lb) + 600))
        off = rnd.randint(0, 500)
        err = rnd.choice([0.0, 0.01, 0.04, 0.08])

        def noisy(s):
            return "".join((rnd.choice("ACGT") if rnd.random() < err else c) for c in s)
        a, b = noisy(g[:la]), noisy(g[off:off + lb])
        if trial % 7 == 0:
            b = O.rc(b)                                   # wrong strand: (nearly) EMPTY
        if trial % 5 == 1:
            rep = "".join(rnd.choice("ACGT") for _ in range(12))
            a = a[:150] + rep * rnd.randint(3, 30) + a[150:]
            b = b[:90] + rep * rnd.randint(3, 30) + b[90:]
        if trial % 9 == 4:
            a, b = "A" * 400 + a, "A" * 300 + b           # long equal-hash runs
        S = rnd.choice([32, 128, 1536])
        _, A, lenA = O.ordered(a, 12, S)
        _, B, lenB = O.ordered(b, 12, S)
        ms = rnd.choice([0.2, 0.05, 0.5])
        want = O.overlap(A, lenA, B, lenB, max_shift=ms)
        got = _lane(A, lenA, B, lenB, max_shift=ms, stride=rnd.choice([1, 3, 64]))
        assert got["empty"] == want["empty"], trial
        if not want["empty"]:
            nonempty += 1
            for key in ("a1", "a2", "b1", "b2", "inter", "k"):
                assert got[key] == want[key], (trial, key)
            assert float(got["raw"]) == want["raw"]
    assert nonempty >= 15


def test_record_text_format_matches_java_semantics():
    rnd = random.Random(1)
    rec = {"from_id": 12, "to_id": 7, "score": 0.78, "raw": 57.0, "a1": 3, "a2": 4000, "alen": 5000, "b1": 10, "b2": 3900,
           "blen": 4800, "to_rc": 1}
    assert mhap_amd.format_record(rec) == "12 7 0.220000 57.000000 0 3 4000 5000 1 10 3900 4800"
    rec["score"] = 1.0 - 5e-7      # error column 5e-7 (approx): compare with the oracle's formatter
    for _ in range(300):
        rec["score"] = rnd.choice([rnd.random(), 1.0, 1.2, 0.0, 1 - 1e-6 * rnd.randint(0, 9) - 5e-7])
        rec["raw"] = float(rnd.randint(0, 3000))
        assert mhap_amd.format_record(rec) == O.format_record(rec)
    assert mhap_amd.format_record({**rec, "score": 1.5}).split()[2] == "0.000000"      # clamp (MatchResult.java:61-64)
    # the formatter's fast path (multiply + floor away from HALF_UP ties) against the oracle's digit-string formatter: random values,
    # values on and next to ties at the sixth decimal, negative ids / coordinates, large counts
    import struct
    for i in range(20000):
        kind = i % 5
        if kind == 0:
            sc = rnd.random()
        elif kind == 1:
            sc = 1.0 - (rnd.randint(0, 999999) + 0.5) * 1e-6                      # error column on a tie
        elif kind == 2:
            t = 1.0 - (rnd.randint(0, 999999) + 0.5) * 1e-6
            sc = struct.unpack("<d", struct.pack("<q", struct.unpack("<q", struct.pack("<d", t))[0] + rnd.randint(-3, 3)))[0]   # a few ulps off a tie
        elif kind == 3:
            sc = rnd.choice([0.0, 1.0, 0.5, 0.999999, 0.9999995, 1e-7, 1 - 1e-7, 0.78])
        else:
            sc = round(rnd.random(), rnd.randint(1, 9))
        r2 = {"from_id": rnd.choice([1, -5, 2**40 + 3, rnd.randint(0, 10**7)]), "to_id": rnd.randint(0, 10**9), "score": sc,
              "raw": float(rnd.choice([0, 1, 57, 1536, rnd.randint(0, 10**6)])), "a1": rnd.randint(0, 70000), "a2": rnd.randint(0, 70000),
              "alen": rnd.randint(0, 2**31 - 1), "b1": rnd.randint(-3, 70000), "b2": rnd.randint(0, 70000), "blen": rnd.randint(0, 70000),
              "to_rc": rnd.randint(0, 1)}
        assert mhap_amd.format_record(r2) == O.format_record(r2), r2


def test_fasta_reader_follows_fastadata(tmp_path):
    p = tmp_path / "x.fasta"
    p.write_bytes(b">r1 desc\nacgt\nACGN\n>empty\n>r2\r\nTTTT\r\n\r\nGG\n>r3\nA")
    fa = mhap_amd.FastaData.from_file(str(p))
    assert len(fa) == 3
    assert [fa.sequence(i) for i in range(3)] == ["ACGTACGN", "TTTTGG", "A"]
    assert fa.ids.tolist() == [1, 2, 3]            # empty record does not consume an id (FastaData.java:180-181)
    fa = mhap_amd.FastaData.from_file(str(p), id_offset=10)
    assert fa.ids.tolist() == [11, 12, 13]
    bad = tmp_path / "bad.fasta"
    bad.write_bytes(b"ACGT\n>r\nAC\n")
    with pytest.raises(mhap_amd.MhapError):
        mhap_amd.FastaData.from_file(str(bad))
    # Utils.getFile: *gz / *bz2 are decompressed, anything else needs a FASTA suffix (FastaData.java:50)
    import bz2
    import gzip
    raw = p.read_bytes()
    gz, bz, txt = tmp_path / "x.fasta.gz", tmp_path / "y.fa.bz2", tmp_path / "z.txt"
    gz.write_bytes(gzip.compress(raw)); bz.write_bytes(bz2.compress(raw)); txt.write_bytes(raw)
    for q in (gz, bz):
        f2 = mhap_amd.FastaData.from_file(str(q))
        assert [f2.sequence(i) for i in range(3)] == ["ACGTACGN", "TTTTGG", "A"]
    with pytest.raises(mhap_amd.MhapError):
        mhap_amd.FastaData.from_file(str(txt))
    g = mhap_amd.FastaData.from_file(os.path.join(ROOT, "tests", "golden", "small_reads.fasta"))
    assert len(g) == 30 and g.sequence(7) == g.sequence(7).upper()


def test_synthetic_reads_are_deterministic_and_overlap():
    a = mhap_amd.synth_reads(50, 2000, seed=123)
    b = mhap_amd.synth_reads(50, 2000, seed=123)
    c = mhap_amd.synth_reads(50, 2000, seed=124)
    assert np.array_equal(a.bases, b.bases) and not np.array_equal(a.bases, c.bases)
    assert set(np.unique(a.bases).tolist()) <= set(b"ACGT")
    assert a.lengths.tolist() == [2000] * 50 and a.ids.tolist() == list(range(1, 51))
    # 30x coverage of a 3.3 kb genome at 5 % error: the oracle must find many overlaps between these reads
    d = mhap_amd.synth_reads(50, 2000, seed=123, error_rate=0.05)
    res = O.run_self(d, H=128, S=512, nthreads=4)
    assert len(res["records"]) > 50


def _join_overlap(A, lenA, B, lenB, max_shift=0.2):
    """numpy restatement of what overlap_join_kernel computes for a pair whose joined hashes are unique in both sketches:
    everything getOverlapInfo derives from the two sorted sketches is a function of their equal-hash join."""
    hA, pA, hB, pB = A[:, 0], A[:, 1], B[:, 0], B[:, 1]
    common, ia, ib = np.intersect1d(hA, hB, return_indices=True)        # sorted by (signed) hash = merge order
    if any((hA == h).sum() > 1 or (hB == h).sum() > 1 for h in common):
        return None                                                      # duplicated joined hash: the kernel replays the merge
    p1, p2 = pA[ia].astype(np.int64), pB[ib].astype(np.int64)

    def stats(shifts):
        if len(shifts) == 0:
            return 0, max(lenA, lenB) + 1
        med = int(np.sort(shifts)[len(shifts) // 2])                      # Utils.quickSelect(k = n / 2)
        ov = max(min(lenA, lenB - med) - max(0, -med), 10)
        return med, min(max(lenA, lenB), int(ov * max_shift))

    def one_pass(med, absmax):
        v1lo, v1hi = max(0, -med - absmax), min(lenA, lenB - med + absmax)
        v2lo, v2hi = max(0, med - absmax), min(lenB, lenA + med + absmax)
        d = (p2 - p1) - med
        return (p1 >= v1lo) & (p1 < v1hi) & (p2 >= v2lo) & (p2 < v2hi) & (d <= absmax) & (d >= -absmax)

    empty = dict(empty=1)
    keep = one_pass(*stats(np.zeros(0)))
    if not keep.any():
        return empty
    keep = one_pass(*stats((p2 - p1)[keep]))
    if not keep.any():
        return empty
    med, absmax = stats((p2 - p1)[keep])
    ok = keep & (np.abs((p2 - p1) - med) <= absmax)
    valid = int(ok.sum())
    if valid < 3:
        return empty
    le1, re1, le2, re2 = p1[ok].min(), p1[ok].max(), p2[ok].min(), p2[ok].max()
    den = float(valid - 1)
    a1 = max(0, O.java_round((valid * le1 - re1) / den)); a2 = min(lenA, O.java_round((valid * re1 - le1) / den))
    b1 = max(0, O.java_round((valid * le2 - re2) / den)); b2 = min(lenB, O.java_round((valid * re2 - le2) / den))
    inA, inB = (pA >= a1) & (pA <= a2), (pB >= b1) & (pB <= b2)
    kk = int(min(inA.sum(), inB.sum()))
    rA, rB = np.cumsum(inA) - inA, np.cumsum(inB) - inB                  # in-window entries ahead of each entry
    both = inA[ia] & inB[ib]
    ahead = np.cumsum(both) - both
    inter = int((both & (rA[ia] + rB[ib] - ahead < kk)).sum())
    return dict(empty=0, raw=valid, a1=int(a1), a2=int(a2), b1=int(b1), b2=int(b2), inter=inter, k=kk)


def test_join_formulation_of_the_second_stage_matches_the_literal_merge():
    """The wave-per-pair kernel's formulation (equal-hash join + filters + rank arithmetic), restated in numpy, against the
    oracle's literal getOverlapInfo on overlapping synthetic reads."""
    fa = mhap_amd.synth_reads(60, 4000, seed=5, error_rate=0.08)
    sk = []
    for i in range(len(fa)):
        s = fa.sequence(i)
        for strand in (s, O.rc(s)):
            rc_, od, _ = O.ordered(strand, 12, 400)
            sk.append((od, len(strand)))
    checked = nonempty = 0
    for i in range(0, len(sk), 2):
        for j in range(len(sk)):
            if j // 2 == i // 2:
                continue
            got = _join_overlap(sk[i][0], sk[i][1], sk[j][0], sk[j][1])
            if got is None:
                continue
            want = O.overlap(sk[i][0], sk[i][1], sk[j][0], sk[j][1])
            checked += 1
            if want["empty"]:
                assert got["empty"] == 1
                continue
            nonempty += 1
            assert got == {k: (int(want[k]) if k != "empty" else 0) for k in got}, (i, j, got, want)
    assert checked > 5000 and nonempty > 100


def test_fasta_reader_line_conventions(tmp_path):
    """mhap_fasta_read against a line-by-line restatement of FastaData.enqueueNextSequenceInFile: \n, \r and \r\n line
    ends, multi-line records, lower case, '>' inside a sequence line, empty records (skipped, no id), no final newline."""
    rnd = random.Random(8)

    def reference(text):
        seqs, cur = [], None
        i, lines = 0, []
        while i < len(text):                      # BufferedReader.readLine
            j = i
            while j < len(text) and text[j] not in "\n\r":
                j += 1
            lines.append(text[i:j])
            i = j + 2 if text[j:j + 2] == "\r\n" else j + 1
        for ln in lines:
            if ln.startswith(">"):
                if cur is not None and cur:
                    seqs.append(cur)
                cur = ""
            else:
                cur += ln.upper()
        if cur:
            seqs.append(cur)
        return seqs

    for trial in range(30):
        eol = rnd.choice(["\n", "\r\n", "\r"])
        parts = []
        for r in range(rnd.randrange(1, 12)):
            parts.append(">read%d some text" % r + eol)
            for _ in range(rnd.randrange(0, 4)):
                ln = "".join(rnd.choice("ACGTacgtNn") for _ in range(rnd.randrange(0, 70)))
                if ln and rnd.random() < 0.1:
                    ln = ln[:len(ln) // 2] + ">" + ln[len(ln) // 2:]
                parts.append(ln + eol)
            if rnd.random() < 0.2:
                parts.append(eol)
        text = "".join(parts)
        if rnd.random() < 0.5:
            text = text.rstrip("\r\n")
        p = tmp_path / ("t%d.fa" % trial)
        p.write_bytes(text.encode())
        f = mhap_amd.FastaData.from_file(str(p))
        want = reference(text)
        assert [f.sequence(i) for i in range(len(f))] == want, (trial, text)
        assert f.ids.tolist() == list(range(1, len(want) + 1))
    bad = tmp_path / "bad.fa"
    bad.write_bytes(b"ACGT\n>r\nACGT\n")
    with pytest.raises(mhap_amd.MhapError):
        mhap_amd.FastaData.from_file(str(bad))


def test_jni_shim_and_java_class_agree():
    """jni/HipMinHashSearch.java (extends AbstractMatchSearch) and jni/mhap_jni.c are shipped uncompiled (no JDK here): every
    native method must have its Java_... function with the same number of arguments, every mhap_* function the shim calls must be
    declared in include/mhap_hip.h and exported by the library, the Java class must override the reference's driver methods, and
    the C file must at least compile (syntax + types against the real mhap_hip.h and a stand-in jni.h)."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    java = open(os.path.join(root, "jni", "HipMinHashSearch.java")).read()
    csrc = open(os.path.join(root, "jni", "mhap_jni.c")).read()
    header = open(os.path.join(root, "include", "mhap_hip.h")).read()
    natives = {m.group(2): len([a for a in m.group(3).split(",") if a.strip()])
               for m in re.finditer(r"private static native\s+([\w\[\]]+)\s+(\w+)\(([^)]*)\)\s*;", java, flags=re.S)}
    assert len(natives) >= 8
    cfuncs = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()]) - 2      # minus JNIEnv*, jclass
              for m in re.finditer(r"Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_(\w+)\(\s*([^)]*)\)\s*\{", csrc, flags=re.S)}
    assert natives == cfuncs, (natives, cfuncs)
    called = set(re.findall(r"\b(mhap_[a-z_0-9]+)\s*\(", csrc)) - {"mhap_record", "mhap_handle", "mhap_params", "mhap_stats"}
    declared = set(re.findall(r"\b(mhap_[a-z_0-9]+)\s*\(", header))
    assert called and called <= declared, called - declared
    assert called <= set(api.EXPORTED_SYMBOLS)
    # the engine behind the Java object is a group of ranks (one per GPU, a group of one for a single GPU)
    for needed in ("mhap_group_create", "mhap_group_add_reads", "mhap_group_find_matches_self", "mhap_group_find_matches_reads",
                   "mhap_find_matches_sketches", "mhap_set_filter_file", "mhap_group_get_stats", "mhap_group_destroy", "mhap_group_rank"):
        assert needed in called, needed
    # records cross in bounded chunks, and no critical region is held across a library call
    assert "nativeTakeRecords" in natives and "GetPrimitiveArrayCritical(" not in csrc.split("*/", 1)[1]
    assert "RECORDS_PER_TAKE" in java and "int[] devices" in java
    # the reference's seams (AbstractMatchSearch.java:119,121,201,203,312,314,340) are all overridden
    assert "extends AbstractMatchSearch" in java
    body = java.split("public final class HipMinHashSearch", 1)[1]
    assert " ... " not in body and "/* ... */" not in body          # complete code, no elisions
    for sig in ("protected boolean addSequence(SequenceSketch", "public ArrayList<MatchResult> findMatches()",
                "public ArrayList<MatchResult> findMatches(final SequenceSketchStreamer", "protected List<MatchResult> findMatches(SequenceSketch",
                "public List<SequenceId> getStoredForwardSequenceIds()", "public SequenceSketch getStoredSequenceHash(SequenceId", "public int size()"):
        assert sig in java, sig
    assert "fromNative" not in java
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(root, "tests", "jni_stub"), "-I", os.path.join(root, "include"),
                        os.path.join(root, "jni", "mhap_jni.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_bit_sliced_step_order_matches_its_generator():
    """bs_step() in sketch_kernels.hip is generated.  The shipped form (92 operations): tools/gen_bs_step92.py drops the intermediate
    planes whose readers have a free xor input, orders the rest in place (five temporaries, no copy), checks the statements against the
    64-bit xorshift step on random values and prints them.  The 107-operation form kept behind -DMH_BS_STEP=107: tools/gen_bs_step.py
    (topological sort, one saved plane).  The kernel source must hold exactly those statements."""
    pytest.importorskip("networkx")
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "mhap_amd", "csrc", "sketch_kernels.hip")).read()
    bodies = src[src.index("#if MH_BS_STEP == 92"):]
    new_body, old_body = bodies[:bodies.index("#else")], bodies[bodies.index("#else"):bodies.index("#endif")]

    def statements(text):
        return [ln.strip() for ln in text.splitlines() if ln.strip().startswith(("const uint32_t t", "const uint32_t T", "P["))]
    gen = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_bs_step92.py")], capture_output=True, text=True, check=True).stdout
    stmts = statements(gen)
    assert len(stmts) == 92 and sum("bs_xor3" in x for x in stmts) == 61 and sum(x.startswith("const uint32_t t") for x in stmts) == 5
    assert statements(new_body) == stmts
    gen = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_bs_step.py")], capture_output=True, text=True, check=True).stdout
    stmts = [ln.strip() for ln in gen.splitlines() if ln.strip()]
    assert len(stmts) == 65 and stmts[0] == "const uint32_t T = P[35];"
    assert statements(old_body) == stmts


def test_bench_preflight_without_a_gpu_reports_instead_of_hanging():
    """bench.py --preflight --gpus 2 on a box without GPUs: exit code 1 and a JSON report that names what is missing."""
    import json, subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preflight", "--gpus", "2"], capture_output=True, text=True, timeout=300)
    rep = json.loads(r.stdout.strip().split("\n")[-1])
    import torch
    if not torch.cuda.is_available():
        assert r.returncode == 1 and not rep["ready"] and not rep["checks"]["devices"]["ok"]
    assert set(rep["checks"]) >= {"devices", "torch_rccl_backend", "library_rccl_entry_points", "peer_access", "ipc_mode"}


def test_early_reject_table_of_the_second_stage_is_exact():
    """The join kernel ends a pair early when its joined k-mers inside both windows (an upper bound of the intersection) are fewer than
    pass_min[lower bound of its k]: pass_min[k] must be the smallest intersection that reaches --threshold for ANY k' >= k, read off the
    identity table the kernel scores with — and that table must be the oracle's jaccardToIdentity bit for bit (BottomOverlapSketch.java:391-395)."""
    lib = api.load_library()
    S, k2 = 200, 12
    f = O.lib().orc_jaccard_to_identity
    f.restype = C.c_double
    for thr in (0.0, 0.3, 0.78, 0.95, 1.0):
        scores = np.zeros((S + 1) * (S + 2) // 2, dtype=np.float64)
        pm = np.zeros(S + 2, dtype=np.int32)
        assert lib.mhap_selftest_pass_min(C.c_int32(S), C.c_int32(k2), C.c_double(thr), scores.ctypes.data_as(C.c_void_p), pm.ctypes.data_as(C.c_void_p)) == 0
        first = np.full(S + 1, np.iinfo(np.int32).max, dtype=np.int64)         # per k: the smallest inter with score >= thr (brute force)
        for k in range(S + 1):
            row = scores[k * (k + 1) // 2:k * (k + 1) // 2 + k + 1]
            if k in (0, 1, 7, 64, S):
                want = [f(C.c_double(0.0 if k == 0 else i / k), C.c_int(k2)) for i in range(k + 1)]
                assert np.array_equal(row.view(np.uint64), np.array(want, dtype=np.float64).view(np.uint64)), k
            ok = np.nonzero(row >= thr)[0]
            if len(ok):
                first[k] = ok[0]
        suffix = np.minimum.accumulate(first[::-1])[::-1]                      # min over k' >= k
        assert np.array_equal(pm[:S + 1].astype(np.int64), suffix), thr
        # the property the kernel relies on: an accepted (inter, k) is never below pass_min of any lower bound of k
        for k in range(S + 1):
            row = scores[k * (k + 1) // 2:k * (k + 1) // 2 + k + 1]
            acc = np.nonzero(row >= thr)[0]
            if len(acc):
                assert acc.min() >= pm[:k + 1].max() or acc.min() >= pm[k], (thr, k)
                assert all(acc.min() >= pm[kl] for kl in range(0, k + 1, max(1, k // 7)))
    assert pm[S + 1] == np.iinfo(np.int32).max


def test_second_stage_lane_on_error_free_overlaps_takes_the_bounded_selection():
    """Two error-free reads that overlap share every k-mer of the overlap at ONE shift: hundreds to 1 536 records with equal shifts, the input
    on which Utils.quickSelect (J/utils/Utils.java:445-494) narrows its range by one element per pass.  The per-lane second stage runs the
    literal loop on a step budget and then selects the same order statistic bit by bit (overlap_lane.hpp): results equal the oracle's."""
    rnd = random.Random(5)
    g = "".join(rnd.choice("ACGT") for _ in range(26000))
    for off, la, lb in ((0, 10000, 10000), (1500, 10000, 12000), (9000, 10000, 10000), (4000, 15000, 3000)):
        a, b = g[:la], g[off:off + lb]
        _, A, lenA = O.ordered(a, 12, 1536)
        _, B, lenB = O.ordered(b, 12, 1536)
        want = O.overlap(A, lenA, B, lenB, max_shift=0.2)
        got = _lane(A, lenA, B, lenB, max_shift=0.2, stride=64)
        assert not want["empty"] and got["empty"] == 0
        for key in ("a1", "a2", "b1", "b2", "inter", "k"):
            assert got[key] == want[key], (off, key)
        assert float(got["raw"]) == want["raw"] and want["raw"] > 100
