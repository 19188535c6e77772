"""ctypes binding of oracle/liboracle.so — the CPU restatement used ONLY as the parity checker."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

ORC_RECORD = np.dtype([("from_id", "<i8"), ("to_id", "<i8"), ("score", "<f8"), ("raw", "<f8"), ("a1", "<i4"), ("a2", "<i4"),
                       ("alen", "<i4"), ("b1", "<i4"), ("b2", "<i4"), ("blen", "<i4"), ("to_rc", "<i4"), ("pad", "<i4")])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        src = os.path.join(ORACLE_DIR, "mhap_oracle.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
        L = C.CDLL(so)
        L.orc_murmur3_x86_32.restype = C.c_uint32
        L.orc_java_round.restype = C.c_int64
        L.orc_jaccard_to_identity.restype = C.c_double
        L.orc_filter_create.restype = C.c_void_p
        L.orc_filter_create2.restype = C.c_void_p
        L.orc_filter_scaled_idf.restype = C.c_double
        L.orc_run_self.restype = C.c_int64
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def murmur32(b, seed=0):
    return lib().orc_murmur3_x86_32(b, C.c_int64(len(b)), C.c_uint32(seed))


def murmur128(b, seed=0):
    out = (C.c_uint64 * 2)()
    lib().orc_murmur3_x64_128(b, C.c_int64(len(b)), C.c_uint32(seed), out)
    return out[0], out[1]


def kmer_hashes64(seq, k, do_rc=False):
    s = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
    out = np.zeros(max(len(s) - k + 1, 1), dtype=np.int64)
    n = lib().orc_kmer_hashes64(s, len(s), k, 1 if do_rc else 0, _p(out))
    return out[:n]


def kmer_hashes32(seq, k, do_rc=False):
    s = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
    out = np.zeros(max(len(s) - k + 1, 1), dtype=np.int32)
    n = lib().orc_kmer_hashes32(s, len(s), k, 1 if do_rc else 0, _p(out))
    return out[:n]


def rc(seq):
    s = seq.encode("latin-1")
    out = C.create_string_buffer(len(s))
    lib().orc_rc(s, len(s), out)
    return out.raw.decode("latin-1")


class Filter:
    def __init__(self, hashes, fractions, cutoff, offset, rng, no_tf=False, remove_unique=0, whitelist=None, size_bloom=0):
        h = np.ascontiguousarray(hashes, dtype=np.int64)
        f = np.ascontiguousarray(fractions, dtype=np.float64)
        if remove_unique:
            # --supress-noise: the Bloom filter holds every k-mer of the file; k-mers without a fraction carry one below the cutoff
            wl = np.ascontiguousarray(whitelist if whitelist is not None else h, dtype=np.int64)
            extra = np.setdiff1d(wl, h)
            h2 = np.concatenate([h, extra]); f2 = np.concatenate([f, np.full(len(extra), -1.0)])
            self.h = C.c_void_p(lib().orc_filter_create2(_p(h2), _p(f2), C.c_int64(len(h2)), C.c_double(cutoff), C.c_double(offset),
                                                         C.c_double(rng), 1 if no_tf else 0, C.c_int(remove_unique),
                                                         C.c_int64(size_bloom or max(1, len(wl)))))
        else:
            self.h = C.c_void_p(lib().orc_filter_create(_p(h), _p(f), C.c_int64(len(h)), C.c_double(cutoff), C.c_double(offset),
                                                        C.c_double(rng), 1 if no_tf else 0))

    def might_contain(self, key):
        return bool(lib().orc_filter_might_contain(self.h, C.c_int64(key)))

    def scaled_idf(self, key):
        return lib().orc_filter_scaled_idf(self.h, C.c_int64(key))

    def __del__(self):
        try:
            lib().orc_filter_destroy(self.h)
        except Exception:
            pass


def bloom_params(n, p=1.0e-5):
    bits, k = C.c_int64(), C.c_int32()
    lib().orc_bloom_params(C.c_int64(n), C.c_double(p), C.byref(bits), C.byref(k))
    return bits.value, k.value


def minhash(seq, k, H, repeat_weight=0.9, flt=None):
    s = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
    out = np.zeros(max(1, H), dtype=np.int32)
    rc_ = lib().orc_minhash(s, len(s), k, H, flt.h if flt else None, C.c_double(repeat_weight), _p(out))
    return rc_, out


def ordered(seq, k2, S):
    s = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
    out = np.zeros((max(min(S, max(len(s) - k2 + 1, 0)), 1), 2), dtype=np.int32)
    size, seqlen = C.c_int32(), C.c_int32()
    rc_ = lib().orc_ordered(s, len(s), k2, S, _p(out), C.byref(size), C.byref(seqlen))
    return rc_, out[:size.value], seqlen.value


def overlap(A, lenA, B, lenB, k2=12, max_shift=0.2):
    A = np.ascontiguousarray(A, dtype=np.int32)
    B = np.ascontiguousarray(B, dtype=np.int32)
    score, raw, empty = C.c_double(), C.c_double(), C.c_int32()
    out6 = np.zeros(6, dtype=np.int32)
    lib().orc_overlap(_p(A), A.shape[0], lenA, _p(B), B.shape[0], lenB, k2, C.c_double(max_shift), C.byref(score), C.byref(raw),
                      _p(out6), C.byref(empty))
    return {"score": score.value, "raw": raw.value, "a1": int(out6[0]), "a2": int(out6[1]), "b1": int(out6[2]), "b2": int(out6[3]),
            "inter": int(out6[4]), "k": int(out6[5]), "empty": empty.value}


def quickselect(arr, k):
    a = np.ascontiguousarray(arr, dtype=np.int32).copy()
    return lib().orc_quickselect(_p(a), k, len(a))


def java_round(x):
    return lib().orc_java_round(C.c_double(x))


def java_fmt6(v):
    buf = C.create_string_buffer(128)
    lib().orc_java_fmt6(C.c_double(v), buf, 128)
    return buf.value.decode()


def format_record(rec):
    a = np.zeros(1, dtype=ORC_RECORD)
    for k in ORC_RECORD.names:
        if k != "pad":
            a[0][k] = rec[k]
    buf = C.create_string_buffer(256)
    lib().orc_format_record(_p(a), buf, 256)
    return buf.value.decode()


def run_self(fasta, k=16, H=512, k2=12, S=1536, num_min_matches=3, min_store_length=0, min_olap_length=116, threshold=0.78,
             max_shift=0.2, repeat_weight=0.9, flt=None, nthreads=8, cap=1 << 22, want_minhash=False):
    n = len(fasta)
    out = np.zeros(cap, dtype=ORC_RECORD)
    timings = np.zeros(2, dtype=np.float64)
    stats = np.zeros(3, dtype=np.int64)
    mh = np.zeros((2 * n, max(1, H)), dtype=np.int32) if want_minhash else None
    status = np.zeros(2 * n, dtype=np.int32)
    cnt = lib().orc_run_self(_p(fasta.bases), _p(fasta.offsets), _p(fasta.lengths), _p(fasta.ids), C.c_int64(n), k, H, k2, S,
                             num_min_matches, min_store_length, min_olap_length, C.c_double(threshold), C.c_double(max_shift),
                             C.c_double(repeat_weight), flt.h if flt else None, nthreads, _p(out), C.c_int64(cap), _p(timings),
                             _p(stats), _p(mh), _p(status))
    if cnt > cap:
        raise RuntimeError("oracle record buffer too small")
    res = {"records": out[:cnt].copy(), "sketch_s": float(timings[0]), "search_s": float(timings[1]), "strands": int(stats[0]),
           "compared": int(stats[1]), "elements": int(stats[2]), "status": status}
    if want_minhash:
        res["minhash"] = mh
    return res


def record_lines(records):
    return sorted(format_record(r) for r in records)
