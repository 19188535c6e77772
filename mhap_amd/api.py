"""Host-side mirror of MHAP's operator interface over the libmhaphip C ABI (include/mhap_hip.h).

Names follow the reference (J/ = src/main/java/edu/umd/marbl/mhap/):
  MinHashSearch   <- J/impl/MinHashSearch.java + J/impl/AbstractMatchSearch.java (addData/findMatches)
  FastaData       <- J/impl/FastaData.java
  FrequencyCounts <- J/sketch/FrequencyCounts.java (file parsing; the table is applied on the GPU)
  MatchResult     <- J/impl/MatchResult.java (record + text format)
Error behaviour: every failure raises MhapError (the MhapRuntimeException analogue).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("MHAP_LIB_PATH") or os.path.join(_HERE, "lib", "libmhaphip.so")   # MHAP_LIB_PATH: A/B builds of the kernels
_lib = None

KERNEL_NAMES = ["hash_kmers", "kmer_weight", "minhash", "ordered", "candidate", "overlap", "index_build", "index_query"]


class MhapError(RuntimeError):
    """MhapRuntimeException analogue (J/impl/MhapRuntimeException.java:32)."""


class _Params(C.Structure):
    _fields_ = [("kmer_size", C.c_int32), ("num_hashes", C.c_int32), ("ordered_kmer_size", C.c_int32),
                ("ordered_sketch_size", C.c_int32), ("num_min_matches", C.c_int32), ("min_store_length", C.c_int32),
                ("min_olap_length", C.c_int32), ("device", C.c_int32), ("threshold", C.c_double),
                ("max_shift", C.c_double), ("repeat_weight", C.c_double)]


class _Stats(C.Structure):
    _fields_ = [("strands_indexed", C.c_int64), ("queries_searched", C.c_int64), ("candidates_compared", C.c_int64),
                ("matches_found", C.c_int64), ("slot_compares", C.c_int64), ("table_elements", C.c_int64),
                ("slow_pairs", C.c_int64), ("index_splits", C.c_int64)]


class _KTimes(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_int64 * 8)]


class _Fasta(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("offsets", C.c_void_p), ("lengths", C.c_void_p), ("ids", C.c_void_p),
                ("n", C.c_int64), ("total_bases", C.c_int64), ("headers", C.c_void_p), ("headers_bytes", C.c_int64)]


RECORD_DTYPE = np.dtype([("from_id", "<i8"), ("to_id", "<i8"), ("score", "<f8"), ("raw", "<f8"), ("a1", "<i4"),
                         ("a2", "<i4"), ("alen", "<i4"), ("b1", "<i4"), ("b2", "<i4"), ("blen", "<i4"),
                         ("to_rc", "<i4"), ("pad", "<i4")])
_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p)
_GATE = C.CFUNCTYPE(C.c_int, C.c_void_p)

# every symbol include/mhap_hip.h declares
EXPORTED_SYMBOLS = [
    "mhap_create", "mhap_destroy", "mhap_last_error", "mhap_default_params", "mhap_set_filter", "mhap_index_add_reads",
    "mhap_sketch_batch", "mhap_index_add_sketches", "mhap_index_size", "mhap_index_export", "mhap_index_clear", "mhap_index_prepare",
    "mhap_sketch_reads_device", "mhap_index_set_device", "mhap_find_matches_self", "mhap_find_matches_reads",
    "mhap_get_stats", "mhap_get_kernel_times", "mhap_reset_kernel_times", "mhap_set_stream", "mhap_synchronize",
    "mhap_format_record", "mhap_fasta_read", "mhap_fasta_free", "mhap_synth_reads", "mhap_hash_kmer",
    "mhap_selftest_hash_windows", "mhap_selftest_overlap_lane", "mhap_stage_reads", "mhap_index_add_staged",
    "mhap_sketch_staged_device", "mhap_find_matches_self_shard", "mhap_synth_reads_shard", "mhap_selftest_transpose32", "mhap_selftest_pass_min", "mhap_selftest_xorshift_jump", "mhap_selftest_xorshift_unjump", "mhap_find_matches_sketches",
    "mhap_synth_reads_repeats", "mhap_synth_reads_genome", "mhap_find_matches_device", "mhap_set_filter_whitelist", "mhap_set_filter_file", "mhap_selftest_bloom", "mhap_set_second_stage_gate",
    "mhap_dist_unique_id", "mhap_dist_init", "mhap_dist_finalize", "mhap_dist_find_matches_self", "mhap_dist_find_matches_reads", "mhap_dist_last_timing", "mhap_dist_exchange_timing", "mhap_dist_info", "mhap_dist_selftest", "mhap_dist_set_eager", "mhap_dist_eager_searches",
    "mhap_group_create", "mhap_group_destroy", "mhap_group_size", "mhap_group_rank", "mhap_group_last_error", "mhap_group_add_reads", "mhap_group_clear",
    "mhap_group_find_matches_self", "mhap_group_find_matches_reads", "mhap_group_get_stats", "mhap_abi_version", "mhap_abi_sizes",
    "mhap_index_reserve", "mhap_fasta_scan_open", "mhap_fasta_scan_free", "mhap_fasta_scan_reads", "mhap_fasta_scan_bases", "mhap_fasta_scan_info",
    "mhap_index_add_scan", "mhap_find_matches_scan", "mhap_group_add_scan",
]
ABI_VERSION = 3   # MHAP_ABI_VERSION of include/mhap_hip.h this binding was written against


def load_library(build_if_missing=True):
    """Load libmhaphip.so (building it in-tree with hipcc if absent). Fails loudly: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        if not build_if_missing:
            raise MhapError(f"{_LIB_PATH} is missing: run `python -m mhap_amd.build`")
        from . import build as _b
        _b.build()
    try:
        # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Importing torch first makes the
        # process use ONE HIP/HSA runtime; loading ours first and torch later leaves the second runtime without devices.
        import torch  # noqa: F401
    except Exception:
        pass
    try:
        lib = C.CDLL(_LIB_PATH)
    except OSError as e:
        raise MhapError(f"cannot load the HIP extension {_LIB_PATH}: {e}") from e
    lib.mhap_last_error.restype = C.c_char_p
    lib.mhap_last_error.argtypes = [C.c_void_p]
    lib.mhap_destroy.restype = None
    lib.mhap_destroy.argtypes = [C.c_void_p]
    lib.mhap_fasta_free.restype = None
    lib.mhap_default_params.restype = None
    lib.mhap_group_destroy.restype = None
    lib.mhap_group_destroy.argtypes = [C.c_void_p]
    lib.mhap_group_rank.restype = C.c_void_p
    lib.mhap_group_rank.argtypes = [C.c_void_p, C.c_int32]
    lib.mhap_group_last_error.restype = C.c_char_p
    lib.mhap_group_last_error.argtypes = [C.c_void_p]
    lib.mhap_group_size.argtypes = [C.c_void_p]
    lib.mhap_fasta_scan_free.restype = None
    lib.mhap_fasta_scan_free.argtypes = [C.c_void_p]
    lib.mhap_fasta_scan_reads.restype = C.c_int64
    lib.mhap_fasta_scan_reads.argtypes = [C.c_void_p]
    lib.mhap_fasta_scan_bases.restype = C.c_int64
    lib.mhap_fasta_scan_bases.argtypes = [C.c_void_p]
    for name in EXPORTED_SYMBOLS:
        getattr(lib, name)  # AttributeError here = header/library mismatch
    # a stale library next to newer host code (or the reverse) must not get as far as a struct copy
    sizes = (C.c_int32 * 4)()
    lib.mhap_abi_sizes(sizes)
    want = [C.sizeof(_Params), RECORD_DTYPE.itemsize, C.sizeof(_Stats), C.sizeof(_KTimes)]
    if lib.mhap_abi_version() != ABI_VERSION or list(sizes) != want:
        raise MhapError(f"{_LIB_PATH} was built from another include/mhap_hip.h (ABI {lib.mhap_abi_version()}, struct sizes {list(sizes)}; "
                        f"this binding: ABI {ABI_VERSION}, {want}): rebuild it with `python -m mhap_amd.build`")
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class MhapParams:
    """Flag table of J/main/MhapMain.java:67-125 (defaults identical)."""

    def __init__(self, kmer_size=16, num_hashes=512, ordered_kmer_size=12, ordered_sketch_size=1536, num_min_matches=3,
                 min_store_length=0, min_olap_length=116, threshold=0.78, max_shift=0.2, repeat_weight=0.9, device=-1):
        self.kmer_size = kmer_size
        self.num_hashes = num_hashes
        self.ordered_kmer_size = ordered_kmer_size
        self.ordered_sketch_size = ordered_sketch_size
        self.num_min_matches = num_min_matches
        self.min_store_length = min_store_length
        self.min_olap_length = min_olap_length
        self.threshold = threshold
        self.max_shift = max_shift
        self.repeat_weight = repeat_weight
        self.device = device

    def _c(self):
        return _Params(self.kmer_size, self.num_hashes, self.ordered_kmer_size, self.ordered_sketch_size,
                       self.num_min_matches, self.min_store_length, self.min_olap_length, self.device,
                       self.threshold, self.max_shift, self.repeat_weight)


class FastaData:
    """Reads (upper-cased, concatenated) + 1-based ids, as J/impl/FastaData.java:125-204 produces them."""

    def __init__(self, bases, offsets, lengths, ids):
        self.bases = np.ascontiguousarray(bases, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        self.ids = np.ascontiguousarray(ids, dtype=np.int64)

    def __len__(self):
        return int(self.lengths.shape[0])

    @classmethod
    def from_file(cls, path, id_offset=0):
        lib = load_library()
        f = _Fasta()
        err = C.create_string_buffer(512)
        rc = lib.mhap_fasta_read(path.encode(), C.c_int64(id_offset), C.byref(f), err, C.c_size_t(512))
        if rc != 0:
            raise MhapError(err.value.decode() or f"mhap_fasta_read failed ({rc})")
        try:
            n, tb = f.n, f.total_bases
            bases = np.ctypeslib.as_array(C.cast(f.bases, C.POINTER(C.c_uint8)), shape=(max(tb, 1),))[:tb].copy()
            offsets = np.ctypeslib.as_array(C.cast(f.offsets, C.POINTER(C.c_int64)), shape=(max(n, 1),))[:n].copy()
            lengths = np.ctypeslib.as_array(C.cast(f.lengths, C.POINTER(C.c_int32)), shape=(max(n, 1),))[:n].copy()
            ids = np.ctypeslib.as_array(C.cast(f.ids, C.POINTER(C.c_int64)), shape=(max(n, 1),))[:n].copy()
        finally:
            lib.mhap_fasta_free(C.byref(f))
        return cls(bases, offsets, lengths, ids)

    @classmethod
    def from_strings(cls, seqs, id_offset=0):
        """Ids count only non-empty records, 1-based (FastaData.java:180-181)."""
        seqs = [s.upper() for s in seqs if len(s) > 0]
        lengths = np.array([len(s) for s in seqs], dtype=np.int32)
        offsets = np.zeros(len(seqs), dtype=np.int64)
        if len(seqs):
            offsets[1:] = np.cumsum(lengths[:-1], dtype=np.int64)
        bases = np.frombuffer("".join(seqs).encode("latin-1"), dtype=np.uint8).copy() if seqs else np.zeros(0, np.uint8)
        ids = np.arange(1, len(seqs) + 1, dtype=np.int64) + id_offset
        return cls(bases, offsets, lengths, ids)

    def sequence(self, i):
        o, n = int(self.offsets[i]), int(self.lengths[i])
        return self.bases[o:o + n].tobytes().decode("latin-1")

    def subset(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        lengths = self.lengths[idx]
        offsets = np.zeros(len(idx), dtype=np.int64)
        if len(idx):
            offsets[1:] = np.cumsum(lengths[:-1], dtype=np.int64)
        bases = np.empty(int(lengths.sum()), dtype=np.uint8)
        for j, i in enumerate(idx):
            bases[offsets[j]:offsets[j] + lengths[j]] = self.bases[self.offsets[i]:self.offsets[i] + self.lengths[i]]
        return FastaData(bases, offsets, lengths, self.ids[idx])


def synth_reads(n, length, seed=0x4D484150, coverage=30.0, error_rate=0.15, shard=0, nshards=1, repeats=None):
    """Deterministic synthetic PacBio-style reads as a FastaData.

    With nshards > 1 only reads shard, shard+nshards, ... of the same n-read data set are generated (ids kept).
    repeats = (element length, spacing, divergence) plants a repeat family in the genome (BASELINE configs[4])."""
    lib = load_library()
    idx = np.arange(shard, n, nshards, dtype=np.int64)
    m = len(idx)
    bases = np.empty(max(m * length, 1), dtype=np.uint8)
    rl, rs, rd = repeats if repeats else (0, 0, 0.0)
    rc = lib.mhap_synth_reads_repeats(C.c_uint64(seed), C.c_int64(n), C.c_int32(length), C.c_double(coverage),
                                      C.c_double(error_rate), C.c_int64(shard), C.c_int64(nshards), C.c_int32(rl), C.c_int32(rs),
                                      C.c_double(rd), _ptr(bases))
    if rc != 0:
        raise MhapError(f"mhap_synth_reads failed ({rc})")
    offsets = np.arange(m, dtype=np.int64) * length
    lengths = np.full(m, length, dtype=np.int32)
    return FastaData(bases[:m * length], offsets, lengths, idx + 1)


def synth_reads_from_genome(genome, lengths, seed=0x4D484150, error_rate=0.15):
    """Reads of the given lengths drawn from a supplied circular genome (uint8 codes 0..3): mhap_synth_reads_genome."""
    lib = load_library()
    genome = np.ascontiguousarray(genome, dtype=np.uint8)
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    n = len(lengths)
    offsets = np.zeros(n, dtype=np.int64)
    if n > 1:
        np.cumsum(lengths[:-1].astype(np.int64), out=offsets[1:])
    bases = np.empty(max(int(lengths.astype(np.int64).sum()), 1), dtype=np.uint8)
    rc = lib.mhap_synth_reads_genome(C.c_uint64(seed), _ptr(genome), C.c_int64(len(genome)), C.c_int64(n), _ptr(lengths), _ptr(offsets),
                                     C.c_double(error_rate), _ptr(bases))
    if rc != 0:
        raise MhapError(f"mhap_synth_reads_genome failed ({rc})")
    return FastaData(bases, offsets, lengths, np.arange(1, n + 1, dtype=np.int64))


class FrequencyCounts:
    """Parsed `-f` filter file (J/sketch/FrequencyCounts.java:63-229): k-mer hash -> fraction (+ the --supress-noise whitelist)."""

    def __init__(self, hashes, fractions, filter_cutoff=1.0e-5, offset=0.0, repeat_idf_scale=3.0, no_tf=False, supress_noise=0,
                 whitelist=None, size_bloom=None):
        self.hashes = np.ascontiguousarray(hashes, dtype=np.int64)
        self.fractions = np.ascontiguousarray(fractions, dtype=np.float64)
        self.filter_cutoff = filter_cutoff
        self.offset = offset
        self.range = repeat_idf_scale
        self.no_tf = no_tf
        self.supress_noise = supress_noise            # removeUnique: 1 drop k-mers absent from the file, 2 give them idf 1
        self.whitelist = np.ascontiguousarray(whitelist if whitelist is not None else self.hashes, dtype=np.int64)
        # first number of the file's first line; only an object built WITHOUT a file falls back to the whitelist length
        self.size_bloom = max(1, len(self.whitelist)) if size_bloom is None else (1 if size_bloom == 0 else int(size_bloom))

    @classmethod
    def from_file(cls, path, filter_cutoff=1.0e-5, repeat_weight=0.9, repeat_idf_scale=3.0, no_tf=False, do_rc=True,
                  supress_noise=0):
        """The parser of mhap_set_filter_file (host_util.cpp), line for line: header "sizeBloom sizeRepeat" (both >= 0; a sizeBloom
        of 0 counts as 1, FrequencyCounts.java:102-121), then `kmer [fraction [ignored]]`; a malformed fraction drops the whole line."""
        lib = load_library()
        offset = repeat_weight if 0.0 <= repeat_weight < 1.0 else 0.0   # MhapMain.java:346-350
        hs, fr, allh = [], [], []
        out = C.c_int64()
        with open(path, "r") as fh:
            first = fh.readline().split()
            try:
                size_bloom, size_repeat = int(first[0]), int(first[1])
            except (IndexError, ValueError):
                raise MhapError("K-mer filter file first line must contain estimated number of k-mers in the file (long).")
            if size_bloom < 0 or size_repeat < 0:
                raise MhapError("K-mer filter file first line must contain estimated number of k-mers in the file (long).")
            for line in fh:
                parts = line.split(None, 2)
                if len(parts) < 1:
                    continue
                kmer = parts[0].encode("latin-1")
                if lib.mhap_hash_kmer(kmer, C.c_int32(len(kmer)), C.c_int32(1 if do_rc else 0), C.byref(out)) != 0:
                    continue
                frac = None
                if len(parts) >= 2:
                    try:
                        frac = float(parts[1])
                    except ValueError:
                        continue
                allh.append(out.value)
                if frac is not None:
                    hs.append(out.value)
                    fr.append(frac)
        return cls(np.array(hs, dtype=np.int64), np.array(fr, dtype=np.float64), filter_cutoff, offset,
                   repeat_idf_scale, no_tf, supress_noise, np.array(allh, dtype=np.int64), size_bloom)


class FastaScan:
    """A FASTA file mapped and scanned, not copied (mhap_fasta_scan_*): ids, lengths and names of its records; the index is fed from
    the mapped text in groups, host threads packing one group while the GPU sketches the previous one (MinHashSearch.add_scan)."""

    def __init__(self, path, id_offset=0):
        self._lib = load_library()
        self._s = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self._lib.mhap_fasta_scan_open(path.encode(), C.c_int64(id_offset), C.byref(self._s), err, C.c_size_t(512))
        if rc != 0:
            self._s = C.c_void_p()
            raise MhapError(err.value.decode() or f"mhap_fasta_scan_open failed ({rc})")
        self._lib.mhap_fasta_scan_reads.restype = C.c_int64
        self._lib.mhap_fasta_scan_bases.restype = C.c_int64
        self.n = self._lib.mhap_fasta_scan_reads(self._s)
        self.total_bases = self._lib.mhap_fasta_scan_bases(self._s)

    def __len__(self):
        return int(self.n)

    def info(self):
        """(ids, lengths, names)"""
        ids = np.zeros(max(self.n, 1), np.int64)
        lens = np.zeros(max(self.n, 1), np.int32)
        hp, hb = C.c_char_p(), C.c_int64()
        rc = self._lib.mhap_fasta_scan_info(self._s, _ptr(ids), _ptr(lens), C.byref(hp), C.byref(hb))
        if rc != 0:
            raise MhapError(f"mhap_fasta_scan_info failed ({rc})")
        raw = C.string_at(hp, hb.value) if hb.value else b""
        names = [x.decode("latin-1") for x in raw.split(b"\0")[:self.n]]
        return ids[:self.n], lens[:self.n], names

    def close(self):
        if getattr(self, "_s", None) and self._s.value:
            self._lib.mhap_fasta_scan_free(self._s)
            self._s = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MatchResult:
    """One overlap record (J/impl/MatchResult.java)."""
    __slots__ = ("from_id", "to_id", "score", "raw", "a1", "a2", "alen", "b1", "b2", "blen", "to_rc")

    def __init__(self, rec):
        for k in self.__slots__:
            setattr(self, k, rec[k].item() if hasattr(rec[k], "item") else rec[k])

    def __str__(self):
        return format_record(self)


def format_record(rec):
    """MatchResult.toString (J/impl/MatchResult.java:98-113) through the library's Java-compatible formatter."""
    lib = load_library()
    arr = np.zeros(1, dtype=RECORD_DTYPE)
    for k in RECORD_DTYPE.names:
        if k != "pad":
            arr[0][k] = rec[k] if not isinstance(rec, MatchResult) else getattr(rec, k)
    buf = C.create_string_buffer(256)
    n = lib.mhap_format_record(_ptr(arr), buf, C.c_size_t(256))
    if n < 0:
        raise MhapError("mhap_format_record failed")
    return buf.value.decode()


def records_to_lines(records):
    lib = load_library()
    records = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
    buf = C.create_string_buffer(256)
    out = []
    base = records.ctypes.data
    for i in range(records.shape[0]):
        lib.mhap_format_record(C.c_void_p(base + i * RECORD_DTYPE.itemsize), buf, C.c_size_t(256))
        out.append(buf.value.decode())
    return out


def _collect_records(call, chk):
    """Run a search entry point with a sink that appends every batch of records to ONE growing array (each record is copied once out
    of the library's buffer; the array doubles when it fills up — a list of per-batch copies concatenated at the end moved every
    record twice, 2 x 1.8 GB a step on one rank's share of BASELINE configs[4]).  The sink is called from a library thread, one batch
    at a time."""
    # (untouched pages of np.empty cost nothing: room for 2^17 records up front saves the first doublings of an ordinary search)
    state = {"buf": np.empty(1 << 17, dtype=RECORD_DTYPE), "n": 0}
    isz = RECORD_DTYPE.itemsize

    def sink(recs, cnt, user):
        n, buf = state["n"], state["buf"]
        need = n + cnt
        if need > buf.shape[0]:
            # grow in place: realloc of a large block is a remap of its pages, not a copy (a fresh array + copy moved gigabytes
            # per doubling on one rank's share of BASELINE configs[4] and made the library's sink thread wait)
            buf.resize(max(need, 2 * buf.shape[0]), refcheck=False)
        # one memmove out of the library's buffer (wrapping the pointer in a NumPy array first cost 1.4 ms per call: C2's 41 915 records)
        C.memmove(buf.ctypes.data + n * isz, recs, cnt * isz)
        state["n"] = need
        return 0

    cb = _SINK(sink)
    chk(call(cb))
    return state["buf"][:state["n"]]


class MinHashSearch:
    """GPU counterpart of J/impl/MinHashSearch.java (+ the drivers of AbstractMatchSearch.java).

    add_data(fasta)            ~ new MinHashSearch(streamer, ...) / addData   (sketch fwd+rc, index)
    find_matches()             ~ AbstractMatchSearch.findMatches()            (self, toSelf=true)
    find_matches_stream(fasta) ~ AbstractMatchSearch.findMatches(streamer)    (-q, toSelf=false)
    """

    def __init__(self, params=None, kmer_filter=None):
        self._lib = load_library()
        self.params = params or MhapParams()
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        p = self.params._c()
        rc = self._lib.mhap_create(C.byref(p), C.byref(self._h), err, C.c_size_t(512))
        if rc != 0:
            self._h = C.c_void_p()
            raise MhapError(err.value.decode() or f"mhap_create failed ({rc})")
        if kmer_filter is not None:
            self.set_filter(kmer_filter)

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.mhap_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise MhapError(f"{self._lib.mhap_last_error(self._h).decode()} (code {rc})")

    # -- configuration --------------------------------------------------------------------------
    def set_filter(self, fc):
        hashes, fractions = fc.hashes, fc.fractions
        if len(hashes) == 0:      # a filter whose table is empty is still a filter (every k-mer gets idf = range): non-NULL pointers say so
            hashes, fractions = np.zeros(1, np.int64), np.zeros(1, np.float64)
        self._chk(self._lib.mhap_set_filter(self._h, _ptr(hashes), _ptr(fractions), C.c_int64(len(fc.hashes)),
                                            C.c_double(fc.filter_cutoff), C.c_double(fc.offset), C.c_double(fc.range),
                                            C.c_int(1 if fc.no_tf else 0)))
        if getattr(fc, "supress_noise", 0):
            self._chk(self._lib.mhap_set_filter_whitelist(self._h, _ptr(fc.whitelist), C.c_int64(len(fc.whitelist)),
                                                          C.c_int64(fc.size_bloom), C.c_int32(fc.supress_noise)))

    def set_stream(self, hip_stream_ptr):
        self._chk(self._lib.mhap_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    # -- index ----------------------------------------------------------------------------------
    def add_data(self, fasta):
        self._chk(self._lib.mhap_index_add_reads(self._h, _ptr(fasta.bases), _ptr(fasta.offsets), _ptr(fasta.lengths),
                                                 _ptr(fasta.ids), C.c_int64(len(fasta))))

    def add_scan(self, scan):
        """Streamed ingest of a scanned FASTA file (mhap_index_add_scan): parse, pack, upload and sketch overlap."""
        self._chk(self._lib.mhap_index_add_scan(self._h, scan._s))

    def reserve(self, total_reads):
        """The reads an empty index is about to receive over several add_data calls (mhap_index_reserve)."""
        self._chk(self._lib.mhap_index_reserve(self._h, C.c_int64(total_reads)))

    def find_matches_scan(self, scan):
        """-q mode with the query reads of a scanned file."""
        return self._collect(lambda cb: self._lib.mhap_find_matches_scan(self._h, scan._s, cb, None))

    def stage(self, fasta):
        """Pack + upload reads so that they are resident in HBM (bench: outside the timed region)."""
        self._chk(self._lib.mhap_stage_reads(self._h, _ptr(fasta.bases), _ptr(fasta.offsets), _ptr(fasta.lengths),
                                             _ptr(fasta.ids), C.c_int64(len(fasta))))

    def add_staged(self):
        self._chk(self._lib.mhap_index_add_staged(self._h))

    def sketch_staged_device(self, d_minhash_ptr, d_ordered_ptr, d_meta_ptr):
        self._chk(self._lib.mhap_sketch_staged_device(self._h, C.c_void_p(d_minhash_ptr), C.c_void_p(d_ordered_ptr),
                                                      C.c_void_p(d_meta_ptr)))

    def size(self):
        n = C.c_int64()
        self._chk(self._lib.mhap_index_size(self._h, C.byref(n)))
        return n.value

    def clear(self):
        self._chk(self._lib.mhap_index_clear(self._h))

    def sketch(self, fasta):
        """SequenceSketchStreamer.getSketch for a batch: returns dict of host arrays (both strands)."""
        n = len(fasta)
        H, S = max(1, self.params.num_hashes), self.params.ordered_sketch_size
        mh = np.zeros((2 * n, H), dtype=np.int32)
        od = np.zeros((2 * n, S, 2), dtype=np.int32)
        osz = np.zeros(2 * n, dtype=np.int32)
        st = np.zeros(2 * n, dtype=np.uint8)
        self._chk(self._lib.mhap_sketch_batch(self._h, _ptr(fasta.bases), _ptr(fasta.offsets), _ptr(fasta.lengths),
                                              C.c_int64(n), _ptr(mh), _ptr(od), _ptr(osz), _ptr(st)))
        return {"minhash": mh, "ordered": od, "ordered_size": osz, "status": st}

    def export(self, first=0, count=None):
        total = self.size()
        count = total - first if count is None else count
        H, S = max(1, self.params.num_hashes), self.params.ordered_sketch_size
        out = {"ids": np.zeros(count, np.int64), "is_fwd": np.zeros(count, np.uint8), "seq_length": np.zeros(count, np.int32),
               "minhash": np.zeros((count, H), np.int32), "ordered": np.zeros((count, S, 2), np.int32),
               "ordered_size": np.zeros(count, np.int32), "ordered_seqlen": np.zeros(count, np.int32),
               "status": np.zeros(count, np.uint8)}
        self._chk(self._lib.mhap_index_export(self._h, C.c_int64(first), C.c_int64(count), _ptr(out["ids"]), _ptr(out["is_fwd"]),
                                              _ptr(out["seq_length"]), _ptr(out["minhash"]), _ptr(out["ordered"]),
                                              _ptr(out["ordered_size"]), _ptr(out["ordered_seqlen"]), _ptr(out["status"])))
        return out

    def add_sketches(self, sk):
        m = len(sk["ids"])
        a = {k: np.ascontiguousarray(v) for k, v in sk.items()}
        self._chk(self._lib.mhap_index_add_sketches(self._h, _ptr(a["ids"].astype(np.int64)), _ptr(a["is_fwd"].astype(np.uint8)),
                                                    _ptr(a["seq_length"].astype(np.int32)), _ptr(a["minhash"].astype(np.int32)),
                                                    _ptr(a["ordered"].astype(np.int32)), _ptr(a["ordered_size"].astype(np.int32)),
                                                    _ptr(a["ordered_seqlen"].astype(np.int32)), C.c_int64(m)))

    # -- multi-GPU plumbing (device pointers owned by the caller, e.g. torch tensors) -----------
    def sketch_reads_device(self, fasta, d_minhash_ptr, d_ordered_ptr, d_meta_ptr):
        self._chk(self._lib.mhap_sketch_reads_device(self._h, _ptr(fasta.bases), _ptr(fasta.offsets), _ptr(fasta.lengths),
                                                     C.c_int64(len(fasta)), C.c_void_p(d_minhash_ptr), C.c_void_p(d_ordered_ptr),
                                                     C.c_void_p(d_meta_ptr)))

    def set_device_index(self, ids, is_fwd, d_minhash_ptr, d_ordered_ptr, d_meta_ptr):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        is_fwd = np.ascontiguousarray(is_fwd, dtype=np.uint8)
        self._chk(self._lib.mhap_index_set_device(self._h, _ptr(ids), _ptr(is_fwd), C.c_void_p(d_minhash_ptr),
                                                  C.c_void_p(d_ordered_ptr), C.c_void_p(d_meta_ptr), C.c_int64(len(ids))))

    # -- search ---------------------------------------------------------------------------------
    def _collect(self, call):
        return _collect_records(call, self._chk)

    def prepare_index(self):
        """Build the inverted index now (reads the MinHash/meta tables only; see mhap_index_prepare)."""
        self._chk(self._lib.mhap_index_prepare(self._h))

    def find_matches(self, q_first=0, q_count=-1):
        """Self overlap of forward entries [q_first, q_first+q_count) against the whole index."""
        return self._collect(lambda cb: self._lib.mhap_find_matches_self(self._h, C.c_int64(q_first), C.c_int64(q_count), cb, None))

    def find_matches_shard(self, shard, nshards):
        """This rank's share of the self overlap (reads with ordinal % nshards == shard are the queries)."""
        return self._collect(lambda cb: self._lib.mhap_find_matches_self_shard(self._h, C.c_int64(shard), C.c_int64(nshards), cb, None))

    def find_matches_stream(self, fasta):
        return self._collect(lambda cb: self._lib.mhap_find_matches_reads(self._h, _ptr(fasta.bases), _ptr(fasta.offsets),
                                                                          _ptr(fasta.lengths), _ptr(fasta.ids),
                                                                          C.c_int64(len(fasta)), cb, None))

    def find_matches_sketches(self, sk):
        """-q x.dat: precomputed (forward) query sketches against the index, toSelf=false."""
        a = {k: np.ascontiguousarray(v) for k, v in sk.items()}
        ids = a["ids"].astype(np.int64); sl = a["seq_length"].astype(np.int32); mh = a["minhash"].astype(np.int32)
        od = a["ordered"].astype(np.int32); osz = a["ordered_size"].astype(np.int32); osl = a["ordered_seqlen"].astype(np.int32)
        return self._collect(lambda cb: self._lib.mhap_find_matches_sketches(self._h, _ptr(ids), _ptr(sl), _ptr(mh), _ptr(od), _ptr(osz),
                                                                            _ptr(osl), C.c_int64(len(ids)), cb, None))

    def find_matches_device(self, d_q_minhash_ptr, d_q_ordered_ptr, d_q_meta_ptr, ids, to_self=True, before_second_stage=None, count_only=False):
        """Device-resident query sketches (forward rows of the ranks' tables) against this handle's index.
        before_second_stage: callable run once the candidates are known and before the ordered rows are read (e.g. the wait
        for their asynchronous all-gather).  count_only: no sink — the library reads the records back and converts them as always,
        nobody keeps them; returns their number (10^8 records and more per search: what a driver would stream to a file)."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        gate = None
        if before_second_stage is not None:
            done = [False]

            def _gate(user):
                if not done[0]:
                    before_second_stage()
                    done[0] = True
                return 0
            gate = _GATE(_gate)
            self._chk(self._lib.mhap_set_second_stage_gate(self._h, gate, None))
        try:
            def call(cb):
                return self._lib.mhap_find_matches_device(self._h, C.c_void_p(d_q_minhash_ptr), C.c_void_p(d_q_ordered_ptr), C.c_void_p(d_q_meta_ptr),
                                                          _ptr(ids), C.c_int64(len(ids)), C.c_int(1 if to_self else 0), cb, None)
            if count_only:
                before = self.stats()["matches_found"]
                self._chk(call(_SINK(0)))
                return self.stats()["matches_found"] - before
            return self._collect(call)
        finally:
            if gate is not None:
                self._chk(self._lib.mhap_set_second_stage_gate(self._h, _GATE(0), None))
                if not done[0]:
                    before_second_stage()       # no candidates at all: the caller still expects the wait to have happened

    # -- one process per GPU: this handle as rank `rank` of `nranks` (the exchange runs inside the library, RCCL over xGMI) ----
    @staticmethod
    def dist_unique_id():
        """ncclGetUniqueId: 128 bytes that rank 0 hands to the other ranks (any channel) before dist_init."""
        buf = C.create_string_buffer(128)
        rc = load_library().mhap_dist_unique_id(buf, C.c_size_t(128))
        if rc != 0:
            raise MhapError(f"mhap_dist_unique_id failed ({rc}): RCCL is not loadable")
        return buf.raw

    def dist_init(self, rank, nranks, unique_id):
        self._chk(self._lib.mhap_dist_init(self._h, C.c_int32(rank), C.c_int32(nranks), C.c_char_p(bytes(unique_id))))

    def dist_set_eager(self, on=True):
        """Eager exchange (mhap_dist_set_eager): the add that fills this rank's empty index becomes collective and gathers the rank's
        forward rows while it computes; the sharded search then starts with all rows in place."""
        self._chk(self._lib.mhap_dist_set_eager(self._h, C.c_int32(1 if on else 0)))

    def dist_eager_searches(self):
        self._lib.mhap_dist_eager_searches.restype = C.c_int64
        return int(self._lib.mhap_dist_eager_searches(self._h))

    def dist_finalize(self):
        self._chk(self._lib.mhap_dist_finalize(self._h))

    def dist_find_matches(self):
        """Collective: self overlap of the union of the ranks' indexes; returns this rank's records."""
        return self._collect(lambda cb: self._lib.mhap_dist_find_matches_self(self._h, cb, None))

    def dist_find_matches_stream(self, fasta):
        """Collective, -q mode: `fasta` = the query reads dealt to this rank."""
        return self._collect(lambda cb: self._lib.mhap_dist_find_matches_reads(self._h, _ptr(fasta.bases), _ptr(fasta.offsets), _ptr(fasta.lengths),
                                                                              _ptr(fasta.ids), C.c_int64(len(fasta)), cb, None))

    def dist_last_timing(self):
        t = (C.c_double * 3)()
        self._chk(self._lib.mhap_dist_last_timing(self._h, t))
        return {"gather_small_ms": t[0], "wait_ordered_ms": t[1], "total_ms": t[2]}

    def dist_info(self):
        """The transport's own view of this rank (mhap_dist_info): what a launcher checks to see that N ranks on N devices formed one communicator."""
        out = (C.c_int32 * 5)()
        pci = C.create_string_buffer(64)
        self._chk(self._lib.mhap_dist_info(self._h, out, pci, C.c_size_t(64)))
        ver = int(out[4])
        return {"comm_count": int(out[0]), "comm_user_rank": int(out[1]), "comm_device": int(out[2]), "handle_device": int(out[3]),
                "rccl_version": (f"{ver // 10000}.{ver // 100 % 100}.{ver % 100}" if ver else None), "pci_bus_id": pci.value.decode() or None}

    def dist_exchange_timing(self):
        """The eager exchange of the last add as the exchange stream saw it: mhap_dist_exchange_timing."""
        out = (C.c_double * 4)()
        self._chk(self._lib.mhap_dist_exchange_timing(self._h, out))
        return {"ordered_gather_ms": out[0], "ordered_bytes_received": out[1], "small_gather_ms": out[2], "small_bytes_received": out[3]}

    def dist_selftest(self, nbytes=1 << 20):
        """Collective: all-gather `nbytes` of a known pattern per rank through the handle's transport and check every block; returns the gather's ms."""
        ms = C.c_double(0.0)
        self._chk(self._lib.mhap_dist_selftest(self._h, C.c_size_t(nbytes), C.byref(ms)))
        return float(ms.value)

    # -- counters -------------------------------------------------------------------------------
    def stats(self):
        s = _Stats()
        self._chk(self._lib.mhap_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _Stats._fields_}

    def kernel_times(self):
        t = _KTimes()
        self._chk(self._lib.mhap_get_kernel_times(self._h, C.byref(t)))
        return {KERNEL_NAMES[i]: {"ms": t.ms[i], "launches": t.launches[i]} for i in range(len(KERNEL_NAMES))}

    def reset_kernel_times(self):
        self._chk(self._lib.mhap_reset_kernel_times(self._h))

    def synchronize(self):
        self._chk(self._lib.mhap_synchronize(self._h))


class MinHashSearchGroup:
    """One process, N GPUs: N ranks of one sharded index (mhap_group_*: AbstractMatchSearch's drivers over several devices).
    Reads are dealt round-robin; every rank sketches and indexes its share; a search gathers the forward query rows of all
    ranks (peer-to-peer copies over xGMI, or RCCL with MHAP_GROUP_TRANSPORT=rccl) and each rank searches them against its shard."""

    def __init__(self, params=None, n=1, devices=None, kmer_filter=None):
        self._lib = load_library()
        self.params = params or MhapParams()
        self._g = C.c_void_p()
        err = C.create_string_buffer(512)
        p = self.params._c()
        dev = (C.c_int32 * n)(*devices) if devices is not None else None
        rc = self._lib.mhap_group_create(C.byref(p), dev, C.c_int32(n), C.byref(self._g), err, C.c_size_t(512))
        if rc != 0:
            self._g = C.c_void_p()
            raise MhapError(err.value.decode() or f"mhap_group_create failed ({rc})")
        self.n = n
        if kmer_filter is not None:
            for r in range(n):
                self.rank(r).set_filter(kmer_filter)

    def rank(self, r):
        """A non-owning MinHashSearch view of rank r's handle (filters, counters)."""
        v = MinHashSearch.__new__(MinHashSearch)
        v._lib, v.params = self._lib, self.params
        h = self._lib.mhap_group_rank(self._g, C.c_int32(r))
        if not h:
            raise MhapError("rank out of range")
        v._h = C.c_void_p(h)
        v.close = lambda: None
        return v

    def close(self):
        if getattr(self, "_g", None) and self._g.value:
            self._lib.mhap_group_destroy(self._g)
            self._g = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise MhapError(f"{self._lib.mhap_group_last_error(self._g).decode()} (code {rc})")

    def add_data(self, fasta):
        self._chk(self._lib.mhap_group_add_reads(self._g, _ptr(fasta.bases), _ptr(fasta.offsets), _ptr(fasta.lengths), _ptr(fasta.ids),
                                                 C.c_int64(len(fasta))))

    def add_scan(self, scan):
        self._chk(self._lib.mhap_group_add_scan(self._g, scan._s))

    def clear(self):
        self._chk(self._lib.mhap_group_clear(self._g))

    def _collect(self, call):
        return _collect_records(call, self._chk)

    def find_matches(self):
        return self._collect(lambda cb: self._lib.mhap_group_find_matches_self(self._g, cb, None))

    def find_matches_stream(self, fasta):
        return self._collect(lambda cb: self._lib.mhap_group_find_matches_reads(self._g, _ptr(fasta.bases), _ptr(fasta.offsets), _ptr(fasta.lengths),
                                                                               _ptr(fasta.ids), C.c_int64(len(fasta)), cb, None))

    def stats(self):
        s = _Stats()
        self._chk(self._lib.mhap_group_get_stats(self._g, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _Stats._fields_}
