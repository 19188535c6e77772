/*
 * mhap_hip.h — C ABI of libmhaphip.so, the MI355X (gfx950) MinHash overlap engine.
 *
 * This is the drop-in boundary for MHAP's hot path.  The reference (marbl/MHAP, Java)
 * has no FFI; its only operator seam is the abstract class
 *   J/impl/AbstractMatchSearch.java:47   (J/ = src/main/java/edu/umd/marbl/mhap/)
 * with MinHashSearch as the sole implementation.  Per-read JNI calls would serialise
 * the GPU, so every entry point below is batch-granular.  Each entry point cites the
 * reference interface it replaces.  INTEGRATION.md shows the JNI stub + the Java
 * subclass (HipMinHashSearch extends AbstractMatchSearch) a maintainer would add.
 *
 * Conventions: C linkage, plain pointers and sizes, no exceptions cross the boundary.
 * Every call returns 0 on success or a negative MHAP_E_* code; the message is
 * available from mhap_last_error().  A handle is NOT re-entrant: the caller
 * serialises calls on one handle (the library overlaps work on its own HIP stream).
 * There is no CPU fallback: without a usable HIP device mhap_create() fails.
 */
#ifndef MHAP_HIP_H
#define MHAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHAP_OK 0
#define MHAP_E_INVALID (-1)     /* bad argument / unsupported option (MhapRuntimeException analogue) */
#define MHAP_E_HIP (-2)         /* HIP runtime error, no device, kernel failure */
#define MHAP_E_NOMEM (-3)
#define MHAP_E_STATE (-4)       /* call sequence error (e.g. search before index) */
#define MHAP_E_IO (-5)          /* a file could not be opened / read */

/* Per-strand sketch status (mirrors the streamer's skip rules). */
#define MHAP_STRAND_OK 0
#define MHAP_STRAND_ZERO_NGRAMS 1 /* ZeroNGramsFoundException, J/impl/SequenceSketchStreamer.java:235-238 */
#define MHAP_STRAND_TOO_SHORT 2   /* L < --min-olap-length,    J/impl/SequenceSketchStreamer.java:129-133 */

/* Flag table of J/main/MhapMain.java:67-125 that reaches the hot path. */
typedef struct mhap_params {
  int32_t kmer_size;           /* -k                      (16)   MhapMain.java:75,107  */
  int32_t num_hashes;          /* --num-hashes            (512)  :87,108               */
  int32_t ordered_kmer_size;   /* --ordered-kmer-size     (12)   :89,116               */
  int32_t ordered_sketch_size; /* --ordered-sketch-size   (1536) :91,117               */
  int32_t num_min_matches;     /* --num-min-matches       (3)    :83,112               */
  int32_t min_store_length;    /* --min-store-length      (0)    :79,118               */
  int32_t min_olap_length;     /* --min-olap-length       (116)  :81,119               */
  int32_t device;              /* HIP device ordinal; -1 = current device              */
  double threshold;            /* --threshold             (0.78) :67,109               */
  double max_shift;            /* --max-shift             (0.2)  :77,111               */
  double repeat_weight;        /* --repeat-weight         (0.9)  :69,114               */
} mhap_params;

/* One overlap = the fields of J/impl/MatchResult.java:46-65 (before text formatting). */
typedef struct mhap_record {
  int64_t from_id, to_id;  /* SequenceId.getHeaderId()  */
  double score;            /* OverlapInfo.score (identity); the text column is 1-min(score,1) */
  double raw;              /* OverlapInfo.rawScore = number of valid shared k-mers */
  int32_t a1, a2, alen;    /* query interval + full length (fromLength)       */
  int32_t b1, b2, blen;    /* match interval (already flipped if to_rc) + toLength */
  int32_t to_rc;           /* 0 fwd / 1 reverse-complement entry              */
  int32_t pad;
} mhap_record;

/* Counters behind MhapMain.outputFinalStat (J/main/MhapMain.java:572-590). */
typedef struct mhap_stats {
  int64_t strands_indexed;        /* MinHashSearch.size()                        */
  int64_t queries_searched;       /* getNumberSequencesSearched()                */
  int64_t candidates_compared;    /* getNumberSequencesFullyCompared()           */
  int64_t matches_found;          /* getMatchesProcessed()                       */
  int64_t slot_compares;          /* slot comparisons done by the brute-force candidate kernel (MHAP_CANDIDATES=bruteforce) */
  int64_t table_elements;         /* getNumberElementsProcessed(): inverted-index hits walked             */
  int64_t slow_pairs;             /* candidates the wave-per-pair second stage handed to the per-lane merge */
  int64_t index_splits;           /* query hit sets too large for the LDS count table that were split into hash-partition passes */
} mhap_stats;

/* Per-kernel HIP-event timings accumulated on the handle's stream (for bench/roofline). */
#define MHAP_K_HASH 0      /* hash_kmers_kernel: murmur3 of every k-mer / k2-mer into HBM — only for reads the kernels cannot hash from
                              their 2-bit codes (raw bytes, k != 16 or k2 != 12, reads beyond 24 591 bases); 0 on the default path */
#define MHAP_K_WEIGHT 1    /* kmer_weight_kernel: per-strand k-mer multiplicity / tf-idf weight, weight classes, MinHash work lists */
#define MHAP_K_DEDUP MHAP_K_WEIGHT   /* (name of rounds 1-2) */
#define MHAP_K_MINHASH 2   /* minhash_kernel: weighted xorshift MinHash (hashes its k-mers from the 2-bit codes)            */
#define MHAP_K_ORDERED 3   /* ordered_kernel: 12-mer hashes + bottom-S select + sort                                     */
#define MHAP_K_CANDIDATE 4 /* candidate_kernel: brute-force all-pairs slot-equality count (MHAP_CANDIDATES=bruteforce only) */
#define MHAP_K_OVERLAP 5   /* overlap_join_kernel (+ overlap_kernel for the pairs it hands over): second-stage getOverlapInfo */
#define MHAP_K_INDEX_BUILD 6 /* index_build + index_finalize: inverted index (MinHashSearch.addSequence) */
#define MHAP_K_INDEX_QUERY 7 /* index_query_kernel: inverted index lookups + per-query hit counting  */
#define MHAP_K_COUNT 8
typedef struct mhap_kernel_times {
  double ms[MHAP_K_COUNT];      /* summed kernel time, milliseconds */
  int64_t launches[MHAP_K_COUNT];
} mhap_kernel_times;

/* Interface version of this header (bumped whenever a struct layout or a signature changes) and the sizes of its structs as the
 * library was compiled: a binding built against another header — a stale libmhaphip.so shipped next to newer host code — finds
 * out at load time instead of overrunning a buffer. */
#define MHAP_ABI_VERSION 3
int mhap_abi_version(void);
int mhap_abi_sizes(int32_t* out4);   /* {sizeof mhap_params, mhap_record, mhap_stats, mhap_kernel_times} */

typedef struct mhap_handle mhap_handle;

/* Record sink: called from the calling thread, one batch at a time; `recs` is owned by
 * the library and valid only during the call.  Replaces AbstractMatchSearch.outputResults
 * (J/impl/AbstractMatchSearch.java:316-338).  Return non-zero to abort the search. */
typedef int (*mhap_record_sink)(const mhap_record* recs, int64_t n, void* user);
/* Optional gate between the two stages of a search: called (from the calling thread) after the candidates of a batch of queries
 * are known and before their ordered sketches are read.  A multi-GPU host uses it to wait for the asynchronous exchange of the
 * ordered-sketch rows, which then overlaps the candidate stage.  Non-zero aborts the search.  NULL removes the gate. */
typedef int (*mhap_stage_gate)(void* user);
int mhap_set_second_stage_gate(mhap_handle* h, mhap_stage_gate gate, void* user);

/* Replaces `new MinHashSearch(...)` argument plumbing (J/impl/MinHashSearch.java:63-98). */
int mhap_create(const mhap_params* params, mhap_handle** out, char* err, size_t errcap);
void mhap_destroy(mhap_handle* h);
const char* mhap_last_error(const mhap_handle* h);
void mhap_default_params(mhap_params* p);

/* Host-built repeat filter = J/sketch/FrequencyCounts.java:63-229 after parsing.
 * hashes[i] = murmur3_x64_128 h1 of the (canonicalised per --no-rc) k-mer, fractions[i] its
 * column-2 value; entries with fraction < filter_cutoff are dropped here (:176-184).
 * offset = repeat_weight if 0<=rw<1 else 0 (MhapMain.java:346-350); range = --repeat-idf-scale.
 * Only --supress-noise 0 is supported.  n == 0 clears the filter. */
int mhap_set_filter(mhap_handle* h, const int64_t* hashes, const double* fractions, int64_t n,
                    double filter_cutoff, double offset, double range, int no_tf);
/* --supress-noise 1|2 (FrequencyCounts removeUnique, J/sketch/FrequencyCounts.java:66-68,137,192,272-278,297): the Bloom filter
 * over EVERY k-mer of the filter file (hashes: all n lines, not only the ones above the cutoff), built exactly like Guava 19.0's
 * BloomFilter.create(funnel(putLong), size_bloom, 1e-5) (strategy MURMUR128_MITZ_64).  mode 1: k-mers that are not in the
 * file are dropped from the MinHash sketch; mode 2: they get idf 1; mode 0 removes the whitelist.  Call after mhap_set_filter. */
int mhap_set_filter_whitelist(mhap_handle* h, const int64_t* hashes, int64_t n, int64_t size_bloom, int32_t mode);
/* new FrequencyCounts(reader, filterCutoff, offset, removeUnique, noTf, numThreads, range, doReverseCompliment)
 * (J/sketch/FrequencyCounts.java:63-229, called from J/main/MhapMain.java:337-361): reads the -f file (first line
 * "sizeBloom sizeRepeat", then `kmer fraction ...` lines), hashes the k-mers (canonical when do_rc) and installs the table
 * (+ the whitelist when remove_unique > 0).  kmer_sizes (may be NULL) receives the distinct k-mer lengths, e.g. "16". */
int mhap_set_filter_file(mhap_handle* h, const char* path, double filter_cutoff, double offset, int32_t remove_unique, int32_t no_tf,
                         double range, int32_t do_rc, char* kmer_sizes, size_t kmer_sizes_cap);

/* Sketch `n` reads (both strands) and append them to the index.  Replaces
 * SequenceSketchStreamer.enqueue/getSketch (J/impl/SequenceSketchStreamer.java:123-177,262-266)
 * + MinHashSearch.addSequence (J/impl/MinHashSearch.java:100-147).
 * bases: concatenated upper-cased sequence bytes (one byte per Java char);
 * offsets[i]/lengths[i] locate read i; ids[i] = SequenceId.getHeaderId (1-based FASTA order).
 * Every read takes two entries (fwd = 2*j, rc = 2*j+1, j = running read count); entries that
 * the reference would skip are kept as non-matchable placeholders (see mhap_index_status). */
int mhap_index_add_reads(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths,
                         const int64_t* ids, int64_t n);

/* Two-step form of mhap_index_add_reads for callers that want the reads resident in HBM before the compute
 * starts (the benchmark's timed region): mhap_stage_reads packs the reads to 2 bits/base (raw bytes for reads
 * with non-ACGT chars) and uploads them once; mhap_index_add_staged then only launches kernels.  Staged reads
 * stay staged until the next mhap_stage_reads / mhap_index_add_reads / mhap_sketch_* call. */
int mhap_stage_reads(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths,
                     const int64_t* ids, int64_t n);
int mhap_index_add_staged(mhap_handle* h);

/* The reads an EMPTY index is about to receive over the coming mhap_index_add_* calls (a file added in batches, as
 * AbstractMatchSearch.addData does, J/impl/AbstractMatchSearch.java:67-117): the sketch tables are sized once for all of
 * them, so that no batch makes them grow (a reallocation + copy of every row sketched so far).  The inverted index is built
 * from all rows by the first search after the last add (a counting sort of the postings: 3 ms per 10^8). */
int mhap_index_reserve(mhap_handle* h, int64_t total_reads);

/* Streamed FASTA ingest (FastaData + SequenceSketchStreamer.enqueueFullFile, J/impl/FastaData.java:101-204,
 * J/impl/SequenceSketchStreamer.java:179-222: the reference reads and sketches through a queue with T threads).
 * mhap_fasta_scan_open maps the file (plain text; gz / bz2 are inflated into memory first) and finds its records on all host
 * threads: ids (1-based count of non-empty records + id_offset), lengths, whether a read is pure ACGT — no copy of the bases.
 * mhap_index_add_scan then feeds the index in groups of <= 256 Mbase: host threads pack group g+1 from the text straight into
 * pinned 2-bit staging while the GPU sketches and indexes group g (two staging buffers), so parsing, packing, the upload and the
 * kernels overlap and the 1-byte-per-base copy of the reads never exists. */
typedef struct mhap_fasta_scan mhap_fasta_scan;
int mhap_fasta_scan_open(const char* path, int64_t id_offset, mhap_fasta_scan** out, char* err, size_t errcap);
void mhap_fasta_scan_free(mhap_fasta_scan* s);
int64_t mhap_fasta_scan_reads(const mhap_fasta_scan* s);          /* non-empty records */
int64_t mhap_fasta_scan_bases(const mhap_fasta_scan* s);
/* ids[n], lengths[n] (either may be NULL); headers: the n NUL-terminated names back to back (mhap_fasta.headers), valid until the scan is freed */
int mhap_fasta_scan_info(mhap_fasta_scan* s, int64_t* ids, int32_t* lengths, const char** headers, int64_t* headers_bytes);
int mhap_index_add_scan(mhap_handle* h, const mhap_fasta_scan* s);
/* the reads of the scan as query reads against the index (-q mode, mhap_find_matches_reads), in groups */
int mhap_find_matches_scan(mhap_handle* h, const mhap_fasta_scan* s, mhap_record_sink sink, void* user);

/* Sketch only (no index change); outputs to caller-allocated HOST arrays, any may be NULL:
 * minhash[2n][max(1,H)], ordered[2n][S][2] (hash,pos), ordered_size[2n], status[2n].
 * Strand order: 2*i = forward, 2*i+1 = reverse complement.  Used by parity tests and the
 * `.dat` writer (J/impl/SequenceSketch.java:123-148). */
int mhap_sketch_batch(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, int64_t n,
                      int32_t* minhash, int32_t* ordered, int32_t* ordered_size, uint8_t* status);

/* Ingest precomputed sketches (from `.dat`, J/impl/SequenceSketch.java:61-96): `m` entries, host arrays.
 * is_fwd[e], seq_length[e] = full base length, ordered_seqlen[e] = L-k2+1 as stored in the file. */
int mhap_index_add_sketches(mhap_handle* h, const int64_t* ids, const uint8_t* is_fwd, const int32_t* seq_length,
                            const int32_t* minhash, const int32_t* ordered, const int32_t* ordered_size,
                            const int32_t* ordered_seqlen, int64_t m);

/* Index introspection: number of entries (incl. placeholders); copy-out of the tables (any NULL). */
int mhap_index_size(mhap_handle* h, int64_t* entries);
int mhap_index_export(mhap_handle* h, int64_t first, int64_t count, int64_t* ids, uint8_t* is_fwd, int32_t* seq_length,
                      int32_t* minhash, int32_t* ordered, int32_t* ordered_size, int32_t* ordered_seqlen,
                      uint8_t* status);
int mhap_index_clear(mhap_handle* h);
/* Build the inverted index (MinHashSearch.addSequence's per-slot maps, J/impl/MinHashSearch.java:123-141) for the current
   entries now instead of at the first search; only the MinHash and meta tables are read, so a caller that adopted device
   tables with mhap_index_set_device may still be filling the ordered-sketch table (e.g. an all-gather in flight). */
int mhap_index_prepare(mhap_handle* h);

/* Multi-GPU plumbing (one process per GPU): the per-rank shard tables live in device memory
 * owned by the CALLER (e.g. torch tensors that RCCL all-gathers over xGMI).
 *  - mhap_sketch_reads_device: sketch n reads into caller device buffers
 *      d_minhash int32[2n][Hrow], d_ordered int32[2n][S][2], d_meta int32[2n][4] =
 *      {ordered_size, ordered_seqlen, seq_length, status}; Hrow = max(1,H).
 *  - mhap_index_set_device: adopt (no copy) gathered tables of `m` entries as the index;
 *      ids/is_fwd are host arrays.  The buffers must outlive the handle's use of them. */
int mhap_sketch_reads_device(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths,
                             int64_t n, void* d_minhash, void* d_ordered, void* d_meta);
int mhap_index_set_device(mhap_handle* h, const int64_t* ids, const uint8_t* is_fwd, void* d_minhash,
                          void* d_ordered, void* d_meta, int64_t m);
/* Same as mhap_sketch_reads_device for reads previously staged with mhap_stage_reads (kernels only). */
int mhap_sketch_staged_device(mhap_handle* h, void* d_minhash, void* d_ordered, void* d_meta);

/* Self-overlap: every forward entry in [q_first, q_first+q_count) is searched against the whole
 * index with toSelf=true.  Replaces AbstractMatchSearch.findMatches()
 * (J/impl/AbstractMatchSearch.java:121-199) + MinHashSearch.findMatches(sketch,true)
 * (J/impl/MinHashSearch.java:150-251).  q_count < 0 means "to the end".  Entry indices, not ids. */
int mhap_find_matches_self(mhap_handle* h, int64_t q_first, int64_t q_count, mhap_record_sink sink, void* user);

/* Sharded self-overlap for one-process-per-GPU runs: this call searches the forward entries whose read
 * ordinal (position among the index's reads) is congruent to `shard` modulo `nshards`; the union over all
 * shards equals mhap_find_matches_self(h, 0, -1).  Round-robin balances the triangular id rule. */
int mhap_find_matches_self_shard(mhap_handle* h, int64_t shard, int64_t nshards, mhap_record_sink sink, void* user);

/* Index-vs-stream (-q mode, toSelf=false): sketch the `n` query reads (forward only,
 * J/impl/AbstractMatchSearch.java:203-285) and search them against the index. */
int mhap_find_matches_reads(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths,
                            const int64_t* ids, int64_t n, mhap_record_sink sink, void* user);

/* Index-vs-stream with PRECOMPUTED query sketches (a `.dat` file given to -q: only its forward entries are
 * queries, J/impl/SequenceSketchStreamer.java:291-303 with fwdOnly=true): `m` query entries as host arrays laid out
 * like mhap_index_add_sketches; toSelf=false. */
int mhap_find_matches_sketches(mhap_handle* h, const int64_t* ids, const int32_t* seq_length, const int32_t* minhash,
                               const int32_t* ordered, const int32_t* ordered_size, const int32_t* ordered_seqlen, int64_t m,
                               mhap_record_sink sink, void* user);
/* Query sketches that already sit in device memory (the layout mhap_sketch_staged_device / mhap_sketch_reads_device write:
 * minhash int32[m][--num-hashes], ordered int32[m][--ordered-sketch-size][2], meta int32[m][4]) against the index — the
 * multi-GPU exchange step: every rank keeps the index of its OWN reads and the forward query sketches of the other ranks
 * visit it one after the other.  ids: host array, one id per query row; rows whose meta status is not 0 are
 * skipped.  to_self != 0 applies findMatches(hashes, toSelf = true)'s id rules (J/impl/MinHashSearch.java:200-225), so every
 * unordered pair of one data set is reported once however its two reads are spread over ranks. */
int mhap_find_matches_device(mhap_handle* h, const void* d_q_minhash, const void* d_q_ordered, const void* d_q_meta, const int64_t* ids,
                             int64_t m, int to_self, mhap_record_sink sink, void* user);

/* ---- several GPUs: one sharded index, the exchange inside the library -------------------------------------
 * The reference has one JVM, one index and a thread pool (AbstractMatchSearch.addData :67-117, findMatches() :121-199,
 * J/impl/AbstractMatchSearch.java).  Over N GPUs the reads are dealt round-robin (read i of the data set -> rank i % N, which
 * balances the id < id rule of J/impl/MinHashSearch.java:215-219), every rank sketches and indexes ITS reads only, and a search
 * all-gathers the forward-strand query rows of all ranks (MinHash rows, meta and ids first; the 6x larger ordered rows behind
 * the candidate stage) and runs every query against the rank's own index shard with the toSelf id rules — each unordered pair is
 * reported exactly once, by the rank that stores its lower-id read.  No collective after the gather; records go to the sink.
 *
 * Two ways to form the ranks, same code underneath:
 *  (a) one process per GPU (bench.py under torchrun, an MPI-style host): every process creates its handle; rank 0 calls
 *      mhap_dist_unique_id and the host hands the 128 bytes to the other ranks (any channel); all call mhap_dist_init, which
 *      creates an RCCL communicator over the handles' devices (ncclCommInitRank; collectives run over xGMI).
 *  (b) one process, N devices (mhap-hip --gpus N, the JNI host of INTEGRATION.md): mhap_group_* below owns N handles and one
 *      host thread per rank; the gather is direct peer-to-peer copies over xGMI (hipMemcpyPeerAsync: each rank pulls the other
 *      ranks' rows), or RCCL (ncclCommInitAll) with MHAP_GROUP_TRANSPORT=rccl. */
#define MHAP_DIST_ID_BYTES 128
int mhap_dist_unique_id(void* id, size_t cap);                       /* ncclGetUniqueId; cap >= MHAP_DIST_ID_BYTES */
int mhap_dist_init(mhap_handle* h, int32_t rank, int32_t nranks, const void* id);   /* collective over the nranks handles */
int mhap_dist_finalize(mhap_handle* h);
/* Collective: self-overlap of the union of the ranks' indexes (every rank calls it; each gets the records of the pairs whose
 * lower-id read it stores).  The index must consist of sketched reads (mhap_index_add_reads / _staged): forward and reverse
 * entries in pairs.  Replaces AbstractMatchSearch.findMatches() for the sharded index. */
int mhap_dist_find_matches_self(mhap_handle* h, mhap_record_sink sink, void* user);
/* Collective, -q mode (AbstractMatchSearch.findMatches(streamer), toSelf = false): this rank sketches the n query reads it was
 * dealt (forward strands only), the query rows of all ranks are gathered, and every rank searches them against its shard. */
int mhap_dist_find_matches_reads(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths,
                                 const int64_t* ids, int64_t n, mhap_record_sink sink, void* user);
/* wall-clock split of the last collective search on this rank, milliseconds: {pack + small gathers, index + candidate stage wait
 * on the ordered rows, whole call} */
/* Eager exchange: with on != 0, the add that fills an EMPTY index of this rank (mhap_index_add_reads / _staged / _scan) becomes a collective
 * call — every rank must make it, in the same order as its searches — and gathers the rank's forward rows while it is still computing:
 * the ordered rows under the MinHash kernel, the MinHash rows under the index build.  mhap_dist_find_matches_self then starts with every
 * rank's rows in place.  An add that cannot take part (not the first one, more than one launch group, any rank saying so) falls back
 * to the exchange at search time on all ranks together.  Replaces nothing in the reference (it has one index in one JVM,
 * J/impl/AbstractMatchSearch.java:67-117); it is the overlap of SURVEY.md §8(e)'s all-gather with the sketch phase. */
int mhap_dist_set_eager(mhap_handle* h, int32_t on);
int64_t mhap_dist_eager_searches(mhap_handle* h);   /* searches of this rank that found every rank's rows already gathered by the add */
int mhap_dist_last_timing(mhap_handle* h, double* out3);
/* The eager exchange of the last add as the exchange stream saw it (the gathers run UNDER the add's kernels): out4 = {ms of the ordered
 * rows' all-gather, bytes this rank received in it, ms of the MinHash + meta + id rows' all-gathers, bytes received}; -1 ms = not run
 * since the last call.  Waits for the gathers.  With mhap_dist_selftest (the same volume with no kernel beside it) this separates fabric
 * time from the interaction with a power-bound compute kernel (bench.py --exchange-only).  No reference counterpart (§8e tooling). */
int mhap_dist_exchange_timing(mhap_handle* h, double* out4);
/* The transport's own view of this rank (RCCL: ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion), for a launcher that
 * wants to confirm that its N processes formed ONE communicator over N devices — the reference has nothing to compare: one JVM, one index
 * (AbstractMatchSearch.java:67-117).  out5 = {ranks, this rank, communicator's device, handle's device, RCCL version code};
 * pci = PCI bus id of the device ("0000:05:00.0"). */
int mhap_dist_info(mhap_handle* h, int32_t* out5, char* pci, size_t pci_cap);
/* The exchange by itself: all-gathers `bytes` (<= 1 GiB) of a known pattern per rank through the handle's transport, under the same
 * watchdog as a search, and checks every rank's block.  Collective over the ranks.  ms_out: wall time of the gather (may be NULL). */
int mhap_dist_selftest(mhap_handle* h, size_t bytes, double* ms_out);

typedef struct mhap_group mhap_group;
/* N handles on the given devices (NULL: devices 0..n-1; repeats allowed — several ranks may share a device, which is how a
 * one-GPU box tests the N > 1 path). */
int mhap_group_create(const mhap_params* params, const int32_t* devices, int32_t n, mhap_group** out, char* err, size_t errcap);
void mhap_group_destroy(mhap_group* g);
int32_t mhap_group_size(const mhap_group* g);
mhap_handle* mhap_group_rank(mhap_group* g, int32_t rank);         /* for per-rank set-up (filters) and counters */
const char* mhap_group_last_error(const mhap_group* g);
/* AbstractMatchSearch.addData over N GPUs: read i of this call goes to rank (reads added so far + i) % N; the ranks sketch and
 * index their shares concurrently.  May be called repeatedly (batches of one file, several files). */
int mhap_group_add_reads(mhap_group* g, const char* bases, const int64_t* offsets, const int32_t* lengths, const int64_t* ids, int64_t n);
int mhap_group_clear(mhap_group* g);
/* mhap_index_add_scan over the ranks: record i of the scan goes to rank (reads added so far + i) % N; every rank runs its own
 * pack / upload / sketch pipeline over its share of the mapped text */
int mhap_group_add_scan(mhap_group* g, const mhap_fasta_scan* s);
/* findMatches() / findMatches(streamer) over the sharded index; the sink is called from the ranks' threads, one call at a time. */
int mhap_group_find_matches_self(mhap_group* g, mhap_record_sink sink, void* user);
int mhap_group_find_matches_reads(mhap_group* g, const char* bases, const int64_t* offsets, const int32_t* lengths, const int64_t* ids,
                                  int64_t n, mhap_record_sink sink, void* user);
int mhap_group_get_stats(mhap_group* g, mhap_stats* sum);           /* counters summed over the ranks */

int mhap_get_stats(mhap_handle* h, mhap_stats* out);
int mhap_get_kernel_times(mhap_handle* h, mhap_kernel_times* out);
int mhap_reset_kernel_times(mhap_handle* h);
/* Use an externally created hipStream_t (e.g. torch's current stream); NULL = library stream. */
int mhap_set_stream(mhap_handle* h, void* hip_stream);
int mhap_synchronize(mhap_handle* h);

/* ---- host-side helpers (no GPU): the reference's IO conventions --------------------------- */

/* Java String.format("%s %s %.6f %.6f %d %d %d %d %d %d %d %d") of MatchResult.toString
 * (J/impl/MatchResult.java:98-113) with numeric headers; returns bytes written (no NUL counted). */
int mhap_format_record(const mhap_record* r, char* out, size_t cap);

/* FASTA ingest with FastaData semantics (J/impl/FastaData.java:125-204): lines concatenated,
 * upper-cased, ids = 1-based running count of non-empty records (+id_offset).  The returned
 * object owns the arrays; free with mhap_fasta_free. */
typedef struct mhap_fasta {
  char* bases; int64_t* offsets; int32_t* lengths; int64_t* ids; int64_t n; int64_t total_bases;
  char* headers;            /* n NUL-terminated names back to back: the header line after '>' up to the first white space or comma
                               (what --store-full-id prints, FastaData.java:155-156) */
  int64_t headers_bytes;
} mhap_fasta;
int mhap_fasta_read(const char* path, int64_t id_offset, mhap_fasta* out, char* err, size_t errcap);
void mhap_fasta_free(mhap_fasta* f);

/* Deterministic synthetic PacBio-style reads: xoshiro256** seeded by
 * splitmix64(seed), random genome of n*len/coverage bp, reads at uniform positions/strands with
 * i.i.d. errors (ins:del:sub = 0.1188:0.0183:0.0129 scaled to error_rate), exactly `len` bases each.
 * bases must hold n*len bytes. */
int mhap_synth_reads(uint64_t seed, int64_t n, int32_t len, double coverage, double error_rate, char* bases);
/* Only reads shard, shard+nshards, ... of that same n-read data set (bases holds ceil((n-shard)/nshards)*len bytes). */
int mhap_synth_reads_shard(uint64_t seed, int64_t n, int32_t len, double coverage, double error_rate, int64_t shard,
                           int64_t nshards, char* bases);
/* The same generator with a planted repeat family in the genome (BASELINE configs[4], the workload of the -f k-mer filter,
 * J/sketch/FrequencyCounts.java): one random element of rep_len bases, one copy with per-base substitution rate rep_div in
 * every rep_spacing-base stretch.  rep_len = 0 gives exactly mhap_synth_reads_shard's reads. */
int mhap_synth_reads_repeats(uint64_t seed, int64_t n, int32_t len, double coverage, double error_rate, int64_t shard,
                             int64_t nshards, int32_t rep_len, int32_t rep_spacing, double rep_div, char* bases);
/* Reads of given lengths from a SUPPLIED circular genome (one code 0..3 per byte), same error model; read r is written to
 * bases[offsets[r] .. offsets[r] + lengths[r]).  Test / bench tooling (the E. coli-shaped stand-in of BASELINE configs[2]). */
int mhap_synth_reads_genome(uint64_t seed, const uint8_t* genome, int64_t G, int64_t n, const int32_t* lengths, const int64_t* offsets,
                            double error_rate, char* bases);

/* murmur3_x64_128(seed 0).h1 of one k-mer line of a `-f` filter file, canonicalised when do_rc != 0
 * (HashUtils.computeSequenceHashesLong(str, len, 0, doRC)[0], J/sketch/FrequencyCounts.java:169). */
int mhap_hash_kmer(const char* kmer, int32_t len, int32_t do_rc, int64_t* out);

/* ---- test hooks: the kernels' __host__ __device__ arithmetic executed on the host (never used by the
 * product path; lets a GPU-less container check the hash and second-stage lane logic) --------------------- */
int mhap_selftest_hash_windows(const char* seq, int32_t len, int32_t k, int32_t k2, int64_t* out64, int32_t* out32);
int mhap_selftest_bloom(const int64_t* hashes, int64_t n, int64_t size_bloom, const int64_t* probes, int64_t np, uint8_t* out_flags,
                        int64_t* out2);
int mhap_selftest_transpose32(uint32_t* a32);
/* the (inter, k) identity table (J/sketch/BottomOverlapSketch.java:391-395; scores[k (k + 1) / 2 + inter], k <= S; may be NULL) and the
 * second stage's early-reject table derived from it: pass_min[k] = smallest inter with score >= threshold for any k' >= k (S + 2 entries) */
int mhap_selftest_pass_min(int32_t S, int32_t k2, double threshold, double* scores, int32_t* pass_min);
int mhap_selftest_xorshift_jump(uint64_t key, int32_t nsteps, uint64_t* out);
int mhap_selftest_xorshift_unjump(uint64_t x, int32_t nsteps, uint64_t* out);   /* the key nsteps steps before chain value x */
/* out8 = {empty, valid(rawScore), a1, a2, b1, b2, inter, k} */
int mhap_selftest_overlap_lane(const int32_t* A, int32_t nA, int32_t lenA, const int32_t* B, int32_t nB, int32_t lenB,
                               double max_shift, int32_t stride, int32_t* out8);

#ifdef __cplusplus
}
#endif
#endif /* MHAP_HIP_H */
