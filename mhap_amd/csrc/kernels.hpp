// kernels.hpp — launch interface between the C-ABI orchestration (mhap_capi.hip) and the gfx950 kernels.
#pragma once
#include "device_common.hpp"

namespace mhap {

constexpr int HASH_TILE = 1024;      // window starts per hash workgroup
constexpr int WEIGHT_THREADS = 1024;
constexpr int WEIGHT_MAXIT = 24;     // k-mers per thread on the LDS path (24 x 1024 = 24576 = 0.75 x 32768 table slots)  // threads per k-mer-weight workgroup
constexpr int MH_U = 4;              // k-mers per lane in the MinHash kernel's per-chain rows
constexpr int ORD_THREADS = 256;
constexpr int ORD_BINS = 2048;
constexpr int CAND_TQ = 128;         // queries per candidate tile
constexpr int CAND_TM = 128;         // index entries per candidate tile
constexpr int CAND_KS = 32;          // slots staged per LDS chunk
constexpr int OVL_THREADS = 128;
#ifndef MH_INV_CT
#define MH_INV_CT 2048   // 4096 / 2048 / 1024: C2 4.73 / 4.31 / 4.39 ms, C5 slice 20.3 / 12.8 / 12.2 (a smaller table = more workgroups per CU, less to zero and to scan, and an earlier hand-over to the dense tier)
#endif
constexpr int INV_CT = MH_INV_CT;    // LDS hit-count table entries per query (first tier of the inverted-index path)

// Weight classes of a strand's distinct k-mers (MinHashSketch.java:98-128).  mode > 0: every k-mer position carries the
// weight `mode` (no repeated k-mer, one tf-idf weight) and neither wts[] nor the class list is read.  mode == 0: the class
// list holds the positions of the first occurrences sorted by class — cnt[c] positions of weight c+1 for c < BS_WCLASSES,
// then cnt[BS_WCLASSES] positions of any larger weight (their weights are in wts[]).
constexpr int BS_WCLASSES = 6;
constexpr int BS_WMAX = 32;          // largest uniform weight the bit-sliced rows take (queue entries keep the sub-step in 6 bits)
struct StrandInfo { int32_t valid; int32_t mode; int32_t cnt[BS_WCLASSES + 1]; int32_t pad; };

// Host-built FrequencyCounts table (open addressing; vals[slot]==0.0 marks empty; vals = scaledIdf).
struct FilterTable {
  const int64_t* keys;
  const double* vals;
  uint32_t mask;
  uint32_t size;
  int32_t enabled;
  int32_t no_tf;
  double range;
  // --supress-noise (FrequencyCounts removeUnique 1|2): Bloom filter over every k-mer of the filter file
  const unsigned long long* bloom; uint64_t bloom_bits; int32_t bloom_k; int32_t bloom_mode;
};

// Per-entry metadata row: {ordered_size, ordered_seqlen (L-k2+1), seq_length (bases), status}
constexpr int META_W = 4;

// Candidate pair produced by the all-pairs count and consumed by the second stage.
struct Candidate { int32_t q; int32_t m; };

// Device-side accepted overlap (entry indices; host maps to ids/lengths, MatchResult.java:46-65).
struct DevRecord { int32_t q, m; double score; int32_t raw, a1, a2, b1, b2, pad; };

struct SearchParams {
  int32_t H, S, k2;
  int32_t num_min_matches, min_store_length;
  int32_t to_self;
  int32_t own_queries;   // the query tables ARE the index's (a self search of one index): query entry q is stored entry q — its own H postings,
                         // the one pair the id rule always drops (MinHashSearch.java:200-201), need not be counted (round 6)
  double max_shift, threshold;
};

// ---- sketch_kernels.hip ----
void launch_hash_kmers(hipStream_t st, const ReadDesc* descs, int64_t nstrands, int max_len, const uint8_t* store, int64_t* keys,
                       int32_t* h32, int k, int k2, const uint64_t* luts, int only_mat);
// block-mix tables of the k = 16 / k2 = 12 fast path (768 words: murmur3_x64_128 k1 mix, k2 mix, murmur3_x86_32 pair)
void build_kmer_hash_luts(uint64_t* out);
int weight_grid(int num_cus, int64_t nstrands, int max_len, int k);   // persistent workgroups = HBM slabs needed
// A read's hashes can be recomputed from its 2-bit codes (no MHAP_RD_MAT) when k = 16, k2 = 12, the read is pure ACGT and its
// k-mers fit the weight kernel's LDS path
bool strand_hashes_from_codes(int length, int k, int k2);
void launch_kmer_weights(hipStream_t st, int num_cus, const ReadDesc* descs, int64_t nstrands, int max_len, const int64_t* keys,
                         uint32_t* wts, uint32_t* perm, uint32_t* slabs, int64_t slab_entries, unsigned long long* counter, int k,
                         const FilterTable& ft, double repeat_weight, StrandInfo* info, bool fused, const uint8_t* store,
                         const uint64_t* luts, const int32_t* order, int32_t* slist);   // order: read indices longest first (or null);
                         // slist: 2 x nstrands work lists of the MinHash launches (counts in counter[4], counter[5])
bool launch_minhash(hipStream_t st, hipStream_t st_weighted, int nblocks, int64_t n_unweighted, int64_t n_weighted, const ReadDesc* descs, int64_t nstrands, const int64_t* keys, const uint32_t* wts,
                    const uint32_t* perm, const StrandInfo* info, const uint8_t* store, const uint64_t* luts, int k, int k2, int H,
                    unsigned long long* counter, int32_t* out_rows, int64_t out_stride, int32_t* out_status, int64_t status_stride,
                    const uint64_t* jump, int jump_na, const int32_t* slist, uint32_t* qbuf, const uint64_t* unjump, const uint64_t* jump_w1, unsigned long long* merge,
                    int max_nk);   // counter: base of the sketch phase's counter block; qbuf: minhash_queue_bytes(2 * nblocks) of device memory (the
                    // waves' deferred-candidate queues); unjump: inverse jump tables; merge: minhash_merge_bytes(nblocks, H); max_nk: k-mers of the longest strand
size_t minhash_merge_bytes(int nblocks, int H);
void build_xorshift_unjump_tables(int na, int nq, uint64_t* out);   // M^-(g a), a = 1..na, then M^-(g na q), q = 1..nq: laid out like the forward tables
int minhash_wgs_per_cu(int H);   // resident MinHash workgroups per CU (LDS and register budget)
size_t minhash_queue_bytes(int nblocks_total, int H);
int minhash_queue_words(int H);
// GF(2) jump-ahead tables of the xorshift64 step, two levels: na tables of 8x256 words for M^(g a), a = 1..na (g = 2^XS_JUMP_LOG2),
// then nq tables for M^(g na q), q = 1..nq (weighted chains run past H steps: one coarse + one fine table application)
constexpr int XS_JUMP_LOG2 = 2;   // measured 0 / 1 / 2 / 3 / 4: 86.9 / 84.3 / 83.9 / 84.5 / 85.8 ms MinHash at C2 (2 MB of tables at H = 512)
constexpr int XS_JUMP_NQ = BS_WMAX;
#ifndef MH_W1_JUMP_NA
#define MH_W1_JUMP_NA 16
#endif
constexpr int W1_JUMP_NA = MH_W1_JUMP_NA;     // fine tables of the weight-1 kernel's own (small) set: M^4 .. M^64, then coarse ones M^(64 q)
inline int w1_jump_tables(int H) { return W1_JUMP_NA + ((H + 1) >> XS_JUMP_LOG2) / W1_JUMP_NA + 1; }
void build_xorshift_jump_tables(int na, int nq, uint64_t* out);
void launch_fix_status(hipStream_t st, int32_t* meta, int64_t nreads);
size_t ordered_lds_bytes(int cap, int code_words, int stage_wide);
void launch_ordered(hipStream_t st, const ReadDesc* descs, int64_t nstrands, int max_len_codes, int max_len, const int32_t* h32,
                    const uint8_t* store, const uint64_t* luts, int k2, int S, int cap, int32_t* out_rows, int64_t out_stride,
                    int32_t* out_meta, int64_t meta_stride, int64_t first = 0, int64_t count = -1);

// ---- search_kernels.hip ----
// All-pairs slot-equality count between query entries qlist[0..nq) and index entries [0..ne).
// rowstart != NULL enables triangular tile skipping (device array of ntq+1 linear tile offsets).
// Appends (q,m) with count >= num_min_matches that pass the MinHashSearch filters to cand[] via *cand_count.
void launch_candidates(hipStream_t st, const int32_t* minhash, int64_t row_stride, const int32_t* qminhash, int64_t qrow_stride,
                       const int32_t* qlist, int nq, int ne, const int64_t* ids, const int64_t* qids, const int32_t* meta,
                       const int32_t* qmeta, const SearchParams& sp, const long long* rowstart, long long nblocks_tri, Candidate* cand,
                       unsigned long long* cand_count, unsigned long long cand_cap);
// Inverted index: one open-addressing table of 2^k (value, entry+1) words per MinHash slot; a value's entries beyond the run cap
// are stored contiguously in the overflow pool (CSR: per-(slot, value) start / count in a second hash table).  During the build
// they are appended to `tmp` and counted; launch_index_finalize lays them out once all entries are in.
constexpr int MHAP_MAX_NUM_HASHES = 8192;   // --num-hashes limit (mhap_create); the MinHash kernels' queue entries keep the slot in 16 bits
int minhash_waves_per_workgroup(int H);     // 4, fewer when --num-hashes is so large that four waves' minima do not fit a workgroup's LDS
struct InvIndex {
  uint32_t* ends;        // [H][nb + 1]: postings of bucket b of slot s are items[s][ends[s][b] .. ends[s][b + 1]), ends[s][0] = 0
  uint2* items;          // [H][slot_stride]: (mix of the value, entry), grouped by bucket
  uint32_t nb, shift;    // buckets per slot (a power of two, 1024 .. 2^20), bucket = mix >> shift
  uint64_t slot_stride;  // postings one slot has room for
  uint32_t ne;           // entries the index was sized for (an upper bound of any query's distinct hits)
  uint32_t grouped;      // != 0: every bucket longer than group_t postings is ordered by entry class (entry >> class_log): the dense query
                         // tier then streams, per pass over a range of entries, only the part of a long bucket that lies in it
  uint32_t group_t, class_log;   // (index_group_params: 256 and 15 unless MHAP_INDEX_GROUP_T / MHAP_INDEX_CLASS_LOG say otherwise — tests)
  // round 6: the LINE table in front of ends / items — a lookup of the first query tier reads ONE 64-byte line (index_lines_kernel)
  uint32_t* lines;       // [H][nl][16] or nullptr: line l of slot s packs the postings of buckets [l << lb, (l + 1) << lb)
  uint32_t nl_log, line_lb, line_ebits;   // nl = 1 << nl_log lines per slot, lb = log2(buckets per line), entry bits of a packed posting
  // scratch of the build
  uint2* staged;         // [H][slot_stride]: the postings grouped by coarse bin
  uint32_t* tile_counts; // [H][tiles][coarse bins]
  uint32_t* bin_start;   // [H][coarse bins + 1]
  uint32_t* bin_long;    // [H][coarse bins]: the bin holds a bucket longer than group_t (written by step 4 for step 5)
};
void launch_query_iota(hipStream_t st, int32_t* qlist, int first, int n);
int index_tiles(int ne);
void index_group_params(int64_t entries, InvIndex& ix);   // sets grouped / group_t / class_log for an index of this many entries
int index_coarse_bins();
bool index_line_params(int64_t entries, uint32_t nb, uint32_t& nl_log, uint32_t& lb, uint32_t& ebits);   // false: no line table for this index
size_t index_line_bytes(int H, uint32_t nl_log);
int index_max_buckets_log();
void launch_index_verify(hipStream_t st, const int32_t* minhash, int64_t row_stride, const int32_t* meta, int ne, int H, const InvIndex& ix,
                         unsigned long long* missing);   // self-check: postings that are not where a lookup would find them
void launch_index_build(hipStream_t st, const int32_t* minhash, int64_t row_stride, const int32_t* meta, int ne, int H, const InvIndex& ix);
void launch_index_query(hipStream_t st, const InvIndex& ix, const int32_t* qminhash, int64_t qrow_stride,
                        const int32_t* qlist, int nq, const int64_t* ids, const int64_t* qids, const int32_t* meta, const int32_t* qmeta,
                        const SearchParams& sp, Candidate* cand, unsigned long long* cand_count, unsigned long long cand_cap,
                        unsigned long long* split_count, unsigned long long* elements_total, int32_t* big, unsigned long long* big_count, int tier,
                        unsigned long long* elements_spread);   // elements_spread: index_elements_spread_words() zeroed words (scratch of the count)
int index_elements_spread_words();
int index_query_dense_ranges(int64_t entries);   // passes the dense tier makes over an index of this size
bool index_query_tiers();
bool index_query_tier_ok(int tier, int64_t entries, int num_min_matches);   // tier 0 / 1: can its packed hit-count words hold this index and threshold?   // false: the build has no second tier (-DMH_IQ_BIG_CT=0)
// (tiers: 0 the first, 1 the same kernel with a large table, 2 dense counters — launch_index_query in search_kernels.hip)
// Second stage: one lane per candidate.
void oj_stats_dump();   // -DMH_OJ_STATS builds: print and reset the join kernel's exit statistics
void launch_overlap(hipStream_t st, int nblocks, const Candidate* cand, const unsigned long long* cand_count, unsigned long long cand_cap,
                    const int32_t* ordered, int64_t ord_stride, const int32_t* meta, const int32_t* qordered, int64_t qord_stride,
                    const int32_t* qmeta, const SearchParams& sp, const double* score_table, int32_t* scratch, int64_t scratch_per_lane,
                    DevRecord* recs, unsigned long long* rec_count, unsigned long long rec_cap, unsigned long long* compared, int spread);
// Second stage: one wavefront per candidate (equal-hash join); pairs it cannot decide exactly are appended to `slow`.
// search_kernels_wide.hip: the join kernel compiled a second time with room for 512 joined k-mers per pair (second pass over what the first hands over)
size_t overlap_join_wide_lds_bytes(int S, int shape);
int overlap_join_wide_blocks_per_cu(int S, int shape);
int overlap_join_wide_waves_per_block(int shape);
int overlap_join_wide_capacity();
void launch_overlap_join_wide(hipStream_t st, int shape, int nblocks, int chunk, const void* cand, const unsigned long long* cand_count,
                              unsigned long long cand_cap, const int32_t* ordered, int64_t ord_stride, const int32_t* meta, const int32_t* qordered,
                              int64_t qord_stride, const int32_t* qmeta, const void* sp, const double* score_table, void* recs,
                              unsigned long long* rec_count, unsigned long long rec_cap, unsigned long long* compared, void* slow,
                              unsigned long long* slow_count, unsigned long long* work, const uint16_t* ph, const uint16_t* qph, const int32_t* pass_min);
size_t overlap_join_wide2_lds_bytes(int S, int shape);   // search_kernels_wide2.hip: ... and with room for 1 536 (third pass)
int overlap_join_wide2_blocks_per_cu(int S, int shape);
int overlap_join_wide2_waves_per_block(int shape);
int overlap_join_wide2_capacity();
void launch_overlap_join_wide2(hipStream_t st, int shape, int nblocks, int chunk, const void* cand, const unsigned long long* cand_count,
                               unsigned long long cand_cap, const int32_t* ordered, int64_t ord_stride, const int32_t* meta, const int32_t* qordered,
                               int64_t qord_stride, const int32_t* qmeta, const void* sp, const double* score_table, void* recs,
                               unsigned long long* rec_count, unsigned long long rec_cap, unsigned long long* compared, void* slow,
                               unsigned long long* slow_count, unsigned long long* work, const uint16_t* ph, const uint16_t* qph, const int32_t* pass_min);
constexpr int OJ_MAX_S = 8192;   // largest ordered sketch the join path stages in LDS
size_t overlap_join_lds_bytes(int S, int shape);     // shape: 0 every wave alone, 1 pairs of waves share a query, 2 teams of four + bucket table
int overlap_join_blocks_per_cu(int S, int shape);
int overlap_join_waves_per_block(int shape);
int overlap_join_table_slots(int S);
void launch_overlap_join(hipStream_t st, int shape, int nblocks, int chunk, const Candidate* cand, const unsigned long long* cand_count,
                         unsigned long long cand_cap, const int32_t* ordered, int64_t ord_stride, const int32_t* meta, const int32_t* qordered,
                         int64_t qord_stride, const int32_t* qmeta, const SearchParams& sp, const double* score_table, DevRecord* recs,
                         unsigned long long* rec_count, unsigned long long rec_cap, unsigned long long* compared, Candidate* slow,
                         unsigned long long* slow_count, unsigned long long* work, const uint16_t* ph = nullptr, const uint16_t* qph = nullptr,
                         const int32_t* pass_min = nullptr);
// position histograms of an ordered table (64 cumulative 16-bit counts per entry) and the join kernel's early "below the threshold":
// ph / qph = histograms of the stored / the query table, pass_min[kk] = smallest inter with score_table[inter, kk'] >= threshold for any kk' >= kk
void launch_poshist(hipStream_t st, const int32_t* ordered, int64_t stride, const int32_t* meta, int64_t n, uint16_t* out);
constexpr int POSHIST_BINS = 64;

}  // namespace mhap
