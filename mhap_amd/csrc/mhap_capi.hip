// mhap_capi.hip — C-ABI implementation (include/mhap_hip.h): handle, HBM tables, batch orchestration.
// No CPU fallback lives here: every compute entry point launches the gfx950 kernels or fails.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mhap_hip.h"
static double hp_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define HPROF(tag) do { if (getenv("MHAP_HOST_PROF")) fprintf(stderr, "[host] %-28s %.3f ms\n", tag, hp_now()); } while (0)
// MHAP_DEBUG_SYNC=1: wait for every sketch kernel and say so (which launch hangs or faults)
#define DBGSYNC(h, tag) do { if (getenv("MHAP_DEBUG_SYNC")) { fprintf(stderr, "[dbg] %s launched\n", tag); hipError_t _e = hipStreamSynchronize((h)->stream); fprintf(stderr, "[dbg] %s done: %s\n", tag, hipGetErrorString(_e)); } } while (0)
#include "kernels.hpp"
#include "mhap_internal.hpp"
#include "overlap_lane.hpp"

using namespace mhap;

namespace {

struct TimedLaunch { hipEvent_t a, b; int kind; };

inline int64_t align4(int64_t v) { return (v + 3) & ~(int64_t)3; }

void parallel_for(int64_t n, int nthreads, const std::function<void(int64_t, int64_t)>& fn, int64_t grain = 16384) {
  if (n <= 0) return;
  nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, n / grain + 1));   // spawning a thread costs more than `grain` cheap items
  if (nthreads == 1) { fn(0, n); return; }
  std::vector<std::thread> th;
  int64_t chunk = (n + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; t++) {
    int64_t lo = t * chunk, hi = std::min(n, lo + chunk);
    if (lo >= hi) break;
    th.emplace_back([=, &fn]() { fn(lo, hi); });
  }
  for (auto& t : th) t.join();
}

// Guava 19.0 BloomFilter.create(funnel(putLong), expectedInsertions, fpp = 1e-5) + put of every value (strategy MURMUR128_MITZ_64):
// optimalNumOfBits / optimalNumOfHashFunctions, BitArray of ceil(bits / 64) longs (J/sketch/FrequencyCounts.java:137,192; a zero
// size is bumped to 1, :117-121)
void build_bloom(const int64_t* hashes, int64_t n, int64_t size_bloom, std::vector<unsigned long long>& out, uint64_t& bit_size, int& k);

int host_threads() { return usable_host_threads(32); }

}  // namespace

namespace {
void build_bloom(const int64_t* hashes, int64_t n, int64_t size_bloom, std::vector<unsigned long long>& out, uint64_t& bit_size, int& k) {
  const double nn = (double)(size_bloom <= 0 ? 1 : size_bloom), p = 1.0e-5;
  const int64_t m = (int64_t)(-nn * std::log(p) / (std::log(2.0) * std::log(2.0)));
  k = std::max(1, (int)java_round((double)m / nn * std::log(2.0)));
  const size_t nwords = (size_t)((m + 63) / 64);
  bit_size = (uint64_t)nwords * 64;
  std::vector<std::atomic<unsigned long long>> words(nwords);
  for (auto& w : words) w.store(0ULL, std::memory_order_relaxed);
  const uint64_t bs = bit_size; const int kk = k;
  parallel_for(n, host_threads(), [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; i++) {
      uint64_t h1, h2;
      murmur128_long((uint64_t)hashes[i], h1, h2);
      uint64_t c = h1;
      for (int t = 0; t < kk; t++) {
        const uint64_t bit = (c & 0x7fffffffffffffffULL) % bs;
        words[(size_t)(bit >> 6)].fetch_or(1ULL << (bit & 63), std::memory_order_relaxed);
        c += h2;
      }
    }
  }, 4096);
  out.resize(nwords);
  for (size_t i = 0; i < nwords; i++) out[i] = words[i].load(std::memory_order_relaxed);
}
}  // namespace

struct mhap_handle {
  mhap_params P;
  int device = 0;
  int num_cus = 256;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  int oj_per_cu[3] = {0, 0, 0}, oj_per_cu_S = -1;   // resident join-kernel workgroups per CU (per shape) at ordered sketch size oj_per_cu_S
  int join_mode = 0;                      // MHAP_JOIN_MODE: 0 = by the candidates per query, 1 = alone, 2 = pair, 3 = team
  int join_wide = 2;                      // MHAP_JOIN_WIDE: further passes of the join kernel (512, then 1 536 joined k-mers) over the pairs handed over: 0, 1 or 2 of them
  int ojw_per_cu[2] = {1, 1}, ojw_per_cu_S[2] = {-1, -1};
  hipStream_t mh_stream = nullptr;        // MinHash launch of the weighted strands, next to the launch of the weight-1 strands
  hipEvent_t ev_mh_fork = nullptr, ev_mh_join = nullptr, ev_ix_fork = nullptr, ev_ix_join = nullptr;
  // inverted index state: inv_ends / inv_items hold the index of entries [0, inv_ne) when inv_ready
  bool inv_ready = false; int64_t inv_ne = 0;
  uint64_t index_gen = 0;   // bumped by every change of the entry set (the eager exchange's rows describe one generation)
  bool ph_ready = false; int64_t ph_ne = 0;   // poshist holds the position histograms of entries [0, ph_ne) (reset wherever inv_ready is)
  bool iq_start_mid = false;   // queries of this index start in the middle query tier (set by a chunk that mostly ended up there)
  int64_t reserve_reads = 0;       // mhap_index_reserve: reads the empty index is about to receive, over one or more adds
  std::string err;
  int Hrow = 1;      // minhash row stride (ints)
  int ord_cap = 0;   // ordered-kernel sort capacity

  // filter
  DevBuf f_keys, f_vals, f_bloom;
  FilterTable ft{};
  DevBuf score_tbl, jump_tbl, unjump_tbl, jump_w1_tbl, hash_luts, pass_min_tbl, poshist, q_poshist;

  // index (owned or external)
  bool external = false;
  int64_t n_entries = 0;
  DevBuf own_minhash, own_ordered, own_meta;
  int32_t *d_minhash = nullptr, *d_ordered = nullptr, *d_meta = nullptr;
  DevBuf d_ids;
  std::vector<int64_t> ids;
  std::vector<uint8_t> fwd;
  std::vector<int32_t> seqlen;   // full base length per entry (host mirror)
  std::vector<uint8_t> status;   // per entry status (host mirror)

  // sketch scratch
  DevBuf store, descs, keys, wts, perm, h32, info, slist, slabs, counters, order, mhq, mhmerge;
  int jump_na = 0;   // fine xorshift jump tables (coarse ones follow them in jump_tbl)
  std::vector<int32_t> h_order;
  uint8_t* pin_store = nullptr;   // pinned host staging buffer of stage_reads
  size_t pin_cap = 0;
  uint8_t* pin_io = nullptr;      // pinned bounce buffer of the small device-to-host read-backs (meta rows, records)
  size_t pin_io_cap = 0;
  ReadDesc* h_descs = nullptr; size_t h_descs_cap = 0;   // a batch's descriptors (pinned, grow-only: their upload is a DMA, not a staged copy)
  // ids of an add in flight (mhap_index_add_*): mirrored and uploaded by sketch_staged while the GPU runs the add's kernels, so that
  // finish_add has only the meta rows left to fetch; and whether the ids rise with the entries, kept for the searches of this generation
  struct { int64_t first = 0, n = 0; const int64_t* ids = nullptr; bool done = false; } pend_ids;
  uint64_t mono_gen = ~0ULL; bool mono_val = false, mono_pending = false;   // (mono_val belongs to generation mono_gen: set by finish_add only)
  std::vector<ReadDesc> st_descs;   // staged reads (base_off/length/flags); packed bases resident in `store`
  std::vector<int64_t> st_ids;
  int64_t st_n = 0, st_bytes = 0;

  // query tables for -q mode
  DevBuf q_minhash, q_ordered, q_meta, q_ids;

  // search scratch
  DevBuf qlist, rowstart, cand, slow_cand, slow_cand2, slow_cand3, recs, recs2, ovl_scratch, inv_ends, inv_items, inv_staged, inv_scratch, inv_big, inv_lines, iq_espread;
  // post stage of a search chunk (record read-back, conversion, sink) on a worker thread, one chunk behind the kernels: two sets of buffers
  hipStream_t copy_stream = nullptr;
  uint8_t* pin_rec[2] = {nullptr, nullptr};
  size_t pin_rec_cap[2] = {0, 0};
  mhap_record* out_recs2[2] = {nullptr, nullptr};   // (grow-only raw buffers: a vector's resize would zero-fill hundreds of MB per chunk)
  size_t out_recs2_cap[2] = {0, 0};
  InvIndex inv{};   // device view of the inverted index in inv_ends / inv_items
  mhap_stage_gate gate = nullptr; void* gate_user = nullptr;   // mhap_set_second_stage_gate
  void* dist = nullptr;   // multi-GPU state (mhap_dist.hip)
  std::vector<mhap_record> out_recs;

  // timing
  std::vector<TimedLaunch> pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
  mhap_kernel_times ktimes{};
  mhap_stats stats{};
};

namespace {

int fail(mhap_handle* h, int code, const std::string& msg) { h->err = msg; return code; }

#define HIPCHK(h, expr)                                                                              \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      return fail((h), _e == hipErrorOutOfMemory ? MHAP_E_NOMEM : MHAP_E_HIP,                        \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                                \
  } while (0)

size_t time_begin(mhap_handle* h, int kind, hipStream_t st = nullptr) {
  TimedLaunch t; t.kind = kind;
  if (!h->free_events.empty()) { t.a = h->free_events.back().first; t.b = h->free_events.back().second; h->free_events.pop_back(); }
  else { (void)hipEventCreate(&t.a); (void)hipEventCreate(&t.b); }
  (void)hipEventRecord(t.a, st ? st : h->stream);
  h->pending.push_back(t);
  return h->pending.size() - 1;
}
void time_end(mhap_handle* h, hipStream_t st = nullptr) { (void)hipEventRecord(h->pending.back().b, st ? st : h->stream); }
void time_end_at(mhap_handle* h, size_t idx, hipStream_t st = nullptr) { (void)hipEventRecord(h->pending[idx].b, st ? st : h->stream); }   // (when another timed launch began in between)

// call after the stream is synchronised
void time_collect(mhap_handle* h) {
  for (auto& t : h->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { h->ktimes.ms[t.kind] += ms; h->ktimes.launches[t.kind]++; }
    h->free_events.emplace_back(t.a, t.b);
  }
  h->pending.clear();
}

int sync_stream(mhap_handle* h) {
  HIPCHK(h, hipStreamSynchronize(h->stream));
  time_collect(h);
  return MHAP_OK;
}

// identity score for every reachable (inter, k): BottomOverlapSketch.jaccardToIdentity
// (J/sketch/BottomOverlapSketch.java:391-395) evaluated on the host so libm lives in one place.
// pass_min[kk] = the smallest number of shared k-mers that reaches the threshold for ANY k' >= kk, read off the score table itself
// (no monotonicity assumed): a pair with fewer joined k-mers than pass_min[lower bound of its k] cannot be accepted
// (search_kernels.hip, poshist_kernel).  MinHashSearch.java:229 accepts score >= acceptScore.
void build_pass_min(const std::vector<double>& tbl, int S, double threshold, std::vector<int32_t>& pm) {
  pm.assign((size_t)S + 2, INT32_MAX);
  for (int kk = S; kk >= 0; kk--) {
    int32_t best = kk < S ? pm[(size_t)kk + 1] : INT32_MAX;
    for (int it = 0; it <= kk; it++)
      if (tbl[(size_t)score_index(it, kk)] >= threshold) { best = std::min<int32_t>(best, it); break; }
    pm[(size_t)kk] = best;
  }
}
void fill_score_table(int S, int k2, std::vector<double>& tbl) {
  const int64_t n = score_index(S, S) + 1;
  tbl.resize((size_t)n);
  // row kk has kk + 1 entries: the rows are dealt to the threads in pairs (kk, S - kk) of equal total length — contiguous chunks of a
  // triangle gave the last thread a third of the work (42 ms of mhap_create at S = 1536 on a 16-thread box)
  const int64_t half = (S + 2) / 2;
  parallel_for(half, host_threads(), [&](int64_t lo, int64_t hi) {
    auto row = [&](int64_t kk) {
      for (int64_t it = 0; it <= kk; it++) {
        double j = (kk == 0) ? 0.0 : (double)it / (double)kk;
        double d = -1.0 / (double)k2 * std::log(2.0 * j / (1.0 + j));
        tbl[(size_t)score_index((int)it, (int)kk)] = std::exp(-d);
      }
    };
    for (int64_t a = lo; a < hi; a++) { row(a); if (S - a != a && S - a >= half) row(S - a); }
  }, 4);
}

// (the two tables are computed on the host — mhap_create starts that before it touches the device — and uploaded here)
int upload_score_table(mhap_handle* h, const std::vector<double>& tbl, const std::vector<int32_t>& pm) {
  HIPCHK(h, h->score_tbl.ensure(tbl.size() * 8));
  HIPCHK(h, hipMemcpy(h->score_tbl.p, tbl.data(), tbl.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(h, h->pass_min_tbl.ensure(pm.size() * 4));
  HIPCHK(h, hipMemcpy(h->pass_min_tbl.p, pm.data(), pm.size() * 4, hipMemcpyHostToDevice));
  return MHAP_OK;
}

inline int code_of(char c) {
  switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

// ------------------------------------------------------------------------------------------------
// Staging: pack reads (2 bits/base; raw bytes for reads with non-ACGT chars) and upload them once.
// After staging the packed reads are resident in HBM; sketch_staged() only launches kernels.
// ------------------------------------------------------------------------------------------------
// 2-bit code of an upper-case base without a table: (c >> 1) & 3 maps A,C,G,T to 0,1,3,2; x ^ (x >> 1) turns that into
// 0,1,2,3.  A char is a base iff the code maps back to it.  Both loops below are branch-free so that they vectorise.
static inline uint32_t base_code(uint8_t c) { const uint32_t x = (c >> 1) & 3u; return x ^ (x >> 1); }
static inline bool is_base(uint8_t c) { return ((0x54474341u >> (8 * base_code(c))) & 0xFFu) == c; }

int stage_reads(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, int64_t n, bool fwd_only) {
  HPROF("stage_reads begin");
  const int nthreads = host_threads();
  h->st_descs.resize((size_t)n);
  h->st_n = n;
  std::vector<uint8_t> israw((size_t)n);
  parallel_for(n, nthreads, [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; i++) {
      const uint8_t* s = (const uint8_t*)bases + offsets[i];
      const int L = lengths[i];
      uint32_t bad = 0;
      for (int j = 0; j < L; j++) bad |= is_base(s[j]) ? 0u : 1u;
      israw[(size_t)i] = (uint8_t)bad;
    }
  }, 256);
  int64_t store_bytes = 0;
  for (int64_t i = 0; i < n; i++) {
    ReadDesc& d = h->st_descs[(size_t)i];
    const int L = std::max(0, lengths[i]);
    d.length = L;
    d.flags = 0;
    if (L < h->P.min_olap_length) d.flags |= MHAP_RD_SKIP;          // SequenceSketchStreamer.java:129-133
    if (israw[(size_t)i]) d.flags |= MHAP_RD_RAW;
    if (fwd_only) d.flags |= MHAP_RD_FWDONLY;
    d.base_off = store_bytes;
    d.key_off = d.h2_off = 0; d.key_stride = d.h2_stride = 0;
    if (!(d.flags & MHAP_RD_SKIP)) store_bytes += (d.flags & MHAP_RD_RAW) ? align4(L) : align4((L + 3) / 4);
  }
  // pinned staging buffer (kept across calls): no zero-fill pass, and the upload runs at PCIe speed
  const size_t need = (size_t)std::max<int64_t>(store_bytes, 4);
  if (h->pin_cap < need) {
    if (h->pin_store) (void)hipHostFree(h->pin_store);
    h->pin_store = nullptr; h->pin_cap = 0;
    HIPCHK(h, hipHostMalloc((void**)&h->pin_store, need + need / 8, hipHostMallocDefault));
    h->pin_cap = need + need / 8;
  }
  uint8_t* hs = h->pin_store;
  parallel_for(n, nthreads, [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; i++) {
      const ReadDesc& d = h->st_descs[(size_t)i];
      if (d.flags & MHAP_RD_SKIP) continue;
      const uint8_t* s = (const uint8_t*)bases + offsets[i];
      uint8_t* dst = hs + d.base_off;
      const int L = d.length;
      if (d.flags & MHAP_RD_RAW) {
        memcpy(dst, s, (size_t)L);
        for (int j = L; j < (int)align4(L); j++) dst[j] = 0;
      } else {
        const int full = L >> 2;
        for (int q = 0; q < full; q++)
          dst[q] = (uint8_t)(base_code(s[4 * q]) | (base_code(s[4 * q + 1]) << 2) | (base_code(s[4 * q + 2]) << 4) | (base_code(s[4 * q + 3]) << 6));
        const int nb = (int)align4((L + 3) / 4);
        for (int q = full; q < nb; q++) {
          uint32_t v = 0;
          for (int j = 4 * q; j < L && j < 4 * q + 4; j++) v |= base_code(s[j]) << (2 * (j & 3));
          dst[q] = (uint8_t)v;
        }
      }
    }
  }, 256);
  HIPCHK(h, h->store.ensure(need));
  HIPCHK(h, hipMemcpyAsync(h->store.p, hs, need, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->st_bytes = (int64_t)need;
  HPROF("stage_reads end (packed + h2d)");
  return MHAP_OK;
}

// Sketch the staged reads into device rows [0, 2*st_n) of the given tables (kernels only + tiny descriptor uploads).
// Batches run back to back on the handle's stream.  (A two-stream variant that overlapped hash/weight/ordered of
// batch b+1 with MinHash of batch b was measured at 358 -> 355..365 ms/step: the kernels compete for the same VALU
// issue slots and LDS, so it was removed.)
// (Re)allocate the inverted index for `ne` entries: per MinHash slot an ends table of one bucket per entry (rounded up to a power
// of two: 0.5-1 postings per bucket), room for ne postings of 8 bytes, and the sort's scratch (as much again + the tile counts).
static int inv_alloc(mhap_handle* h, int64_t ne) {
  const int H = h->P.num_hashes;
  uint32_t lg = 10;
#ifndef MH_INV_BUCKETS_PER_ENTRY
#define MH_INV_BUCKETS_PER_ENTRY 1
#endif
  while (lg < (uint32_t)index_max_buckets_log() && (1ULL << lg) < (uint64_t)MH_INV_BUCKETS_PER_ENTRY * (uint64_t)ne) lg++;
  const size_t nb = (size_t)1 << lg, stride = (size_t)std::max<int64_t>(ne, 1), tiles = (size_t)index_tiles((int)ne), cb = (size_t)index_coarse_bins();
  HIPCHK(h, h->inv_ends.ensure((size_t)H * (nb + 1) * 4));
  HIPCHK(h, h->inv_items.ensure((size_t)H * stride * 8));
  HIPCHK(h, h->inv_staged.ensure((size_t)H * stride * 8));
  HIPCHK(h, h->inv_scratch.ensure((size_t)H * (tiles * cb + 2 * cb + 1) * 4));
  h->inv.ends = h->inv_ends.as<uint32_t>(); h->inv.items = h->inv_items.as<uint2>(); h->inv.staged = h->inv_staged.as<uint2>();
  h->inv.tile_counts = h->inv_scratch.as<uint32_t>(); h->inv.bin_start = h->inv.tile_counts + (size_t)H * tiles * cb;
  h->inv.bin_long = h->inv.bin_start + (size_t)H * (cb + 1);
  h->inv.nb = (uint32_t)nb; h->inv.shift = 32 - lg;
  h->inv.slot_stride = (uint64_t)stride;
  h->inv.ne = (uint32_t)std::min<int64_t>(ne, 0xFFFFFFFFLL);
  index_group_params(ne, h->inv);
  // the line table of the first query tier (round 6); an index it does not cover, or no memory for it: the tier reads ends / items
  h->inv.lines = nullptr; h->inv.nl_log = 0; h->inv.line_lb = 0; h->inv.line_ebits = 0;
  uint32_t nl_log = 0, lb = 0, eb = 0;
  if (index_line_params(ne, (uint32_t)nb, nl_log, lb, eb)) {
    if (h->inv_lines.ensure(index_line_bytes(H, nl_log)) == hipSuccess) {
      h->inv.lines = h->inv_lines.as<uint32_t>(); h->inv.nl_log = nl_log; h->inv.line_lb = lb; h->inv.line_ebits = eb;
    } else (void)hipGetLastError();
  }
  return MHAP_OK;
}

int ensure_inverted_index(mhap_handle* h, hipStream_t st = nullptr, int64_t ne_override = -1);
// the ids of entries [first, first + 2 n): host mirror, device copy, and whether the ids of the whole index rise with the entries
int fill_ids(mhap_handle* h, int64_t first, const int64_t* ids, int64_t n) {
  h->ids.resize((size_t)(first + 2 * n)); h->fwd.resize((size_t)(first + 2 * n));
  for (int64_t i = 0; i < n; i++) {
    h->ids[(size_t)(first + 2 * i)] = ids[i]; h->ids[(size_t)(first + 2 * i + 1)] = ids[i];
    h->fwd[(size_t)(first + 2 * i)] = 1; h->fwd[(size_t)(first + 2 * i + 1)] = 0;
  }
  HPROF("ids built");
  HIPCHK(h, hipMemcpy(h->d_ids.as<int64_t>() + first, h->ids.data() + first, (size_t)(2 * n) * 8, hipMemcpyHostToDevice));
  HPROF("ids h2d");
  bool mono = true;
  for (int64_t e = 1; e < first + 2 * n && mono; e++) if (h->ids[(size_t)e] < h->ids[(size_t)(e - 1)]) mono = false;
  h->mono_pending = mono;   // (finish_add starts the generation these ids belong to)
  return MHAP_OK;
}


// eager_index_entries > 0 (an add that very likely completes the index): the inverted index of entries [0, eager_index_entries) —
// the tables' rows up to the end of this add — is built as soon as the last batch's MinHash rows and statuses exist, on the side
// stream, next to the ordered-sketch kernel (which is bound by the LDS pipe and leaves the memory side idle): at C2 3 ms of index
// build used to follow 4.8 ms of ordered kernel.
int sketch_staged(mhap_handle* h, int32_t* d_minhash, int64_t mh_stride, int32_t* d_ordered, int64_t ord_stride, int32_t* d_meta,
                  int64_t eager_index_entries = 0, bool eager_exchange = false, bool eager_first = false) {
  const int64_t n = h->st_n;
  if (n <= 0) { if (eager_exchange) { const int rx = dist_eager_begin(h, 0, nullptr, false); if (rx < 0) return rx; } return MHAP_OK; }
  HPROF("sketch_staged begin");
  const int k = h->P.kmer_size, k2 = h->P.ordered_kmer_size, H = h->P.num_hashes, S = h->P.ordered_sketch_size;
  int64_t batch_bases = 1LL << 30;   // bases per launch group: 16 B of scratch per base (weights + class lists of both strands) = 16 GB of the 288 GB;
                                     // fewer, larger launches = fewer drain tails of the persistent MinHash waves (a strand takes ~2 ms)
  {   // Round 6: larger launch groups where the HBM is there — a job of more than 1 G bases (configs[3]: 15 G) is cut into groups of up to 4 G bases
      // as long as their scratch stays under a quarter of the memory that is free now: 14 -> 4 groups at C4, 1 469 -> 1 437 ms on one box (fewer
      // drain tails, fewer launches into an idle GPU).  The groups of one rank of configs[4] at N = 8 (137 GB in use) stay where they were.
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const int64_t by_mem = (int64_t)(free_b / 4 / 16);
      batch_bases = std::max<int64_t>(batch_bases, std::min<int64_t>(4LL << 30, by_mem));
    }
  }
  if (const char* e = getenv("MHAP_BATCH_BASES")) { long long v = atoll(e); if (v > 0) batch_bases = v; }
  // Strands whose hashes are recomputed from their 2-bit codes wherever they are consumed need no key / 32-bit hash arrays;
  // raw-byte reads, k != 16 / k2 != 12 and reads beyond the weight kernel's LDS path are hashed by hash_kmers_kernel (MHAP_RD_MAT).
  const char* fenv = getenv("MHAP_FUSED_HASH");
  const bool from_codes_ok = !(fenv && atoi(fenv) == 0);
  auto is_mat = [&](const ReadDesc& d) { return (d.flags & MHAP_RD_RAW) || !from_codes_ok || !strand_hashes_from_codes(d.length, k, k2); };
  // ---- batch plan + worst-case scratch sizes (allocated once) ----
  struct Batch { int64_t r0, r1; int max_len; };
  std::vector<Batch> plan;
  int64_t max_key = 4, max_h2 = 4, max_w = 4, max_nb = 1;
  int max_len_all = 0;
  for (int64_t r0 = 0; r0 < n;) {
    int64_t r1 = r0, tot = 0;
    while (r1 < n && (r1 == r0 || tot + h->st_descs[(size_t)r1].length <= batch_bases) && (r1 - r0) < (1 << 22)) { tot += h->st_descs[(size_t)r1].length; r1++; }
    Batch b{r0, r1, 0};
    int64_t key_elems = 0, h2_elems = 0, w_elems = 0;
    for (int64_t i = r0; i < r1; i++) {
      const ReadDesc& d = h->st_descs[(size_t)i];
      if (d.flags & MHAP_RD_SKIP) continue;
      w_elems += 2 * align4(std::max(0, d.length - k + 1));
      if (is_mat(d)) { key_elems += 2 * align4(std::max(0, d.length - k + 1)); h2_elems += 2 * align4(std::max(0, d.length - k2 + 1)); }
      b.max_len = std::max(b.max_len, d.length);
    }
    max_key = std::max(max_key, key_elems); max_h2 = std::max(max_h2, h2_elems); max_w = std::max(max_w, w_elems); max_nb = std::max(max_nb, r1 - r0);
    max_len_all = std::max(max_len_all, b.max_len);
    plan.push_back(b);
    r0 = r1;
  }
  // Eager exchange (mhap_dist_set_eager): the rendezvous of the ranks — collective, so every rank of the job gets here — says whether
  // this add gathers its rows while it computes.  Only an add of one launch group can (the rows of earlier groups would have to wait).
  bool eager_x = false;
  if (eager_exchange) {
    const int rx = dist_eager_begin(h, n, h->st_ids.data(), plan.size() == 1 && eager_first);
    if (rx < 0) return rx;
    eager_x = rx == 1;
  }
  HIPCHK(h, h->descs.ensure((size_t)max_nb * sizeof(ReadDesc)));
  HIPCHK(h, h->keys.ensure((size_t)max_key * 8));
  HIPCHK(h, h->h32.ensure((size_t)max_h2 * 4));
  HIPCHK(h, h->wts.ensure((size_t)max_w * 4));
  HIPCHK(h, h->perm.ensure((size_t)max_w * 4));
  HIPCHK(h, h->info.ensure((size_t)(2 * max_nb) * sizeof(StrandInfo)));
  HIPCHK(h, h->slist.ensure((size_t)(4 * max_nb) * 4));
  HIPCHK(h, h->counters.ensure(256));
  {
    const int wb = weight_grid(h->num_cus, 2 * max_nb, max_len_all, k);
    int64_t se = 64;
    while (3 * se < 4LL * std::max(1, max_len_all - k + 1)) se <<= 1;
    HIPCHK(h, h->slabs.ensure((size_t)wb * (size_t)se * 4));
  }
  HPROF("sketch: plan + scratch");
  for (const Batch& B : plan) {
    const int64_t nb = B.r1 - B.r0, nstr = 2 * nb;
    if (h->h_descs_cap < (size_t)nb) {
      if (h->h_descs) (void)hipHostFree(h->h_descs);
      h->h_descs = nullptr; h->h_descs_cap = 0;
      const size_t want = (size_t)nb + (size_t)nb / 4 + 64;
      HIPCHK(h, hipHostMalloc((void**)&h->h_descs, want * sizeof(ReadDesc), hipHostMallocDefault));
      h->h_descs_cap = want;
    }
    int64_t key_elems = 0, h2_elems = 0, w_elems = 0;
    bool any_mat = false;
    int min_len_b = INT32_MAX, max_len_codes = 0;
    for (int64_t i = 0; i < nb; i++) {
      ReadDesc d = h->st_descs[(size_t)(B.r0 + i)];
      const bool skip = (d.flags & MHAP_RD_SKIP) != 0;
      if (!skip) min_len_b = std::min(min_len_b, d.length);
      const int64_t nk = align4(std::max(0, d.length - k + 1)), nk2 = align4(std::max(0, d.length - k2 + 1));
      d.key_off = d.h2_off = 0;
      d.w_off = w_elems; d.key_stride = (int32_t)nk; d.h2_stride = (int32_t)nk2;
      if (!skip) {
        w_elems += 2 * nk;
        if (is_mat(d)) {
          d.flags |= MHAP_RD_MAT; any_mat = true;
          d.key_off = key_elems; d.h2_off = h2_elems;
          key_elems += 2 * nk; h2_elems += 2 * nk2;
        } else max_len_codes = std::max(max_len_codes, d.length);
      }
      h->h_descs[(size_t)i] = d;
    }
    // reads of clearly different lengths: hand them out longest first (counting sort on length / 128), so that the persistent
    // workgroups do not end on a long read while the rest of the GPU idles
    const int32_t* d_order = nullptr;
    if (min_len_b * 5 < B.max_len * 4 && nb > 1 && !getenv("MHAP_NO_LENGTH_ORDER")) {
      const int nbk = B.max_len / 128 + 2;
      std::vector<int64_t> start((size_t)nbk + 1, 0);
      for (int64_t i = 0; i < nb; i++) start[(size_t)(nbk - 1 - h->h_descs[(size_t)i].length / 128)]++;
      int64_t acc = 0;
      for (int b = 0; b <= nbk; b++) { const int64_t c = start[(size_t)b]; start[(size_t)b] = acc; acc += c; }
      h->h_order.resize((size_t)nb);
      for (int64_t i = 0; i < nb; i++) h->h_order[(size_t)start[(size_t)(nbk - 1 - h->h_descs[(size_t)i].length / 128)]++] = (int32_t)i;
      HIPCHK(h, h->order.ensure((size_t)nb * 4));
      HIPCHK(h, hipMemcpyAsync(h->order.p, h->h_order.data(), (size_t)nb * 4, hipMemcpyHostToDevice, h->stream));
      d_order = h->order.as<int32_t>();
    }
    int64_t slab_entries = 64;
    while (3 * slab_entries < 4LL * std::max(1, B.max_len - k + 1)) slab_entries <<= 1;
    int32_t* mh_rows = d_minhash + (2 * B.r0) * mh_stride;
    int32_t* ord_rows = d_ordered + (2 * B.r0) * ord_stride;
    int32_t* meta_rows = d_meta + (2 * B.r0) * META_W;
    unsigned long long* ctr = h->counters.as<unsigned long long>();
    const ReadDesc* dd = h->descs.as<ReadDesc>();
    HIPCHK(h, hipMemcpyAsync(h->descs.p, h->h_descs, (size_t)nb * sizeof(ReadDesc), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(h->counters.p, 0, 256, h->stream));
    const bool fused = k == 16 && k2 == 12;   // the table path exists: strands without MHAP_RD_MAT are hashed where their hashes are used
    if (any_mat) {
      time_begin(h, MHAP_K_HASH);
      launch_hash_kmers(h->stream, dd, nstr, B.max_len, h->store.as<uint8_t>(), h->keys.as<int64_t>(), h->h32.as<int32_t>(), k, k2,
                        h->hash_luts.as<uint64_t>(), 1);
      time_end(h);
      DBGSYNC(h, "hash_kmers");
    }
    HPROF("sketch: descriptors up");
    time_begin(h, MHAP_K_WEIGHT);
    launch_kmer_weights(h->stream, h->num_cus, dd, nstr, B.max_len, h->keys.as<int64_t>(), h->wts.as<uint32_t>(), h->perm.as<uint32_t>(),
                        h->slabs.as<uint32_t>(), slab_entries, ctr + 0, k, h->ft, h->P.repeat_weight, h->info.as<StrandInfo>(), fused,
                        h->store.as<uint8_t>(), h->hash_luts.as<uint64_t>(), d_order, h->slist.as<int32_t>());
    time_end(h);
    DBGSYNC(h, "kmer_weights");
    int per_cu = minhash_wgs_per_cu(H);   // the launch is persistent: exactly the workgroups that can be resident
    if (const char* e = getenv("MHAP_MINHASH_WGS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) per_cu = v; }
    int mblocks = h->num_cus * per_cu;   // (each launch is trimmed to the workgroups its work list can feed)
    if (eager_x) mblocks = std::max(per_cu, mblocks - dist_eager_reserve_wgs(h));   // (room for the all-gather's own kernels: mhap_dist.hip)
    // (the slack factor only where the workgroups really have fewer than four waves — --num-hashes beyond ~3 000 — instead of 2 GB
    //  of queue scratch at --num-hashes 2048 and 8.6 GB at 8192 for everybody)
    HIPCHK(h, h->mhq.ensure(minhash_queue_bytes(2 * std::max(mblocks, 1), H) * (size_t)(minhash_waves_per_workgroup(H) < 4 ? 4 : 1)));
    HIPCHK(h, h->mhmerge.ensure(minhash_merge_bytes(std::max(mblocks, 1), H)));
    // The strands with weighted k-mers are a second launch (own instantiation).  On the same stream a handful of such strands (C2:
    // under 1 %) hold the GPU for one strand's duration (3 ms) after the weight-1 launch has drained.  So the two list lengths are
    // read back (one 16-byte copy: the weight kernel has to be complete anyway), each launch gets only the workgroups its list
    // can feed, and the weighted one goes first, on its own stream: its few workgroups take their slots, the weight-1 launch fills
    // the rest of the GPU.  (Full grids side by side do not work: whichever launch is first in line takes every slot before most of
    // its workgroups find their list empty, +8 ms; the weighted launch second in line only moves into the draining tail, -1 ms.)
    unsigned long long lens[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(lens, ctr + 4, 16, hipMemcpyDeviceToHost, h->stream));
    { const int rs = sync_stream(h); if (rs != MHAP_OK) return rs; }
    HPROF("sketch: list lengths back");
    // (Round 4 measured the ordered-sketch kernel — it only reads the strands — on a side stream next to the MinHash launch, with equal
    //  and with lowest priority, and next to the weight kernel: 107.4 -> 107.6-108.4 and 108.1 -> 110-111.6 ms at C2.  It trickles
    //  through the persistent MinHash grid's slots for 60-90 ms and slows that kernel by what it gains; the weight kernel and it both
    //  live on the LDS pipe.  One stream it is.)
    const hipStream_t ost = h->stream;
    auto do_ordered = [&](int64_t first, int64_t count) -> int {
      time_begin(h, MHAP_K_ORDERED, ost);
      launch_ordered(ost, dd, nstr, max_len_codes, B.max_len, h->h32.as<int32_t>(), h->store.as<uint8_t>(), h->hash_luts.as<uint64_t>(), k2, S, h->ord_cap,
                     ord_rows, ord_stride, meta_rows, META_W, first, count);
      time_end(h, ost);
      return MHAP_OK;
    };
    // Round 6: PART of the ordered kernel runs between the weight kernel's read-back and the weight-1 MinHash launch, and the weighted strands'
    // MinHash launch (its own stream) runs NEXT TO THAT PART instead of next to the weight-1 launch's start.  Measured, not derived (EXPERIMENTS.md,
    // round 6, "the MinHash launch's clock"): the weight-1 launch of C2 — 75 ms at the card's power limit — holds whatever clock the power
    // management settles on when it starts.  Started into an idle GPU (the read-back is a host round trip) together with the weighted launch it held
    // 1 820-2 126 MHz from step to step; started alone, right behind a running, moderately busy kernel, 2 164-2 179 MHz every time
    // (MHAP_MINHASH_PROF): 76.2-76.8 -> 72.1-72.4 ms, the C2 step 93.4-94.6 -> 90.5-91.3 ms on one box.  With the weighted launch made to wait for
    // the ordered part the gain is gone (76.1-76.3); behind memory fills the clock is 1 800; with an idle gap behind the ordered part 1 850-2 030.
    // The rest of the strands' ordered rows are made behind the MinHash launch, next to the index build as before (the build takes about as long as
    // 0.45 of the ordered kernel beside it).  Only where a LONG weight-1 launch is the job: a launch group of 0.9 G bases of reads or more (a
    // MinHash launch of 60 ms and more: C2 on one GPU, the launch groups of C4) with at most a fifth of the strands weighted.  Shorter launches
    // lose by it — one rank's share of an N-GPU C2 job, same box, alternating: N = 2 / 4 / 8 rank step 50.5 -> 50.7 / 26.85 -> 27.6 / 14.8 -> 15.2 ms —
    // and a -f run's strands are all weighted (its MinHash launch would run under the ordered part from the start: c5slice 180.3 -> 180.5 / 184.6).
    // MHAP_ORDERED_SPLIT = per cent of the strands in the first part (0 = rounds 1-5's order, 100 = all of them); MHAP_ORDERED_NOWAIT=0 makes the
    // weighted launch wait for the first part; MHAP_ORDERED_FIRST=1 / 2 = all of it first / and an idle gap behind it (the experiments).
    // (read at every launch group: tests switch them inside one process)
    const int ord_first_env = []() { const char* e = getenv("MHAP_ORDERED_FIRST"); return e ? atoi(e) : 0; }();
    const int ord_split_env = []() { const char* e = getenv("MHAP_ORDERED_SPLIT"); const int v = e ? atoi(e) : -1; return v > 100 ? 100 : v; }();
    int64_t batch_read_bases = 0;
    for (int64_t i = 0; i < nb; i++) batch_read_bases += h->h_descs[(size_t)i].length;
    const bool split_pays = batch_read_bases >= 900000000LL && (int64_t)lens[0] >= 4 * (int64_t)lens[1];
    int64_t ord_pre = ord_first_env ? nstr : (nstr * (int64_t)(ord_split_env >= 0 ? ord_split_env : (split_pays ? 55 : 0))) / 100;
    ord_pre &= ~(int64_t)1;   // (both strands of a read in one part)
    bool ordered_done = false;
    if (ord_pre > 0 && !eager_x) {
      (void)do_ordered(0, ord_pre);
      ordered_done = ord_pre >= nstr;
      const int nowait = []() { const char* e = getenv("MHAP_ORDERED_NOWAIT"); return e ? atoi(e) : 1; }();
      if (!nowait) {
        HIPCHK(h, hipEventRecord(h->ev_mh_fork, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->mh_stream, h->ev_mh_fork, 0));
      }
    } else ord_pre = 0;
    if (eager_x) {
      // eager exchange: the ordered rows first (they do not depend on the MinHash rows), so that their all-gather — 6/7 of the bytes
      // a rank sends — runs under the MinHash kernel; its copy engines / RCCL workgroups are in place before the persistent grid starts
      (void)do_ordered(0, nstr);
      ordered_done = true;
      const int rxo = dist_eager_ordered(h, h->stream, ord_rows);
      if (rxo != MHAP_OK) return rxo;
    }
    {   // experiments (round 6, EXPERIMENTS.md "the MinHash launch's clock"): MHAP_ORDERED_FIRST=2 = an idle gap between the ordered kernel and the
        // MinHash launch; MHAP_W1_PREFILL=n = n fills of the queue buffer in front of the MinHash launch (it is enqueued behind running work)
      static const int prefill = []() { const char* e = getenv("MHAP_W1_PREFILL"); return e ? atoi(e) : 0; }();
      if (ord_first_env == 2) (void)hipStreamSynchronize(h->stream);
      for (int i = 0; i < prefill; i++) (void)hipMemsetAsync(h->mhq.p, 0, h->mhq.cap, h->stream);
    }
    const size_t t_mh = time_begin(h, MHAP_K_MINHASH);
    const bool mh_forked = launch_minhash(h->stream, h->mh_stream, mblocks, (int64_t)lens[0], (int64_t)lens[1], dd, nstr, h->keys.as<int64_t>(), h->wts.as<uint32_t>(),
                   h->perm.as<uint32_t>(), h->info.as<StrandInfo>(), h->store.as<uint8_t>(), h->hash_luts.as<uint64_t>(), k, k2, H, ctr, mh_rows, mh_stride,
                   meta_rows + 3, META_W, h->jump_tbl.as<uint64_t>(), h->jump_na, h->slist.as<int32_t>(), h->mhq.as<uint32_t>(), h->unjump_tbl.as<uint64_t>(), h->jump_w1_tbl.as<uint64_t>(),
                   h->mhmerge.as<unsigned long long>(), std::max(0, B.max_len - k + 1));
    if (mh_forked) {   // (the strands with weighted k-mers went to their own stream)
      HIPCHK(h, hipEventRecord(h->ev_mh_join, h->mh_stream));
      HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_mh_join, 0));
    }
    time_end_at(h, t_mh);
    DBGSYNC(h, "minhash");
    launch_fix_status(h->stream, meta_rows, nb);   // statuses are final here (the ordered kernel only writes sizes)
    if (eager_x) { const int rxm = dist_eager_minhash(h, h->stream, mh_rows, meta_rows); if (rxm != MHAP_OK) return rxm; }
    bool eager_launched = false;
    if (eager_index_entries > 0 && &B == &plan.back() && !getenv("MHAP_NO_EAGER_INDEX")) {
      HIPCHK(h, hipEventRecord(h->ev_ix_fork, h->stream));
      HIPCHK(h, hipStreamWaitEvent(h->mh_stream, h->ev_ix_fork, 0));
      const int rce = ensure_inverted_index(h, h->mh_stream, eager_index_entries);
      if (rce != MHAP_OK) return rce;
      HIPCHK(h, hipEventRecord(h->ev_ix_join, h->mh_stream));
      eager_launched = true;
    }
    if (!ordered_done) (void)do_ordered(ord_pre, nstr - ord_pre);
    DBGSYNC(h, "ordered");
    HIPCHK(h, hipGetLastError());
    if (eager_launched) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_ix_join, 0));
    HPROF("sketch: all launched");
    if (&B == &plan.back() && h->pend_ids.n > 0 && !h->pend_ids.done) {   // (host work and a copy the kernels do not touch: under the GPU's time)
      const int rf = fill_ids(h, h->pend_ids.first, h->pend_ids.ids, h->pend_ids.n);
      if (rf != MHAP_OK) { (void)sync_stream(h); return rf; }
      h->pend_ids.done = true;
    }
    int rc = sync_stream(h);   // h_descs is reused by the next batch
    if (rc != MHAP_OK) return rc;
    HPROF("sketch: stream drained");
  }
  return MHAP_OK;
}

int sketch_into(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, int64_t n, bool fwd_only,
                int32_t* d_minhash, int64_t mh_stride, int32_t* d_ordered, int64_t ord_stride, int32_t* d_meta) {
  if (n <= 0) return MHAP_OK;
  int rc = stage_reads(h, bases, offsets, lengths, n, fwd_only);
  if (rc != MHAP_OK) return rc;
  rc = sketch_staged(h, d_minhash, mh_stride, d_ordered, ord_stride, d_meta);
  h->st_n = 0;   // staging consumed (explicit mhap_stage_reads keeps it; see mhap_index_add_staged)
  return rc;
}


int ensure_index_capacity(mhap_handle* h, int64_t entries) {
  if (h->external) return fail(h, MHAP_E_STATE, "index tables are externally owned (mhap_index_set_device); clear the index first");
  const int S = h->P.ordered_sketch_size;
  HIPCHK(h, h->own_minhash.ensure((size_t)entries * h->Hrow * 4, true, h->stream));
  HIPCHK(h, h->own_ordered.ensure((size_t)entries * S * 8, true, h->stream));
  HIPCHK(h, h->own_meta.ensure((size_t)entries * META_W * 4, true, h->stream));
  HIPCHK(h, h->d_ids.ensure((size_t)entries * 8, true, h->stream));
  h->d_minhash = h->own_minhash.as<int32_t>();
  h->d_ordered = h->own_ordered.as<int32_t>();
  h->d_meta = h->own_meta.as<int32_t>();
  return MHAP_OK;
}

// pull meta rows [first, first+count) into the host mirrors
// pinned scratch of at least `bytes` (grow-only): device-to-host copies into it run at PCIe speed without a staging copy
static void* pinned_io(mhap_handle* h, size_t bytes) {
  if (h->pin_io_cap < bytes) {
    if (h->pin_io) (void)hipHostFree(h->pin_io);
    h->pin_io = nullptr; h->pin_io_cap = 0;
    if (hipHostMalloc((void**)&h->pin_io, bytes + bytes / 4 + 4096, hipHostMallocDefault) != hipSuccess) return nullptr;
    h->pin_io_cap = bytes + bytes / 4 + 4096;
  }
  return h->pin_io;
}

int mirror_meta(mhap_handle* h, const int32_t* d_meta, int64_t first, int64_t count) {
  int32_t* m = (int32_t*)pinned_io(h, (size_t)count * META_W * 4);
  if (!m) return fail(h, MHAP_E_HIP, "cannot allocate pinned host memory");
  HIPCHK(h, hipMemcpy(m, d_meta + first * META_W, (size_t)count * META_W * 4, hipMemcpyDeviceToHost));
  if ((int64_t)h->seqlen.size() < first + count) { h->seqlen.resize((size_t)(first + count)); h->status.resize((size_t)(first + count)); }
  for (int64_t e = 0; e < count; e++) { h->seqlen[(size_t)(first + e)] = m[(size_t)e * META_W + 2]; h->status[(size_t)(first + e)] = (uint8_t)m[(size_t)e * META_W + 3]; }
  return MHAP_OK;
}

struct QuerySide {
  const int32_t* d_minhash; int64_t mh_stride;
  const int32_t* d_ordered; int64_t ord_stride;
  const int32_t* d_meta;
  const int64_t* d_ids;
  const int64_t* h_ids; const int32_t* h_seqlen;
  int64_t n_rows;   // rows of the query tables
  // Round 6 — device-resident query rows (mhap_find_matches_device, the sharded search): `identity` = every row is a query unless its status
  // word says "not sketched", which the first tier checks itself; the list is then made on the device and the first kernels are launched
  // BEFORE the rows' meta words have reached the host.  `prep` (run once, behind the first launch) waits for them and fills h_seqlen / h_valid.
  bool identity = false;
  std::function<int(QuerySide&)> prep;
  const uint8_t* h_valid = nullptr;
};

// Run candidate + second stage for the query entries in `ql` (entry indices into the query side).
// (re)build the inverted index for the current entries unless the table in place already covers them
int ensure_inverted_index(mhap_handle* h, hipStream_t st, int64_t ne_override) {
  const int ne = (int)(ne_override >= 0 ? ne_override : h->n_entries), H = h->P.num_hashes;
  if (!st) st = h->stream;
  if (h->inv_ready && h->inv_ne == (int64_t)ne) return MHAP_OK;
  { const int rr = inv_alloc(h, ne); if (rr != MHAP_OK) return rr; }
  HPROF("index build launch");
  time_begin(h, MHAP_K_INDEX_BUILD, st);
  launch_index_build(st, h->d_minhash, h->Hrow, h->d_meta, ne, H, h->inv);
  time_end(h, st);
  HIPCHK(h, hipGetLastError());
  if (getenv("MHAP_DEBUG_INDEX")) {   // self-check: every stored (entry, slot) must find its own posting
    HIPCHK(h, h->counters.ensure(256));
    unsigned long long* ctr = h->counters.as<unsigned long long>();
    unsigned long long missing = 0;
    HIPCHK(h, hipMemsetAsync(ctr + 15, 0, 8, st));
    launch_index_verify(st, h->d_minhash, h->Hrow, h->d_meta, ne, H, h->inv, ctr + 15);
    HIPCHK(h, hipMemcpyAsync(&missing, ctr + 15, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    fprintf(stderr, "[index] self-check: %llu of %lld postings missing\n", missing, (long long)ne * H);
    if (missing) return fail(h, MHAP_E_STATE, "inverted index self-check failed");
  }
  h->inv_ready = true; h->inv_ne = ne; h->iq_start_mid = false;
  return MHAP_OK;
}

// The tail of a search chunk — device records to pinned host memory, id / length / strand conversion (MatchResult.java:46-65), the
// caller's sink — runs on a worker thread while the main thread launches the NEXT chunk's kernels (one chunk ahead, two sets of
// buffers).  On one rank's share of configs[4] (28.8 M records a step) that tail was 1.4 s of a 3.3 s search.  The sink is still
// called by one thread at a time, in chunk order; it is a library thread, as the header says.
struct PostStage {
  mhap_handle* h; const QuerySide* qs; mhap_record_sink sink; void* user;
  std::thread th; std::mutex mu; std::condition_variable cv;
  bool started = false, has_job = false, busy = false, stop = false;
  int slot = 0; unsigned long long nrec = 0;
  int rc = MHAP_OK; std::string err; int64_t matches = 0;

  int process(int sl, unsigned long long n) {
    const size_t bytes = (size_t)n * sizeof(DevRecord);
    if (h->pin_rec_cap[sl] < bytes) {
      if (h->pin_rec[sl]) (void)hipHostFree(h->pin_rec[sl]);
      h->pin_rec[sl] = nullptr; h->pin_rec_cap[sl] = 0;
      const size_t want = bytes + bytes / 4 + 4096;
      if (hipHostMalloc((void**)&h->pin_rec[sl], want, hipHostMallocDefault) != hipSuccess) { err = "cannot allocate pinned host memory"; return MHAP_E_HIP; }
      h->pin_rec_cap[sl] = want;
    }
    const DevRecord* hrecs = (const DevRecord*)h->pin_rec[sl];
    const void* src = sl ? h->recs2.p : h->recs.p;
    if (hipMemcpyAsync((void*)hrecs, src, bytes, hipMemcpyDeviceToHost, h->copy_stream) != hipSuccess || hipStreamSynchronize(h->copy_stream) != hipSuccess) {
      err = "record read-back failed"; return MHAP_E_HIP;
    }
    HPROF("tail: records on the host");
    if (h->out_recs2_cap[sl] < (size_t)n) {
      free(h->out_recs2[sl]);
      h->out_recs2_cap[sl] = (size_t)n + (size_t)n / 4 + 1024;
      h->out_recs2[sl] = (mhap_record*)malloc(h->out_recs2_cap[sl] * sizeof(mhap_record));
      if (!h->out_recs2[sl]) { h->out_recs2_cap[sl] = 0; err = "out of host memory (record buffer)"; return MHAP_E_NOMEM; }
    }
    mhap_record* out = h->out_recs2[sl];
    const QuerySide& q = *qs;
    parallel_for((int64_t)n, host_threads(), [&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; i++) {
        const DevRecord& d = hrecs[(size_t)i];
        mhap_record& r = out[(size_t)i];
        r.from_id = q.h_ids[d.q]; r.to_id = h->ids[(size_t)d.m];
        r.score = d.score; r.raw = (double)d.raw;
        r.alen = q.h_seqlen[d.q]; r.blen = h->seqlen[(size_t)d.m];
        r.a1 = d.a1; r.a2 = d.a2;                                        // from is always a forward entry
        r.to_rc = h->fwd[(size_t)d.m] ? 0 : 1;
        if (r.to_rc) { r.b1 = r.blen - d.b2 - 1; r.b2 = r.blen - d.b1 - 1; }   // MatchResult.java:56-57
        else { r.b1 = d.b1; r.b2 = d.b2; }
        r.pad = 0;
      }
    }, 65536);   // (spawning threads for C2's 41 915 records took longer than converting them)
    matches += (int64_t)n;
    HPROF("tail: records converted");
    if (sink && sink(out, (int64_t)n, user) != 0) { err = "record sink aborted the search"; return MHAP_E_STATE; }
    HPROF("tail: sink returned");
    return MHAP_OK;
  }
  void loop() {
    (void)hipSetDevice(h->device);
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&]() { return has_job || stop; });
      if (!has_job) return;
      has_job = false; busy = true;
      const int sl = slot; const unsigned long long n = nrec;
      lk.unlock();
      const int r = rc == MHAP_OK ? process(sl, n) : rc;
      lk.lock();
      if (rc == MHAP_OK) rc = r;
      busy = false;
      cv.notify_all();
    }
  }
  // wait until the previous chunk's tail is done, then hand this one over (inline = no worker: run it here)
  int submit(int sl, unsigned long long n, bool inline_run) {
    if (inline_run && !started) { if (rc == MHAP_OK) rc = process(sl, n); return rc; }
    std::unique_lock<std::mutex> lk(mu);
    if (!started) { started = true; th = std::thread([this]() { loop(); }); }
    cv.wait(lk, [&]() { return !busy && !has_job; });
    if (rc != MHAP_OK) return rc;
    slot = sl; nrec = n; has_job = true;
    cv.notify_all();
    return MHAP_OK;
  }
  int drain() {
    if (started) {
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return !busy && !has_job; }); stop = true; cv.notify_all(); }
      th.join(); started = false;
    }
    return rc;
  }
  ~PostStage() { (void)drain(); }
};

int search_core(mhap_handle* h, QuerySide& qs, const std::vector<int32_t>& ql_in, bool to_self, bool triangular_ok,
                mhap_record_sink sink, void* user) {
  // candidate generation: GPU inverted index (default) or brute-force all-pairs (MHAP_CANDIDATES=bruteforce)
  const char* cmode = getenv("MHAP_CANDIDATES");
  const bool use_index = !(cmode && strcmp(cmode, "bruteforce") == 0);
  bool prep_done = !qs.prep;
  auto run_prep = [&]() -> int { if (prep_done) return MHAP_OK; prep_done = true; return qs.prep(qs); };
  auto count_valid = [&](int64_t a, int64_t n) { int64_t c = 0; for (int64_t e = a; e < a + n; e++) c += qs.h_valid[(size_t)e] ? 1 : 0; return c; };
  // the list on the device (identity) unless the brute-force tiles want it sorted out on the host, the index is empty, or MHAP_QUERY_LIST_HOST=1 (A/B, tests)
  bool ident = qs.identity;
  std::vector<int32_t> ql_made;
  if (ident && (!use_index || h->n_entries == 0 || getenv("MHAP_QUERY_LIST_HOST"))) {
    const int rp = run_prep();
    if (rp != MHAP_OK) return rp;
    ql_made.reserve((size_t)qs.n_rows);
    for (int64_t e = 0; e < qs.n_rows; e++) if (qs.h_valid[(size_t)e]) ql_made.push_back((int32_t)e);
    ident = false;
  }
  const std::vector<int32_t>& ql = qs.identity && !ident ? ql_made : ql_in;
  const int64_t nql = ident ? qs.n_rows : (int64_t)ql.size();
  if (nql == 0 || h->n_entries == 0) { h->stats.queries_searched += nql; return MHAP_OK; }
  const int S = h->P.ordered_sketch_size;
  SearchParams sp;
  sp.H = h->P.num_hashes; sp.S = S; sp.k2 = h->P.ordered_kmer_size;
  sp.num_min_matches = h->P.num_min_matches; sp.min_store_length = h->P.min_store_length; sp.to_self = to_self ? 1 : 0;
  sp.own_queries = (to_self && qs.d_minhash == h->d_minhash && qs.d_meta == h->d_meta && !getenv("MHAP_COUNT_OWN")) ? 1 : 0;
  sp.max_shift = h->P.max_shift; sp.threshold = h->P.threshold;
  const int ne = (int)h->n_entries;
  int64_t qchunk = 262144;   // queries per candidate/overlap launch pair (bounds the candidate buffer)
  if (const char* e = getenv("MHAP_QUERY_CHUNK")) { long long v = atoll(e); if (v >= CAND_TQ) qchunk = (v / CAND_TQ) * CAND_TQ; }
  HIPCHK(h, h->qlist.ensure((size_t)nql * 4));
  if (ident) { launch_query_iota(h->stream, h->qlist.as<int32_t>(), 0, (int)nql); HIPCHK(h, hipGetLastError()); }
  else HIPCHK(h, hipMemcpyAsync(h->qlist.p, ql.data(), ql.size() * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, h->counters.ensure(256));
  unsigned long long* ctr = h->counters.as<unsigned long long>();
  size_t cand_cap = std::max<size_t>(h->cand.cap / sizeof(Candidate), (size_t)4 << 20);
  // second stage: one lane per candidate; it is latency-bound, so put as many lanes in flight as the candidates allow
  // (per-lane scratch = 3 int arrays of 2S+2 entries: 37 KB at S=1536; 16 workgroups/CU = 19 GB of the 288 GB HBM)
  int64_t ovl_max_blocks = (int64_t)h->num_cus * 16;
  if (const char* e = getenv("MHAP_OVERLAP_BLOCKS")) { long long v = atoll(e); if (v > 0) ovl_max_blocks = v; }
  const int64_t per_lane = 3LL * (2LL * S + 2);
  const char* omode = getenv("MHAP_OVERLAP");
  const bool lane_only = omode && strcmp(omode, "lane") == 0;
  { const char* jw = getenv("MHAP_JOIN_WIDE"); h->join_wide = !jw ? 2 : (jw[0] == '0' ? 0 : (jw[0] == '1' ? 1 : 2)); }
  { const char* jm = getenv("MHAP_JOIN_MODE"); h->join_mode = !jm ? 0 : (strcmp(jm, "alone") == 0 ? 1 : (strcmp(jm, "pair") == 0 ? 2 : (strcmp(jm, "team") == 0 ? 3 : 0))); }
  const int ntu = (ne + CAND_TM - 1) / CAND_TM;
  if (use_index) {
    int rcb = ensure_inverted_index(h);
    if (rcb != MHAP_OK) return rcb;
  }

  h->iq_start_mid = false;   // decided anew by the first chunks of every search
  bool q_ph_done = false;    // the query side's position histograms exist (second stage, early "below the threshold")
  PostStage post; post.h = h; post.qs = &qs; post.sink = sink; post.user = user;
#define SCHK(expr) do { const hipError_t _e = (expr); if (_e != hipSuccess) return leave(fail(h, _e == hipErrorOutOfMemory ? MHAP_E_NOMEM : MHAP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); } while (0)
  // every error return below leaves through here: the worker is drained first (its buffers belong to the handle), the records it has
  // already delivered are credited, and — if the search itself was fine — the worker's own error is the one reported
  auto leave = [&](int code) { const int pr = post.drain(); h->stats.matches_found += post.matches; post.matches = 0;
                               if (code == MHAP_OK && pr != MHAP_OK) return fail(h, pr, post.err); return code; };
  int64_t chunk_no = 0;
  const char* ppe = getenv("MHAP_SEARCH_PIPELINE");   // "0": the tail of every chunk inline (tests / A-B)
  const bool pipeline = !(ppe && ppe[0] == '0');
  int slot = 0;   // the record buffers (device + host) this chunk's kernels and tail use; flips with every tail handed over
  for (int64_t c0 = 0, adv = 0; c0 < nql; c0 += adv, chunk_no++) {
    const int nq = (int)std::min<int64_t>(qchunk, nql - c0);
    adv = nq;
    const int ntq = (nq + CAND_TQ - 1) / CAND_TQ;
    const long long* d_rowstart = nullptr;
    long long nblocks_tri = 0;
    if (triangular_ok && !use_index) {
      std::vector<long long> rs((size_t)ntq + 1);
      long long acc = 0;
      for (int t = 0; t < ntq; t++) {
        rs[(size_t)t] = acc;
        const int32_t maxq = ql[(size_t)(c0 + std::min<int64_t>((int64_t)(t + 1) * CAND_TQ, nq) - 1)];
        acc += std::min<long long>(ntu, ((long long)maxq + CAND_TM - 1) / CAND_TM);
      }
      rs[(size_t)ntq] = acc; nblocks_tri = acc;
      SCHK(h->rowstart.ensure(rs.size() * 8));
      SCHK(hipMemcpyAsync(h->rowstart.p, rs.data(), rs.size() * 8, hipMemcpyHostToDevice, h->stream));
      SCHK(hipStreamSynchronize(h->stream));  // rs is a stack vector
      d_rowstart = h->rowstart.as<long long>();
      if (nblocks_tri == 0) { h->stats.queries_searched += nq; continue; }   // (never the identity list: that needs the index)
    }
    unsigned long long ncand = 0;
    for (;;) {
      SCHK(h->cand.ensure(cand_cap * sizeof(Candidate)));
      SCHK(hipMemsetAsync(ctr, 0, 160, h->stream));
      if (use_index) {
        SCHK(h->inv_big.ensure((size_t)nq * 8));   // two lists: handed on by the first tier / by the middle tier
        // (the spread words of the "elements processed" count: zeroed once, every launch folds them into ctr[4] and leaves them zero)
        if (!h->iq_espread.p) { SCHK(h->iq_espread.ensure((size_t)index_elements_spread_words() * 8)); SCHK(hipMemsetAsync(h->iq_espread.p, 0, (size_t)index_elements_spread_words() * 8, h->stream)); }
        unsigned long long* espread = h->iq_espread.as<unsigned long long>();
        const char* tv = getenv("MHAP_INDEX_TIERS");   // "1": first tier only (large hit sets are split right away; tests)
        const bool tiers = index_query_tiers() && !(tv && tv[0] == '1');
        // (an index or a numMinMatches the first tier's packed hit-count words cannot hold: every query takes the dense tier)
        const char* dv = getenv("MHAP_INDEX_DENSE");    // "1": every query takes the dense tier (tests)
        const bool first_ok = index_query_tier_ok(0, h->n_entries, sp.num_min_matches) && !(dv && dv[0] == '1');
        int32_t* listA = h->inv_big.as<int32_t>();
        int32_t* listB = listA + nq;
        const char* midv = getenv("MHAP_INDEX_MID");   // "1" / "0": with / without the middle tier whatever the index size (tests)
        const bool use_mid = tiers && first_ok && (midv ? midv[0] == '1' : index_query_dense_ranges(h->n_entries) > 4) &&
                             index_query_tier_ok(1, h->n_entries, sp.num_min_matches);
        // Tiers (search_kernels.hip).  Queries whose hits outgrow the first tier's table: a small index (a few dense ranges) counts them
        // with dense counters at once; a large one in the middle tier's 8192-entry table first — ordinary reads of a big data set have
        // thousands of hits, and the dense tier would make a pass per 32 768 stored entries for each of them — and dense counters for
        // what outgrows that too (repeats).  When more than half of a chunk's queries left the first tier for the middle one (all of
        // C4), the next chunks START there: the first tier's look at the bucket lengths was their only use of it.
        const int t0 = !first_ok ? 2 : (use_mid && h->iq_start_mid ? 1 : 0);
        time_begin(h, MHAP_K_INDEX_QUERY);
        launch_index_query(h->stream, h->inv, qs.d_minhash, qs.mh_stride, h->qlist.as<int32_t>() + c0, nq,
                           h->d_ids.as<int64_t>(), qs.d_ids, h->d_meta, qs.d_meta, sp, h->cand.as<Candidate>(), ctr + 0,
                           (unsigned long long)cand_cap, ctr + 3, ctr + 4, tiers && t0 < 2 ? listA : nullptr, ctr + 6, t0, espread);
        time_end(h);
        SCHK(hipGetLastError());
        unsigned long long c5[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        SCHK(hipMemcpyAsync(c5, ctr, 72, hipMemcpyDeviceToHost, h->stream));
        if (!prep_done) {   // (the rows' meta words, under the first tier's time)
          const int rp = run_prep();
          if (rp != MHAP_OK) { (void)sync_stream(h); return leave(rp); }
          HPROF("query meta on the host");
        }
        int rc = sync_stream(h);
        if (rc != MHAP_OK) return leave(rc);
        HPROF("first query tier done");
        const int32_t* dense_list = listA;
        unsigned long long n_dense = c5[6];
        if (t0 == 0 && use_mid && c5[0] <= cand_cap) {
          // (a chunk of a few thousand queries at least: one repeat-rich read of a small -q batch must not send every later
          //  query of this search through the 256-lane tier, which costs ordinary queries about twice the first tier's time)
          if (nq >= 4096 && 2 * c5[6] > (unsigned long long)nq) h->iq_start_mid = true;
          if (c5[6] > 0) {
            time_begin(h, MHAP_K_INDEX_QUERY);
            launch_index_query(h->stream, h->inv, qs.d_minhash, qs.mh_stride, listA, (int)c5[6], h->d_ids.as<int64_t>(), qs.d_ids, h->d_meta, qs.d_meta, sp,
                               h->cand.as<Candidate>(), ctr + 0, (unsigned long long)cand_cap, ctr + 3, ctr + 4, listB, ctr + 8, 1, espread);
            time_end(h);
            SCHK(hipGetLastError());
            SCHK(hipMemcpyAsync(c5, ctr, 72, hipMemcpyDeviceToHost, h->stream));
            rc = sync_stream(h);
            if (rc != MHAP_OK) return leave(rc);
            dense_list = listB; n_dense = c5[8];
          }
        }
        if (n_dense > 0 && c5[0] <= cand_cap) {
          time_begin(h, MHAP_K_INDEX_QUERY);
          launch_index_query(h->stream, h->inv, qs.d_minhash, qs.mh_stride, dense_list, (int)n_dense,
                             h->d_ids.as<int64_t>(), qs.d_ids, h->d_meta, qs.d_meta, sp, h->cand.as<Candidate>(), ctr + 0,
                             (unsigned long long)cand_cap, ctr + 3, ctr + 4, nullptr, nullptr, 2, espread);
          time_end(h);
          SCHK(hipGetLastError());
          SCHK(hipMemcpyAsync(c5, ctr, 72, hipMemcpyDeviceToHost, h->stream));
          rc = sync_stream(h);
          if (rc != MHAP_OK) return leave(rc);
        }
        HPROF("candidates known");
        ncand = c5[0];
        if (ncand <= cand_cap) { h->stats.table_elements += (int64_t)c5[4]; h->stats.index_splits += (int64_t)c5[3]; }
      } else {
        time_begin(h, MHAP_K_CANDIDATE);
        launch_candidates(h->stream, h->d_minhash, h->Hrow, qs.d_minhash, qs.mh_stride, h->qlist.as<int32_t>() + c0, nq, ne,
                          h->d_ids.as<int64_t>(), qs.d_ids, h->d_meta, qs.d_meta, sp, d_rowstart, nblocks_tri, h->cand.as<Candidate>(),
                          ctr + 0, (unsigned long long)cand_cap);
        time_end(h);
        SCHK(hipGetLastError());
        SCHK(hipMemcpyAsync(&ncand, ctr + 0, 8, hipMemcpyDeviceToHost, h->stream));
        int rc = sync_stream(h);
        if (rc != MHAP_OK) return leave(rc);
        if (ncand <= cand_cap) {
          const long long tiles = d_rowstart ? nblocks_tri : (long long)ntq * ntu;
          h->stats.slot_compares += tiles * (long long)CAND_TQ * CAND_TM * sp.H;
        }
      }
      if (ncand <= cand_cap) break;
      cand_cap = (size_t)ncand + (size_t)(ncand / 4) + 1024;   // overflow: grow and redo this chunk
    }
    h->stats.queries_searched += ident ? count_valid(c0, nq) : (int64_t)nq;
    if (ncand == 0) continue;
    // a candidate-rich search (repeats): smaller chunks from here on — smaller candidate / record buffers, and the tail of a chunk
    // (read-back, conversion, sink: PostStage) hides behind the next chunk's kernels, so less of it is left over at the end
    if (chunk_no == 0 && (int64_t)ncand >= 64LL * nq && !getenv("MHAP_QUERY_CHUNK")) qchunk = 65536;
    DevBuf& recbuf = slot ? h->recs2 : h->recs;
    SCHK(recbuf.ensure((size_t)ncand * sizeof(DevRecord)));
    // second stage: one wavefront per candidate from the equal-hash join (MHAP_OVERLAP=lane: the literal per-lane merge for
    // every pair); pairs the join cannot decide exactly come back in slow_cand and take the per-lane merge
    if (h->gate && h->gate(h->gate_user) != 0) return leave(fail(h, MHAP_E_STATE, "second-stage gate aborted the search"));
    // the join kernel's three shapes (search_kernels.hip): every wave alone, pairs of waves or teams of four sharing a staged query and
    // its filter — by the candidates per query.  MHAP_JOIN_MODE=alone|pair|team pins one.
    bool fits[3];
    for (int i = 0; i < 3; i++) fits[i] = overlap_join_lds_bytes(S, i) <= 64 * 1024;
    const bool use_join = !lane_only && S <= OJ_MAX_S && (fits[0] || fits[1] || fits[2]);
    unsigned long long nslow = use_join ? 0 : ncand;
    const Candidate* slow_list = nullptr; unsigned long long* slow_count_ptr = nullptr;   // (set when the wide second pass ran)
    unsigned long long cj[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool have_counts = false;
    if (use_join) {
      SCHK(h->slow_cand.ensure((size_t)ncand * sizeof(Candidate)));
      // (round 3, every entry looked up — C2, 4.5 candidates per query: alone 4.94, pair 4.83, team 5.03 ms; C5 slice, 79: alone 93, pair 87,
      //  team 77; a rank of eight, 0.56 per query: alone 0.86, pair 1.24.  Round 4, the shared shapes with the filter — C2: pair 3.74, team 3.65;
      //  C5 slice: pair 66.8, team 49.8; ranks of two / four / eight: alone 2.29 / 1.19 / 0.69, pair 2.09 / 1.29 / 0.83, team 2.29 / 1.69 / 1.33)
      int shape = (int64_t)ncand >= 4LL * nq ? 2 : ((int64_t)ncand >= 2LL * nq ? 1 : 0);
      if (h->join_mode > 0) shape = h->join_mode - 1;
      while (!fits[shape]) shape = (shape + 1) % 3;
      if (h->oj_per_cu_S != S) {
        for (int i = 0; i < 3; i++) h->oj_per_cu[i] = fits[i] ? overlap_join_blocks_per_cu(S, i) : 0;
        h->oj_per_cu_S = S;
      }
      // candidates per pull of a wave: 8 amortise the counter and keep a query's hashes staged across its candidates — unless the
      // candidates are few (a small batch of -q reads, one rank's share of a small job): then every resident wave should get some
      const int per_cu = h->oj_per_cu[shape], wpb = overlap_join_waves_per_block(shape);
      const int64_t resident_waves = (int64_t)h->num_cus * per_cu * wpb;
      const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(8, (int64_t)ncand / std::max<int64_t>(resident_waves, 1)));
      const int64_t want = ((int64_t)ncand + (int64_t)wpb * chunk - 1) / ((int64_t)wpb * chunk);
      const int jblocks = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)h->num_cus * per_cu, want));
      // Early "below the threshold" from position histograms (search_kernels.hip, poshist_kernel): pays when the joined k-mers of a pair
      // are few next to its windows — repeat-induced candidates, i.e. many candidates per query (sixteen and more); at C2 (40
      // joined k-mers per pair) it rejects nothing and the histograms would cost a pass over the ordered table.  MHAP_OVERLAP_PRUNE=0|1.
      const char* pe = getenv("MHAP_OVERLAP_PRUNE");
      const bool prune = pe ? pe[0] == '1' : (int64_t)ncand >= 16LL * nq;
      const uint16_t *ph = nullptr, *qph = nullptr;
      if (prune) {
        time_begin(h, MHAP_K_OVERLAP);
        if (!(h->ph_ready && h->ph_ne == h->n_entries)) {
          SCHK(h->poshist.ensure((size_t)h->n_entries * POSHIST_BINS * 2));
          launch_poshist(h->stream, h->d_ordered, 2LL * S, h->d_meta, h->n_entries, h->poshist.as<uint16_t>());
          h->ph_ready = true; h->ph_ne = h->n_entries;
        }
        ph = h->poshist.as<uint16_t>();
        if (qs.d_ordered == h->d_ordered && qs.d_meta == h->d_meta) qph = ph;
        else {
          if (!q_ph_done) {      // (after the gate: the query rows of a sharded search have all arrived)
            SCHK(h->q_poshist.ensure((size_t)qs.n_rows * POSHIST_BINS * 2));
            launch_poshist(h->stream, qs.d_ordered, qs.ord_stride, qs.d_meta, qs.n_rows, h->q_poshist.as<uint16_t>());
            q_ph_done = true;
          }
          qph = h->q_poshist.as<uint16_t>();
        }
        time_end(h);
        SCHK(hipGetLastError());
      }
      time_begin(h, MHAP_K_OVERLAP);
      launch_overlap_join(h->stream, shape, jblocks, chunk, h->cand.as<Candidate>(), ctr + 0, (unsigned long long)cand_cap, h->d_ordered, 2LL * S, h->d_meta,
                          qs.d_ordered, qs.ord_stride, qs.d_meta, sp, h->score_tbl.as<double>(), recbuf.as<DevRecord>(), ctr + 1,
                          (unsigned long long)ncand, ctr + 2, h->slow_cand.as<Candidate>(), ctr + 5, ctr + 7, ph, qph, h->pass_min_tbl.as<int32_t>());
      time_end(h);
      SCHK(hipGetLastError());
      // (one read-back for the pairs handed over and for the counts the tail needs: nothing else changes them when none were)
      SCHK(hipMemcpyAsync(cj, ctr, 160, hipMemcpyDeviceToHost, h->stream));
      int rcj = sync_stream(h);
      if (rcj != MHAP_OK) return leave(rcj);
      nslow = cj[5];
      have_counts = nslow == 0;
      // Further passes: the pairs the join kernel handed over — nearly all of them for MORE THAN 128 JOINED K-MERS, i.e. true overlaps of reads
      // better than the 15 %-error ones MHAP was built for — go through the same kernel compiled with room for 512, and what that hands
      // over through the one with room for 1 536 (search_kernels_wide.hip / _wide2.hip), every wave alone; only what is left then (the
      // duplicated-hash group caps) takes the per-lane merge.  MHAP_JOIN_WIDE=0 switches the passes off, =1 keeps the first of them only.
      unsigned long long group_bad = cj[13];   // of the pairs handed over: for the duplicated-hash group caps (ctr + 13 = the first pass's slow_count + 8)
      for (int level = 0; level < h->join_wide && nslow > 0; level++) {
        // (pairs handed over for their duplicated-hash groups — repeat-rich reads of low error without a -f filter: hundreds of millions of
        //  them — would be handed on by a wider pass too: 1.4 s of 9.6 wasted on 20 000 such reads)
        if (group_bad * 10 > nslow * 9) break;
        const size_t wl = level == 0 ? overlap_join_wide_lds_bytes(S, 0) : overlap_join_wide2_lds_bytes(S, 0);
        if (wl > 64 * 1024) break;
        DevBuf& in = level == 0 ? h->slow_cand : h->slow_cand2;
        DevBuf& outb = level == 0 ? h->slow_cand2 : h->slow_cand3;
        unsigned long long* in_count = level == 0 ? ctr + 5 : ctr + 10;
        unsigned long long* out_count = level == 0 ? ctr + 10 : ctr + 11;
        unsigned long long* work = level == 0 ? ctr + 9 : ctr + 12;
        SCHK(outb.ensure((size_t)nslow * sizeof(Candidate)));
        if (h->ojw_per_cu_S[level] != S) {
          h->ojw_per_cu[level] = level == 0 ? overlap_join_wide_blocks_per_cu(S, 0) : overlap_join_wide2_blocks_per_cu(S, 0);
          h->ojw_per_cu_S[level] = S;
        }
        const int wpbw = level == 0 ? overlap_join_wide_waves_per_block(0) : overlap_join_wide2_waves_per_block(0);
        const int64_t residentw = (int64_t)h->num_cus * h->ojw_per_cu[level] * wpbw;
        const int chunkw = (int)std::max<int64_t>(1, std::min<int64_t>(8, (int64_t)nslow / std::max<int64_t>(residentw, 1)));
        const int64_t wantw = ((int64_t)nslow + (int64_t)wpbw * chunkw - 1) / ((int64_t)wpbw * chunkw);
        const int wblocks = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)h->num_cus * h->ojw_per_cu[level], wantw));
        time_begin(h, MHAP_K_OVERLAP);
        (level == 0 ? launch_overlap_join_wide : launch_overlap_join_wide2)(
            h->stream, 0, wblocks, chunkw, in.as<Candidate>(), in_count, (unsigned long long)ncand, h->d_ordered, 2LL * S, h->d_meta, qs.d_ordered, qs.ord_stride,
            qs.d_meta, &sp, h->score_tbl.as<double>(), recbuf.as<DevRecord>(), ctr + 1, (unsigned long long)ncand, ctr + 2, outb.as<Candidate>(), out_count, work,
            nullptr, nullptr, h->pass_min_tbl.as<int32_t>());
        time_end(h);
        SCHK(hipGetLastError());
        unsigned long long cw[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        SCHK(hipMemcpyAsync(cw, ctr, 160, hipMemcpyDeviceToHost, h->stream));
        const int rcw = sync_stream(h);
        if (rcw != MHAP_OK) return leave(rcw);
        for (int i = 0; i < 8; i++) cj[i] = cw[i];
        const unsigned long long before = nslow;
        nslow = level == 0 ? cw[10] : cw[11];
        group_bad = level == 0 ? cw[18] : cw[19];
        slow_list = outb.as<Candidate>(); slow_count_ptr = out_count;
        have_counts = nslow == 0;
        if (nslow * 10 > before * 9) break;   // (handed over for the duplicated-hash group caps, not for their number of k-mers: a wider pass would hand them on again)
      }
      h->stats.slow_pairs += (int64_t)nslow;
    }
    if (nslow > 0) {
      // (few pairs — what the join kernel hands over — are spread over more wavefronts while there is at most one per SIMD: 600 pairs of a
      //  c5rank chunk, one per wave, 9 -> 3.5 ms; the 76 000 of the C5 slice at sixteen per wave instead of 64: 3 ms SLOWER — four times
      //  the instructions on a machine that is then full: search_kernels.hip, overlap_kernel)
      int spread = 1;
      while (spread < 64 && (int64_t)nslow * spread * 2 <= (int64_t)h->num_cus * 4 * 64) spread *= 2;
      const int oblocks = (int)std::max<int64_t>(1, std::min<int64_t>(ovl_max_blocks, ((int64_t)nslow * spread + OVL_THREADS - 1) / OVL_THREADS));
      SCHK(h->ovl_scratch.ensure((size_t)oblocks * OVL_THREADS / (size_t)spread * (size_t)per_lane * 4));
      time_begin(h, MHAP_K_OVERLAP);
      launch_overlap(h->stream, oblocks, slow_list ? slow_list : (use_join ? h->slow_cand.as<Candidate>() : h->cand.as<Candidate>()), slow_list ? slow_count_ptr : (use_join ? ctr + 5 : ctr + 0),
                     use_join ? (unsigned long long)ncand : (unsigned long long)cand_cap, h->d_ordered, 2LL * S, h->d_meta,
                     qs.d_ordered, qs.ord_stride, qs.d_meta, sp, h->score_tbl.as<double>(), h->ovl_scratch.as<int32_t>(), per_lane,
                     recbuf.as<DevRecord>(), ctr + 1, (unsigned long long)ncand, ctr + 2, spread);
      time_end(h);
      SCHK(hipGetLastError());
    }
    unsigned long long counts[3] = {cj[0], cj[1], cj[2]};
    if (!have_counts) {
      SCHK(hipMemcpyAsync(counts, ctr, 24, hipMemcpyDeviceToHost, h->stream));
      int rc = sync_stream(h);
      if (rc != MHAP_OK) return leave(rc);
    }
    HPROF("overlap done");
    oj_stats_dump();
    const unsigned long long nrec = counts[1];
    h->stats.candidates_compared += (int64_t)counts[2];
    if (nrec == 0) continue;
    // the chunk's tail: inline when it is the last chunk with no worker running (nothing left to hide it behind), else on the worker
    const bool last = c0 + nq >= nql;
    const int rp = post.submit(slot, nrec, !pipeline || last);
    if (rp != MHAP_OK) return leave(fail(h, rp, post.err));
    slot ^= 1;
    HPROF("chunk tail handed over");
  }
  return leave(MHAP_OK);
#undef SCHK
}

}  // namespace

namespace mhap {
HandleView handle_view(mhap_handle* h) {
  HandleView v;
  v.device = h->device; v.stream = h->stream; v.Hrow = h->Hrow; v.S = h->P.ordered_sketch_size; v.k = h->P.kmer_size; v.min_olap_length = h->P.min_olap_length;
  v.n_entries = h->n_entries; v.index_gen = h->index_gen;
  v.d_minhash = h->d_minhash; v.d_ordered = h->d_ordered; v.d_meta = h->d_meta;
  v.h_ids = h->ids.data(); v.h_fwd = h->fwd.data(); v.err = &h->err; v.dist = &h->dist;
  return v;
}
int internal_stage_packed(mhap_handle* h, const ReadDesc* descs, const int64_t* ids, int64_t n, const void* packed, size_t bytes) {
  (void)hipSetDevice(h->device);
  h->st_descs.assign(descs, descs + n);
  h->st_ids.assign(ids, ids + n);
  h->st_n = n;
  const size_t need = std::max<size_t>(bytes, 4);
  HIPCHK(h, h->store.ensure(need));
  if (bytes) HIPCHK(h, hipMemcpyAsync(h->store.p, packed, bytes, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));   // the caller reuses its staging buffer
  h->st_bytes = (int64_t)need;
  return MHAP_OK;
}

// the sharded search's call of mhap_find_matches_device: the gathered ids are in device memory already
int internal_find_matches_device(mhap_handle* h, const void* d_q_minhash, const void* d_q_ordered, const void* d_q_meta, const int64_t* ids,
                                 const int64_t* d_ids_dev, int64_t m, int to_self, mhap_record_sink sink, void* user);

// -q mode of the sharded search: sketch n query reads (forward strands only, AbstractMatchSearch.java:225,236) into caller tables
// laid out like an index's (row 2i = read i's forward strand)
int internal_sketch_queries(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, int64_t n, void* d_mh, void* d_od, void* d_mt) {
  for (int64_t i = 0; i < n; i++) if (lengths[i] < 0) return fail(h, MHAP_E_INVALID, "negative read length");
  return sketch_into(h, bases, offsets, lengths, n, true, (int32_t*)d_mh, h->Hrow, (int32_t*)d_od, 2LL * h->P.ordered_sketch_size, (int32_t*)d_mt);
}
}  // namespace mhap

// =================================================================================================
extern "C" {

void mhap_default_params(mhap_params* p) {
  p->kmer_size = 16; p->num_hashes = 512; p->ordered_kmer_size = 12; p->ordered_sketch_size = 1536;
  p->num_min_matches = 3; p->min_store_length = 0; p->min_olap_length = 116; p->device = -1;
  p->threshold = 0.78; p->max_shift = 0.2; p->repeat_weight = 0.9;
}

int mhap_create(const mhap_params* params, mhap_handle** out, char* err, size_t errcap) {
  auto seterr = [&](const std::string& m) { if (err && errcap) snprintf(err, errcap, "%s", m.c_str()); };
  if (!params || !out) { seterr("null argument"); return MHAP_E_INVALID; }
  const mhap_params& P = *params;
  if (P.kmer_size < 1 || P.kmer_size > 255) { seterr("k-mer size must be in [1,255]"); return MHAP_E_INVALID; }
  if (P.ordered_kmer_size < 1 || P.ordered_kmer_size > 255) { seterr("ordered k-mer size must be in [1,255]"); return MHAP_E_INVALID; }
  if (P.num_hashes < 1 || P.num_hashes > MHAP_MAX_NUM_HASHES) { seterr("num-hashes must be in [1,8192]"); return MHAP_E_INVALID; }
  if (P.ordered_sketch_size < 1 || P.ordered_sketch_size > 8192) { seterr("ordered-sketch-size must be in [1,8192]"); return MHAP_E_INVALID; }
  if (P.num_min_matches < 1) { seterr("Minimum number of matches must be positive."); return MHAP_E_INVALID; }
  if (P.min_store_length < 0) { seterr("The minimum read length stored must be >=0."); return MHAP_E_INVALID; }
  if (P.max_shift < -1.0) { seterr("The minimum shift must be greater than -1."); return MHAP_E_INVALID; }
  if (P.threshold < 0.0 || P.threshold > 1.0) { seterr("The second stage filter threshold must be 0<=threshold<=1.0."); return MHAP_E_INVALID; }
  HPROF("create begin");
  // The handle's constant tables are functions of the flags alone (S, k2, threshold, H): they are computed on host threads WHILE the HIP
  // runtime and the device come up (0.18 s in a fresh process, the larger part of mhap_create), and uploaded afterwards.
  struct HostTables { std::vector<double> score; std::vector<int32_t> pass_min; std::vector<uint64_t> jump, jump_w1, unjump, luts; int na = 0; } ht;
  std::thread th_score([&]() {
    fill_score_table(P.ordered_sketch_size, P.ordered_kmer_size, ht.score);
    build_pass_min(ht.score, P.ordered_sketch_size, P.threshold, ht.pass_min);
  });
  std::thread th_jump([&]() {
    ht.na = ((P.num_hashes + 1) >> XS_JUMP_LOG2) + 1;
    ht.jump.resize((size_t)(ht.na + XS_JUMP_NQ) * 2048);
    build_xorshift_jump_tables(ht.na, XS_JUMP_NQ, ht.jump.data());
  });
  std::thread th_jump2([&]() {
    const int nt = w1_jump_tables(P.num_hashes);
    ht.jump_w1.resize((size_t)nt * 2048); ht.unjump.resize((size_t)nt * 2048); ht.luts.resize(768);
    build_xorshift_jump_tables(W1_JUMP_NA, nt - W1_JUMP_NA, ht.jump_w1.data());
    build_xorshift_unjump_tables(W1_JUMP_NA, nt - W1_JUMP_NA, ht.unjump.data());
    build_kmer_hash_luts(ht.luts.data());
  });
  struct Join { std::thread& a; std::thread& b; std::thread& c; ~Join() { if (a.joinable()) a.join(); if (b.joinable()) b.join(); if (c.joinable()) c.join(); } } join_tables{th_score, th_jump, th_jump2};
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) { seterr(std::string("no HIP device available: ") + hipGetErrorString(e)); return MHAP_E_HIP; }
  mhap_handle* h = new mhap_handle();
  h->P = P;
  int dev = P.device;
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
  if (dev >= ndev) { seterr("device ordinal out of range"); delete h; return MHAP_E_INVALID; }
  h->device = dev;
  if ((e = hipSetDevice(dev)) != hipSuccess) { seterr(hipGetErrorString(e)); delete h; return MHAP_E_HIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess) h->num_cus = std::max(1, prop.multiProcessorCount);
  if ((e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess) { seterr(hipGetErrorString(e)); delete h; return MHAP_E_HIP; }
  h->stream = h->own_stream;
  if (hipStreamCreateWithFlags(&h->mh_stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_mh_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_mh_join, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_ix_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_ix_join, hipEventDisableTiming) != hipSuccess) { seterr("cannot create the side streams"); mhap_destroy(h); return MHAP_E_HIP; }
  h->Hrow = std::max(1, P.num_hashes);
  int cap = 1; while (cap < P.ordered_sketch_size) cap <<= 1;
  h->ord_cap = cap;
  h->ft = FilterTable{nullptr, nullptr, 0, 0, 0, 0, 3.0};
  HPROF("create: device + streams");
  th_score.join(); th_jump.join(); th_jump2.join();
  HPROF("create: host tables joined");
  int rc = upload_score_table(h, ht.score, ht.pass_min);
  if (rc != MHAP_OK) { seterr(h->err); mhap_destroy(h); return rc; }
  HPROF("create: score table");
  // xorshift jump-ahead tables for slots up to H (the general MinHash kernel's deferred-candidate drain); the weight-1 kernel's own small
  // two-level set (its drains look tables up a thousand times per strand: they must stay in L2) and their inverses (a slot's minimal
  // chain value back into the winning key); the block-mix tables of the k = 16 / k2 = 12 hash path
  h->jump_na = ht.na;
  struct Up { DevBuf* b; const std::vector<uint64_t>* v; } ups[] = {{&h->jump_tbl, &ht.jump}, {&h->jump_w1_tbl, &ht.jump_w1}, {&h->unjump_tbl, &ht.unjump}, {&h->hash_luts, &ht.luts}};
  for (const Up& u : ups)
    if (u.b->ensure(u.v->size() * 8) != hipSuccess || hipMemcpy(u.b->p, u.v->data(), u.v->size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
      seterr("cannot allocate the jump / hash tables"); mhap_destroy(h); return MHAP_E_HIP;
    }
  HPROF("create: jump tables");
  *out = h;
  return MHAP_OK;
}

void mhap_destroy(mhap_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->dist) { mhap_dist_release(h->dist); h->dist = nullptr; }
  for (auto& t : h->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (auto& p : h->free_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  DevBuf* bufs[] = {&h->f_keys, &h->f_vals, &h->f_bloom, &h->score_tbl, &h->jump_tbl, &h->hash_luts, &h->own_minhash, &h->own_ordered, &h->own_meta, &h->d_ids, &h->store, &h->descs,
                    &h->keys, &h->wts, &h->perm, &h->h32, &h->slist, &h->info, &h->slabs, &h->counters, &h->order, &h->mhq, &h->mhmerge, &h->unjump_tbl, &h->jump_w1_tbl, &h->q_minhash, &h->q_ordered, &h->q_meta, &h->q_ids,
                    &h->qlist, &h->rowstart, &h->cand, &h->slow_cand, &h->slow_cand2, &h->slow_cand3, &h->recs, &h->recs2, &h->ovl_scratch, &h->inv_ends, &h->inv_items, &h->inv_staged, &h->inv_scratch, &h->inv_big, &h->inv_lines, &h->iq_espread,
                    &h->pass_min_tbl, &h->poshist, &h->q_poshist};
  for (DevBuf* b : bufs) b->release();
  if (h->pin_store) (void)hipHostFree(h->pin_store);
  if (h->pin_io) (void)hipHostFree(h->pin_io);
  if (h->h_descs) (void)hipHostFree(h->h_descs);
  for (int i = 0; i < 2; i++) { if (h->pin_rec[i]) (void)hipHostFree(h->pin_rec[i]); free(h->out_recs2[i]); }
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  if (h->ev_mh_fork) (void)hipEventDestroy(h->ev_mh_fork);
  if (h->ev_mh_join) (void)hipEventDestroy(h->ev_mh_join);
  if (h->ev_ix_fork) (void)hipEventDestroy(h->ev_ix_fork);
  if (h->ev_ix_join) (void)hipEventDestroy(h->ev_ix_join);
  if (h->mh_stream) (void)hipStreamDestroy(h->mh_stream);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

const char* mhap_last_error(const mhap_handle* h) { return h ? h->err.c_str() : "null handle"; }

int mhap_set_filter(mhap_handle* h, const int64_t* hashes, const double* fractions, int64_t n, double filter_cutoff, double offset,
                    double range, int no_tf) {
  if (!h) return MHAP_E_INVALID;
  (void)hipSetDevice(h->device);
  // n == 0 with a NULL table clears the filter; n == 0 with a non-NULL pointer is a filter file that yielded no entry (header only, or
  // k-mer-only lines): the reference still has kmerFilter != null then, and every k-mer gets idf = range (FrequencyCounts.java:295-300)
  if (n <= 0 && !hashes) { h->ft = FilterTable{nullptr, nullptr, 0, 0, 0, 0, range}; return MHAP_OK; }
  if (offset < 0.0 || offset >= 1.0) return fail(h, MHAP_E_INVALID, "Offset can only be between 0 and 1.0.");   // FrequencyCounts.java:74-75
  // FrequencyCounts.java:176-184 keep fraction >= cutoff; maxValue = max kept fraction
  std::vector<int64_t> kk; std::vector<double> ff;
  double maxv = -INFINITY;
  for (int64_t i = 0; i < n; i++) if (fractions[i] >= filter_cutoff) { kk.push_back(hashes[i]); ff.push_back(fractions[i]); maxv = std::max(maxv, fractions[i]); }
  const double minv = filter_cutoff;
  auto idf = [&](double f) { return std::log(maxv / f - offset); };   // :250-254
  const double minIdf = idf(maxv), maxIdf = idf(minv);                // :226-227
  uint32_t ts = 16; while (ts < 2 * kk.size() + 2) ts <<= 1;
  std::vector<int64_t> tk(ts, 0); std::vector<double> tv(ts, 0.0);
  for (size_t i = 0; i < kk.size(); i++) {
    double id = idf(ff[i]);
    double scale = (maxIdf - minIdf) / (double)(range - 1.0);
    double v = 1.0 + (id - minIdf) / scale;                            // scaledIdf :290-311
    if (v == 0.0) v = 4.9406564584124654e-324;                         // keep 0.0 reserved for "empty"
    uint32_t slot = (uint32_t)fmix64((uint64_t)kk[i]) & (ts - 1);
    for (;;) {
      if (tv[slot] == 0.0) { tk[slot] = kk[i]; tv[slot] = v; break; }
      if (tk[slot] == kk[i]) { tv[slot] = v; break; }                  // duplicate k-mer line: last writer wins (:181-184)
      slot = (slot + 1) & (ts - 1);
    }
  }
  HIPCHK(h, h->f_keys.ensure((size_t)ts * 8));
  HIPCHK(h, h->f_vals.ensure((size_t)ts * 8));
  HIPCHK(h, hipMemcpy(h->f_keys.p, tk.data(), (size_t)ts * 8, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->f_vals.p, tv.data(), (size_t)ts * 8, hipMemcpyHostToDevice));
  h->ft.keys = h->f_keys.as<int64_t>(); h->ft.vals = h->f_vals.as<double>();
  h->ft.mask = ts - 1; h->ft.size = (uint32_t)kk.size(); h->ft.enabled = 1; h->ft.no_tf = no_tf ? 1 : 0; h->ft.range = range;
  h->ft.bloom = nullptr; h->ft.bloom_bits = 0; h->ft.bloom_k = 0; h->ft.bloom_mode = 0;   // a new filter has no whitelist until one is set
  return MHAP_OK;
}

int mhap_set_filter_whitelist(mhap_handle* h, const int64_t* hashes, int64_t n, int64_t size_bloom, int32_t mode) {
  if (!h) return MHAP_E_INVALID;
  if (mode < 0 || mode > 2) return fail(h, MHAP_E_INVALID, "Unknown removeUnique option.");   // FrequencyCounts.java:71-72
  if (mode == 0) { h->ft.bloom = nullptr; h->ft.bloom_bits = 0; h->ft.bloom_k = 0; h->ft.bloom_mode = 0; return MHAP_OK; }
  if (!h->ft.enabled) return fail(h, MHAP_E_STATE, "set the k-mer filter (mhap_set_filter) before its whitelist");
  if (n < 0 || (n > 0 && !hashes) || size_bloom < 0) return fail(h, MHAP_E_INVALID, "bad whitelist arguments");
  (void)hipSetDevice(h->device);
  std::vector<unsigned long long> words;
  uint64_t bit_size = 0; int k = 0;
  build_bloom(hashes, n, size_bloom, words, bit_size, k);
  const size_t nwords = words.size();
  HIPCHK(h, h->f_bloom.ensure(nwords * 8));
  HIPCHK(h, hipMemcpy(h->f_bloom.p, words.data(), nwords * 8, hipMemcpyHostToDevice));
  h->ft.bloom = h->f_bloom.as<unsigned long long>(); h->ft.bloom_bits = bit_size; h->ft.bloom_k = k; h->ft.bloom_mode = mode;
  return MHAP_OK;
}

static int finish_add(mhap_handle* h, int64_t first, const int64_t* ids, int64_t n);

// With the eager exchange on (mhap_dist_set_eager) an add is a COLLECTIVE call: its rendezvous (dist_eager_begin, inside sketch_staged)
// is where the ranks agree whether this add gathers its rows.  A rank that leaves the add before it gets there — nothing to add, a bad
// argument, no memory for the tables — must still take part, saying "not this time": otherwise the other ranks wait in the rendezvous
// until the watchdog's time-out (MHAP_DIST_TIMEOUT_S).  Returns `code`, or the rendezvous' own error if it failed.
static int leave_collective_add(mhap_handle* h, int code) {
  if (h->dist != nullptr && dist_eager_wanted(h)) {
    const std::string keep = h->err;
    const int rx = dist_eager_begin(h, 0, nullptr, false);
    if (code != MHAP_OK) h->err = keep;                 // (the caller's reason, not the rendezvous')
    else if (rx < 0) return rx;
  }
  return code;
}

int mhap_index_add_reads(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, const int64_t* ids, int64_t n) {
  if (!h) return MHAP_E_INVALID;
  (void)hipSetDevice(h->device);
  if (n > 0 && (!bases || !offsets || !lengths || !ids)) return leave_collective_add(h, fail(h, MHAP_E_INVALID, "null argument"));
  if (n <= 0) return leave_collective_add(h, MHAP_OK);
  if (h->n_entries + 2 * n > (int64_t)INT32_MAX / 2) return leave_collective_add(h, fail(h, MHAP_E_INVALID, "index too large for 32-bit entry indices"));
  for (int64_t i = 0; i < n; i++) if (lengths[i] < 0) return leave_collective_add(h, fail(h, MHAP_E_INVALID, "negative read length"));
  // = mhap_stage_reads + mhap_index_add_staged (a fresh index is filled while its reads are being sketched), staging released afterwards
  int rc = mhap_stage_reads(h, bases, offsets, lengths, ids, n);
  if (rc != MHAP_OK) return leave_collective_add(h, rc);
  rc = mhap_index_add_staged(h);      // (from here on the add takes care of the rendezvous itself)
  h->st_n = 0;
  return rc;
}

// shared tail of mhap_index_add_reads / mhap_index_add_staged: host mirrors after the kernels ran
static int finish_add(mhap_handle* h, int64_t first, const int64_t* ids, int64_t n) {
  HPROF("finish_add begin");
  const bool early = h->pend_ids.done && h->pend_ids.first == first && h->pend_ids.n == n && h->pend_ids.ids == ids;
  h->pend_ids.done = false; h->pend_ids.n = 0;
  // (nothing is committed before the add has succeeded — a failed id upload or meta mirror leaves the host mirrors at the
  //  length of the entries that exist, and the generation / "ids rise with the entries" pair describes only ids that were committed)
  auto undo = [&](int code) { h->ids.resize((size_t)first); h->fwd.resize((size_t)first); return code; };
  if (!early) { const int rf = fill_ids(h, first, ids, n); if (rf != MHAP_OK) return undo(rf); }
  h->inv_ready = false; h->ph_ready = false;                    // the entry set changes
  int rc = mirror_meta(h, h->d_meta, first, 2 * n);
  HPROF("meta mirrored");
  if (rc != MHAP_OK) return undo(rc);
  h->index_gen++;
  h->mono_val = h->mono_pending; h->mono_gen = h->index_gen;
  h->n_entries = first + 2 * n;
  h->stats.strands_indexed = 0;
  for (int64_t e = 0; e < h->n_entries; e++) if (h->status[(size_t)e] == 0) h->stats.strands_indexed++;
  return MHAP_OK;
}

int mhap_stage_reads(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, const int64_t* ids, int64_t n) {
  if (!h || n < 0 || (n > 0 && (!bases || !offsets || !lengths || !ids))) return h ? fail(h, MHAP_E_INVALID, "null argument") : MHAP_E_INVALID;
  (void)hipSetDevice(h->device);
  for (int64_t i = 0; i < n; i++) if (lengths[i] < 0) return fail(h, MHAP_E_INVALID, "negative read length");
  h->st_ids.assign(ids, ids + n);
  if (n == 0) { h->st_n = 0; return MHAP_OK; }
  return stage_reads(h, bases, offsets, lengths, n, false);
}

int mhap_index_reserve(mhap_handle* h, int64_t total_reads) {
  if (!h || total_reads < 0) return h ? fail(h, MHAP_E_INVALID, "negative read count") : MHAP_E_INVALID;
  if (h->n_entries != 0) return fail(h, MHAP_E_STATE, "mhap_index_reserve applies to an empty index");
  if (2 * total_reads > (int64_t)INT32_MAX / 2) return fail(h, MHAP_E_INVALID, "index too large for 32-bit entry indices");
  h->reserve_reads = total_reads;
  return MHAP_OK;
}

int mhap_index_add_staged(mhap_handle* h) {
  if (!h) return MHAP_E_INVALID;
  (void)hipSetDevice(h->device);
  const int64_t n = h->st_n;
  // (every exit in front of sketch_staged's rendezvous goes through leave_collective_add: the add is collective under the eager exchange)
  if (n <= 0 || (int64_t)h->st_ids.size() != n) return leave_collective_add(h, fail(h, MHAP_E_STATE, "no staged reads (call mhap_stage_reads first)"));
  if (h->n_entries + 2 * n > (int64_t)INT32_MAX / 2) return leave_collective_add(h, fail(h, MHAP_E_INVALID, "index too large for 32-bit entry indices"));
  const int64_t first = h->n_entries;
  // mhap_index_reserve: the tables are sized once for every read that is still to come
  const int64_t want_entries = std::max<int64_t>(first + 2 * n, first == 0 ? 2 * h->reserve_reads : 0);
  int rc = ensure_index_capacity(h, want_entries);
  if (rc != MHAP_OK) return leave_collective_add(h, rc);
  const int S = h->P.ordered_sketch_size;
  h->ph_ready = false; h->inv_ready = false;
  // The inverted index is built by the first search — unless this add very likely completes the index (the first add of an index
  // that was not announced to be larger, or the add that reaches the announced size): then it is built here, next to the ordered kernel
  const int64_t after = first + 2 * n;
  const bool likely_last = (first == 0 && 2 * h->reserve_reads <= after) || (h->reserve_reads > 0 && after == 2 * h->reserve_reads);
  // (a rank of a multi-GPU job with the eager exchange on: the add is collective; only the first add of an empty index can gather)
  const bool eager_exchange = h->dist != nullptr && dist_eager_wanted(h);
  h->pend_ids.first = first; h->pend_ids.n = n; h->pend_ids.ids = h->st_ids.data(); h->pend_ids.done = false;
  rc = sketch_staged(h, h->d_minhash + first * h->Hrow, h->Hrow, h->d_ordered + first * 2LL * S, 2LL * S, h->d_meta + first * META_W,
                     likely_last ? after : 0, eager_exchange, first == 0);
  if (rc != MHAP_OK) {   // (the early fill_ids inside sketch_staged may have grown the mirrors: back to the committed entries)
    h->pend_ids.n = 0; h->pend_ids.done = false;
    h->ids.resize((size_t)first); h->fwd.resize((size_t)first);
    return rc;
  }
  const bool built = h->inv_ready && h->inv_ne == after;
  rc = finish_add(h, first, h->st_ids.data(), n);
  if (rc == MHAP_OK && built) h->inv_ready = true;      // (finish_add drops the index of the OLD entry set; this one covers the new one)
  if (rc == MHAP_OK && eager_exchange) dist_eager_commit(h);
  return rc;
}

int mhap_sketch_staged_device(mhap_handle* h, void* d_minhash, void* d_ordered, void* d_meta) {
  if (!h || !d_minhash || !d_ordered || !d_meta) return h ? fail(h, MHAP_E_INVALID, "null argument") : MHAP_E_INVALID;
  (void)hipSetDevice(h->device);
  if (h->st_n <= 0) return fail(h, MHAP_E_STATE, "no staged reads (call mhap_stage_reads first)");
  return sketch_staged(h, (int32_t*)d_minhash, h->Hrow, (int32_t*)d_ordered, 2LL * h->P.ordered_sketch_size, (int32_t*)d_meta);
}

int mhap_sketch_batch(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, int64_t n, int32_t* minhash,
                      int32_t* ordered, int32_t* ordered_size, uint8_t* status) {
  if (!h) return MHAP_E_INVALID;
  if (n <= 0) return MHAP_OK;
  if (!bases || !offsets || !lengths) return fail(h, MHAP_E_INVALID, "null argument");
  (void)hipSetDevice(h->device);
  const int S = h->P.ordered_sketch_size;
  HIPCHK(h, h->q_minhash.ensure((size_t)(2 * n) * h->Hrow * 4));
  HIPCHK(h, h->q_ordered.ensure((size_t)(2 * n) * S * 8));
  HIPCHK(h, h->q_meta.ensure((size_t)(2 * n) * META_W * 4));
  int rc = sketch_into(h, bases, offsets, lengths, n, false, h->q_minhash.as<int32_t>(), h->Hrow, h->q_ordered.as<int32_t>(), 2LL * S,
                       h->q_meta.as<int32_t>());
  if (rc != MHAP_OK) return rc;
  std::vector<int32_t> meta((size_t)(2 * n) * META_W);
  HIPCHK(h, hipMemcpy(meta.data(), h->q_meta.p, meta.size() * 4, hipMemcpyDeviceToHost));
  if (minhash) HIPCHK(h, hipMemcpy(minhash, h->q_minhash.p, (size_t)(2 * n) * h->Hrow * 4, hipMemcpyDeviceToHost));
  if (ordered) HIPCHK(h, hipMemcpy(ordered, h->q_ordered.p, (size_t)(2 * n) * S * 8, hipMemcpyDeviceToHost));
  for (int64_t e = 0; e < 2 * n; e++) {
    if (ordered_size) ordered_size[e] = meta[(size_t)e * META_W + 3] == 0 ? meta[(size_t)e * META_W + 0] : 0;
    if (status) status[e] = (uint8_t)meta[(size_t)e * META_W + 3];
  }
  return MHAP_OK;
}

// Sketches that did not come from this library's kernels (a `.dat` file, a caller's arrays): the second stage ranks and medians positions
// by their bits and keeps no per-entry window test in its first pass (search_kernels.hip: FIRST => ok), which is only the reference's
// arithmetic for 0 <= pos < seqLength (J/sketch/BottomOverlapSketch.java:246-276: valid1Upper = seqLength; :548-558 writes pos = an
// index into the hash array).  A position outside that range is therefore refused here instead of being silently mis-ranked.
static bool ordered_positions_ok(const int32_t* ordered, const int32_t* ordered_size, const int32_t* ordered_seqlen, int64_t m, int S, int64_t* bad_entry) {
  std::atomic<int64_t> bad{-1};
  parallel_for(m, host_threads(), [&](int64_t lo, int64_t hi) {
    for (int64_t e = lo; e < hi && bad.load(std::memory_order_relaxed) < 0; e++) {
      const int32_t n = ordered_size[e], len = ordered_seqlen[e];
      if (n < 0 || n > S) continue;                       // (reported by the caller's own size check)
      const int32_t* row = ordered + (size_t)e * (size_t)S * 2;
      uint32_t out = 0;
      for (int32_t i = 0; i < n; i++) out |= (uint32_t)((uint32_t)row[2 * i + 1] >= (uint32_t)len);   // (one unsigned compare: pos < 0 or pos >= len)
      if (out) { int64_t want = -1; bad.compare_exchange_strong(want, e); }
    }
  }, 256);
  *bad_entry = bad.load();
  return *bad_entry < 0;
}

int mhap_index_add_sketches(mhap_handle* h, const int64_t* ids, const uint8_t* is_fwd, const int32_t* seq_length, const int32_t* minhash,
                            const int32_t* ordered, const int32_t* ordered_size, const int32_t* ordered_seqlen, int64_t m) {
  if (!h) return MHAP_E_INVALID;
  if (m <= 0) return MHAP_OK;
  if (!ids || !is_fwd || !seq_length || !minhash || !ordered || !ordered_size || !ordered_seqlen) return fail(h, MHAP_E_INVALID, "null argument");
  (void)hipSetDevice(h->device);
  if (h->n_entries + m > (int64_t)INT32_MAX / 2) return fail(h, MHAP_E_INVALID, "index too large for 32-bit entry indices");
  for (int64_t e = 0; e < m; e++)
    if (seq_length[e] < 0 || ordered_seqlen[e] < 0) return fail(h, MHAP_E_INVALID, "negative sequence length in a precomputed sketch");
  {
    int64_t bad = -1;
    if (!ordered_positions_ok(ordered, ordered_size, ordered_seqlen, m, h->P.ordered_sketch_size, &bad))
      return fail(h, MHAP_E_INVALID, "precomputed sketch " + std::to_string(bad) + ": an ordered-sketch position outside [0, its sequence's k-mer count)");
  }
  int rc = ensure_index_capacity(h, h->n_entries + m);
  if (rc != MHAP_OK) return rc;
  const int64_t first = h->n_entries;
  const int S = h->P.ordered_sketch_size;
  std::vector<int32_t> meta((size_t)m * META_W);
  for (int64_t e = 0; e < m; e++) {
    if (ordered_size[e] < 0 || ordered_size[e] > S) return fail(h, MHAP_E_INVALID, "ordered sketch larger than --ordered-sketch-size");
    meta[(size_t)e * META_W + 0] = ordered_size[e]; meta[(size_t)e * META_W + 1] = ordered_seqlen[e];
    meta[(size_t)e * META_W + 2] = seq_length[e]; meta[(size_t)e * META_W + 3] = 0;
  }
  HIPCHK(h, hipMemcpy(h->d_minhash + first * h->Hrow, minhash, (size_t)m * h->Hrow * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_ordered + first * 2LL * S, ordered, (size_t)m * S * 8, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_meta + first * META_W, meta.data(), meta.size() * 4, hipMemcpyHostToDevice));
  h->ids.resize((size_t)(first + m)); h->fwd.resize((size_t)(first + m));
  for (int64_t e = 0; e < m; e++) { h->ids[(size_t)(first + e)] = ids[e]; h->fwd[(size_t)(first + e)] = is_fwd[e] ? 1 : 0; }
  HIPCHK(h, hipMemcpy(h->d_ids.as<int64_t>() + first, h->ids.data() + first, (size_t)m * 8, hipMemcpyHostToDevice));
  rc = mirror_meta(h, h->d_meta, first, m);
  if (rc != MHAP_OK) return rc;
  h->n_entries = first + m; h->inv_ready = false; h->ph_ready = false; h->index_gen++;
  h->stats.strands_indexed += m;
  return MHAP_OK;
}

int mhap_index_size(mhap_handle* h, int64_t* entries) {
  if (!h || !entries) return MHAP_E_INVALID;
  *entries = h->n_entries;
  return MHAP_OK;
}

int mhap_index_export(mhap_handle* h, int64_t first, int64_t count, int64_t* ids, uint8_t* is_fwd, int32_t* seq_length, int32_t* minhash,
                      int32_t* ordered, int32_t* ordered_size, int32_t* ordered_seqlen, uint8_t* status) {
  if (!h) return MHAP_E_INVALID;
  if (first < 0 || count < 0 || first + count > h->n_entries) return fail(h, MHAP_E_INVALID, "export range outside the index");
  if (count == 0) return MHAP_OK;
  (void)hipSetDevice(h->device);
  const int S = h->P.ordered_sketch_size;
  std::vector<int32_t> meta((size_t)count * META_W);
  HIPCHK(h, hipMemcpy(meta.data(), h->d_meta + first * META_W, meta.size() * 4, hipMemcpyDeviceToHost));
  if (minhash) HIPCHK(h, hipMemcpy(minhash, h->d_minhash + first * h->Hrow, (size_t)count * h->Hrow * 4, hipMemcpyDeviceToHost));
  if (ordered) HIPCHK(h, hipMemcpy(ordered, h->d_ordered + first * 2LL * S, (size_t)count * S * 8, hipMemcpyDeviceToHost));
  for (int64_t e = 0; e < count; e++) {
    if (ids) ids[e] = h->ids[(size_t)(first + e)];
    if (is_fwd) is_fwd[e] = h->fwd[(size_t)(first + e)];
    if (seq_length) seq_length[e] = meta[(size_t)e * META_W + 2];
    if (ordered_size) ordered_size[e] = meta[(size_t)e * META_W + 0];
    if (ordered_seqlen) ordered_seqlen[e] = meta[(size_t)e * META_W + 1];
    if (status) status[e] = (uint8_t)meta[(size_t)e * META_W + 3];
  }
  return MHAP_OK;
}

int mhap_index_clear(mhap_handle* h) {
  if (!h) return MHAP_E_INVALID;
  h->n_entries = 0; h->external = false; h->inv_ready = false; h->ph_ready = false; h->reserve_reads = 0; h->index_gen++;
  h->ids.clear(); h->fwd.clear(); h->seqlen.clear(); h->status.clear();
  h->d_minhash = h->own_minhash.as<int32_t>(); h->d_ordered = h->own_ordered.as<int32_t>(); h->d_meta = h->own_meta.as<int32_t>();
  h->stats = mhap_stats{};
  return MHAP_OK;
}

int mhap_index_prepare(mhap_handle* h) {
  if (!h) return MHAP_E_INVALID;
  (void)hipSetDevice(h->device);
  if (h->n_entries == 0) return MHAP_OK;
  const int rc = ensure_inverted_index(h);
  if (rc != MHAP_OK) return rc;
  return sync_stream(h);
}

int mhap_sketch_reads_device(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, int64_t n, void* d_minhash,
                             void* d_ordered, void* d_meta) {
  if (!h) return MHAP_E_INVALID;
  if (n <= 0) return MHAP_OK;
  if (!bases || !offsets || !lengths || !d_minhash || !d_ordered || !d_meta) return fail(h, MHAP_E_INVALID, "null argument");
  (void)hipSetDevice(h->device);
  return sketch_into(h, bases, offsets, lengths, n, false, (int32_t*)d_minhash, h->Hrow, (int32_t*)d_ordered, 2LL * h->P.ordered_sketch_size,
                     (int32_t*)d_meta);
}

int mhap_index_set_device(mhap_handle* h, const int64_t* ids, const uint8_t* is_fwd, void* d_minhash, void* d_ordered, void* d_meta, int64_t m) {
  if (!h) return MHAP_E_INVALID;
  if (m < 0 || (m > 0 && (!ids || !is_fwd || !d_minhash || !d_ordered || !d_meta))) return fail(h, MHAP_E_INVALID, "null argument");
  if (m > (int64_t)INT32_MAX / 2) return fail(h, MHAP_E_INVALID, "index too large for 32-bit entry indices");
  (void)hipSetDevice(h->device);
  mhap_index_clear(h);
  h->external = true;
  h->d_minhash = (int32_t*)d_minhash; h->d_ordered = (int32_t*)d_ordered; h->d_meta = (int32_t*)d_meta;
  h->ids.assign(ids, ids + m); h->fwd.assign(is_fwd, is_fwd + m);
  HIPCHK(h, h->d_ids.ensure((size_t)std::max<int64_t>(m, 1) * 8));
  if (m > 0) HIPCHK(h, hipMemcpy(h->d_ids.p, ids, (size_t)m * 8, hipMemcpyHostToDevice));
  if (m > 0) { int rc = mirror_meta(h, h->d_meta, 0, m); if (rc != MHAP_OK) return rc; }
  h->n_entries = m; h->inv_ready = false; h->ph_ready = false; h->index_gen++;
  for (int64_t e = 0; e < m; e++) if (h->status[(size_t)e] == 0) h->stats.strands_indexed++;
  return MHAP_OK;
}

static int self_search(mhap_handle* h, int64_t q_first, int64_t q_count, int64_t shard, int64_t nshards, mhap_record_sink sink, void* user) {
  HPROF("self_search begin");
  (void)hipSetDevice(h->device);
  if (q_first < 0 || q_first > h->n_entries) return fail(h, MHAP_E_INVALID, "query range outside the index");
  if (nshards < 1 || shard < 0 || shard >= nshards) return fail(h, MHAP_E_INVALID, "bad shard");
  int64_t q_end = (q_count < 0) ? h->n_entries : std::min(h->n_entries, q_first + q_count);
  std::vector<int32_t> ql;
  int64_t ordinal = 0;   // running count of forward entries = the read's position in the index
  for (int64_t e = 0; e < q_end; e++) {
    if (!h->fwd[(size_t)e]) continue;
    const int64_t o = ordinal++;
    if (e < q_first || (o % nshards) != shard) continue;
    if (h->status[(size_t)e] == 0) ql.push_back((int32_t)e);   // AbstractMatchSearch.java:128-129
  }
  bool mono = true;
  if (h->mono_gen == h->index_gen) mono = h->mono_val;   // (looked at when the ids arrived)
  else for (int64_t e = 1; e < h->n_entries && mono; e++) if (h->ids[(size_t)e] < h->ids[(size_t)(e - 1)]) mono = false;
  // tile skipping needs: ids sorted with entry order, every entry "long" (minStore == 0 -> only m.id < q.id survives)
  const bool tri = mono && h->P.min_store_length == 0 && !getenv("MHAP_NO_TRIANGULAR");
  QuerySide qs{h->d_minhash, h->Hrow, h->d_ordered, 2LL * h->P.ordered_sketch_size, h->d_meta, h->d_ids.as<int64_t>(), h->ids.data(), h->seqlen.data(), h->n_entries};
  HPROF("search_core begin");
  const int rcs = search_core(h, qs, ql, true, tri, sink, user);
  HPROF("search_core end");
  return rcs;
}

int mhap_find_matches_self(mhap_handle* h, int64_t q_first, int64_t q_count, mhap_record_sink sink, void* user) {
  if (!h) return MHAP_E_INVALID;
  return self_search(h, q_first, q_count, 0, 1, sink, user);
}

int mhap_find_matches_self_shard(mhap_handle* h, int64_t shard, int64_t nshards, mhap_record_sink sink, void* user) {
  if (!h) return MHAP_E_INVALID;
  return self_search(h, 0, -1, shard, nshards, sink, user);
}

int mhap_find_matches_reads(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, const int64_t* ids, int64_t n,
                            mhap_record_sink sink, void* user) {
  if (!h) return MHAP_E_INVALID;
  if (n <= 0) return MHAP_OK;
  if (!bases || !offsets || !lengths || !ids) return fail(h, MHAP_E_INVALID, "null argument");
  (void)hipSetDevice(h->device);
  const int S = h->P.ordered_sketch_size;
  HIPCHK(h, h->q_minhash.ensure((size_t)(2 * n) * h->Hrow * 4));
  HIPCHK(h, h->q_ordered.ensure((size_t)(2 * n) * S * 8));
  HIPCHK(h, h->q_meta.ensure((size_t)(2 * n) * META_W * 4));
  HIPCHK(h, h->q_ids.ensure((size_t)(2 * n) * 8));
  int rc = sketch_into(h, bases, offsets, lengths, n, true, h->q_minhash.as<int32_t>(), h->Hrow, h->q_ordered.as<int32_t>(), 2LL * S,
                       h->q_meta.as<int32_t>());
  if (rc != MHAP_OK) return rc;
  std::vector<int32_t> meta((size_t)(2 * n) * META_W);
  HIPCHK(h, hipMemcpy(meta.data(), h->q_meta.p, meta.size() * 4, hipMemcpyDeviceToHost));
  std::vector<int64_t> qids((size_t)(2 * n));
  std::vector<int32_t> qlen((size_t)(2 * n));
  std::vector<int32_t> ql;
  for (int64_t i = 0; i < n; i++) {
    qids[(size_t)(2 * i)] = qids[(size_t)(2 * i + 1)] = ids[i];
    qlen[(size_t)(2 * i)] = qlen[(size_t)(2 * i + 1)] = lengths[i];
    if (meta[(size_t)(2 * i) * META_W + 3] == 0) ql.push_back((int32_t)(2 * i));   // forward only (AbstractMatchSearch.java:225,236)
  }
  HIPCHK(h, hipMemcpy(h->q_ids.p, qids.data(), qids.size() * 8, hipMemcpyHostToDevice));
  QuerySide qs{h->q_minhash.as<int32_t>(), h->Hrow, h->q_ordered.as<int32_t>(), 2LL * S, h->q_meta.as<int32_t>(), h->q_ids.as<int64_t>(),
               qids.data(), qlen.data(), 2 * n};
  return search_core(h, qs, ql, false, false, sink, user);
}

int mhap_find_matches_sketches(mhap_handle* h, const int64_t* ids, const int32_t* seq_length, const int32_t* minhash, const int32_t* ordered,
                               const int32_t* ordered_size, const int32_t* ordered_seqlen, int64_t m, mhap_record_sink sink, void* user) {
  if (!h) return MHAP_E_INVALID;
  if (m <= 0) return MHAP_OK;
  if (!ids || !seq_length || !minhash || !ordered || !ordered_size || !ordered_seqlen) return fail(h, MHAP_E_INVALID, "null argument");
  (void)hipSetDevice(h->device);
  const int S = h->P.ordered_sketch_size;
  for (int64_t e = 0; e < m; e++)
    if (seq_length[e] < 0 || ordered_seqlen[e] < 0) return fail(h, MHAP_E_INVALID, "negative sequence length in a precomputed sketch");
  {
    int64_t bad = -1;
    if (!ordered_positions_ok(ordered, ordered_size, ordered_seqlen, m, S, &bad))
      return fail(h, MHAP_E_INVALID, "query sketch " + std::to_string(bad) + ": an ordered-sketch position outside [0, its sequence's k-mer count)");
  }
  std::vector<int32_t> meta((size_t)m * META_W);
  std::vector<int32_t> ql((size_t)m);
  for (int64_t e = 0; e < m; e++) {
    if (ordered_size[e] < 0 || ordered_size[e] > S) return fail(h, MHAP_E_INVALID, "ordered sketch larger than --ordered-sketch-size");
    meta[(size_t)e * META_W + 0] = ordered_size[e]; meta[(size_t)e * META_W + 1] = ordered_seqlen[e];
    meta[(size_t)e * META_W + 2] = seq_length[e]; meta[(size_t)e * META_W + 3] = 0;
    ql[(size_t)e] = (int32_t)e;
  }
  HIPCHK(h, h->q_minhash.ensure((size_t)m * h->Hrow * 4));
  HIPCHK(h, h->q_ordered.ensure((size_t)m * S * 8));
  HIPCHK(h, h->q_meta.ensure((size_t)m * META_W * 4));
  HIPCHK(h, h->q_ids.ensure((size_t)m * 8));
  HIPCHK(h, hipMemcpy(h->q_minhash.p, minhash, (size_t)m * h->Hrow * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->q_ordered.p, ordered, (size_t)m * S * 8, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->q_meta.p, meta.data(), meta.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->q_ids.p, ids, (size_t)m * 8, hipMemcpyHostToDevice));
  QuerySide qs{h->q_minhash.as<int32_t>(), h->Hrow, h->q_ordered.as<int32_t>(), 2LL * S, h->q_meta.as<int32_t>(), h->q_ids.as<int64_t>(), ids,
               seq_length, m};
  return search_core(h, qs, ql, false, false, sink, user);
}

// (d_ids_dev: the ids as they already sit in device memory — the sharded search has them from its gather — or NULL: uploaded here)
static int find_matches_device_impl(mhap_handle* h, const void* d_q_minhash, const void* d_q_ordered, const void* d_q_meta, const int64_t* ids,
                                    const int64_t* d_ids_dev, int64_t m, int to_self, mhap_record_sink sink, void* user) {
  if (!h) return MHAP_E_INVALID;
  if (m <= 0) return MHAP_OK;
  if (!d_q_minhash || !d_q_ordered || !d_q_meta || !ids) return fail(h, MHAP_E_INVALID, "null argument");
  if (m > (int64_t)INT32_MAX / 2) return fail(h, MHAP_E_INVALID, "too many query rows");
  (void)hipSetDevice(h->device);
  const int S = h->P.ordered_sketch_size;
  // the rows' meta words come to the host through the pinned bounce buffer (lengths for the records, statuses for the query list), and
  // the ids go up — unless they are on the device already — while the host walks them: on one rank of eight this preparation was
  // 0.4 ms of a 3.9 ms search
  // Round 6: the first kernels no longer wait for any of it — the ids are staged through the pinned buffer (an asynchronous copy), the query
  // list is made on the device, and the meta words are waited for behind the first tier's launch (QuerySide::prep): 0.35 -> 0.06 ms between
  // the call and the first kernel on one rank of eight.
  const size_t mbytes = (size_t)m * META_W * 4, idbytes = d_ids_dev ? 0 : (size_t)m * 8;
  char* pin = (char*)pinned_io(h, mbytes + idbytes);
  if (!pin) return fail(h, MHAP_E_HIP, "cannot allocate pinned host memory");
  const int32_t* meta = (const int32_t*)pin;
  if (!d_ids_dev) {
    HIPCHK(h, h->q_ids.ensure((size_t)m * 8));
    memcpy(pin + mbytes, ids, idbytes);
    HIPCHK(h, hipMemcpyAsync(h->q_ids.p, pin + mbytes, idbytes, hipMemcpyHostToDevice, h->stream));   // (ordered before the search's kernels)
  }
  HIPCHK(h, hipMemcpyAsync((void*)meta, d_q_meta, mbytes, hipMemcpyDeviceToHost, h->copy_stream));
  std::vector<int32_t> qlen((size_t)m), ql;
  std::vector<uint8_t> valid((size_t)m);
  QuerySide qs{(const int32_t*)d_q_minhash, h->Hrow, (const int32_t*)d_q_ordered, 2LL * S, (const int32_t*)d_q_meta,
               d_ids_dev ? d_ids_dev : h->q_ids.as<int64_t>(), ids, qlen.data(), m};
  qs.identity = true;
  qs.prep = [&](QuerySide& q) -> int {
    const hipError_t e = hipStreamSynchronize(h->copy_stream);
    if (e != hipSuccess) return fail(h, MHAP_E_HIP, std::string("read-back of the query rows' meta words: ") + hipGetErrorString(e));
    for (int64_t i = 0; i < m; i++) { qlen[(size_t)i] = meta[(size_t)i * META_W + 2]; valid[(size_t)i] = meta[(size_t)i * META_W + 3] == 0 ? 1 : 0; }
    q.h_valid = valid.data();
    return MHAP_OK;
  };
  const int rcs = search_core(h, qs, ql, to_self != 0, false, sink, user);
  if (qs.prep) (void)hipStreamSynchronize(h->copy_stream);   // (the copy targets this call's buffers: never left in flight)
  return rcs;
}

int mhap_find_matches_device(mhap_handle* h, const void* d_q_minhash, const void* d_q_ordered, const void* d_q_meta, const int64_t* ids,
                             int64_t m, int to_self, mhap_record_sink sink, void* user) {
  return find_matches_device_impl(h, d_q_minhash, d_q_ordered, d_q_meta, ids, nullptr, m, to_self, sink, user);
}

int mhap_set_second_stage_gate(mhap_handle* h, mhap_stage_gate gate, void* user) {
  if (!h) return MHAP_E_INVALID;
  h->gate = gate; h->gate_user = user;
  return MHAP_OK;
}

int mhap_get_stats(mhap_handle* h, mhap_stats* out) { if (!h || !out) return MHAP_E_INVALID; *out = h->stats; return MHAP_OK; }
int mhap_get_kernel_times(mhap_handle* h, mhap_kernel_times* out) { if (!h || !out) return MHAP_E_INVALID; *out = h->ktimes; return MHAP_OK; }
int mhap_reset_kernel_times(mhap_handle* h) { if (!h) return MHAP_E_INVALID; h->ktimes = mhap_kernel_times{}; return MHAP_OK; }
int mhap_set_stream(mhap_handle* h, void* hip_stream) {
  if (!h) return MHAP_E_INVALID;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  return MHAP_OK;
}
int mhap_synchronize(mhap_handle* h) { if (!h) return MHAP_E_INVALID; (void)hipSetDevice(h->device); return sync_stream(h); }

}  // extern "C"

// =================================================================================================
// Host-side entry points that share the kernels' __host__ __device__ arithmetic.
// =================================================================================================
extern "C" {

// murmur3_x64_128(seed 0).h1 of one k-mer string, canonicalised when do_rc != 0 exactly like
// HashUtils.computeSequenceHashesLong(str, len, 0, doRC)[0] (J/sketch/HashUtils.java:237-258).  This is what
// the host driver uses to hash the lines of a `-f` filter file (J/sketch/FrequencyCounts.java:169).
int mhap_hash_kmer(const char* kmer, int32_t len, int32_t do_rc, int64_t* out) {
  if (!kmer || !out || len < 1 || len > 4096) return MHAP_E_INVALID;
  std::string s(kmer, (size_t)len);
  if (do_rc) {
    std::string r((size_t)len, 'N');
    for (int i = 0; i < len; i++) r[(size_t)i] = (char)rc_char((uint32_t)(uint8_t)s[(size_t)(len - 1 - i)]);
    if (r.compare(s) < 0) s = r;   // String.compareTo < 0 -> use the reverse complement (:246-251)
  }
  std::vector<uint32_t> W((size_t)len / 4 + 3, 0u);
  memcpy(W.data(), s.data(), (size_t)len);
  *out = (int64_t)murmur128_h1_chars<0>(W.data(), 0, len);
  return MHAP_OK;
}

// Test hooks: run the SAME __host__ __device__ functions the kernels use on the host, so that the hash and
// second-stage lane logic can be checked in a GPU-less container.  Never called by the product path.
int mhap_selftest_hash_windows(const char* seq, int32_t len, int32_t k, int32_t k2, int64_t* out64, int32_t* out32) {
  if (!seq || len < 0 || k < 1 || k2 < 1) return MHAP_E_INVALID;
  std::vector<uint32_t> W((size_t)len / 4 + 3, 0u);
  memcpy(W.data(), seq, (size_t)len);
  if (out64) for (int p = 0; p + k <= len; p++) out64[p] = (k == 16) ? (int64_t)murmur128_h1_chars<16>(W.data(), p, k) : (int64_t)murmur128_h1_chars<0>(W.data(), p, k);
  if (out32) for (int p = 0; p + k2 <= len; p++) out32[p] = (k2 == 12) ? (int32_t)murmur32_chars<12>(W.data(), p, k2) : (int32_t)murmur32_chars<0>(W.data(), p, k2);
  return MHAP_OK;
}

// chain value after `nsteps` xorshift64 steps computed the way the MinHash kernel's candidate drain does it
// (GF(2) byte tables for the multiple of 64, single steps for the rest)
int mhap_selftest_xorshift_jump(uint64_t key, int32_t nsteps, uint64_t* out) {
  if (!out || nsteps < 0 || nsteps > (1 << 16)) return MHAP_E_INVALID;
  const int a = nsteps >> XS_JUMP_LOG2, r = nsteps & ((1 << XS_JUMP_LOG2) - 1);
  uint64_t x = key;
  if (a > 0) {
    // two levels like the drain: fine tables for a <= na, one coarse table application (M^(g na q)) before them beyond that
    const int na = a < 24 ? a : 24;
    int qa = 0, af = a;
    if (af > na) { qa = (af - 1) / na; af -= qa * na; }
    std::vector<uint64_t> jt((size_t)(na + qa + 1) * 2048);
    build_xorshift_jump_tables(na, qa + 1, jt.data());
    if (qa > 0) {
      const uint64_t* T = jt.data() + (size_t)(na + qa - 1) * 2048;
      uint64_t y = 0;
      for (int i = 0; i < 8; i++) y ^= T[i * 256 + (int)((x >> (8 * i)) & 255u)];
      x = y;
    }
    const uint64_t* T = jt.data() + (size_t)(af - 1) * 2048;
    uint64_t y = 0;
    for (int i = 0; i < 8; i++) y ^= T[i * 256 + (int)((x >> (8 * i)) & 255u)];
    x = y;
  }
  for (int t = 0; t < r; t++) x = xorshift_step(x);
  *out = x;
  return MHAP_OK;
}

// the key behind chain value x after nsteps >= 1 steps, recovered the way the weight-1 MinHash kernel does it (inverse byte tables
// for the next multiple of 4 steps, then up to 3 steps forward)
int mhap_selftest_xorshift_unjump(uint64_t x, int32_t nsteps, uint64_t* out) {
  if (!out || nsteps < 1 || nsteps > (1 << 15)) return MHAP_E_INVALID;
  int a = (nsteps + (1 << XS_JUMP_LOG2) - 1) >> XS_JUMP_LOG2;
  const int r = (a << XS_JUMP_LOG2) - nsteps;
  const int qa = a > W1_JUMP_NA ? (a - 1) / W1_JUMP_NA : 0;      // two levels, as in w1_key_of
  a -= qa * W1_JUMP_NA;
  std::vector<uint64_t> ut((size_t)(W1_JUMP_NA + qa) * 2048);
  build_xorshift_unjump_tables(W1_JUMP_NA, qa, ut.data());
  uint64_t y = x;
  if (qa > 0) {
    const uint64_t* T = ut.data() + (size_t)(W1_JUMP_NA + qa - 1) * 2048;
    uint64_t z = 0;
    for (int i = 0; i < 8; i++) z ^= T[i * 256 + (int)((y >> (8 * i)) & 255u)];
    y = z;
  }
  {
    const uint64_t* T = ut.data() + (size_t)(a - 1) * 2048;
    uint64_t z = 0;
    for (int i = 0; i < 8; i++) z ^= T[i * 256 + (int)((y >> (8 * i)) & 255u)];
    y = z;
  }
  for (int t = 0; t < r; t++) y = xorshift_step(y);
  *out = y;
  return MHAP_OK;
}

// The --supress-noise whitelist built the way mhap_set_filter_whitelist builds it, probed with the kernels' own lookup:
// out_flags[i] = mightContain(probes[i]); out2 = {bit size, number of hash functions}
int mhap_selftest_bloom(const int64_t* hashes, int64_t n, int64_t size_bloom, const int64_t* probes, int64_t np, uint8_t* out_flags, int64_t* out2) {
  if ((n > 0 && !hashes) || (np > 0 && (!probes || !out_flags)) || !out2) return MHAP_E_INVALID;
  std::vector<unsigned long long> words; uint64_t bit_size = 0; int k = 0;
  build_bloom(hashes, n, size_bloom, words, bit_size, k);
  for (int64_t i = 0; i < np; i++) out_flags[i] = bloom_might_contain(words.data(), bit_size, k, (uint64_t)probes[i]) ? 1 : 0;
  out2[0] = (int64_t)bit_size; out2[1] = k;
  return MHAP_OK;
}

// in-place 32x32 bit-matrix transpose used by the bit-sliced MinHash rows (device_common.hpp: transpose32)
// host-side: the identity table of (inter, k) and the early-reject table derived from it (GPU-less unit tests): scores[(k (k + 1) / 2) + inter]
// for k <= S, pass_min[0 .. S + 1]
int mhap_selftest_pass_min(int32_t S, int32_t k2, double threshold, double* scores, int32_t* pass_min) {
  if (S < 1 || S > 8192 || k2 < 1 || !pass_min) return MHAP_E_INVALID;
  std::vector<double> tbl; std::vector<int32_t> pm;
  fill_score_table(S, k2, tbl);
  build_pass_min(tbl, S, threshold, pm);
  if (scores) memcpy(scores, tbl.data(), tbl.size() * 8);
  memcpy(pass_min, pm.data(), pm.size() * 4);
  return MHAP_OK;
}

int mhap_selftest_transpose32(uint32_t* a32) {
  if (!a32) return MHAP_E_INVALID;
  uint32_t a[32];
  memcpy(a, a32, sizeof a);
  transpose32(a);
  memcpy(a32, a, sizeof a);
  return MHAP_OK;
}

// out8 = {empty, valid, a1, a2, b1, b2, inter, k}
int mhap_selftest_overlap_lane(const int32_t* A, int32_t nA, int32_t lenA, const int32_t* B, int32_t nB, int32_t lenB, double max_shift,
                               int32_t stride, int32_t* out8) {
  if (!A || !B || !out8 || stride < 1) return MHAP_E_INVALID;
  const int maxrec = 2 * std::max(nA, nB) + 2;
  std::vector<int32_t> scratch((size_t)3 * (size_t)maxrec * (size_t)stride, 0);
  LaneScratch sc; sc.base = scratch.data(); sc.stride = stride; sc.maxrec = maxrec;
  PlainView va{A, nA}, vb{B, nB};
  const LaneOverlap r = lane_overlap(va, lenA, vb, lenB, max_shift, sc);
  out8[0] = r.empty; out8[1] = r.valid; out8[2] = r.a1; out8[3] = r.a2; out8[4] = r.b1; out8[5] = r.b2; out8[6] = r.inter; out8[7] = r.kk;
  return MHAP_OK;
}

}  // extern "C"

namespace mhap {
int internal_find_matches_device(mhap_handle* h, const void* d_q_minhash, const void* d_q_ordered, const void* d_q_meta, const int64_t* ids,
                                 const int64_t* d_ids_dev, int64_t m, int to_self, mhap_record_sink sink, void* user) {
  return find_matches_device_impl(h, d_q_minhash, d_q_ordered, d_q_meta, ids, d_ids_dev, m, to_self, sink, user);
}
}  // namespace mhap
