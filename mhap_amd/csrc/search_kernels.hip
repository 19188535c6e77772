// search_kernels.hip — gfx950 kernels for candidate generation and second-stage scoring
// (hot loops E-F of SURVEY §3.1).
//
//   candidate_kernel : tiled all-pairs MinHash slot-equality count.  The reference walks an inverted
//                      index (J/impl/MinHashSearch.java:100-147,161-181); the hit count of a pair is by
//                      construction the number of slots whose int sketches are equal
//                      (J/sketch/MinHashSketch.java:237-252), so a brute-force count with the same
//                      `>= numMinMatches` cut and the same id/length filters (:200-225) yields the identical
//                      candidate set.  Integer VALU work (v_cmp_eq + v_addc per slot pair), tiles staged
//                      through LDS, 8x8 register micro-tile per lane, triangular tile skipping in self mode.
//   overlap_join_kernel : BottomOverlapSketch.getOverlapInfo per candidate, one wavefront each, from the equal-hash join.
//   overlap_kernel   : the same per candidate, one lane each, literal merge (overlap_lane.hpp): pairs the join path hands back.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "kernels.hpp"
#include "overlap_lane.hpp"

namespace mhap {

// MH_OJ_WIDE_UNIT (search_kernels_wide.hip / _wide2.hip): this file compiled once more for the join kernel with room for more joined k-mers
// per pair — ONLY the join kernel, its ALONE shape: the candidate, index and per-lane kernels are not compiled a second and third time
// (round 5 built the whole unit three times: every search kernel triplicated in a 2.4-MB library)
#ifndef MH_OJ_WIDE_UNIT

// count += (a == b): exactly two VALU issues per slot pair (v_cmp_eq -> vcc, v_addc consumes vcc).  Left to the
// compiler the compare lands in arbitrary SGPR pairs (v_cmp_e64 + v_cndmask + add) and spills SGPRs through
// v_writelane/v_readlane inside the hot loop.
#define CMP_ACC4(c, q, m)                                                                                   \
  asm("v_cmp_eq_u32 vcc, %1, %5\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"                                    \
      "v_cmp_eq_u32 vcc, %2, %6\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"                                    \
      "v_cmp_eq_u32 vcc, %3, %7\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"                                    \
      "v_cmp_eq_u32 vcc, %4, %8\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc"                                          \
      : "+v"(c)                                                                                             \
      : "v"((q).x), "v"((q).y), "v"((q).z), "v"((q).w), "v"((m).x), "v"((m).y), "v"((m).z), "v"((m).w)      \
      : "vcc")

constexpr int CAND_LD = CAND_KS + 4;  // padded LDS row (ints): 36 -> conflict-free ds_read_b128 across 16 rows

__device__ inline bool pair_passes(const SearchParams& sp, int64_t qid, int64_t mid, int qlen, int mlen) {
  if (sp.to_self && mid == qid) return false;                                               // MinHashSearch.java:200-201
  if (mlen < sp.min_store_length && qlen < sp.min_store_length) return false;               // :211-212
  if (sp.to_self && mid > qid && mlen >= sp.min_store_length && qlen >= sp.min_store_length) return false;  // :215-219
  if (sp.to_self && mlen < sp.min_store_length && qlen >= sp.min_store_length) return false;                // :222-225
  return true;
}

__global__ __launch_bounds__(256, 2) void candidate_kernel(const int32_t* __restrict__ minhash, int64_t row_stride,
                                                        const int32_t* __restrict__ qminhash, int64_t qrow_stride,
                                                        const int32_t* __restrict__ qlist, int nq, int ne,
                                                        const int64_t* __restrict__ ids, const int64_t* __restrict__ qids,
                                                        const int32_t* __restrict__ meta, const int32_t* __restrict__ qmeta,
                                                        SearchParams sp, int triangular, const long long* __restrict__ rowstart, int ntq,
                                                        long long nblocks_valid, int ntu,
                                                        Candidate* __restrict__ cand, unsigned long long* __restrict__ cand_count,
                                                        unsigned long long cand_cap) {
  __shared__ __attribute__((aligned(16))) int32_t qs[CAND_TQ * CAND_LD];
  __shared__ __attribute__((aligned(16))) int32_t ms[CAND_TM * CAND_LD];
  __shared__ int32_t qent[CAND_TQ];
  // XCD-aware remap: block b runs on XCD b%8; give every XCD a contiguous range of tiles so that
  // consecutive tiles (same query tile, neighbouring index tiles) share that XCD's L2.
  const long long nb8 = (long long)gridDim.x;
  const long long b = (long long)blockIdx.x;
  const long long bp = (b % 8) * (nb8 / 8) + b / 8;
  if (bp >= nblocks_valid) return;
  int t, u;
  if (triangular) {
    // rowstart[t] = first linear tile id of query-tile row t; row t only owns the index tiles that can
    // hold an entry with a smaller read id than the row's largest query (MinHashSearch.java:215-219).
    int lo = 0, hi = ntq;  // find t with rowstart[t] <= bp < rowstart[t+1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rowstart[mid] <= bp) lo = mid; else hi = mid; }
    t = lo;
    u = (int)(bp - rowstart[t]);
  } else {
    t = (int)(bp / ntu);
    u = (int)(bp % ntu);
  }
  const int q0 = t * CAND_TQ, m0 = u * CAND_TM;
  const int tid = threadIdx.x;
  if (tid < CAND_TQ) qent[tid] = (q0 + tid < nq) ? qlist[q0 + tid] : -1;
  __syncthreads();
  const int tq = tid >> 4, tm = tid & 15;
  int cnt[8][8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) cnt[i][j] = 0;

  const int H = sp.H;
  const bool vec_ok = ((row_stride & 3) == 0) && ((qrow_stride & 3) == 0);
  // Register-staged software pipeline: the global loads of chunk c+1 are issued before the compare phase of
  // chunk c and land in LDS after it, so HBM/L2 latency hides under ~8k VALU cycles of compares.
  int4 pq[4], pm[4];
  auto fetch = [&](int s0) {
    const bool full = vec_ok && (s0 + CAND_KS <= H);
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
      const int idx = tid + rep * 256;
      const int row = idx >> 3, c4 = (idx & 7) * 4;
      int4 qv = make_int4(0, 0, 0, 0), mv = make_int4(1, 1, 1, 1);   // never-equal sentinels for absent rows/slots
      const int qe = qent[row];
      const int me = m0 + row;
      if (full) {
        if (qe >= 0) qv = *(const int4*)(qminhash + (int64_t)qe * qrow_stride + s0 + c4);
        if (me < ne) mv = *(const int4*)(minhash + (int64_t)me * row_stride + s0 + c4);
      } else {
        int qa[4] = {0, 0, 0, 0}, ma[4] = {1, 1, 1, 1};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int s = s0 + c4 + e;
          if (s < H) {
            if (qe >= 0) qa[e] = qminhash[(int64_t)qe * qrow_stride + s];
            if (me < ne) ma[e] = minhash[(int64_t)me * row_stride + s];
          }
        }
        qv = make_int4(qa[0], qa[1], qa[2], qa[3]);
        mv = make_int4(ma[0], ma[1], ma[2], ma[3]);
      }
      pq[rep] = qv; pm[rep] = mv;
    }
  };
  fetch(0);
  for (int s0 = 0; s0 < H; s0 += CAND_KS) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
      const int idx = tid + rep * 256;
      const int row = idx >> 3, c4 = (idx & 7) * 4;
      *(int4*)&qs[row * CAND_LD + c4] = pq[rep];
      *(int4*)&ms[row * CAND_LD + c4] = pm[rep];
    }
    __syncthreads();
    if (s0 + CAND_KS < H) fetch(s0 + CAND_KS);
#pragma unroll 1
    for (int s4 = 0; s4 < CAND_KS; s4 += 4) {
      int4 qv[8], mv[8];
#pragma unroll
      for (int i = 0; i < 8; i++) qv[i] = *(const int4*)&qs[(i * 16 + tq) * CAND_LD + s4];
#pragma unroll
      for (int j = 0; j < 8; j++) mv[j] = *(const int4*)&ms[(j * 16 + tm) * CAND_LD + s4];
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          CMP_ACC4(cnt[i][j], qv[i], mv[j]);
        }
    }
    __syncthreads();
  }
  // emit: flag the (rare) pairs that reach --num-min-matches, then handle them one at a time
  unsigned long long hitmask = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (cnt[i][j] >= sp.num_min_matches) hitmask |= 1ULL << (i * 8 + j);                   // MinHashSearch.java:204
  while (hitmask) {
    const int bit = __ffsll((long long)hitmask) - 1;
    hitmask &= hitmask - 1;
    const int i = bit >> 3, j = bit & 7;
    const int qe = qent[i * 16 + tq];
    const int me = m0 + j * 16 + tm;
    if (qe < 0 || me >= ne) continue;
    const int32_t* qm = qmeta + (int64_t)qe * META_W;
    const int32_t* mm = meta + (int64_t)me * META_W;
    if (qm[3] != 0 || mm[3] != 0) continue;   // placeholder entries (skipped strands) are not in the index
    if (!pair_passes(sp, qids[qe], ids[me], qm[2], mm[2])) continue;
    const unsigned long long slot = atomicAdd(cand_count, 1ULL);
    if (slot < cand_cap) { cand[slot].q = qe; cand[slot].m = me; }
  }
}

void launch_candidates(hipStream_t st, const int32_t* minhash, int64_t row_stride, const int32_t* qminhash, int64_t qrow_stride,
                       const int32_t* qlist, int nq, int ne, const int64_t* ids, const int64_t* qids, const int32_t* meta,
                       const int32_t* qmeta, const SearchParams& sp, const long long* rowstart, long long nblocks_tri,
                       Candidate* cand, unsigned long long* cand_count, unsigned long long cand_cap) {
  if (nq <= 0 || ne <= 0) return;
  const int ntq = (nq + CAND_TQ - 1) / CAND_TQ;
  const int ntu = (ne + CAND_TM - 1) / CAND_TM;
  const int triangular = rowstart != nullptr;
  const long long nblocks = triangular ? nblocks_tri : (long long)ntq * ntu;
  if (nblocks <= 0) return;
  const long long nb8 = ((nblocks + 7) / 8) * 8;
  hipLaunchKernelGGL(candidate_kernel, dim3((unsigned)nb8), dim3(256), 0, st, minhash, row_stride, qminhash, qrow_stride, qlist, nq, ne,
                     ids, qids, meta, qmeta, sp, triangular, rowstart, ntq, nblocks, ntu, cand, cand_count, cand_cap);
}

// =============================================================================================
// Inverted index on the GPU (the reference's structure, J/impl/MinHashSearch.java:100-147,161-181: per MinHash slot a map
// value -> list of stored sequences).  Here: per slot, the postings (mix(value), entry) of all stored entries GROUPED BY BUCKET,
// bucket = top bits of mix(value) (fmix32, a bijection: equal values <=> equal mixes, and the mixes are uniform whatever the
// distribution of the minima), plus a table of bucket ends.  One bucket per stored entry (rounded up to a power of two), so a
// bucket holds 0.5-1 postings on average and a lookup is two dependent loads: the bucket's bounds, then its postings — the ones
// whose mix equals the query's are the hits.  A query counts hits per stored entry in an LDS count table; the hit count of a pair
// equals the number of slots with equal values, so the candidate set is identical to the brute-force count.
//
// Built by a two-level counting sort whose only atomics are in LDS (a global atomic costs this chip ~50 ps of its memory side:
// 102 M of them at C2 are 4-5 ms per pass, measured with the one-level counting sort that was tried first, 11.2 ms in all):
//   1. index_tile_kernel<0>  per tile of IB_TE entries x IB_S slots, LDS histogram over the 128 coarse bins (top 7 bits of the mix)
//   2. index_offsets_kernel  per slot: exclusive scan over (bin, tile) -> where each tile's share of each bin starts
//   3. index_tile_kernel<1>  same tiles: postings to their bin (LDS cursors), 256 contiguous bytes per (tile, slot, bin) on average
//   4. index_bins_kernel     per (slot, bin): the bin's ~ne/128 postings grouped by bucket through an LDS histogram, ends written
// A value shared by many entries (a repeat) simply makes its bucket long: the postings are contiguous and a workgroup streams them
// with coalesced loads.  Rounds 1-2 kept an open-addressing table of (value, entry) words per slot with runs of equal values, an
// overflow pool and a per-value overflow table: 4.3 GB at C2 and 34 GB at C4 against 1.4 / 12.5 GB here (+ the sort's scratch),
// every insert a CAS chain into a random line, and — what cost the C5 slice 40 of its 54 ms of index_query — a lookup had to walk
// to the end of its CLUSTER, which the 16-entry runs of popular values made hundreds of words long for some lane of nearly every
// workgroup (670 loads per wave in the second tier: r03 PMC pass).
// =============================================================================================
__device__ __forceinline__ uint32_t inv_mix(uint32_t v) { return fmix32(v); }
// "table elements processed" (MinHashSearch.java:173) of a query, added to the launch's total.  Round 6: every LANE used to add its own count
// to ONE word — a wave instruction of up to 64 atomic operations on one address, one per query, all through one L2 channel: the timing build
// without it ran the first query tier of an 8-GPU rank in 1.03 ms instead of 1.98 (profiles/r06_iq_timing_builds.txt).  Now a wave adds its
// sum once, to one of IQ_ELEM_SPREAD words picked by the workgroup; index_elements_sum_kernel folds them into the counter the host reads.
constexpr int IQ_ELEM_SPREAD = 256;
__device__ __forceinline__ void iq_add_elements(unsigned long long* __restrict__ spread, unsigned long long mine) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(spread + (blockIdx.x & (IQ_ELEM_SPREAD - 1)), mine);
}
__global__ __launch_bounds__(IQ_ELEM_SPREAD) void index_elements_sum_kernel(unsigned long long* __restrict__ spread, unsigned long long* __restrict__ total) {
  unsigned long long v = spread[threadIdx.x];
  spread[threadIdx.x] = 0ULL;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(total, v);
}
int index_elements_spread_words() { return IQ_ELEM_SPREAD; }
// (Round 4 tried an occupancy bitmap in front of the lookups — 4 bits per bucket, a clear bit ends a lookup after one load instead of two
//  dependent ones; 82 % of the lookups of a C2 query find nothing.  It made the first tier SLOWER: 3.3 -> 4.8 ms at C2, 70.8 -> 78.4 at
//  C4, 2.50 -> 2.42 for one rank of eight.  A wave is one query with eight slots per lane; nearly every wave holds some lane whose bit
//  is set, and for those the chain is three dependent loads instead of two — the wave's time is its slowest lane's.  Removed.)

#ifndef MH_IB_BINS_LOG
#define MH_IB_BINS_LOG 7
#endif
constexpr int IB_BINS_LOG = MH_IB_BINS_LOG, IB_BINS = 1 << IB_BINS_LOG;   // coarse bins per slot: a tile's share of a bin is IB_TE / IB_BINS postings = 256
                                                                          // contiguous bytes (512 bins, 64-byte shares: the scatter took 2.6 ms at C2, its
                                                                          // half-written lines leaving the L2 before their neighbours arrived)
constexpr int IB_S = 8;                                      // slots of one workgroup (32 bytes of every MinHash row it reads)
constexpr int IB_TE = 4096;                                  // entries of one tile (index_tile_entries: 1024 for a small index)
constexpr int IB_THREADS = 256;
// Steps 1 and 3.  Slot-group-major grid: consecutive workgroups are the tiles of one group of IB_S slots.
template <bool SCATTER>
__global__ __launch_bounds__(IB_THREADS) void index_tile_kernel(const int32_t* __restrict__ minhash, int64_t row_stride, const int32_t* __restrict__ meta,
                                                                int ne, int H, int tiles, int te, InvIndex ix) {
  __shared__ uint32_t bins[IB_S][IB_BINS];
  const int g = (int)(blockIdx.x / (unsigned)tiles), tile = (int)(blockIdx.x % (unsigned)tiles);
  const int sl = (int)(threadIdx.x % IB_S), s = g * IB_S + sl;
  for (int i = threadIdx.x; i < IB_S * IB_BINS; i += IB_THREADS) {
    const int ss = g * IB_S + i / IB_BINS;
    (&bins[0][0])[i] = (SCATTER && ss < H) ? ix.tile_counts[((size_t)ss * tiles + tile) * IB_BINS + (i % IB_BINS)] : 0u;
  }
  __syncthreads();
  if (s < H) {
    const int e1 = min(ne, (tile + 1) * te);
    for (int e = tile * te + (int)(threadIdx.x / IB_S); e < e1; e += IB_THREADS / IB_S) {
      if (meta[(int64_t)e * META_W + 3] != 0) continue;                  // skipped strands are not stored (addSequence never sees them)
      const uint32_t hv = inv_mix((uint32_t)minhash[(int64_t)e * row_stride + s]);
      const uint32_t at = atomicAdd(&bins[sl][hv >> (32 - IB_BINS_LOG)], 1u);
      if (SCATTER) ix.staged[(size_t)s * ix.slot_stride + at] = make_uint2(hv, (uint32_t)e);
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (int i = threadIdx.x; i < IB_S * IB_BINS; i += IB_THREADS) {
      const int ss = g * IB_S + i / IB_BINS;
      if (ss < H) ix.tile_counts[((size_t)ss * tiles + tile) * IB_BINS + (i % IB_BINS)] = (&bins[0][0])[i];
    }
  }
}

// Step 2, one workgroup per slot, one thread per bin: tile_counts[s][t][bin] becomes the first posting (within the slot) of tile t's
// share of the bin; bin_start[s][bin] (IB_BINS + 1 words) the bins' bounds.
__global__ __launch_bounds__(IB_BINS) void index_offsets_kernel(InvIndex ix, int tiles) {
  __shared__ uint32_t wsum[IB_BINS / 64];
  const int s = blockIdx.x, bin = threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t* c = ix.tile_counts + (size_t)s * tiles * IB_BINS + bin;
  uint32_t total = 0;
  for (int t = 0; t < tiles; t++) total += c[(size_t)t * IB_BINS];
  uint32_t incl = total;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off); if (lane >= off) incl += v; }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  uint32_t run = incl - total;
  for (int w = 0; w < wv; w++) run += wsum[w];
  uint32_t* bs = ix.bin_start + (size_t)s * (IB_BINS + 1);
  bs[bin] = run;
  if (bin == IB_BINS - 1) bs[IB_BINS] = run + total;
  for (int t = 0; t < tiles; t++) { const uint32_t v = c[(size_t)t * IB_BINS]; c[(size_t)t * IB_BINS] = run; run += v; }
}

// Step 6 (round 6): the LINE table.  A lookup through ends / items is two dependent random accesses — the bucket's bounds, then its
// postings — and in an N-GPU job every rank looks ALL N n queries up against its shard: the one term of a rank's step that does not
// shrink with N.  Line l of slot s packs the postings of the buckets [l << lb, (l + 1) << lb) — 3.5 to 7 of them on average — into ONE
// 64-byte line: the first-tier query reads that line and nothing else.  A packed posting is exact: the line index is the mix's top
// nl_log bits, the posting keeps the other 32 - nl_log (the "tag") above the entry's ebits bits — at most 36 bits (index_line_params):
//   words 0..13   the low 32 bits of 14 postings
//   word 14       the top 4 bits of postings 0..7, word 15 bits 0..23 those of postings 8..13
//   word 15 >> 24 the header: n = postings of the line's buckets (0..28), or 0xFF "look the bucket up in ends / items" (a long bucket: a
//                 repeat — or, one line in ten thousand, more than a pair of lines holds)
// n > 14: the postings beyond the 14th sit in the PARTNER line (l ^ 1: the other half of the same 128-byte block), behind the partner's
// own (header of the partner = its own count < 14), as long as the pair's postings fit the pair's 28 places.
constexpr int IL_CAP = 14;
constexpr uint32_t IL_FALLBACK = 0xFFu;
// one line: its own postings P[me_lo .. me_lo + n_me), and — behind them — what its partner line (n_pt postings from pt_lo) cannot hold
__device__ __forceinline__ void il_build_line(const uint2* __restrict__ P, uint32_t me_lo, uint32_t n_me, uint32_t pt_lo, uint32_t n_pt, uint32_t tmask, uint32_t eb,
                                              uint4* __restrict__ out) {
  const bool fits = n_me + n_pt <= 2u * IL_CAP;
  const uint32_t h = n_me <= (uint32_t)IL_CAP ? n_me : (fits ? n_me : IL_FALLBACK);
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = 0u;
  const uint32_t m = n_me < (uint32_t)IL_CAP ? n_me : (uint32_t)IL_CAP;
  const uint32_t x = (fits && n_pt > (uint32_t)IL_CAP) ? n_pt - IL_CAP : 0u;   // postings of the partner that sit here
  if (h != IL_FALLBACK) {
#pragma unroll
    for (int i = 0; i < IL_CAP; i++) {
      uint32_t src = 0xFFFFFFFFu;
      if ((uint32_t)i < m) src = me_lo + (uint32_t)i; else if ((uint32_t)i - m < x) src = pt_lo + (uint32_t)IL_CAP + ((uint32_t)i - m);
      if (src != 0xFFFFFFFFu) {
        const uint2 v = P[src];
        const unsigned long long W = ((unsigned long long)(v.x & tmask) << eb) | (unsigned long long)v.y;
        w[i] = (uint32_t)W;
        if (i < 8) w[14] |= ((uint32_t)(W >> 32) & 15u) << (4 * i); else w[15] |= ((uint32_t)(W >> 32) & 15u) << (4 * (i - 8));
      }
    }
  }
  w[15] |= h << 24;
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// One thread per line.  (Built by index_bins_kernel's own workgroups right after their scatter — the postings come back out of the L2, the
// buckets' ends are the scatter's cursors — it was SLOWER: a (slot, bin) workgroup has 16 to 128 pairs of lines for its 256 threads and every
// thread walks its pair serially: index build 0.91 -> 1.11 ms for a rank of eight at C2, the ordered kernel beside it 0.80 -> 1.0; C1
// 0.148 -> 0.242.  Round 6, EXPERIMENTS.md.)
constexpr int IL_THREADS = 256;
__global__ __launch_bounds__(IL_THREADS) void index_lines_kernel(InvIndex ix, int H) {
  const size_t t = (size_t)blockIdx.x * IL_THREADS + threadIdx.x;
  if (t >= ((size_t)H << ix.nl_log)) return;
  const size_t s = t >> ix.nl_log, l = t & (((size_t)1 << ix.nl_log) - 1);
  const uint32_t* E = ix.ends + s * ((size_t)ix.nb + 1) + ((l & ~(size_t)1) << ix.line_lb);   // ends at the pair's first bucket
  const uint32_t a = E[0], b = E[(size_t)1 << ix.line_lb], c = E[(size_t)2 << ix.line_lb];
  const bool odd = (l & 1) != 0;
  il_build_line(ix.items + s * ix.slot_stride, odd ? b : a, odd ? c - b : b - a, odd ? a : b, odd ? b - a : c - b, 0xFFFFFFFFu >> ix.nl_log, ix.line_ebits,
                (uint4*)(ix.lines + ((s << ix.nl_log) + l) * 16));
}
// Step 4, one workgroup per (slot, bin): the bin's postings move from `staged` to `items` grouped by bucket; the buckets' ends are
// written (ends[s][0] = 0 by bin 0).  sub = buckets per bin (a power of two, <= IB_SUB_MAX).
// (Round 6: the kernel is compiled for three sizes of a bin.  It used to reserve the counters of the largest — 32 KB, five workgroups of 256 lanes
//  per CU — whatever the index: one rank's shard of an 8-GPU job at C2 has 256 buckets and ~195 postings per (slot, bin), 65 536 workgroups of
//  which 1 280 were resident, most of their lanes without a posting: 0.46 ms where an eighth of the one-GPU build is 0.16.  Small bins now take one
//  wavefront and 2 KB each, 32 of them per CU.)
constexpr int IB_SUB_MAX = 8192;
template <int SUBCAP, int IB_FIN_THREADS>
__global__ __launch_bounds__(IB_FIN_THREADS) void index_bins_kernel(InvIndex ix) {
  __shared__ uint32_t cnt[SUBCAP];
  __shared__ uint32_t wsum[IB_FIN_THREADS / 64];
  __shared__ uint32_t s_carry, s_long;
  const int s = blockIdx.x >> IB_BINS_LOG, bin = blockIdx.x & (IB_BINS - 1);
  const int sub = (int)(ix.nb >> IB_BINS_LOG), lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t* bs = ix.bin_start + (size_t)s * (IB_BINS + 1) + bin;
  const uint32_t lo = bs[0], n = bs[1] - lo;
  const uint2* in = ix.staged + (size_t)s * ix.slot_stride + lo;
  uint2* out = ix.items + (size_t)s * ix.slot_stride + lo;
  uint32_t* ends = ix.ends + (size_t)s * ((size_t)ix.nb + 1) + (size_t)bin * sub + 1;
  for (int i = threadIdx.x; i < sub; i += IB_FIN_THREADS) cnt[i] = 0;
  if (threadIdx.x == 0) { s_carry = 0; s_long = 0; if (bin == 0) ends[-1] = 0; }
  __syncthreads();
  const uint32_t smask = (uint32_t)sub - 1u;
  for (uint32_t i = threadIdx.x; i < n; i += IB_FIN_THREADS) atomicAdd(&cnt[(in[i].x >> ix.shift) & smask], 1u);
  __syncthreads();
  for (int base = 0; base < sub; base += IB_FIN_THREADS) {      // exclusive scan of the counts; ends = inclusive
    const int i = base + (int)threadIdx.x;
    const uint32_t v = i < sub ? cnt[i] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t before = s_carry, total = 0;
    for (int w = 0; w < IB_FIN_THREADS / 64; w++) { const uint32_t t = wsum[w]; if (w < wv) before += t; total += t; }
    if (i < sub) { cnt[i] = before + incl - v; ends[i] = lo + before + incl; }
    if (v > ix.group_t) s_long = 1;                      // (benign race: every writer stores 1)
    __syncthreads();
    if (threadIdx.x == 0) s_carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) ix.bin_long[blockIdx.x] = s_long;
  for (uint32_t i = threadIdx.x; i < n; i += IB_FIN_THREADS) {
    const uint2 x = in[i];
    out[atomicAdd(&cnt[(x.x >> ix.shift) & smask], 1u)] = x;
  }
}

// Step 5, large indexes only (index_wants_grouping): the postings of every LONG bucket — a value thousands of stored entries share:
// a repeat — are put in ascending order of their entry CLASS (entry >> IB_CLASS_LOG), by a counting sort through `staged` (free again
// once step 4 is done).  The dense query tier counts hits in passes over ranges of stored entries; before this step every pass of a
// repeat-carrying query streamed all of its buckets again and kept the 1/npass that fell into its range (39 passes at 1.25 M entries:
// 14 us per query on one rank's share of BASELINE configs[4], 560 ms for 40 000 queries); with the classes in order a pass finds its
// part of a long bucket by bisection and streams only that.  One workgroup per (slot, bin), like step 4.
constexpr int IB_GROUP_T = 256;          // buckets up to this long are streamed whole by every pass (InvIndex::group_t)
constexpr int IB_CLASS_LOG = 15;         // entries per class: the finest range a dense pass covers (ranges are power-of-two multiples of it; InvIndex::class_log)
constexpr int IB_MAX_CLASSES = 8192;     // class counters in LDS (2^28 entries); an index beyond that is left ungrouped
constexpr int IB_GRP_THREADS = 256;
__global__ __launch_bounds__(IB_GRP_THREADS) void index_group_kernel(InvIndex ix) {
  __shared__ uint32_t cls[IB_MAX_CLASSES];
  __shared__ uint32_t longs[IB_GRP_THREADS];
  __shared__ uint32_t s_nlong, s_carry;
  __shared__ uint32_t wsum[IB_GRP_THREADS / 64];
  const int s = blockIdx.x >> IB_BINS_LOG, bin = blockIdx.x & (IB_BINS - 1);
  const int sub = (int)(ix.nb >> IB_BINS_LOG), lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t* ends = ix.ends + (size_t)s * ((size_t)ix.nb + 1) + (size_t)bin * sub;   // ends[i] .. ends[i + 1]: bucket i of this bin
  uint2* items = ix.items + (size_t)s * ix.slot_stride;
  uint2* staged = ix.staged + (size_t)s * ix.slot_stride;
  const int nclass = (int)((ix.ne + (1u << ix.class_log) - 1) >> ix.class_log);
  if (!ix.bin_long[blockIdx.x]) return;                  // (step 4 saw no long bucket in this bin: nearly every bin of an index without repeats)
  for (int w0 = 0; w0 < sub; w0 += IB_GRP_THREADS) {     // windows of IB_GRP_THREADS buckets: the list of a window's long ones cannot overflow
    __syncthreads();
    if (threadIdx.x == 0) s_nlong = 0;
    __syncthreads();
    {
      const int i = w0 + (int)threadIdx.x;
      if (i < sub && ends[i + 1] - ends[i] > ix.group_t) longs[atomicAdd(&s_nlong, 1u)] = (uint32_t)i;
    }
    __syncthreads();
    const uint32_t nl = s_nlong;
    for (uint32_t li = 0; li < nl; li++) {
      const uint32_t b = longs[li], lo = ends[b], n = ends[b + 1] - lo;
      for (int c = threadIdx.x; c < nclass; c += IB_GRP_THREADS) cls[c] = 0;
      if (threadIdx.x == 0) s_carry = 0;
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < n; i += IB_GRP_THREADS) {
        const uint2 x = items[lo + i];
        staged[lo + i] = x;
        atomicAdd(&cls[x.y >> ix.class_log], 1u);
      }
      __syncthreads();
      for (int cb = 0; cb < nclass; cb += IB_GRP_THREADS) {        // exclusive scan of the class counts
        const int c = cb + (int)threadIdx.x;
        const uint32_t v = c < nclass ? cls[c] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t before = s_carry, total = 0;
        for (int w = 0; w < IB_GRP_THREADS / 64; w++) { const uint32_t t = wsum[w]; if (w < wv) before += t; total += t; }
        if (c < nclass) cls[c] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += total;
        __syncthreads();
      }
      for (uint32_t i = threadIdx.x; i < n; i += IB_GRP_THREADS) {
        const uint2 x = staged[lo + i];
        items[lo + atomicAdd(&cls[x.y >> ix.class_log], 1u)] = x;
      }
      __syncthreads();
    }
  }
}

// lines per slot and posting layout for an index of `entries` entries in nb buckets per slot; false: no line table (a posting would not
// fit its 36 bits, or MHAP_INDEX_LINES=0).  Average postings per line: 3.5 .. 7 (IL_CAP = 14 places, 28 with the partner's)
bool index_line_params(int64_t entries, uint32_t nb, uint32_t& nl_log, uint32_t& lb, uint32_t& ebits) {
  // Default: a line table for an index of 8 192 to 262 144 entries.  Measured in round 6 (EXPERIMENTS.md; hits queued, one build thread per
  // line), index_query / index_build / step, without -> with: rank of eight at C2 (25 000 entries) 1.47 -> 1.03 / 0.78 -> 0.88 / 14.9 -> 14.4 ms;
  // C2 on one GPU (200 000) 2.29 -> 1.79 / 2.94 -> 3.80 / 95.2 -> 94.8 (even: the build runs next to the ordered kernel); rank of eight at C4
  // (250 000) 19.4 -> 11.7 / 3.7 -> 5.0 / 194 -> 189; C4 on one GPU (2 M entries: the middle tier) 76.8 -> 73.5 / 35.3 -> 44.4 / 1466 -> 1471;
  // one rank's share of configs[4] (1.25 M, repeat-rich: the dense tier) 392 -> 405 / 23 -> 29 / 3314 -> 3362; C1 (2 000) 0.041 -> 0.051 /
  // 0.139 -> 0.147 / 0.93 -> 0.98.  So: the sizes where the first tier does the work and the table stays near the caches.
  // MHAP_INDEX_LINES=0 / 1: never / whenever the layout allows.
  const char* e = getenv("MHAP_INDEX_LINES");
  if (e && e[0] == '0') return false;
  if (!(e && e[0] == '1') && (entries < 8192 || entries > 262144)) return false;
  const int64_t per_line = []() { const char* v = getenv("MHAP_INDEX_LINE_LOAD"); const int x = v ? atoi(v) : 0; return (int64_t)(x >= 1 && x <= 14 ? x : 7); }();
  uint32_t nbl = 0;
  while ((1u << nbl) < nb) nbl++;
  uint32_t l = 1;
  while (l < nbl && ((int64_t)1 << l) * per_line < entries) l++;
  if (((int64_t)1 << l) * per_line < entries) return false;          // (an index beyond nb * 7 entries: more than 7 M)
  ebits = 1;
  while (ebits < 32 && ((int64_t)1 << ebits) < entries) ebits++;     // entry < entries <= 2^ebits
  if ((32 - l) + ebits > 36) return false;
  nl_log = l; lb = nbl - l;
  return true;
}
size_t index_line_bytes(int H, uint32_t nl_log) { return (size_t)H * ((size_t)64 << nl_log); }

// Self-check (MHAP_DEBUG_INDEX=1, tests): every stored (entry, slot) finds its posting in its bucket; counts the ones that do not.
__global__ void index_verify_kernel(const int32_t* __restrict__ minhash, int64_t row_stride, const int32_t* __restrict__ meta, int ne, int H, InvIndex ix,
                                    unsigned long long* __restrict__ missing) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)ne * H) return;
  const int e = (int)(t / H), s = (int)(t % H);
  if (meta[(int64_t)e * META_W + 3] != 0) return;
  const uint32_t hv = inv_mix((uint32_t)minhash[(int64_t)e * row_stride + s]);
  const uint32_t* E = ix.ends + (size_t)s * ((size_t)ix.nb + 1) + (hv >> ix.shift);
  const uint2* P = ix.items + (size_t)s * ix.slot_stride;
  // (a grouped index: a long bucket's postings are in ascending class order — checked from this entry's own posting backwards)
  for (uint32_t i = E[0]; i < E[1]; i++)
    if (P[i].x == hv && P[i].y == (uint32_t)e) {
      if (ix.grouped && E[1] - E[0] > ix.group_t && i > E[0] && (P[i - 1].y >> ix.class_log) > (P[i].y >> ix.class_log)) break;
      return;
    }
  atomicAdd(missing, 1ULL);
}
// ... and, through the line table, by the rules of the first query tier (a line that says "see the buckets" counts as found)
__global__ void index_verify_lines_kernel(const int32_t* __restrict__ minhash, int64_t row_stride, const int32_t* __restrict__ meta, int ne, int H, InvIndex ix,
                                          unsigned long long* __restrict__ missing) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)ne * H) return;
  const int e = (int)(t / H), s = (int)(t % H);
  if (meta[(int64_t)e * META_W + 3] != 0) return;
  const uint32_t hv = inv_mix((uint32_t)minhash[(int64_t)e * row_stride + s]);
  const uint32_t lsh = 32u - ix.nl_log, tmask = 0xFFFFFFFFu >> ix.nl_log, eb = ix.line_ebits, emk = (1u << eb) - 1u;
  const uint32_t line = hv >> lsh;
  const uint32_t* L = ix.lines + (((size_t)s << ix.nl_log) + line) * 16;
  const uint32_t hdr = L[15] >> 24;
  if (hdr == IL_FALLBACK) return;
  int found = 0;
  auto scan = [&](const uint32_t* W, uint32_t from, uint32_t to) {
    for (uint32_t i = from; i < to && i < (uint32_t)IL_CAP; i++) {
      const uint32_t nib = ((i < 8 ? W[14] >> (4 * i) : W[15] >> (4 * (i - 8))) & 15u);
      const unsigned long long P = ((unsigned long long)nib << 32) | W[i];
      if ((uint32_t)(P >> eb) == (hv & tmask) && (uint32_t)(P & emk) == (uint32_t)e) found++;
    }
  };
  scan(L, 0u, hdr);
  if (hdr > (uint32_t)IL_CAP) {
    const uint32_t* Q = ix.lines + (((size_t)s << ix.nl_log) + (line ^ 1u)) * 16;
    const uint32_t pn = Q[15] >> 24;
    if (pn >= (uint32_t)IL_CAP) { atomicAdd(missing, 1ULL); return; }
    scan(Q, pn, pn + hdr - (uint32_t)IL_CAP);
  }
  if (found != 1) atomicAdd(missing, 1ULL);
}
void launch_index_verify(hipStream_t st, const int32_t* minhash, int64_t row_stride, const int32_t* meta, int ne, int H, const InvIndex& ix,
                         unsigned long long* missing) {
  const long long total = (long long)ne * H;
  if (total <= 0) return;
  hipLaunchKernelGGL(index_verify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, minhash, row_stride, meta, ne, H, ix, missing);
  if (ix.lines)
    hipLaunchKernelGGL(index_verify_lines_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, minhash, row_stride, meta, ne, H, ix, missing);
}

// (Round 6: a small index — one rank's shard of an 8-GPU job at C2: 25 000 entries — had 7 tiles x 64 slot groups = 448 workgroups for 256 CUs;
//  tiles of 1 024 entries give it four times as many.  Their shares of a bin are 64 bytes: the scatter's half-written lines — what made 512
//  coarse bins slow at C2 — meet in the caches while the whole staging area is a few hundred MB.)
static int index_tile_entries(int ne) {
  static const int force = []() { const char* e = getenv("MHAP_INDEX_TILE"); const int v = e ? atoi(e) : 0; return v >= 256 && v <= IB_TE ? v : 0; }();
  return force ? force : (ne <= 65536 ? 1024 : IB_TE);
}
int index_tiles(int ne) { const int te = index_tile_entries(ne); return (ne + te - 1) / te; }
int index_coarse_bins() { return IB_BINS; }
int index_max_buckets_log() { return IB_BINS_LOG + 13; }   // IB_SUB_MAX buckets per coarse bin
// (re)build the index for entries [0, ne): ix.ends / items / staged / tile_counts / bin_start sized by the caller (index_tiles)
void launch_index_build(hipStream_t st, const int32_t* minhash, int64_t row_stride, const int32_t* meta, int ne, int H, const InvIndex& ix) {
  if ((int64_t)ne * H <= 0) return;
  const int tiles = index_tiles(ne), te = index_tile_entries(ne);
  const unsigned grid = (unsigned)tiles * (unsigned)((H + IB_S - 1) / IB_S);
  hipLaunchKernelGGL(index_tile_kernel<false>, dim3(grid), dim3(IB_THREADS), 0, st, minhash, row_stride, meta, ne, H, tiles, te, ix);
  hipLaunchKernelGGL(index_offsets_kernel, dim3((unsigned)H), dim3(IB_BINS), 0, st, ix, tiles);
  hipLaunchKernelGGL(index_tile_kernel<true>, dim3(grid), dim3(IB_THREADS), 0, st, minhash, row_stride, meta, ne, H, tiles, te, ix);
  {
    const int sub = (int)(ix.nb >> IB_BINS_LOG);
    const dim3 g((unsigned)H << IB_BINS_LOG);
    static const int force = []() { const char* e = getenv("MHAP_INDEX_BINS_SHAPE"); return e ? atoi(e) : -1; }();   // 0 / 1 / 2: pin a shape that fits (A/B)
    const int shape = force >= 0 ? force : (sub <= 512 ? 0 : (sub <= 2048 ? 1 : 2));
    if (shape == 0 && sub <= 512) hipLaunchKernelGGL((index_bins_kernel<512, 64>), g, dim3(64), 0, st, ix);
    else if (shape <= 1 && sub <= 2048) hipLaunchKernelGGL((index_bins_kernel<2048, 256>), g, dim3(256), 0, st, ix);
    else hipLaunchKernelGGL((index_bins_kernel<IB_SUB_MAX, 256>), g, dim3(256), 0, st, ix);
  }
  if (ix.grouped) hipLaunchKernelGGL(index_group_kernel, dim3((unsigned)H << IB_BINS_LOG), dim3(IB_GRP_THREADS), 0, st, ix);
  if (ix.lines) {
    const size_t threads = (size_t)H << ix.nl_log;
    hipLaunchKernelGGL(index_lines_kernel, dim3((unsigned)((threads + IL_THREADS - 1) / IL_THREADS)), dim3(IL_THREADS), 0, st, ix, H);
  }
}
// (called when the index is sized: an index the compact dense tier covers in one pass gains nothing from the order, and the class
//  counters bound the size from above.  MHAP_INDEX_GROUP=0|1 never / always, MHAP_INDEX_GROUP_T, MHAP_INDEX_CLASS_LOG: tests, which
//  have to make a few hundred entries look like a million)
void index_group_params(int64_t entries, InvIndex& ix) {
  // (read at every build: tests switch them inside one process)
  const int force = []() { const char* e = getenv("MHAP_INDEX_GROUP"); return e ? atoi(e) : -1; }();
  const int gt = []() { const char* e = getenv("MHAP_INDEX_GROUP_T"); const int v = e ? atoi(e) : 0; return v > 0 ? v : IB_GROUP_T; }();
  const int cl = []() { const char* e = getenv("MHAP_INDEX_CLASS_LOG"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 24 ? v : IB_CLASS_LOG; }();
  ix.group_t = (uint32_t)gt; ix.class_log = (uint32_t)cl;
  const bool fits = entries <= ((int64_t)IB_MAX_CLASSES << cl);
  ix.grouped = (fits && (force >= 0 ? force != 0 : entries > (1LL << 18))) ? 1u : 0u;   // (more than two passes of the compact dense tier)
}

// One workgroup (one wavefront) per query.  LDS: tbl[CT], a hit-count table of packed words: entry + 1 in the low `ebits` bits (as
// many as the index needs), its saturating hit count in the bits above.
#ifndef MH_IQ_THREADS
#define MH_IQ_THREADS 64
#endif
constexpr int IQ_THREADS = MH_IQ_THREADS;   // lanes per query: 64 / 128 / 256 at C2: 4.04 / 4.30 / 6.35 ms with a 2048-entry two-word table, 3.3 with the
                                  // packed one (one rank of eight: 2.5 / 3.4 / 5.7): one wavefront's barriers are free and its 8 KB let 15 of them share a CU
constexpr int IQ_STACK = 48;      // pending (prefix, bits) parts of a query whose hit set is being split
#ifndef MH_IQ_SPT
#define MH_IQ_SPT 1   // 1 / 2 / 4 at C2 with 128 lanes: 4.7 / 4.7 / 5.7 ms; with 64: 4.08 / 4.07 / 4.44 (the queue's LDS costs resident workgroups)
#endif
constexpr int IQ_SPT = MH_IQ_SPT;   // slots a lane of the first tier looks up per trip (their loads are in flight together)
constexpr int IQ_INLINE = 16;     // a bucket up to this long is read by the lane that looked it up; longer ones are streamed by the workgroup
#ifndef MH_IQ_BIG_CT
#define MH_IQ_BIG_CT 16384   // (0: no further tiers — every hit set that outgrows the first tier's table is split there)
#endif
// Tiers (launch_index_query).  <INV_CT, IQ_THREADS> takes every query; one whose buckets hold more postings than the table can
// count (repeats: thousands of stored entries share a MinHash value with the query), or whose distinct hits outgrow the table, is
// appended to `big` and handed on: to the dense tier (index_query_dense_kernel), on a large index through <INV_CT_MID,
// IQ_THREADS_MID> first.  big == nullptr (MHAP_INDEX_TIERS=1, tests): the hit set is split into hash-partition passes over the
// stored entries instead (split in two until every part fits), which bounds a query's cost by its own postings.
#ifndef MH_IQ_TIMING
#define MH_IQ_TIMING 0   // timing builds of the first query tier (results wrong): 1 no `elements` atomic, 2 no hit counting, 4 no global atomic in the emit
#endif
constexpr int IQ_OV = 64;    // line mode: slots of one query that may fall back to ends / items (more: the query is handed on)
#ifndef MH_IQ_QUAD
#define MH_IQ_QUAD 0   // line mode: 1 = four lanes share a line (one 16-byte quarter each), 0 = a line per lane (four 16-byte loads).  Measured (round 6,
                       // emulated rank of eight at C2 / C2 on one GPU / rank of eight at C4): see EXPERIMENTS.md
#endif
#ifndef MH_IQ_LB
#define MH_IQ_LB 4
#endif
constexpr int IQ_LB = MH_IQ_LB;   // line mode: lines a lane has in flight
constexpr int IQ_PRE = 12;   // slots per lane whose first look is kept (H <= 768 with 64 lanes; twelve elements: the compiler addresses such a vector by a uniform index)
typedef int iq_pre_t __attribute__((ext_vector_type(IQ_PRE)));
template <int INV_CT, int IQ_THREADS, int SPT, bool LINES>
__global__ __launch_bounds__(IQ_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8))) void index_query_kernel(InvIndex ix, const int32_t* __restrict__ qminhash, int64_t qrow_stride,
                                                          const int32_t* __restrict__ qlist, int nq, const int64_t* __restrict__ ids,
                                                          const int64_t* __restrict__ qids, const int32_t* __restrict__ meta,
                                                          const int32_t* __restrict__ qmeta, SearchParams sp,
                                                          Candidate* __restrict__ cand, unsigned long long* __restrict__ cand_count,
                                                          unsigned long long cand_cap, unsigned long long* __restrict__ split_count,
                                                          unsigned long long* __restrict__ elements, int32_t* __restrict__ big,
                                                          unsigned long long* __restrict__ big_count) {
  __shared__ uint32_t tbl[INV_CT];
  static_assert(INV_CT <= 32 * IQ_THREADS, "the emit mask holds 32 table slots per lane");
  __shared__ uint32_t s_distinct, s_over, s_top, s_prefix, s_bits;
  __shared__ uint32_t stack[2 * IQ_STACK];
  __shared__ uint32_t s_nseg[2];
  constexpr int QCAP = IQ_THREADS * SPT;   // a trip looks SPT slots per lane up (their loads in flight together) and may queue as many buckets
  __shared__ uint32_t segraw[QCAP * 4];
  uint2* const seglist = (uint2*)segraw;  // queued long buckets: (first posting within the slot, length),
  uint2* const segkey = seglist + QCAP;   // ... (slot, the query's mix there)
  uint32_t* const hitq = segraw;          // (line mode, before any bucket is queued: the entries whose tag matched, waiting to be counted)
  constexpr int IQ_HQ = QCAP * 4;
  __shared__ unsigned long long segpre[QCAP + 1];
  __shared__ unsigned long long wsum[IQ_THREADS / 64];
  __shared__ unsigned long long s_base;
  const int qi = blockIdx.x;
  if (qi >= nq) return;
  const int qe = qlist[qi];
  const int32_t* qm = qmeta + (int64_t)qe * META_W;
  if (qm[3] != 0) return;   // (a strand that was not sketched is no query: a list made on the device — launch_query_iota — names every row)
  const int64_t qid = qids[qe];
  const int qlen = qm[2];
  const int32_t* qrow = qminhash + (int64_t)qe * qrow_stride;
  const size_t eper = (size_t)ix.nb + 1;
  // (what the first look at the buckets found — the mixed value, the bucket's start and length of the lane's first IQ_PRE slots — stays in
  //  registers for the counting loop below, which used to load the query's value and the bucket's bounds a second time: two of the three
  //  dependent round trips of each of its trips.  C2 3.29 -> 2.87 ms, a rank of eight 2.65 -> 2.17.  Requesting the next trip's postings a
  //  trip ahead on top of it — with the mixed value recomputed, or the kernel drops to three waves per SIMD — gave nothing: 3.03 / 2.18)
  iq_pre_t pre_hv = 0, pre_lo = 0, pre_n = 0;
  bool pre_ok = false;
  // round 6: the index has a LINE table (index_lines_kernel) and this launch hands its large hit sets on: one 64-byte line per lookup
  // instead of bucket bounds + postings; only slots whose line says "long bucket" (a repeat) go through ends / items below
  constexpr bool lmode = LINES;   // (launch_index_query: big != nullptr && ix.lines != nullptr)
  __shared__ uint32_t s_nov;
  __shared__ uint32_t ovlist[IQ_OV];
  if (big != nullptr && !lmode) {
    // first tier: the buckets' lengths alone say whether this table can hold the hits — a repeat-rich query is handed over after
    // H loads instead of after counting until the table overflows
    unsigned long long tot = 0;
    // (two loops: the first IQ_PRE trips write the kept vectors with STATIC element indices.  As one loop with `if (itp < IQ_PRE)` the
    //  element write was compiled to an indexed register move executed for every trip — beyond --num-hashes 768 it wrote past the vectors
    //  into whatever registers followed, and from --num-hashes 2 752 on it hit a live pointer: a memory fault in this kernel, found
    //  by a parameter sweep in round 5)
#pragma unroll
    for (int itp = 0; itp < IQ_PRE; itp++) {
      const int s = (int)threadIdx.x + itp * IQ_THREADS;
      if (s < sp.H) {
        const uint32_t hvp = inv_mix((uint32_t)qrow[s]);
        const uint32_t* E = ix.ends + (size_t)s * eper + (hvp >> ix.shift);
        const uint32_t e0 = E[0], e1 = E[1];
        tot += e1 - e0;
        pre_hv[itp] = (int)hvp; pre_lo[itp] = (int)e0; pre_n[itp] = (int)(e1 - e0);
      }
    }
    for (int s = (int)threadIdx.x + IQ_PRE * IQ_THREADS; s < sp.H; s += IQ_THREADS) {
      const uint32_t hvp = inv_mix((uint32_t)qrow[s]);
      const uint32_t* E = ix.ends + (size_t)s * eper + (hvp >> ix.shift);
      tot += E[1] - E[0];
    }
    pre_ok = true;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = tot;
    __syncthreads();
    tot = 0;
    for (unsigned w = 0; w < IQ_THREADS / 64; w++) tot += wsum[w];
    // (tot counts every posting of the H buckets: the hits plus ~ne/nb strangers per bucket; distinct hits <= hits.  Beyond a
    // quarter over the table's capacity the count would very likely overflow half-way and be thrown away)
    // (in a self search the query's own strand is stored: its H postings are ONE distinct entry — without this term every query of a run at
    //  --num-hashes 2048 was handed over on its own postings alone: index query 4.9 ms for 20 000 queries where the first tier takes 2.4)
    const unsigned long long strangers = (unsigned long long)sp.H * ix.ne / ix.nb + (sp.to_self ? (unsigned long long)sp.H : 0ULL);
    if (tot > strangers + (5ULL * (INV_CT * 3 / 4)) / 4) {
      if (threadIdx.x == 0) big[atomicAdd(big_count, 1ULL)] = qe;
      return;
    }
  }
  if (threadIdx.x == 0) { stack[0] = 0; stack[1] = 0; s_top = 1; }
  __syncthreads();
  for (;;) {
    // ---- next part of the stored entries: those whose hash has `bits` low bits equal to `prefix` (the whole index first) ----
    if (threadIdx.x == 0) {
      if (s_top == 0) s_bits = 0xFFFFFFFFu;
      else { s_top--; s_prefix = stack[2 * s_top]; s_bits = stack[2 * s_top + 1]; }
      s_distinct = 0; s_over = 0; s_nseg[0] = 0; s_nseg[1] = 0; s_nov = 0;
    }
    for (int j = threadIdx.x; j < INV_CT; j += IQ_THREADS) tbl[j] = 0;
    __syncthreads();
    const uint32_t bits = s_bits, prefix = s_prefix;
    if (bits == 0xFFFFFFFFu) break;
    constexpr int CT_LOG = __builtin_ctz((unsigned)INV_CT), MAX_BITS = 32 - CT_LOG;   // part bits sit above the home-slot bits
    const uint32_t pmask = bits >= (uint32_t)MAX_BITS ? (0xFFFFFFFFu >> CT_LOG) : ((1u << bits) - 1u);
    unsigned long long mine = 0;
    // count one hit of stored entry `me` (the id/length rules do not depend on the count: they are applied to the few entries that
    // reach numMinMatches, below, so that the lookup loop's only global loads are the index words)
    // The count is read before it is raised and left alone from `sat` on, so it passes `sat` by at most the adds in flight — one per
    // lane of the workgroup — and the field must hold sat + IQ_THREADS: index_query_tier_ok() keeps a launch out of this kernel when
    // the bits the entries leave are too few for that (the middle tier is given 10 bits, i.e. an index below 2^22 entries).
    const int ebits = 32 - __builtin_clz(ix.ne | 1u);                   // entry + 1 <= ne fits
    const uint32_t emask = ebits >= 32 ? 0xFFFFFFFFu : ((1u << ebits) - 1u), cone = ebits >= 32 ? 0u : (1u << ebits);
    const uint32_t cmax = ebits >= 32 ? 0u : (0xFFFFFFFFu >> ebits);
    const uint32_t sat = (uint32_t)sp.num_min_matches < cmax - (uint32_t)IQ_THREADS ? (uint32_t)sp.num_min_matches : cmax - (uint32_t)IQ_THREADS;
    auto count_hit = [&](int me) {
      if (MH_IQ_TIMING & 2) return;
      // (a self search of one index: the query's own strand is stored — H hits on one table word per query, six in seven of all the hits a
      //  C2 query counts, for the one pair the id rule drops anyway)
      if (sp.own_queries && me == qe) return;
      const uint32_t hm = inv_mix((uint32_t)me);
      if (((hm >> CT_LOG) & pmask) != prefix) return;
      const uint32_t id = (uint32_t)me + 1u;
      uint32_t slot = hm & (INV_CT - 1);
      for (int tries = 0; tries < INV_CT; tries++) {
        uint32_t w = __hip_atomic_load(&tbl[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((w & emask) == 0) {
          if (__hip_atomic_load(&s_distinct, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (INV_CT * 3) / 4) { s_over = 1; break; }
          const uint32_t old = atomicCAS(&tbl[slot], 0u, id);
          if (old == 0) { atomicAdd(&s_distinct, 1u); w = id; } else w = old;
        }
        if ((w & emask) == id) { if ((w >> ebits) < sat) atomicAdd(&tbl[slot], cone); break; }
        slot = (slot + 1) & (INV_CT - 1);
      }
    };
    bool handed_over = false;
    int nloop = sp.H;
#if MH_IQ_QUAD
    if (lmode) {
      const uint32_t lsh = 32u - ix.nl_log, tmask = 0xFFFFFFFFu >> ix.nl_log, eb = ix.line_ebits, emk = (1u << eb) - 1u, hsh = 32u - eb;
      const size_t lstride = (size_t)16 << ix.nl_log;   // words of one slot's lines
      // FOUR lanes per lookup, one 16-byte quarter of the line each: a wave's load instruction then touches 16 lines instead of 64 —
      // tools/line_gather_probe.hip: the same 51.2 M lookups out of a 17-GB table (all of C4 on one GPU) take 2.6 ms with a line per lane
      // and 1.06 ms with a line per quad; no difference at 0.13 / 1.1 GB (0.9 ms) — and a lookup in flight costs 4 registers, not 16.
      // Lane 4 g + p holds places 4 p .. 4 p + 3 of group g's line (p = 3: places 12, 13, then the words with the places' top bits
      // and the header, which the other three lanes fetch through a quad broadcast).
      const int part = (int)(threadIdx.x & 3u), grp = (int)(threadIdx.x >> 2);
      constexpr int GPI = IQ_THREADS / 4;     // lookups one wave instruction covers
      auto bcast3 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xFF, 0xF, 0xF, true); };   // quad_perm [3,3,3,3]
      // places [from, to) of the line quarter q (of the lane's part) against the query's tag
      auto scan = [&](const uint4 q, uint32_t w14, uint32_t w15, uint32_t from, uint32_t to, uint32_t qt) {
        const uint32_t nibs = ((part & 2) ? w15 : w14) >> ((part & 1) * 16);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int jx = 0; jx < 4; jx++) {
          const uint32_t place = (uint32_t)(part * 4 + jx);
          const uint32_t tag = (w[jx] >> eb) | (((nibs >> (4 * jx)) & 15u) << hsh);
          if (place < (uint32_t)IL_CAP && place >= from && place < to && tag == qt) { mine++; count_hit((int)(w[jx] & emk)); }
        }
      };
      for (int sb = 0; sb < sp.H; sb += IQ_LB * GPI) {      // (workgroup-uniform bounds; no barrier inside)
        uint4 L[IQ_LB];
        uint32_t hvv[IQ_LB];
#pragma unroll
        for (int u = 0; u < IQ_LB; u++) {
          const int s = sb + u * GPI + grp;
          hvv[u] = 0; L[u] = make_uint4(0u, 0u, 0u, 0u);
          if (s < sp.H) {
            hvv[u] = inv_mix((uint32_t)qrow[s]);
            L[u] = ((const uint4*)(ix.lines + (size_t)s * lstride + ((size_t)(hvv[u] >> lsh) << 4)))[part];
          }
        }
        uint32_t more = 0;   // bit u: lookup u goes on in its partner line
        uint32_t nn[IQ_LB];
#pragma unroll
        for (int u = 0; u < IQ_LB; u++) {
          const int s = sb + u * GPI + grp;
          const uint32_t w14 = bcast3(L[u].z), w15 = bcast3(L[u].w), hdr = w15 >> 24;
          nn[u] = hdr;
          if (s < sp.H && hdr) {
            if (hdr == IL_FALLBACK) {
              if (part == 0) { const uint32_t at = atomicAdd(&s_nov, 1u); if (at < (uint32_t)IQ_OV) ovlist[at] = (uint32_t)s; }
            } else {
              scan(L[u], w14, w15, 0u, hdr, hvv[u] & tmask);
              if (hdr > (uint32_t)IL_CAP) more |= 1u << u;
            }
          }
        }
        if (__builtin_amdgcn_ballot_w64(more != 0u)) {
          // what a line of more than 14 postings could not hold sits in its partner line, behind the partner's own
#pragma unroll
          for (int u = 0; u < IQ_LB; u++)
            if ((more >> u) & 1u)
              L[u] = ((const uint4*)(ix.lines + (size_t)(sb + u * GPI + grp) * lstride + ((size_t)((hvv[u] >> lsh) ^ 1u) << 4)))[part];
#pragma unroll
          for (int u = 0; u < IQ_LB; u++) {
            // (the broadcasts run for the whole wave: a DPP read of a lane that is switched off returns nothing useful)
            const uint32_t w14 = bcast3(L[u].z), w15 = bcast3(L[u].w);
            if ((more >> u) & 1u) { const uint32_t pn = w15 >> 24; scan(L[u], w14, w15, pn, pn + nn[u] - (uint32_t)IL_CAP, hvv[u] & tmask); }
          }
        }
      }
#else
    if (lmode) {
      // (the queue is per WAVE — ballots compact within a wave — so the 256-lane middle tier runs the same code: IQ_HQ / waves words each)
      const uint32_t lsh = 32u - ix.nl_log, tmask = 0xFFFFFFFFu >> ix.nl_log, eb = ix.line_ebits, emk = (1u << eb) - 1u, hsh = 32u - eb;
      const size_t lstride = (size_t)16 << ix.nl_log;   // words of one slot's lines
      // Hits are QUEUED, not counted where they are found: a match is one lane in a hundred per place, and counting it on the spot sent the
      // whole wave through the hit-count table's probe loop (LDS atomics, their waits) 14 places x 8 lookups = 112 times a query — the timing
      // build without the counting ran C2's first tier in 1.05 ms against 2.26 with it (round 6).  The scan is executed by all lanes together
      // (a lane without a line scans the empty range), a ballot compacts the matching entries into hitq, and the queue is counted by a loop in
      // which every lane has a hit to count.
      int hbase = 0;                 // hits waiting in this wave's queue (wave-uniform)
      unsigned long long nhits = 0;  // ... and all hits so far ("table elements processed")
      constexpr int HQW = IQ_HQ / (IQ_THREADS / 64);                 // words of one wave's queue (256)
      static_assert(HQW >= 128, "a place adds up to 64 hits to a queue that is flushed beyond HQW - 64");
      uint32_t* const hq = hitq + (threadIdx.x >> 6) * HQW;
      const int wlane = (int)(threadIdx.x & 63u);
      auto flush_hits = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int i = wlane; i < hbase; i += 64) count_hit((int)hq[i]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        hbase = 0;
      };
      // a posting of place i of line words L: tag above the entry's bits; the place's top 4 bits sit in words 14 / 15
#define IQ_LINE_SCAN(L, from, to, qtag)                                                                                   \
      {                                                                                                                     \
        const uint32_t lw_[14] = {L[0].x, L[0].y, L[0].z, L[0].w, L[1].x, L[1].y, L[1].z, L[1].w, L[2].x, L[2].y, L[2].z, L[2].w, L[3].x, L[3].y}; \
        _Pragma("unroll") for (int i_ = 0; i_ < IL_CAP; i_++) {                                                              \
          const uint32_t nib_ = ((i_ < 8 ? L[3].z >> (4 * i_) : L[3].w >> (4 * (i_ - 8))) & 15u);                            \
          const uint32_t tag_ = (lw_[i_] >> eb) | (nib_ << hsh);                                                             \
          const bool hit_ = (uint32_t)i_ >= (from) && (uint32_t)i_ < (to) && tag_ == (qtag);                                  \
          const unsigned long long bal_ = __builtin_amdgcn_ballot_w64(hit_);                                                  \
          if (bal_) {                                                                                                        \
            if (hit_) hq[hbase + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal_, 0u))] = lw_[i_] & emk; \
            const int np_ = __popcll(bal_);                                                                                  \
            hbase += np_; nhits += (unsigned long long)np_;                                                                  \
            if (hbase > HQW - 64) flush_hits();                                                                              \
          }                                                                                                                 \
        }                                                                                                                   \
      }
      for (int sb0 = 0; sb0 < sp.H; sb0 += IQ_LB * IQ_THREADS) {     // (wave-uniform trip count: the scans below are executed by all lanes)
        const int sb = sb0 + (int)threadIdx.x;
        uint4 L[IQ_LB][4];
        uint32_t hvv[IQ_LB], nn[IQ_LB];
#pragma unroll
        for (int u = 0; u < IQ_LB; u++) {
          const int s = sb + u * IQ_THREADS;
          hvv[u] = 0;
          L[u][0] = L[u][1] = L[u][2] = L[u][3] = make_uint4(0u, 0u, 0u, 0u);
          if (s < sp.H) {
            hvv[u] = inv_mix((uint32_t)qrow[s]);
            const uint4* l4 = (const uint4*)(ix.lines + (size_t)s * lstride + ((size_t)(hvv[u] >> lsh) << 4));
            L[u][0] = l4[0]; L[u][1] = l4[1]; L[u][2] = l4[2]; L[u][3] = l4[3];
          }
        }
        bool partner = false;
#pragma unroll
        for (int u = 0; u < IQ_LB; u++) {
          const int s = sb + u * IQ_THREADS;
          const uint32_t hdr = L[u][3].w >> 24;      // (0 for a lane without a slot: nothing to scan)
          nn[u] = 0;
          if (hdr == IL_FALLBACK) {
            const uint32_t at = atomicAdd(&s_nov, 1u);
            if (at < (uint32_t)IQ_OV) ovlist[at] = (uint32_t)s;
          } else nn[u] = hdr;
          const uint32_t qt = hvv[u] & tmask;
          IQ_LINE_SCAN(L[u], 0u, nn[u], qt);
          partner = partner || nn[u] > (uint32_t)IL_CAP;
        }
        if (__builtin_amdgcn_ballot_w64(partner)) {
          // what a line of more than 14 postings could not hold sits in its partner line, behind the partner's own
#pragma unroll
          for (int u = 0; u < IQ_LB; u++)
            if (nn[u] > (uint32_t)IL_CAP) {
              const uint4* l4 = (const uint4*)(ix.lines + (size_t)(sb + u * IQ_THREADS) * lstride + ((size_t)((hvv[u] >> lsh) ^ 1u) << 4));
              L[u][0] = l4[0]; L[u][1] = l4[1]; L[u][2] = l4[2]; L[u][3] = l4[3];
            }
#pragma unroll
          for (int u = 0; u < IQ_LB; u++) {
            const bool pm = nn[u] > (uint32_t)IL_CAP;
            const uint32_t pn = pm ? L[u][3].w >> 24 : 0u, pe = pm ? pn + nn[u] - (uint32_t)IL_CAP : 0u, qt = hvv[u] & tmask;
            IQ_LINE_SCAN(L[u], pn, pe, qt);
          }
        }
      }
      flush_hits();
      if (wlane == 0) mine += nhits;
#undef IQ_LINE_SCAN
#endif
      __syncthreads();
      const uint32_t nov = s_nov;
      if (nov > (uint32_t)IQ_OV) { if (threadIdx.x == 0) s_over = 1; nloop = 0; }   // (a query with that many long buckets is repeat-rich: handed on)
      else {
        nloop = (int)nov;
        if (nov) {
          // the long buckets' lengths say whether this table can hold their hits — the plain path's early hand-over, for the slots the
          // lines sent here (without it a repeat-rich query with a few dozen long buckets streamed all of them before its table
          // overflowed: C5 slice, first tier 4.6 -> 33 ms with the line table)
          unsigned long long tot = 0;
          for (uint32_t i = threadIdx.x; i < nov; i += IQ_THREADS) {
            const uint32_t so = ovlist[i];
            const uint32_t hvp = inv_mix((uint32_t)qrow[so]);
            const uint32_t* E = ix.ends + (size_t)so * eper + (hvp >> ix.shift);
            tot += E[1] - E[0];
          }
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
          if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = tot;
          __syncthreads();
          tot = 0;
          for (unsigned w = 0; w < IQ_THREADS / 64; w++) tot += wsum[w];
          if (tot + s_distinct > (5ULL * (INV_CT * 3 / 4)) / 4) { if (threadIdx.x == 0) s_over = 1; nloop = 0; }
        }
      }
      __syncthreads();
    }
    for (int s0 = 0, it = 0; s0 < nloop; s0 += QCAP, it++) {   // workgroup-uniform trip count (barriers inside)
      uint32_t hv[SPT], lo[SPT], n[SPT];
      int sl[SPT];   // the slot a lane looks up: all of them in turn, or (line mode) the slots whose line said "see the buckets"
#pragma unroll
      for (int u = 0; u < SPT; u++) {
        const int idx = s0 + u * IQ_THREADS + (int)threadIdx.x;
        const int s = lmode ? (idx < nloop ? (int)ovlist[idx] : sp.H) : idx;
        sl[u] = s;
        hv[u] = 0; lo[u] = 0; n[u] = 0;
        if (pre_ok && it * SPT + u < IQ_PRE) {
          if (s < sp.H) { hv[u] = (uint32_t)pre_hv[it * SPT + u]; lo[u] = (uint32_t)pre_lo[it * SPT + u]; n[u] = (uint32_t)pre_n[it * SPT + u]; }
        } else if (s < sp.H) {
          hv[u] = inv_mix((uint32_t)qrow[s]);
          const uint32_t* E = ix.ends + (size_t)s * eper + (hv[u] >> ix.shift);
          lo[u] = E[0]; n[u] = E[1] - lo[u];
        }
      }
      uint2 w[SPT][4];                                                           // the usual bucket in one round trip
#pragma unroll
      for (int u = 0; u < SPT; u++) {
        const uint2* P = ix.items + (size_t)sl[u] * ix.slot_stride + lo[u];
        const uint32_t m = n[u] <= (uint32_t)IQ_INLINE ? n[u] : 0u;
#pragma unroll
        for (int x = 0; x < 4; x++) w[u][x] = (uint32_t)x < m ? P[x] : make_uint2(~hv[u], 0u);
      }
#pragma unroll
      for (int u = 0; u < SPT; u++) {
        const int s = sl[u];
        if (n[u] > (uint32_t)IQ_INLINE) {
          // a long bucket (a value many entries share): queued for the whole workgroup (a repeat's bucket holds tens of thousands
          // of postings: one lane would stream it alone)
          const uint32_t at = atomicAdd(&s_nseg[it & 1], 1u);
          seglist[at] = make_uint2(lo[u], n[u]); segkey[at] = make_uint2((uint32_t)s, hv[u]);
        } else if (n[u]) {
#pragma unroll
          for (int x = 0; x < 4; x++)
            if (w[u][x].x == hv[u]) { if (bits == 0) mine++; count_hit((int)w[u][x].y); }   // "table elements processed" (:173), counted once
          const uint2* P = ix.items + (size_t)s * ix.slot_stride + lo[u];
          for (uint32_t x = 4; x < n[u]; x++) {
            const uint2 y = P[x];
            if (y.x == hv[u]) { if (bits == 0) mine++; count_hit((int)y.y); }
          }
        }
      }
      __syncthreads();
      const uint32_t nseg = s_nseg[it & 1];
      const uint32_t over_now = s_over;
      if (threadIdx.x == 0) s_nseg[(it + 1) & 1] = 0;
      __syncthreads();                       // every lane holds the same (nseg, over_now) before anyone counts again
      if (big != nullptr && over_now) { handed_over = true; break; }            // first tier: the query is handed over, stop counting
      if (nseg) {
        // all queued buckets as ONE index space (exclusive prefix of their lengths in segpre): a trip of the loop below has
        // 8 x IQ_THREADS loads in flight whatever the buckets' lengths — one bucket per trip cost a memory round trip per
        // bucket, and repeat-rich queries queue a hundred short ones
        // (64-bit sums: a thousand buckets of a huge index can hold more than 2^32 postings between them)
        unsigned long long lens[SPT], len = 0;                                    // lane t holds buckets t * SPT .. t * SPT + SPT - 1
#pragma unroll
        for (int u = 0; u < SPT; u++) { const uint32_t q = threadIdx.x * SPT + u; lens[u] = q < nseg ? seglist[q].y : 0u; len += lens[u]; }
        unsigned long long incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const unsigned long long v = __shfl_up(incl, off); if ((threadIdx.x & 63) >= (unsigned)off) incl += v; }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned long long wbase = 0, total = 0;
        for (unsigned w = 0; w < IQ_THREADS / 64; w++) { const unsigned long long t = wsum[w]; if (w < (threadIdx.x >> 6)) wbase += t; total += t; }
        unsigned long long run = wbase + incl - len;
#pragma unroll
        for (int u = 0; u < SPT; u++) { const uint32_t q = threadIdx.x * SPT + u; if (q < nseg) segpre[q] = run; run += lens[u]; }
        if (threadIdx.x == 0) segpre[nseg] = total;
        __syncthreads();
        // second tier, whole index, every slot looked up: buckets with more than twice the table's capacity between them (and an
        // index with that many entries) are split before they are streamed: the pass would overflow after streaming everything
        // (the postings are still streamed, to count the ones that match — "table elements processed" — but no hit is counted)
        bool count_only = false;
        if (big == nullptr && bits == 0 && s0 + QCAP >= nloop && (total < ix.ne ? total : (unsigned long long)ix.ne) > 2ULL * (INV_CT * 3 / 4)) { if (threadIdx.x == 0) s_over = 1; count_only = true; }
        uint32_t g = 0;   // bucket of this lane's current posting (its postings come in ascending order)
        for (unsigned long long i0 = threadIdx.x; i0 - threadIdx.x < total; i0 += IQ_THREADS * 8) {
          uint2 e[8];
          uint32_t want[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const unsigned long long i = i0 + (unsigned long long)IQ_THREADS * u;
            e[u] = make_uint2(0u, 0u); want[u] = 1u;                               // (never equal)
            if (i < total) {
              while (i >= segpre[g + 1]) g++;
              const uint2 key = segkey[g];
              e[u] = ix.items[(size_t)key.x * ix.slot_stride + (size_t)seglist[g].x + (size_t)(i - segpre[g])];
              want[u] = key.y;
            }
          }
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (e[u].x == want[u]) { if (bits == 0) mine++; if (!count_only) count_hit((int)e[u].y); }
        }
        if (count_only) break;
        __syncthreads();         // the queue is reused by the next trip's lookups
      }
    }
    (void)handed_over;
    __syncthreads();
    if (!(MH_IQ_TIMING & 1)) iq_add_elements(elements, (s_over && big != nullptr) ? 0ULL : mine);   // (every wave of the workgroup gets here)
    if (s_over && big != nullptr) {
      // first tier: hand the query to the launch with the large count table (it starts over; nothing was emitted yet)
      if (threadIdx.x == 0) big[atomicAdd(big_count, 1ULL)] = qe;
      return;
    }
    if (s_over) {
      // the part's distinct hits outgrow the count table: split it in two and run both halves (exact; every stored entry
      // belongs to exactly one leaf part)
      if (threadIdx.x == 0) {
        if (bits < (uint32_t)MAX_BITS && s_top + 2 <= IQ_STACK) {
          stack[2 * s_top] = prefix; stack[2 * s_top + 1] = bits + 1; s_top++;
          stack[2 * s_top] = prefix | (1u << bits); stack[2 * s_top + 1] = bits + 1; s_top++;
        }
        atomicAdd(split_count, 1ULL);
      }
      __syncthreads();
      continue;
    }
    // emit this part's candidates as ONE contiguous block (one global atomic): the second stage then finds
    // the lanes of a wave sharing the query's ordered-sketch row
    uint32_t mymask = 0;   // bit t set -> table slot threadIdx.x + IQ_THREADS*t is a candidate
    int mycount = 0;
#pragma unroll
    for (int t = 0; t < INV_CT / IQ_THREADS; t++) {
      const int j = threadIdx.x + IQ_THREADS * t;
      const uint32_t w = tbl[j];
      if ((w & emask) != 0 && (int)(w >> ebits) >= sp.num_min_matches) {                          // MinHashSearch.java:204
        const int me = (int)(w & emask) - 1;
        if (pair_passes(sp, qid, ids[me], qlen, meta[(int64_t)me * META_W + 2])) { mymask |= 1u << t; mycount++; }   // :200-225
      }
    }
    __syncthreads();            // s_distinct is dead from here on: reuse it as the block's emit counter
    if (threadIdx.x == 0) s_distinct = 0;
    __syncthreads();
    uint32_t local = 0;
    if (mycount) local = atomicAdd(&s_distinct, (uint32_t)mycount);
    __syncthreads();
    if (threadIdx.x == 0) s_base = (s_distinct && !(MH_IQ_TIMING & 4)) ? atomicAdd(cand_count, (unsigned long long)s_distinct) : 0ULL;
    __syncthreads();
    unsigned long long slot = s_base + local;
#pragma unroll
    for (int t = 0; t < INV_CT / IQ_THREADS; t++) {
      if (mymask & (1u << t)) {
        if (slot < cand_cap) { cand[slot].q = qe; cand[slot].m = (int)(tbl[threadIdx.x + IQ_THREADS * t] & emask) - 1; }
        slot++;
      }
    }
    __syncthreads();
  }
}

// Second tier: DENSE counts.  A query the first tier hands over has thousands of distinct hits — on repeat-rich reads most of the
// index — and a hash table of them in LDS (rounds 1-2: 12 288 entries, beyond that hash-partition passes that each stream all of
// the query's postings again; 21 190 splits per step on the C5 slice) is the wrong structure: 64 KB of LDS hold a saturating
// 16-bit counter for each of 32 768 stored entries, indexed by the entry itself — no keys, no probing, no overflow.  An index
// with more entries is covered in passes over entry ranges of 32 768; the first pass notes which ranges the query's postings
// fall into and the others are skipped.
#ifndef MH_DQ_THREADS
#define MH_DQ_THREADS 512    // 1024 lanes + 65 536-entry ranges (one workgroup per CU) / 512 + 32 768 (two): C5 slice 12.2 / 10.8 ms — one
                            // query's zeroing and scan overlap the other's streaming, at the price of a third range pass over its 80 000 entries
#endif
#ifndef MH_DQ_RANGE_LOG
#define MH_DQ_RANGE_LOG 15
#endif
constexpr int DQ_THREADS = MH_DQ_THREADS, DQ_RANGE_LOG = MH_DQ_RANGE_LOG, DQ_MAX_RANGES = 4096;   // (an index of more than 4096 ranges = 2^28 entries runs every pass)
__global__ __launch_bounds__(DQ_THREADS) void index_query_dense_kernel(InvIndex ix, const int32_t* __restrict__ qminhash, int64_t qrow_stride,
                                                                       const int32_t* __restrict__ qlist, int nq, const int64_t* __restrict__ ids,
                                                                       const int64_t* __restrict__ qids, const int32_t* __restrict__ meta,
                                                                       const int32_t* __restrict__ qmeta, SearchParams sp, Candidate* __restrict__ cand,
                                                                       unsigned long long* __restrict__ cand_count, unsigned long long cand_cap,
                                                                       unsigned long long* __restrict__ split_count, unsigned long long* __restrict__ elements) {
  __shared__ uint32_t cnt[1 << (DQ_RANGE_LOG - 1)];     // two counters per word
  __shared__ uint2 seglist[DQ_THREADS];                 // queued long buckets: (first posting within the slot, length),
  __shared__ uint2 segkey[DQ_THREADS];                  // ... (slot, the query's mix there)
  __shared__ unsigned long long segpre[DQ_THREADS + 1];
  __shared__ unsigned long long wsum[DQ_THREADS / 64];
  __shared__ uint32_t rmask[DQ_MAX_RANGES / 32];        // ranges of stored entries the query's postings fall into
  __shared__ uint32_t s_nseg[2], s_emit;
  __shared__ unsigned long long s_base;
  const int qi = blockIdx.x;
  if (qi >= nq) return;
  const int qe = qlist[qi];
  const int32_t* qm = qmeta + (int64_t)qe * META_W;
  if (qm[3] != 0) return;   // (a strand that was not sketched is no query: a list made on the device — launch_query_iota — names every row)
  const int64_t qid = qids[qe];
  const int qlen = qm[2];
  const int32_t* qrow = qminhash + (int64_t)qe * qrow_stride;
  const size_t eper = (size_t)ix.nb + 1;
  const uint32_t npass = (ix.ne + (1u << DQ_RANGE_LOG) - 1) >> DQ_RANGE_LOG;
  const uint32_t sat = (uint32_t)(sp.num_min_matches < 0xF000 ? (sp.num_min_matches > 0 ? sp.num_min_matches : 1) : 0xF000);   // counts stop here
  for (int j = threadIdx.x; j < DQ_MAX_RANGES / 32; j += DQ_THREADS) rmask[j] = 0;
  unsigned long long mine = 0;
  for (uint32_t pass = 0; pass < npass; pass++) {
    __syncthreads();
    if (pass > 0 && !((rmask[(pass >> 5) & (DQ_MAX_RANGES / 32 - 1)] >> (pass & 31)) & 1u) && npass <= (uint32_t)DQ_MAX_RANGES) continue;   // (uniform)
    const uint32_t span = min(1u << DQ_RANGE_LOG, ix.ne - (pass << DQ_RANGE_LOG));   // stored entries of this range (the last one is partial)
    for (uint32_t j = threadIdx.x; j < (span + 1) / 2; j += DQ_THREADS) cnt[j] = 0;
    if (threadIdx.x == 0) { s_nseg[0] = 0; s_nseg[1] = 0; s_emit = 0; }
    __syncthreads();
    // one hit of stored entry `me`: the counter is read first and left alone once it has reached numMinMatches, so it can pass
    // that by no more than the adds in flight (8 x 1024) and never carries into its neighbour
#define DQ_COUNT_HIT(me_)                                                                                             \
    do {                                                                                                              \
      const uint32_t me = (me_);                                                                                      \
      if (pass == 0) { mine++; const uint32_t r = me >> DQ_RANGE_LOG; if (r) atomicOr(&rmask[(r >> 5) & (DQ_MAX_RANGES / 32 - 1)], 1u << (r & 31)); }   \
      if ((me >> DQ_RANGE_LOG) == pass) {                                                                             \
        const uint32_t wi = (me & ((1u << DQ_RANGE_LOG) - 1)) >> 1;                                                   \
        const int sh = (int)(me & 1u) * 16;                                                                           \
        if (((__hip_atomic_load(&cnt[wi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> sh) & 0xFFFFu) < sat) atomicAdd(&cnt[wi], 1u << sh);   \
      }                                                                                                               \
    } while (0)
    for (int s0 = 0, it = 0; s0 < sp.H; s0 += DQ_THREADS, it++) {   // workgroup-uniform trip count (barriers inside)
      const int s = s0 + (int)threadIdx.x;
      if (s < sp.H) {
        const uint32_t hv = inv_mix((uint32_t)qrow[s]);
        const uint32_t* E = ix.ends + (size_t)s * eper + (hv >> ix.shift);
        const uint32_t lo = E[0], n = E[1] - lo;
        if (n > (uint32_t)IQ_INLINE) {
          const uint32_t at = atomicAdd(&s_nseg[it & 1], 1u);
          seglist[at] = make_uint2(lo, n); segkey[at] = make_uint2((uint32_t)s, hv);
        } else {
          const uint2* P = ix.items + (size_t)s * ix.slot_stride + lo;
          for (uint32_t u = 0; u < n; u++) { const uint2 x = P[u]; if (x.x == hv) DQ_COUNT_HIT(x.y); }
        }
      }
      __syncthreads();
      const uint32_t nseg = s_nseg[it & 1];
      if (threadIdx.x == 0) s_nseg[(it + 1) & 1] = 0;
      __syncthreads();
      if (nseg) {
        // all queued buckets as ONE index space (exclusive prefix of their lengths in segpre): a trip of the loop below has
        // 8 x 1024 loads in flight whatever the buckets' lengths
        unsigned long long len = threadIdx.x < nseg ? seglist[threadIdx.x].y : 0u, incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const unsigned long long v = __shfl_up(incl, off); if ((threadIdx.x & 63) >= (unsigned)off) incl += v; }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned long long wbase = 0, total = 0;
        for (unsigned w = 0; w < DQ_THREADS / 64; w++) { const unsigned long long t = wsum[w]; if (w < (threadIdx.x >> 6)) wbase += t; total += t; }
        if (threadIdx.x < nseg) segpre[threadIdx.x] = wbase + incl - len;
        if (threadIdx.x == 0) segpre[nseg] = total;
        __syncthreads();
        uint32_t g = 0;   // bucket of this lane's current posting (its postings come in ascending order)
        for (unsigned long long i0 = threadIdx.x; i0 - threadIdx.x < total; i0 += DQ_THREADS * 8) {
          uint2 e[8];
          uint32_t want[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const unsigned long long i = i0 + (unsigned long long)DQ_THREADS * u;
            e[u] = make_uint2(0u, 0u); want[u] = 1u;                               // (never equal)
            if (i < total) {
              while (i >= segpre[g + 1]) g++;
              const uint2 key = segkey[g];
              e[u] = ix.items[(size_t)key.x * ix.slot_stride + (size_t)seglist[g].x + (size_t)(i - segpre[g])];
              want[u] = key.y;
            }
          }
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (e[u].x == want[u]) DQ_COUNT_HIT(e[u].y);
        }
        __syncthreads();         // the queue is reused by the next trip's lookups
      }
    }
#undef DQ_COUNT_HIT
    __syncthreads();
    if (pass == 0) iq_add_elements(elements, mine);           // "table elements processed" (:173), counted once
    if (pass > 0 && threadIdx.x == 0) atomicAdd(split_count, 1ULL);   // a pass beyond the first = the hit set was split
    // emit this range's candidates as ONE contiguous block (one global atomic)
    constexpr int PER = (1 << DQ_RANGE_LOG) / DQ_THREADS;      // 64 entries per lane: lane t owns entries t * 64 .. t * 64 + 63 of the range
    unsigned long long mymask = 0;
    int mycount = 0;
    const int jmax = threadIdx.x * PER >= span ? 0 : (int)min((uint32_t)PER, span - threadIdx.x * PER + 1) / 2;
    for (int j = 0; j < jmax; j++) {
      const uint32_t w = cnt[threadIdx.x * (PER / 2) + j];
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        if ((int)((w >> (16 * hf)) & 0xFFFFu) >= sp.num_min_matches) {                             // MinHashSearch.java:204
          const uint32_t me = (pass << DQ_RANGE_LOG) + threadIdx.x * PER + 2 * j + hf;
          if (me < ix.ne && pair_passes(sp, qid, ids[me], qlen, meta[(int64_t)me * META_W + 2])) { mymask |= 1ULL << (2 * j + hf); mycount++; }   // :200-225
        }
      }
    }
    uint32_t local = 0;
    if (mycount) local = atomicAdd(&s_emit, (uint32_t)mycount);
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_emit ? atomicAdd(cand_count, (unsigned long long)s_emit) : 0ULL;
    __syncthreads();
    unsigned long long slot = s_base + local;
    while (mymask) {
      const int b = __builtin_ctzll(mymask);
      mymask &= mymask - 1;
      if (slot < cand_cap) { cand[slot].q = qe; cand[slot].m = (int)((pass << DQ_RANGE_LOG) + threadIdx.x * PER + b); }
      slot++;
    }
  }
}

// Dense tier, compact form (round 4): --num-hashes <= 512 (one slot per lane) and numMinMatches <= 4 — every BASELINE configuration.
// Differences from the kernel above:
//  * a THERMOMETER of 4 bits per stored entry instead of a 16-bit counter: hit k of an entry sets bit k (atomicOr; the returned word
//    says whether the bit was already there, in which case the next one is tried), "count >= numMinMatches" is one bit.  OR is
//    idempotent, so no add in flight can carry into a neighbour, and 64 KB of LDS cover 131 072 stored entries per pass instead of
//    32 768: a quarter of the passes;
//  * the H buckets are looked up ONCE (registers), not once per pass;
//  * on a grouped index (index_group_kernel) a pass streams only ITS part of a long bucket: the parts' bounds are found by
//    bisection, all passes and buckets at once, before the first pass.  One rank's share of BASELINE configs[4] (1.25 M entries,
//    repeat-carrying queries with 250 000 postings each): every posting is now streamed once instead of 39 times.
constexpr int DQ2_THREADS = 512, DQ2_RANGE_LOG = 17, DQ2_WORDS = 1 << (DQ2_RANGE_LOG - 3), DQ2_MAX_RANGES = 1024, DQ2_BOUNDS = 512;
constexpr uint32_t DQ2_NOGRP = 0x3FFu;
__global__ __launch_bounds__(DQ2_THREADS) void index_query_dense4_kernel(InvIndex ix, const int32_t* __restrict__ qminhash, int64_t qrow_stride,
                                                                         const int32_t* __restrict__ qlist, int nq, const int64_t* __restrict__ ids,
                                                                         const int64_t* __restrict__ qids, const int32_t* __restrict__ meta,
                                                                         const int32_t* __restrict__ qmeta, SearchParams sp, Candidate* __restrict__ cand,
                                                                         unsigned long long* __restrict__ cand_count, unsigned long long cand_cap,
                                                                         unsigned long long* __restrict__ split_count, unsigned long long* __restrict__ elements,
                                                                         const uint32_t range_log /* <= DQ2_RANGE_LOG: entries per pass (tests shrink it) */) {
  __shared__ __align__(16) uint32_t cnt[DQ2_WORDS];    // eight 4-bit thermometers per word
  __shared__ uint2 seglist[DQ2_THREADS];                // queued buckets (longer than IQ_INLINE): (first posting within the slot, length),
  __shared__ uint2 segkey[DQ2_THREADS];                 // ... (slot | bounds row << 16, the query's mix there)
  __shared__ unsigned long long segpre[DQ2_THREADS + 1];
  __shared__ uint32_t bounds[DQ2_BOUNDS];               // per grouped long bucket: npass + 1 offsets, the starts of the passes' parts
  __shared__ unsigned long long wsum[DQ2_THREADS / 64];
  __shared__ uint32_t rmask[DQ2_MAX_RANGES / 32];       // ranges of stored entries the query's postings fall into
  __shared__ uint32_t s_nseg, s_ng, s_emit;
  __shared__ unsigned long long s_base;
  const int qi = blockIdx.x;
  if (qi >= nq) return;
  const int qe = qlist[qi];
  const int32_t* qm = qmeta + (int64_t)qe * META_W;
  if (qm[3] != 0) return;   // (a strand that was not sketched is no query: a list made on the device — launch_query_iota — names every row)
  const int64_t qid = qids[qe];
  const int qlen = qm[2];
  const int32_t* qrow = qminhash + (int64_t)qe * qrow_stride;
  const size_t eper = (size_t)ix.nb + 1;
  const uint32_t npass = (ix.ne + (1u << range_log) - 1) >> range_log;
  const int nmm = sp.num_min_matches < 1 ? 1 : sp.num_min_matches;     // (<= 4: launch_index_query)
  const uint32_t topmask = 0x11111111u << (nmm - 1);
  // ---- the H lookups, once ----
  const int s = (int)threadIdx.x;
  uint32_t hv = 0, lo = 0, n = 0;
  if (s < sp.H) {
    hv = inv_mix((uint32_t)qrow[s]);
    const uint32_t* E = ix.ends + (size_t)s * eper + (hv >> ix.shift);
    lo = E[0]; n = E[1] - lo;
  }
  if (threadIdx.x < DQ2_MAX_RANGES / 32) rmask[threadIdx.x] = 0;
  if (threadIdx.x == 0) { s_nseg = 0; s_ng = 0; }
  __syncthreads();
  const uint32_t per = npass + 1;
  const uint32_t gcap = (ix.grouped && range_log >= ix.class_log && npass > 1 && per <= (uint32_t)DQ2_BOUNDS) ? (uint32_t)DQ2_BOUNDS / per : 0u;
  if (n > (uint32_t)IQ_INLINE) {
    const uint32_t at = atomicAdd(&s_nseg, 1u);
    uint32_t g = DQ2_NOGRP;
    if (gcap && n > ix.group_t) { const uint32_t g2 = atomicAdd(&s_ng, 1u); if (g2 < gcap) g = g2; }   // (no room: streamed whole by every pass)
    seglist[at] = make_uint2(lo, n); segkey[at] = make_uint2((uint32_t)s | (g << 16), hv);
  }
  __syncthreads();
  const uint32_t nseg = s_nseg;
  if (gcap) {
    // bounds[g][p] = first posting of the bucket whose entry lies in range p or beyond (classes ascend within a long bucket)
    for (uint32_t i = threadIdx.x; i < nseg * per; i += DQ2_THREADS) {
      const uint32_t at = i / per, p = i % per, g = segkey[at].x >> 16;
      if (g == DQ2_NOGRP) continue;
      const uint2 sl = seglist[at];
      uint32_t b = 0;
      if (p == npass) b = sl.y;
      else if (p > 0) {
        const uint2* P = ix.items + (size_t)(segkey[at].x & 0xFFFFu) * ix.slot_stride + sl.x;
        uint32_t a = 0, z = sl.y;                                   // first j in [0, n) with class(j) >= p
        while (a < z) { const uint32_t m = (a + z) >> 1; if ((P[m].y >> range_log) < p) a = m + 1; else z = m; }
        b = a;
      }
      bounds[g * per + p] = b;
    }
    __syncthreads();
    const uint32_t ng = s_ng < gcap ? s_ng : gcap;
    if (npass <= (uint32_t)DQ2_MAX_RANGES)
      for (uint32_t i = threadIdx.x; i < ng * npass; i += DQ2_THREADS) {
        const uint32_t g = i / npass, p = i % npass;
        if (p && bounds[g * per + p + 1] > bounds[g * per + p]) atomicOr(&rmask[p >> 5], 1u << (p & 31));
      }
  }
  // a short bucket (read by the lane that looked it up) is read ONCE: its postings with the query's value — none, one or two, nearly
  // always — stay in registers for the passes.  (Every pass read the bucket again, posting after posting, each load waiting for the
  // one before: with ten passes over 1.25 M entries that was a fifth of this kernel.)
  uint32_t in_me0 = 0, in_me1 = 0;
  int in_cnt = 0;
  if (n && n <= (uint32_t)IQ_INLINE) {
    const uint2* P = ix.items + (size_t)s * ix.slot_stride + lo;
    for (uint32_t u0 = 0; u0 < n; u0 += 4) {
      uint2 x[4];
#pragma unroll
      for (int c = 0; c < 4; c++) { x[c] = make_uint2(~hv, 0u); if (u0 + (uint32_t)c < n) x[c] = P[u0 + (uint32_t)c]; }
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (x[c].x == hv) { if (in_cnt == 0) in_me0 = x[c].y; else if (in_cnt == 1) in_me1 = x[c].y; in_cnt++; }
    }
  }
  unsigned long long mine = 0;
  for (uint32_t pass = 0; pass < npass; pass++) {
    __syncthreads();
    if (pass > 0 && npass <= (uint32_t)DQ2_MAX_RANGES && !((rmask[pass >> 5] >> (pass & 31)) & 1u)) continue;   // (uniform)
    const uint32_t span = min(1u << range_log, ix.ne - (pass << range_log));   // stored entries of this range (the last one is partial)
    for (uint32_t j = threadIdx.x * 4; j < (span + 7) / 8; j += DQ2_THREADS * 4) *(uint4*)&cnt[j] = make_uint4(0u, 0u, 0u, 0u);   // (16 bytes per lane: whole vectors, the last one past the span's words too)
    if (threadIdx.x == 0) s_emit = 0;
    __syncthreads();
    // one hit of stored entry `me` (celem: this posting has not been seen by an earlier pass — "table elements processed", :173)
#define DQ2_HIT(me_, celem)                                                                                           \
    do {                                                                                                              \
      const uint32_t me = (me_);                                                                                      \
      if (celem) mine++;                                                                                              \
      if (pass == 0 && npass > 1) { const uint32_t r = me >> range_log; if (r && r < (uint32_t)DQ2_MAX_RANGES) atomicOr(&rmask[r >> 5], 1u << (r & 31)); }   \
      if ((me >> range_log) == pass) {                                                                            \
        const uint32_t wi = (me & ((1u << range_log) - 1)) >> 3;                                                  \
        const int sh = (int)(me & 7u) * 4;                                                                            \
        for (int k = 0; k < nmm; k++) { const uint32_t old = atomicOr(&cnt[wi], (1u << k) << sh); if (!((old >> sh) & (1u << k))) break; }   \
      }                                                                                                               \
    } while (0)
    if (n && n <= (uint32_t)IQ_INLINE) {
      if (in_cnt <= 2) {
        if (in_cnt > 0) DQ2_HIT(in_me0, pass == 0);
        if (in_cnt > 1) DQ2_HIT(in_me1, pass == 0);
      } else {
        const uint2* P = ix.items + (size_t)s * ix.slot_stride + lo;
        for (uint32_t u = 0; u < n; u++) { const uint2 x = P[u]; if (x.x == hv) DQ2_HIT(x.y, pass == 0); }
      }
    }
    if (nseg) {
      // the queued buckets — of a grouped one this pass's part — as ONE index space (exclusive prefix of the lengths in segpre)
      unsigned long long len = 0;
      if (threadIdx.x < nseg) {
        const uint32_t g = segkey[threadIdx.x].x >> 16;
        len = g == DQ2_NOGRP ? seglist[threadIdx.x].y : bounds[g * per + pass + 1] - bounds[g * per + pass];
      }
      unsigned long long incl = len;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const unsigned long long v = __shfl_up(incl, off); if ((threadIdx.x & 63) >= (unsigned)off) incl += v; }
      if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
      __syncthreads();
      unsigned long long wbase = 0, total = 0;
      for (unsigned w = 0; w < DQ2_THREADS / 64; w++) { const unsigned long long t = wsum[w]; if (w < (threadIdx.x >> 6)) wbase += t; total += t; }
      if (threadIdx.x < nseg) segpre[threadIdx.x] = wbase + incl - len;
      if (threadIdx.x == 0) segpre[nseg] = total;
      __syncthreads();
      // this lane's current queue entry, cached in registers: its span of the index space, where its postings start, the query's mix there.
      // (The first version looked all of that up in LDS for every posting — five dependent LDS reads in front of each load, eight
      //  postings per trip one after the other: the address arithmetic took as long as the memory round trip it was meant to keep busy.)
      uint32_t g = 0;
      unsigned long long seg_beg = 0, seg_end = 0;
      const uint2* seg_base = nullptr;
      uint32_t seg_want = 1u;
      bool seg_fresh = false;
      for (unsigned long long i0 = threadIdx.x; i0 - threadIdx.x < total; i0 += DQ2_THREADS * 8) {
        uint2 e[8];
        uint32_t want[8];
        bool fresh[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const unsigned long long i = i0 + (unsigned long long)DQ2_THREADS * u;
          e[u] = make_uint2(0u, 0u); want[u] = 1u; fresh[u] = false;              // (never equal)
          if (i < total) {
            if (i >= seg_end) {                                                    // (its postings come in ascending order)
              while (i >= segpre[g + 1]) g++;
              seg_beg = segpre[g]; seg_end = segpre[g + 1];
              const uint2 key = segkey[g];
              const uint32_t grp = key.x >> 16;
              const uint32_t off = grp == DQ2_NOGRP ? 0u : bounds[grp * per + pass];
              seg_base = ix.items + (size_t)(key.x & 0xFFFFu) * ix.slot_stride + (size_t)seglist[g].x + (size_t)off;
              seg_want = key.y;
              seg_fresh = grp != DQ2_NOGRP || pass == 0;
            }
            e[u] = seg_base[(size_t)(i - seg_beg)];
            want[u] = seg_want;
            fresh[u] = seg_fresh;
          }
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (e[u].x == want[u]) DQ2_HIT(e[u].y, fresh[u]);
      }
    }
#undef DQ2_HIT
    __syncthreads();
    if (pass > 0 && threadIdx.x == 0) atomicAdd(split_count, 1ULL);   // a pass beyond the first = the hit set was split
    // emit this range's candidates as ONE contiguous block (one global atomic).  Lane t owns the four-word vectors t, t + 512, ... of the
    // counters (16-byte LDS accesses, conflict-free; a pass of a repeat-carrying query touches a few thousand of the 131 072 entries, so
    // nearly every vector is skipped on one test: zeroing and sweeping word by word cost a query more LDS instructions than its postings);
    // entries that reached numMinMatches but fail the id / length rules lose their top bit in the first sweep, so the second one only
    // enumerates bits.
    const uint32_t nwords = (span + 7) / 8;
    int mycount = 0;
    for (uint32_t w4 = threadIdx.x * 4; w4 < nwords; w4 += DQ2_THREADS * 4) {
      const uint4 v = *(const uint4*)&cnt[w4];
      if (((v.x | v.y | v.z | v.w) & topmask) == 0u) continue;
      const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int c = 0; c < 4; c++) {
        uint32_t w = vw[c], m = w & topmask;
        if (!m) continue;
        const uint32_t wi = w4 + (uint32_t)c;
        while (m) {
          const int b = __builtin_ctz(m);
          m &= m - 1;
          const uint32_t me = (pass << range_log) + wi * 8 + (uint32_t)(b >> 2);
          if (me < ix.ne && pair_passes(sp, qid, ids[me], qlen, meta[(int64_t)me * META_W + 2])) mycount++;   // MinHashSearch.java:200-225
          else w &= ~(1u << b);
        }
        cnt[wi] = w;
      }
    }
    uint32_t local = 0;
    if (mycount) local = atomicAdd(&s_emit, (uint32_t)mycount);
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_emit ? atomicAdd(cand_count, (unsigned long long)s_emit) : 0ULL;
    __syncthreads();
    unsigned long long slot = s_base + local;
    if (mycount)
      for (uint32_t w4 = threadIdx.x * 4; w4 < nwords; w4 += DQ2_THREADS * 4) {
        const uint4 v = *(const uint4*)&cnt[w4];
        if (((v.x | v.y | v.z | v.w) & topmask) == 0u) continue;
        const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
          uint32_t m = vw[c] & topmask;
          while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1;
            if (slot < cand_cap) { cand[slot].q = qe; cand[slot].m = (int)((pass << range_log) + (w4 + (uint32_t)c) * 8 + (uint32_t)(b >> 2)); }
            slot++;
          }
        }
      }
  }
  iq_add_elements(elements, mine);           // "table elements processed" (:173): every posting with the query's value, once
}

bool index_query_tiers() { return MH_IQ_BIG_CT != 0; }
// whether the packed table words of tier 0 (IQ_THREADS lanes) / tier 1 (IQ_THREADS_MID lanes) can count to numMinMatches for an index
// of this many entries: the count field (the bits the entry index leaves) must hold numMinMatches + one add in flight per lane
bool index_query_tier_ok(int tier, int64_t entries, int num_min_matches) {
  if (entries < 1 || entries >= (1LL << 31)) return false;
  int ebits = 1;
  while ((1LL << ebits) <= entries) ebits++;            // entry + 1 <= entries < 2^ebits
  const long long cmax = (1LL << (32 - ebits)) - 1;
  return (long long)num_min_matches + (tier == 0 ? IQ_THREADS : 512) <= cmax;   // (512: an upper bound of the middle tier's lanes)
}

// tier 0: the first tier (one wavefront per query, 2048-entry table); tier 1: the same kernel with an 8192-entry table and 256 lanes
// (queries of a LARGE index that outgrow the first table but have thousands, not hundreds of thousands, of hits: the dense tier would
// make a pass per 32 768 stored entries for them); tier 2: the dense tier.  big / big_count: where tiers 0 and 1 list the queries
// they hand on (nullptr: they split hit sets into hash-partition passes instead).
#ifndef MH_IQ_MID_CT
#define MH_IQ_MID_CT 8192    // 16384 entries + 512 lanes / 8192 + 256 / 4096 + 128 on all of C4: 136 / 81 / 78 ms of index query (more workgroups per
                            // CU again); 8192 holds the 2-3 thousand hits of an ordinary read of a large data set with room to spare
#endif
#ifndef MH_IQ_MID_THREADS
#define MH_IQ_MID_THREADS 256
#endif
constexpr int INV_CT_MID = MH_IQ_MID_CT, IQ_THREADS_MID = MH_IQ_MID_THREADS;
// the query list "every row": qlist[i] = first + i (the kernels above skip the rows whose status says "not sketched").  Round 6: the search of
// device-resident query rows used to wait for the rows' meta words on the host, a host loop over them and the upload of the list it made
// before its first kernel: 0.35 ms of a 2.1-ms search on one rank of eight.
__global__ __launch_bounds__(256) void query_iota_kernel(int32_t* __restrict__ qlist, int first, int n) {
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i < n) qlist[i] = first + i;
}
void launch_query_iota(hipStream_t st, int32_t* qlist, int first, int n) {
  if (n > 0) hipLaunchKernelGGL(query_iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, qlist, first, n);
}
void launch_index_query(hipStream_t st, const InvIndex& ix, const int32_t* qminhash, int64_t qrow_stride,
                        const int32_t* qlist, int nq, const int64_t* ids, const int64_t* qids, const int32_t* meta, const int32_t* qmeta,
                        const SearchParams& sp, Candidate* cand, unsigned long long* cand_count, unsigned long long cand_cap,
                        unsigned long long* split_count, unsigned long long* elements_total, int32_t* big, unsigned long long* big_count, int tier,
                        unsigned long long* elements) {
  // elements: index_elements_spread_words() zeroed words the kernels add to; folded into *elements_total (and zeroed again) behind the launch
  if (nq <= 0) return;
  struct Fold { hipStream_t st; unsigned long long* s; unsigned long long* t; ~Fold() { hipLaunchKernelGGL(index_elements_sum_kernel, dim3(1), dim3(IQ_ELEM_SPREAD), 0, st, s, t); } } fold{st, elements, elements_total};
  if (tier == 0 && big != nullptr && ix.lines != nullptr)
    hipLaunchKernelGGL((index_query_kernel<INV_CT, IQ_THREADS, IQ_SPT, true>), dim3((unsigned)nq), dim3(IQ_THREADS), 0, st, ix, qminhash, qrow_stride, qlist, nq, ids, qids,
                       meta, qmeta, sp, cand, cand_count, cand_cap, split_count, elements, big, big_count);
  else if (tier == 0)
    hipLaunchKernelGGL((index_query_kernel<INV_CT, IQ_THREADS, IQ_SPT, false>), dim3((unsigned)nq), dim3(IQ_THREADS), 0, st, ix, qminhash, qrow_stride, qlist, nq, ids, qids,
                       meta, qmeta, sp, cand, cand_count, cand_cap, split_count, elements, big, big_count);
  else if (tier == 1 && big != nullptr && ix.lines != nullptr)
    hipLaunchKernelGGL((index_query_kernel<INV_CT_MID, IQ_THREADS_MID, 1, true>), dim3((unsigned)nq), dim3(IQ_THREADS_MID), 0, st, ix, qminhash, qrow_stride, qlist, nq,
                       ids, qids, meta, qmeta, sp, cand, cand_count, cand_cap, split_count, elements, big, big_count);
  else if (tier == 1)
    hipLaunchKernelGGL((index_query_kernel<INV_CT_MID, IQ_THREADS_MID, 1, false>), dim3((unsigned)nq), dim3(IQ_THREADS_MID), 0, st, ix, qminhash, qrow_stride, qlist, nq,
                       ids, qids, meta, qmeta, sp, cand, cand_count, cand_cap, split_count, elements, big, big_count);
  else {
    const bool old_dense = []() { const char* e = getenv("MHAP_DENSE_TIER"); return e && strcmp(e, "counters") == 0; }();   // (the 16-bit-counter kernel for every case: tests)
    const uint32_t range_log = []() { const char* e = getenv("MHAP_DENSE_RANGE_LOG"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 3 && v <= DQ2_RANGE_LOG ? v : DQ2_RANGE_LOG); }();
    if (sp.H <= DQ2_THREADS && sp.num_min_matches <= 4 && !old_dense)
      hipLaunchKernelGGL(index_query_dense4_kernel, dim3((unsigned)nq), dim3(DQ2_THREADS), 0, st, ix, qminhash, qrow_stride, qlist, nq, ids, qids, meta, qmeta, sp,
                         cand, cand_count, cand_cap, split_count, elements, range_log);
    else
      hipLaunchKernelGGL(index_query_dense_kernel, dim3((unsigned)nq), dim3(DQ_THREADS), 0, st, ix, qminhash, qrow_stride, qlist, nq, ids, qids, meta, qmeta, sp,
                         cand, cand_count, cand_cap, split_count, elements);
  }
}
int index_query_dense_ranges(int64_t entries) { return (int)((entries + (1 << DQ_RANGE_LOG) - 1) >> DQ_RANGE_LOG); }

// =============================================================================================
// Second stage.  Persistent lanes: lane g handles candidates g, g+G, ...  Scratch (3 int arrays of
// maxrec entries per lane) is interleaved across lanes so that lanes of a wave touch adjacent words.
// =============================================================================================
// Per-lane streaming view of one ordered-sketch row: the current 64-byte line (8 entries) sits in LDS
// (lane-interleaved 8-byte words), the next line is already in flight into registers.  lane_overlap only walks
// forward between reset()s, so one line + one prefetch hides the HBM/L2 latency of the otherwise dependent loads.
struct CachedView {
  const uint2* row;      // global row: entry i = (hash, pos)
  int n;
  uint2* lds;            // this lane's slot: entry e of the current line at lds[e * OVL_THREADS]
  int cur, nxt;
  uint2 r[8];
  __device__ inline void init(const int32_t* p, int n_, uint2* lds_) { row = (const uint2*)p; n = n_; lds = lds_; cur = -1; nxt = -1; }
  __device__ inline void reset() {}
  __device__ inline void fetch(int line) {
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = row[line * 8 + e];
  }
  __device__ inline void get(int i, int& h, int& pos) {
    const int line = i >> 3;
    if (line != cur) {
      if (line != nxt) fetch(line);
#pragma unroll
      for (int e = 0; e < 8; e++) lds[e * OVL_THREADS] = r[e];
      cur = line;
      if ((line + 1) * 8 < n) { fetch(line + 1); nxt = line + 1; } else nxt = -1;
    }
    const uint2 v = lds[(i & 7) * OVL_THREADS];
    h = (int)v.x; pos = (int)v.y;
  }
};

__global__ __launch_bounds__(OVL_THREADS) void overlap_kernel(const Candidate* __restrict__ cand, const unsigned long long* __restrict__ cand_count,
                                                              unsigned long long cand_cap, const int32_t* __restrict__ ordered,
                                                              int64_t ord_stride, const int32_t* __restrict__ meta,
                                                              const int32_t* __restrict__ qordered, int64_t qord_stride,
                                                              const int32_t* __restrict__ qmeta, SearchParams sp,
                                                              const double* __restrict__ score_table, int32_t* __restrict__ scratch,
                                                              int64_t scratch_per_lane, DevRecord* __restrict__ recs,
                                                              unsigned long long* __restrict__ rec_count, unsigned long long rec_cap,
                                                              unsigned long long* __restrict__ compared, int spread) {
  __shared__ uint2 lines[2][8 * OVL_THREADS];
  unsigned long long n = *cand_count;
  if (n > cand_cap) n = cand_cap;
  // spread (a power of two <= 64): only every spread-th lane takes pairs.  A handful of pairs — the few the join kernel hands over —
  // packed 64 to a wavefront run in lockstep through each other's branches (4 147 pairs of a c5rank step: 63 ms, on 65 waves of a
  // machine that holds 20 000); one pair per wavefront, they take as long as the longest of them.
  const int64_t gl = (int64_t)blockIdx.x * OVL_THREADS + threadIdx.x;
  if (gl & (int64_t)(spread - 1)) return;
  const int64_t G = (int64_t)gridDim.x * OVL_THREADS / spread;
  const int64_t g = gl / spread;
  LaneScratch sc;
  sc.base = scratch + g;
  sc.stride = G;
  sc.maxrec = (int32_t)(scratch_per_lane / 3);
  unsigned long long mine = 0;
  for (unsigned long long c = (unsigned long long)g; c < n; c += (unsigned long long)G) {
    const Candidate cd = cand[c];
    const int32_t* qm = qmeta + (int64_t)cd.q * META_W;
    const int32_t* mm = meta + (int64_t)cd.m * META_W;
    CachedView A, B;
    A.init(qordered + (int64_t)cd.q * qord_stride, qm[0], &lines[0][threadIdx.x]);
    B.init(ordered + (int64_t)cd.m * ord_stride, mm[0], &lines[1][threadIdx.x]);
    const LaneOverlap r = lane_overlap(A, qm[1], B, mm[1], sp.max_shift, sc);   // MinHashSearch.java:228
    mine++;
    double score = 0.0;
    if (!r.empty) score = score_table[score_index(r.inter, r.kk)];
    if (score >= sp.threshold) {                                                             // :229
      const unsigned long long slot = atomicAdd(rec_count, 1ULL);
      if (slot < rec_cap) {
        DevRecord d;
        d.q = cd.q; d.m = cd.m; d.score = score; d.raw = r.valid; d.a1 = r.a1; d.a2 = r.a2; d.b1 = r.b1; d.b2 = r.b2; d.pad = 0;
        recs[slot] = d;
      }
    }
  }
  if (mine) atomicAdd(compared, mine);
}

#endif   // MH_OJ_WIDE_UNIT

// =============================================================================================
// Second stage, one WAVEFRONT per candidate pair (default path).
//
// Both ordered sketches are sorted by (hash, pos), and everything getOverlapInfo does with them is a function of the
// equal-hash JOIN of the two lists: recordMatchingKmers (both passes) keeps the joined k-mers whose positions pass the
// pass's windows, and the bottom-k Jaccard walk counts the joined k-mers inside [a1,a2]x[b1,b2] whose rank in the
// merged union is below k.  So the wave computes the join once — the query's hashes sit in LDS, every lane binary-
// searches the hash of one entry of the other sketch (coalesced 8-byte loads) — and the rest is a handful of wave-wide
// filters, one rank selection (the median shift = Utils.quickSelect's k-th order statistic) and min/max reductions over
// the few joined k-mers.  Per pair that is ~n log n lane steps instead of the ~4n divergent merge steps per LANE of
// overlap_kernel.
//
// A joined hash that is unique inside both sketches contributes at most one record per pass, independent of all other
// hashes (the two-pointer merge has no run there), and record ORDER only matters to optimizeShifts, which merges
// neighbouring records of one query position, i.e. of one hash.  A hash that is duplicated in either sketch forms a
// "group": the merge's run logic (:460-496), optimizeShifts and the one-to-one pairing of the Jaccard walk are replayed
// literally on the group's few entries (oj_group_merge_lane etc.), and its records join the others.  Pairs
// beyond the caps below (joined k-mers, groups, group length) are appended to `slow` for overlap_kernel's literal merge.
// =============================================================================================
#ifndef MH_OJ_WAVES
#define MH_OJ_WAVES 4
#endif
constexpr int OJ_WAVES = MH_OJ_WAVES;
#ifndef MH_OJ_JCAP
#define MH_OJ_JCAP 128
#endif
constexpr int OJ_JCAP = MH_OJ_JCAP;    // joined k-mers + group records kept per pair (a multiple of 64)
constexpr int OJ_R = OJ_JCAP / 64;     // ... = rounds of one entry per lane
#ifndef MH_OJ_GCAP
#define MH_OJ_GCAP 16   // 12 / 16 with the 6-KB filter: C5 slice 50.6 / 46.9 ms (75 928 / 3 739 pairs handed to the per-lane kernel), c5rank 1004 / 996, C2 3.64 / 3.63
#endif
constexpr int OJ_GCAP = MH_OJ_GCAP;    // duplicated-hash groups per pair
constexpr int OJ_GLEN = 8;             // entries of one sketch in a group
#ifndef MH_OJ_U
#define MH_OJ_U 3
#endif
constexpr int OJ_U = MH_OJ_U;          // 64-entry blocks of the other sketch in flight per wave
#ifndef MH_OJ_PAD
#define MH_OJ_PAD 0   // (experiment: unused ints per wave, to see what the resident waves per CU are worth)
#endif
constexpr int OJ_LDS_EXTRA = 3 * OJ_JCAP + OJ_GCAP * (6 + 2 * OJ_GLEN) + MH_OJ_PAD;   // ints per wave besides the query hashes
// Round 4: the other sketch's POSITIONS stay in registers from the join's pass over its row (one per lane and 64-entry block: 24 at
// S = 1536), and a shared query's positions are staged in LDS next to its hashes — the two later passes over both rows that the
// bottom-k Jaccard ranks need (nine pairs in ten get that far: -DMH_OJ_STATS) then read no memory at all.  Before: 36 KB per pair
// (the row, then both rows' positions again), now 12.
#ifndef MH_OJ_KEEP
#define MH_OJ_KEEP 1
#endif
// (Also tried in round 4, on top of this: the WHOLE row of the other sketch loaded at once — 24 loads per lane in flight — and looked up
//  in groups of eight blocks: one memory round trip and six LDS round trips per pair instead of eight and sixteen.  168 VGPRs, three
//  waves per SIMD: C2 4.78 -> 6.63 ms, C5 slice 69 -> 91; capped at 128 VGPRs (four waves): 5.24 / 74.8; at two waves 9.0 / 126.  The
//  kernel's time stays inversely proportional to the waves a CU holds; instruction-level parallelism inside a wave does not replace them.)
constexpr int OJ_KB = 24;              // blocks of the other sketch whose positions are kept (S <= 64 * OJ_KB)

__device__ __forceinline__ int oj_mbcnt(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ int oj_wave_min(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off); v = o < v ? o : v; }
  return v;
}
__device__ __forceinline__ int oj_wave_max(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off); v = o > v ? o : v; }
  return v;
}
__device__ __forceinline__ void oj_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// MatchData.performUpdate (:191-215) given the median shift of the current records (have = any records)
__device__ __forceinline__ ShiftStats oj_shift_stats(bool have, int med, int len1, int len2, double max_shift) {
  ShiftStats st;
  if (have) {
    st.med = med;
    const int left = 0 > -med ? 0 : -med;
    const int right = len1 < len2 - med ? len1 : len2 - med;
    int ov = right - left; if (ov < 10) ov = 10;
    const int mx = len1 > len2 ? len1 : len2;
    const int lim = (int)((double)ov * max_shift);
    st.absmax = mx < lim ? mx : lim;
  } else {
    st.med = 0;
    st.absmax = (len1 > len2 ? len1 : len2) + 1;
  }
  return st;
}

#ifdef MH_OJ_STATS
// (diagnostic build: where the pairs of the join kernel end — {nj < 3, no record in pass 1, in pass 2, < 3 valid, below the threshold,
//  accepted, sum of nj, sum of in-window joined k-mers of the scored pairs}; printed by launch_overlap_join's caller through oj_stats_dump)
__device__ unsigned long long g_oj_stats[20];
#define OJ_STAT(k, v) do { if (lane == 0) atomicAdd(&g_oj_stats[k], (unsigned long long)(v)); } while (0)
#else
#define OJ_STAT(k, v) do { } while (0)
#endif
struct OjWindows { int v1lo, v1hi, v2lo, v2hi, med, absmax; };
__device__ __forceinline__ OjWindows oj_windows(ShiftStats st, int len1, int len2) {   // MatchData :246-276
  const int med = st.med, absmax = st.absmax;
  const int t1 = -med - absmax, t2 = len2 - med + absmax, t3 = med - absmax, t4 = len1 + med + absmax;
  OjWindows w;
  w.v1lo = 0 > t1 ? 0 : t1; w.v1hi = len1 < t2 ? len1 : t2;
  w.v2lo = 0 > t3 ? 0 : t3; w.v2hi = len2 < t4 ? len2 : t4;
  w.med = med; w.absmax = absmax;
  return w;
}

// recordMatchingKmers restricted to one hash value that is duplicated in at least one sketch: pa[0..m) / pb[0..n) are the
// positions of ALL entries with that hash (ascending), and the loop below is the reference's, run on just those entries
// (entries of other hashes end a run exactly like the end of these arrays does).  Run by ONE LANE for its own group: a pair of
// repeat-rich reads has half a dozen groups, and replayed one after the other by the whole wave they were a third of the join
// kernel's time on the C5 slice (-DMH_OJ_NO_GROUPS timing build: 69.0 -> 44.9 ms).  Returns the number of records written to o1/o2:
// at most two for every three entries the walk consumes, so a group's records fit the m + n words its entries reserve (oj_pass).
// (Which of a group's positions — eight words per sketch, 16-byte aligned, read as vectors — lie in the pass's windows becomes a bit
// mask per sketch, so the walk's skips and its runs of consecutive in-window entries are bit scans; only the positions the walk
// stops at are read again.  The literal loop read one LDS word per step, every read waiting for the one before: two replays per
// pair were 13 % of the kernel on the C5 slice.  Keeping all sixteen positions in registers and picking them by index was tried:
// 114 VGPRs in the PAIR shape, 92 bytes of scratch in TEAM, slower everywhere.)
__device__ __forceinline__ uint32_t oj_mask4(const int4 v, int lo, uint32_t width) {
  return (((uint32_t)(v.x - lo) < width) ? 1u : 0u) | (((uint32_t)(v.y - lo) < width) ? 2u : 0u) | (((uint32_t)(v.z - lo) < width) ? 4u : 0u) |
         (((uint32_t)(v.w - lo) < width) ? 8u : 0u);
}
__device__ __forceinline__ uint32_t oj_mask8(const int32_t* p, int cnt, int lo, int hi) {   // bit x: entry x < cnt lies in [lo, hi)
  const uint32_t width = hi > lo ? (uint32_t)(hi - lo) : 0u;
  uint32_t m = oj_mask4(*(const int4*)p, lo, width);
  if (cnt > 4) m |= oj_mask4(*(const int4*)(p + 4), lo, width) << 4;
  return m & ((1u << cnt) - 1u);
}
static_assert(OJ_GLEN == 8, "a group's positions are two int4 per sketch");
__device__ __forceinline__ int oj_group_merge_lane(const int32_t* pa, int m, const int32_t* pb, int n, const OjWindows& w, int32_t* o1, int32_t* o2) {
  const uint32_t w1 = oj_mask8(pa, m, w.v1lo, w.v1hi), w2 = oj_mask8(pb, n, w.v2lo, w.v2hi);
  int i1 = 0, i2 = 0, cnt = 0;
  for (;;) {
    const uint32_t r1 = w1 >> i1, r2 = w2 >> i2;
    if (r1 == 0u || r2 == 0u) break;                    // (entries outside their window are stepped over one by one in the reference: no record on the way)
    i1 += __builtin_ctz(r1); i2 += __builtin_ctz(r2);
    const int p1 = pa[i1], p2 = pb[i2];
    const int diff = (p2 - p1) - w.med;
    if (diff > w.absmax) { i1++; continue; }
    if (diff < -w.absmax) { i2++; continue; }
    o1[cnt] = p1; o2[cnt] = p2;
    cnt++;
    // the in-window entries that follow without a gap (:476-490): the last of either run makes a second record
    const int e1 = __builtin_ctz(~(w1 >> (i1 + 1))), e2 = __builtin_ctz(~(w2 >> (i2 + 1)));
    if (e1 | e2) {
      i1 += e1; i2 += e2;
      o1[cnt] = pa[i1]; o2[cnt] = pb[i2];
      cnt++;
    }
    i1++; i2++;
  }
  return cnt;
}
static_assert(OJ_GCAP <= 64, "lane g replays group g");

// One recordMatchingKmers pass over the join.  Entries [0, nj) are the unique-hash joined k-mers (kept if they pass the
// pass's windows); behind them every group owns as many words as it has entries ([gi[4], gi[4] + m + n), nx words in all: laid out
// when the groups were collected), lane g replays group g into the first of them and marks the rest unused — the order of the
// records matters inside a group only (optimizeShifts), so nothing has to be counted or compacted first.  (Round 4's first version
// replayed every group twice — to count, then, after a prefix sum over the lanes, to store contiguously — and groups of fewer than
// three one after the other by the whole wave.)  Bit r of the result = entry r*64+lane is a record of this pass; count = their number.
template <bool FIRST>
__device__ __forceinline__ uint32_t oj_pass(int32_t* jp1, int32_t* jp2, int nj, int ng, int nx, int32_t* gi, const int32_t* gpa, const int32_t* gpb,
                                           int len1, int len2, ShiftStats st, int lane, int& count) {
  const OjWindows w = oj_windows(st, len1, len2);
  if (ng) {
    if (lane < ng) {
      const int m = gi[lane * 6 + 2], n = gi[lane * 6 + 3], at = gi[lane * 6 + 4];
#ifdef MH_OJ_NO_GMERGE
      const int k = 0;   // (timing experiment; results are wrong)
#else
      const int k = oj_group_merge_lane(gpa + lane * OJ_GLEN, m, gpb + lane * OJ_GLEN, n, w, jp1 + at, jp2 + at);
#endif
      for (int x = k; x < m + n; x++) jp1[at + x] = INT32_MIN;
      gi[lane * 6 + 5] = k;
    }
    oj_lds_sync();
  }
  uint32_t fl = 0;
  int cnt = 0;
#pragma unroll
  for (int r = 0; r < OJ_R; r++) {
    if (r * 64 < nj + nx) {
      const int t = r * 64 + lane;
      bool ok = false;
      if (t < nj) {
        // (the first pass's windows are the whole strands and its shift bound max(len1, len2) + 1: every position pair of [0, len1) x [0, len2) passes)
        if (FIRST) ok = true;
        else {
          const int p1 = jp1[t], p2 = jp2[t];
          const int diff = (p2 - p1) - w.med;
          ok = p1 >= w.v1lo && p1 < w.v1hi && p2 >= w.v2lo && p2 < w.v2hi && !(diff > w.absmax) && !(diff < -w.absmax);
        }
      } else if (t < nj + nx) ok = jp1[t] != INT32_MIN;
      fl |= (ok ? 1u : 0u) << r;
      cnt += __popcll(__ballot(ok));
    }
  }
  count = cnt;
  return fl;
}

// k-th smallest (k = count / 2) of the records' shifts = Utils.quickSelect(shifts, count / 2, count): the value, bit by bit from the
// top — of the records still in the running, those with a 0 in the bit are the smaller ones; the k-th is among them or k moves past
// them.  The records' lanes are scalar masks, so a bit costs two vector instructions per round of 64 records and a handful of scalar
// ones: 15 bits for 10-kb reads.  (Round 3 counted, for every record, the records below it — one LDS broadcast and four vector
// instructions per record and round: with the 40 records of a typical C2 pair, three times the instructions; and this kernel is
// bound by the instructions it issues — at five waves per SIMD more resident waves no longer help it.  That way stays for pairs
// of a dozen records or fewer, where it is the shorter one.)
// A shift is p2 - p1 with 0 <= p1 < len1, 0 <= p2 < len2: biased by 2^lb > max(len1, len2) it is a positive (lb + 1)-bit number.
constexpr int OJ_MED_SMALL = 12;   // up to this many records the median is found by counting (below)
__device__ __forceinline__ int oj_median_shift(const int32_t* jp1, const int32_t* jp2, int32_t* sh, uint32_t fl, int ntot, int count, int lane, int lb) {
  if (count <= OJ_MED_SMALL) {
    // a handful of records (a pair that shares a repeat's k-mers and nothing else): every record counts the records below it —
    // one LDS broadcast and a few instructions per record, fewer than the lb + 1 bit steps
    int myv[OJ_R], myidx[OJ_R], less[OJ_R];
    int base = 0;
#pragma unroll
    for (int r = 0; r < OJ_R; r++) {
      myv[r] = 0; myidx[r] = 0; less[r] = 0;
      if (r * 64 < ntot) {
        const bool ok = (fl >> r) & 1u;
        const unsigned long long bal = __ballot(ok);
        if (ok) {
          const int t = r * 64 + lane;
          myv[r] = jp2[t] - jp1[t];
          myidx[r] = base + oj_mbcnt(bal);
          sh[myidx[r]] = myv[r];
        }
        base += __popcll(bal);
      }
    }
    oj_lds_sync();
    for (int u = 0; u < count; u++) {
      const int v = sh[u];   // same address in every lane: LDS broadcast
#pragma unroll
      for (int r = 0; r < OJ_R; r++)
        if (r * 64 < ntot) less[r] += (v < myv[r] || (v == myv[r] && u < myidx[r])) ? 1 : 0;
    }
    const int k = count / 2;
    int med = 0;
#pragma unroll
    for (int r = 0; r < OJ_R; r++) {
      if (r * 64 < ntot) {
        const unsigned long long bal = __ballot(((fl >> r) & 1u) && less[r] == k);
        if (bal) med = __builtin_amdgcn_readlane(myv[r], __builtin_amdgcn_readfirstlane(__builtin_ctzll(bal)));
      }
    }
    __builtin_amdgcn_wave_barrier();
    return med;
  }
  uint32_t key[OJ_R];
  unsigned long long in[OJ_R];
  const uint32_t bias = 1u << lb;
#pragma unroll
  for (int r = 0; r < OJ_R; r++) {
    key[r] = 0u; in[r] = 0ULL;
    if (r * 64 < ntot) {
      const bool ok = (fl >> r) & 1u;
      if (ok) { const int t = r * 64 + lane; key[r] = (uint32_t)(jp2[t] - jp1[t]) + bias; }
      in[r] = __builtin_amdgcn_ballot_w64(ok);
    }
  }
  int k = count / 2;
  uint32_t res = 0u;
  for (int b = lb; b >= 0; b--) {
    const uint32_t bit = 1u << b;
    unsigned long long one[OJ_R];
    int c0 = 0;
#pragma unroll
    for (int r = 0; r < OJ_R; r++) {
      one[r] = 0ULL;
      if (r * 64 < ntot) { one[r] = __builtin_amdgcn_ballot_w64((key[r] & bit) != 0u); c0 += __popcll(in[r] & ~one[r]); }
    }
    if (k < c0) {
#pragma unroll
      for (int r = 0; r < OJ_R; r++) in[r] &= ~one[r];
    } else {
      k -= c0; res |= bit;
#pragma unroll
      for (int r = 0; r < OJ_R; r++) in[r] &= one[r];
    }
  }
  return (int)(res - bias);
}

// Rank of entry idx (of the current chunk of OJ_RCH blocks) among the in-window entries ahead of it: lane b of the wave holds block
// b's in-window mask and the in-window count of the blocks before it.
extern "C" __device__ int oj_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");   // v_writelane_b32 (this clang has no builtin for it)
// bit of a hash value in a filter of ts <= 65 536 bits: the low 16 bits of the value (uniform, whatever end of the hash range the sketch keeps)
// scaled onto [0, ts) — ts need not be a power of two, so the filter can take exactly the LDS a workgroup has to spare
__device__ __forceinline__ uint32_t oj_filter_bit(uint32_t h, uint32_t ts) { return (uint32_t)__umul24(h & 0xFFFFu, ts) >> 16; }   // (HIP declares __umul24 as int)
constexpr int OJ_CQ = 256;              // ring of entry indices that passed the query's filter (FILTER shapes; it lives in jp1's words during the join)
static_assert(OJ_CQ * 2 <= OJ_JCAP * 4 && OJ_CQ >= 64 * OJ_U + 64, "the ring takes a trip's entries on top of an unhandled rest");
constexpr int OJ_RCH = (64 / OJ_U) * OJ_U;   // blocks of a chunk: whole trips of OJ_U blocks, one block per lane
// The last block of a sketch whose length is no multiple of 64: its lanes past the end took part in the ballot with whatever they held.
// Trimmed once after the pass (the block's mask sits in lane `blk`; no later block's prefix depends on it) instead of tested in every block.
__device__ __forceinline__ void oj_trim_last_block(uint32_t& mlo, uint32_t& mhi, int& total, int blk, int valid) {
  const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mhi, blk) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mlo, blk);
  const unsigned long long keep = m & ((1ULL << valid) - 1ULL);
  total -= __popcll(m ^ keep);
  mlo = (uint32_t)oj_writelane((int)(uint32_t)keep, blk, (int)mlo);
  mhi = (uint32_t)oj_writelane((int)(uint32_t)(keep >> 32), blk, (int)mhi);
}
__device__ __forceinline__ int oj_rank_from(uint32_t mlo, uint32_t mhi, int mpre, int idx) {
  const int src = (idx >> 6) & 63;
  const uint32_t lo = (uint32_t)__shfl((int)mlo, src), hi = (uint32_t)__shfl((int)mhi, src);
  const int pre = __shfl(mpre, src);
  const unsigned long long m = (((unsigned long long)hi << 32) | (unsigned long long)lo) & ((1ULL << (idx & 63)) - 1ULL);
  return pre + __popcll(m);
}

// Lookup of a hash among the query's sorted hashes, in LDS.  The hashes of a bottom-S sketch are uniform order statistics of
// [first, last], so  bucket(h) = (h - first) * NB / (last - first + 1)  spreads them evenly; st[b] = index of the first entry
// whose bucket is >= b (NB + 1 16-bit words, NB >= 2 S: 0.375 entries per bucket at the defaults).  A lookup reads st[b] and
// st[b + 1], then the bucket's first two hashes — two dependent LDS round trips, the same for every lane — and only a bucket of
// three or more entries (0.7 % of them) costs a lane more.  Equal hashes share a bucket and the scan ascends, so a hit is the
// FIRST entry with that hash, as the lower bound was.  Round 2 found every entry of the other sketch by binary search: eleven
// dependent round trips per entry, 36 % of this kernel at the C5 slice and 24 % at C2 (-DMH_OJ_JOIN_ONLY / -DMH_OJ_NO_SEARCH
// timing builds).  (An open-addressing hash table was tried first: the probe chains' MAXIMUM over the 64 lanes, not their mean,
// sets a wave's time — 8.0 ms at C2 against the binary search's 4.7.)
// ---- position histograms: an exact early "below the threshold" for the join kernel ---------------------------------------------
// Nine in ten pairs of every workload measured end BELOW THE THRESHOLD, after the two extra passes over both rows that the bottom-k
// Jaccard needs (the ranks of the joined k-mers among the in-window entries; -DMH_OJ_STATS: 408 050 of 451 976 pairs at C2,
// 7.1 M of 8.3 M on one rank's share of configs[4]).  A pair's score is score_table[inter, kk] with inter <= J, the joined k-mers
// inside both windows, and kk = min(in-window entries of either sketch).  A cumulative histogram of the positions of a sketch's
// entries (64 bins over the strand, 128 bytes per entry) bounds the in-window counts from below with two 16-bit loads per sketch;
// pass_min[kk] = the smallest inter that reaches the threshold for any kk' >= kk (from the score table itself, on the host: no
// monotonicity is assumed).  J < pass_min[kk_lb] => the pair cannot be accepted whatever the ranks are: it ends EMPTY-scored here.
constexpr int PH_BINS = 64;
__global__ __launch_bounds__(256) void poshist_kernel(const int32_t* __restrict__ ordered, int64_t stride, const int32_t* __restrict__ meta, int64_t n,
                                                      uint16_t* __restrict__ out) {
  __shared__ uint32_t hist[4][PH_BINS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 4 + wv;
  if (e >= n) return;
  const int32_t* mm = meta + e * META_W;
  const int ne = mm[3] == 0 ? mm[0] : 0, len = mm[1];
  const int w = len > 0 ? (len + PH_BINS - 1) / PH_BINS : 1;
  hist[wv][lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint2* row = (const uint2*)(ordered + e * stride);
  for (int j = lane; j < ne; j += 64) {
    const int pos = (int)row[j].y;
    int b = pos > 0 ? pos / w : 0;
    b = b < PH_BINS - 1 ? b : PH_BINS - 1;
    atomicAdd(&hist[wv][b], 1u);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  uint32_t v = hist[wv][lane];
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(v, off); if (lane >= off) v += t; }
  out[e * PH_BINS + lane] = (uint16_t)v;
}
void launch_poshist(hipStream_t st, const int32_t* ordered, int64_t stride, const int32_t* meta, int64_t n, uint16_t* out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(poshist_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, ordered, stride, meta, n, out);
}
// entries with a position in [x, y], from below: the bins that lie inside the window (bin b = positions [b w, (b + 1) w))
__device__ __forceinline__ int ph_count_lb(const uint16_t* __restrict__ ph, int len, int x, int y) {
  if (y < x) return 0;
  const int w = len > 0 ? (len + PH_BINS - 1) / PH_BINS : 1;
  const int fb = (x + w - 1) / w;
  int lb = (y + 1) / w;                      // bins [fb, lb) are inside: whole bins among 0 .. 62 ...
  lb = lb < PH_BINS - 1 ? lb : PH_BINS - 1;
  if (y + 1 >= len) lb = PH_BINS;            // ... and the last one, which holds everything from 63 w on, when the window reaches the strand's end
  if (lb <= fb) return 0;
  return (int)ph[lb - 1] - (fb > 0 ? (int)ph[fb - 1] : 0);
}

struct OjBuckets { int first, last; uint32_t mult; };
__device__ __forceinline__ OjBuckets oj_buckets(int first, int last, int nb) {
  OjBuckets k;
  k.first = first; k.last = last;
  unsigned long long range = (unsigned long long)(uint32_t)(last - first) + 1ULL;
  if (range < 2ULL * (unsigned long long)nb) range = 2ULL * (unsigned long long)nb;   // (keeps mult below 2^32; a degenerate sketch uses fewer buckets)
  k.mult = (uint32_t)(((unsigned long long)nb << 32) / range);
  return k;
}
__device__ __forceinline__ int oj_bucket_of(const OjBuckets& k, int h) { return (int)__umulhi((uint32_t)(h - k.first), k.mult); }
int overlap_join_table_slots(int S) { int t = 1024; while (t < 2 * S) t <<= 1; return t; }
// bits of a query's filter: 8 per entry where LDS bounds the resident waves (PAIR: 12 % of the other sketch's entries pass), 32 where registers do (TEAM: 3 %)
int overlap_join_filter_bits(int S, int waves) {
  int per = waves == 4 ? 32 : 8;   // (power-of-two filters — TEAM, C5 slice: 16 384 / 32 768 / 65 536 bits 57.3 / 55.3 / 53.2 ms; PAIR, C2: 8 192 / 16 384 / 32 768: 4.04 / 3.75 /
                                   //  3.95.  TEAM's 49 152 bits = 6 KB leave the LDS that sixteen groups per pair need with five workgroups on a CU)
  if (const char* e = getenv("MHAP_OJ_FILTER_BPE")) { const int x = atoi(e); if (x >= 1 && x <= 64) per = x; }   // (experiments)
  long long t = ((long long)per * S + 127) & ~127LL;   // (whole 16-byte vectors of LDS)
  if (t < 1024) t = 1024;
  if (t > 65536) t = 65536;                            // (oj_filter_bit maps sixteen bits of the hash)
  return (int)t;
}

typedef int oj_keep_t __attribute__((ext_vector_type(12)));   // (vectors, not an array: with a dynamic index an array of this size goes to scratch in this kernel; two of
                                                              //  twelve: a vector of 24 takes 32 registers, and up to eight elements the compiler picks by compare and select)
static_assert(OJ_KB == 24 && 12 % OJ_U == 0, "two vectors of twelve blocks, whole trips each");
// The kept positions are addressed by a wave-uniform trip number: s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off, three instructions per
// element.  (History: a switch over static indices kept the array in registers too, but the compiler merged its cases through copies of the
// WHOLE array, a dozen to two dozen v_mov per trip of the streaming loop — found in the ISA when a probe showed that loop at 3.5 TB/s where
// bare row gathers reach 6.1, tools/row_gather_probe.hip; and the stores, written as `block < 12 ? first vector : second`, did the same
// between the two vectors until the streaming loops were split by vector — see `trip` in the kernel.)
#define OJ_KEEP_LOAD(it, pbk, posv) { const int b_ = (it) * OJ_U;                                                               \
    if (b_ < 12) { _Pragma("unroll") for (int u_ = 0; u_ < OJ_U; u_++) posv[u_] = pbk[0][b_ + u_]; }                              \
    else if (b_ < 24) { _Pragma("unroll") for (int u_ = 0; u_ < OJ_U; u_++) posv[u_] = pbk[1][b_ - 12 + u_]; }                    \
    else { _Pragma("unroll") for (int u_ = 0; u_ < OJ_U; u_++) posv[u_] = INT32_MIN; } }

// SHARED = true : a WORKGROUP pulls chunks of candidates; for every run of one query inside the chunk its WAVES waves stage the
//                 query's hashes (and, TABLE, build the bucket table) together — one copy in LDS — then take the run's candidates one
//                 by one from an LDS counter.
// SHARED = false: every wave works alone — pulls its own chunks, keeps its own hashes.
// (the shapes in use and what each is for: OJ_ALONE / OJ_PAIR / OJ_TEAM below)
// (the TEAM shape — candidate-rich queries, pairs with many duplicated-hash groups — collects three groups per round, which costs it
//  registers: it is held at 96 VGPRs = five waves per SIMD, 8-16 B of scratch; the other shapes keep the one-group loop and their 93)
#ifndef MH_OJ_TEAM_MINW
#define MH_OJ_TEAM_MINW 5
#endif
#ifndef MH_OJ_MINW
#define MH_OJ_MINW 4   // waves per SIMD the shapes without the table are compiled for
#endif
template <bool SHARED, int WAVES, bool TABLE, bool FILTER>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES == 4 ? MH_OJ_TEAM_MINW : MH_OJ_MINW, 8))) void overlap_join_kernel(const Candidate* __restrict__ cand, const unsigned long long* __restrict__ cand_count,
                                                                     unsigned long long cand_cap, const int32_t* __restrict__ ordered,
                                                                     int64_t ord_stride, const int32_t* __restrict__ meta,
                                                                     const int32_t* __restrict__ qordered, int64_t qord_stride,
                                                                     const int32_t* __restrict__ qmeta, SearchParams sp,
                                                                     const double* __restrict__ score_table, DevRecord* __restrict__ recs,
                                                                     unsigned long long* __restrict__ rec_count, unsigned long long rec_cap,
                                                                     unsigned long long* __restrict__ compared, Candidate* __restrict__ slow,
                                                                     unsigned long long* __restrict__ slow_count, int chunk,
                                                                     unsigned long long* __restrict__ work, int ts,
                                                                     const uint16_t* __restrict__ ph, const uint16_t* __restrict__ qph,
                                                                     const int32_t* __restrict__ pass_min) {
  extern __shared__ int32_t oj_lds[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  constexpr bool APOS = SHARED && MH_OJ_KEEP;                          // the shared query's positions are staged too
  const bool keepb = MH_OJ_KEEP && sp.S <= 64 * OJ_KB;                 // the other sketch's positions stay in registers
  static_assert(!(TABLE && FILTER) && (!FILTER || SHARED), "one lookup aid per shape; the filter is built by a workgroup");
  const int tabw = TABLE ? (ts / 2 + 4) & ~3 : (FILTER ? ts / 32 : 0);  // ints of the table (ts + 1 shorts, padded: what follows stays 16-byte aligned) / of the filter (ts bits)
  const int spad = (sp.S + 3) & ~3, own = spad + tabw + (APOS ? spad : 0);   // ints of the hashes (+ the table / the filter) (+ the positions)
  int32_t* ah = SHARED ? oj_lds : oj_lds + (size_t)wv * (own + OJ_LDS_EXTRA);   // the query sketch's hashes,
  uint16_t* st = (uint16_t*)(ah + spad);                               // ... (TABLE) the bucket starts over them,
  uint32_t* bm = (uint32_t*)(ah + spad);                               // ... (FILTER) a bit per hash value mod ts,
  int32_t* ap = ah + spad + tabw;                                      // ... (APOS) its positions,
  int32_t* svar = oj_lds + own;                                        // (SHARED) {-, next candidate of the run, chunk start lo, hi}
  int32_t* jp1 = SHARED ? svar + 4 + (size_t)wv * OJ_LDS_EXTRA : ah + own;   // per wave — join: position in the query / in the other sketch,
  int32_t* jp2 = jp1 + OJ_JCAP;
  uint16_t* cq = (uint16_t*)jp1;                                       // (FILTER, during the join: jp1 is filled after it) ring of OJ_CQ entry indices that passed the filter
  uint32_t* jij = (uint32_t*)(jp2 + OJ_JCAP);                          // ... entry indices (i | j << 16)
  int32_t* sh = (int32_t*)jij;                                         // (later) shifts of the current records, median by counting
  int32_t* gi = (int32_t*)(jij + OJ_JCAP);                             // groups: {first i, first j, m, n, first record, records}
  int32_t* gpa = gi + OJ_GCAP * 6;                                     // ... positions of the group's entries in the query
  int32_t* gpb = gpa + OJ_GCAP * OJ_GLEN;                              // ... and in the other sketch
  OjBuckets bk = {0, 0, 0u};
  unsigned long long n = *cand_count;
  if (n > cand_cap) n = cand_cap;
  int curq = -1, nA = 0, len1 = 0;
  const int32_t* qrow = nullptr;
  unsigned long long mine = 0;
  constexpr int NT = SHARED ? 64 * WAVES : 64;                      // threads that stage one query
  const int tid = SHARED ? (int)threadIdx.x : lane;
  // one candidate pair, by the wave
  auto one_candidate = [&](const Candidate cd) {
      const int32_t* mm = meta + (int64_t)cd.m * META_W;
      const int nB = __builtin_amdgcn_readfirstlane(mm[0]), len2 = __builtin_amdgcn_readfirstlane(mm[1]);
      const uint2* brow = (const uint2*)(ordered + (int64_t)cd.m * ord_stride);
      // ---- join ----
      int nj = 0, ng = 0;
      bool bad = false, bad_groups = false;   // bad_groups: handed over for the duplicated-hash group caps (a wider pass would hand the pair on again)
      oj_keep_t pbk[2];      // (keepb) positions of the other sketch: entry blk * 64 + lane in pbk[blk / 12][blk % 12]
      if (nA > 0 && nB > 0) {
        if constexpr (FILTER) {
        // Filter, compact, look up.  1 entry in 40 of the other sketch has a partner in the query (a true overlap; a handful for a pair that
        // shares a repeat), but a wave looked all 64 entries of a block up and went through the found-entry code for nearly every block.
        // Now a block costs one LDS word per entry — the query's filter, a bit per hash value mod ts: 5-9 % of the entries pass — and
        // the indices of those that pass are queued (a ring of OJ_CQ 16-bit words); whenever 64 are waiting they are handled as ONE
        // dense block: entry and neighbours re-read (L2), bisection among the query's hashes, run / group detection.  At S = 1536 that
        // is three or four dense blocks per pair instead of twenty-four sparse ones.
        const int p2 = 1 << (31 - __builtin_clz((unsigned)nA));   // largest power of two <= nA
        int qn = 0, qd = 0, kit = 0;   // entries queued / handled (wave-uniform)
        // (loads past the sketch's end read its last entry again — no bounds branch around a load; what such lanes queue is dropped when the
        //  queue is handled, and the rank pass trims the last block's mask)
        uint2 en[OJ_U];
#pragma unroll
        for (int u = 0; u < OJ_U; u++) { const int j = u * 64 + lane; en[u] = brow[j < nB ? j : nB - 1]; }
        // One trip of the streaming loop (returns false after the row's last one).  The kept positions of a trip go to ONE of the two
        // twelve-element vectors, and which one is static at each of the three places the trip is instantiated from below: written as one
        // loop with `block < 12 ? pbk[0] : pbk[1]` the compiler joined the two cases through copies of a whole vector — six to eighteen
        // v_mov_b64 in every trip (read in the ISA, round 5).
        auto trip = [&](const int j0, auto WHICH) -> bool {
          constexpr int which = decltype(WHICH)::value;   // 0 / 1: the vector this trip's positions are kept in; 2: none (beyond OJ_KB blocks)
          const bool more = j0 < nB && !bad;
          if (more) {
            uint2 e[OJ_U];
            uint32_t w[OJ_U];
#pragma unroll
            for (int u = 0; u < OJ_U; u++) {
              e[u] = en[u];
              const int jn = j0 + (u + OJ_U) * 64 + lane;
              en[u] = brow[jn < nB ? jn : nB - 1];
            }
            if (keepb && which < 2) {
              const int b_ = (kit - (12 / OJ_U) * which) * OJ_U;   // 0, 3, 6, 9 inside the vector
#pragma unroll
              for (int u = 0; u < OJ_U; u++) pbk[which < 2 ? which : 0][b_ + u] = (int)e[u].y;
              kit++;
            }
            uint32_t fb[OJ_U];
#pragma unroll
            for (int u = 0; u < OJ_U; u++) { fb[u] = oj_filter_bit(e[u].x, (uint32_t)ts); w[u] = bm[fb[u] >> 5]; }
#pragma unroll
            for (int u = 0; u < OJ_U; u++) {
              const int jb = j0 + u * 64;
              if (jb < nB) {
#ifdef MH_OJ_NO_SEARCH
                const bool c = (e[u].x ^ e[u].y) == 0x7ffffffeu && w[u] == 0x12345u;   // (timing experiment: the rows are streamed, nothing passes the filter; results are wrong)
#else
                const bool c = ((w[u] >> (fb[u] & 31u)) & 1u) != 0u;
#endif
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(c);
                if (bal) {
                  if (c) cq[(qn + oj_mbcnt(bal)) & (OJ_CQ - 1)] = (uint16_t)(jb + lane);
                  qn += __popcll(bal);
                }
              }
            }
          }
          // whole blocks of queued entries — at the row's end whatever is left.  (A trip adds at most 64 OJ_U entries to at most 63.)
          while (!bad && (qn - qd >= 64 || (!more && qn > qd))) {
            oj_lds_sync();
            const int cnt = qn - qd < 64 ? qn - qd : 64;
            bool act = lane < cnt;
            int j = 0, hb = 0, pb = 0, hprev = 0, hnext = 0;
            if (act) j = cq[(qd + lane) & (OJ_CQ - 1)];
            act = act && j < nB;                       // (the last block's lanes past the sketch's end queue themselves too)
            if (act) {
              const uint2 be = brow[j];
              hb = (int)be.x; pb = (int)be.y;
              if (j > 0) hprev = (int)brow[j - 1].x;
              if (j + 1 < nB) hnext = (int)brow[j + 1].x;
            }
            qd += cnt;
            int l = (ah[p2 - 1] < hb) ? nA - p2 : -1;   // lower bound by bisection: fixed probe sequence for a sorted array of any length
            for (int q = p2 >> 1; q > 0; q >>= 1) l = (ah[l + q] < hb) ? l + q : l;
            l += 1;
            const bool found = act && l < nA && ah[l < nA ? l : 0] == hb;
            if (__builtin_amdgcn_ballot_w64(found)) {
              // the first entry of a run of equal hashes in the other sketch speaks for the run
              const bool leader = found && !(j > 0 && hprev == hb);
              bool grp = false;
              if (leader) grp = (l + 1 < nA && ah[l + 1] == hb) || (j + 1 < nB && hnext == hb);
              const bool reg = leader && !grp;
              const unsigned long long balr = __builtin_amdgcn_ballot_w64(reg), balg = __builtin_amdgcn_ballot_w64(grp);
              if (nj + __popcll(balr) > OJ_JCAP || ng + __popcll(balg) > OJ_GCAP) { bad = true; bad_groups = bad_groups || ng + __popcll(balg) > OJ_GCAP; OJ_STAT(ng + __popcll(balg) > OJ_GCAP ? 15 : 14, 1); }
              else {
                if (reg) {
                  const int idx = nj + oj_mbcnt(balr);
                  jp2[idx] = pb;
                  jij[idx] = (uint32_t)l | ((uint32_t)j << 16);
                }
                if (grp) {
                  const int idx = ng + oj_mbcnt(balg);
                  gi[idx * 6 + 0] = l; gi[idx * 6 + 1] = j;
                }
                nj += __popcll(balr);
                ng += __popcll(balg);
              }
            }
          }
          return more;
        };
        {
          static_assert(12 % OJ_U == 0, "whole trips per kept vector");
          constexpr int TPV = 12 / OJ_U;   // trips per kept vector
          int j0 = 0;
          bool go = true;
          for (int t = 0; go && t < TPV; t++, j0 += 64 * OJ_U) go = trip(j0, std::integral_constant<int, 0>());
          for (int t = 0; go && t < TPV; t++, j0 += 64 * OJ_U) go = trip(j0, std::integral_constant<int, 1>());
          for (; go; j0 += 64 * OJ_U) go = trip(j0, std::integral_constant<int, 2>());
        }
        } else {
        int carry = 0;   // hash of the last entry of the previous OJ_U blocks (run detection across blocks)
        int kit = 0;
        uint2 en[OJ_U];
#pragma unroll
        for (int u = 0; u < OJ_U; u++) { const int j = u * 64 + lane; en[u] = make_uint2(0u, 0u); if (j < nB) en[u] = brow[j]; }
#ifdef MH_OJ_NO_SEARCH
        for (int j0 = 0; j0 < nB && !bad; j0 += 64 * OJ_U) {   // (timing experiment: the rows are streamed, nothing is looked up)
          int acc = 0;
#pragma unroll
          for (int u = 0; u < OJ_U; u++) { acc += (int)en[u].x; const int jn = j0 + (u + OJ_U) * 64 + lane; en[u] = make_uint2(0u, 0u); if (jn < nB) en[u] = brow[jn]; }
          if (acc == 0x7fffffff) bad = true;
        }
        const int jstart = nB;
#else
        const int jstart = 0;
#endif
        // (one trip; the vector its positions are kept in is static at each place it is instantiated from, as in the filter branch above)
        auto trip = [&](const int j0, auto WHICH) {
          constexpr int which = decltype(WHICH)::value;
          // OJ_U blocks of 64 entries at a time: their binary searches (dependent LDS reads) overlap each other and the loads
          // of the next OJ_U blocks
          uint2 e[OJ_U];
          int l[OJ_U];
#pragma unroll
          for (int u = 0; u < OJ_U; u++) {
            e[u] = en[u];
            const int jn = j0 + (u + OJ_U) * 64 + lane;
            en[u] = make_uint2(0u, 0u);
            if (jn < nB) en[u] = brow[jn];
          }
          if (keepb && which < 2) {
            const int b_ = (kit - (12 / OJ_U) * which) * OJ_U;
#pragma unroll
            for (int u = 0; u < OJ_U; u++) pbk[which < 2 ? which : 0][b_ + u] = (int)e[u].y;
            kit++;
          }
          bool found[OJ_U];
          bool anyf = false;
          if constexpr (TABLE) {
            // bucket lookup (above): st[b], st[b + 1], then the bucket's first two hashes — the OJ_U entries' reads are independent
            int i0[OJ_U], i1[OJ_U], x0[OJ_U], x1[OJ_U];
#pragma unroll
            for (int u = 0; u < OJ_U; u++) {
              const int hb = (int)e[u].x;
              const int b = (hb >= bk.first && hb <= bk.last) ? oj_bucket_of(bk, hb) : 0;
              i0[u] = st[b]; i1[u] = st[b + 1];
            }
#pragma unroll
            for (int u = 0; u < OJ_U; u++) { x0[u] = ah[i0[u]]; x1[u] = ah[i0[u] + 1]; }   // (past the bucket / the sketch: read, never used)
#pragma unroll
            for (int u = 0; u < OJ_U; u++) {
              const int j = j0 + u * 64 + lane;
              const int hb = (int)e[u].x;
              l[u] = -1;
              if (j < nB && hb >= bk.first && hb <= bk.last && i0[u] < i1[u]) {
                if (x0[u] == hb) l[u] = i0[u];
                else if (i0[u] + 1 < i1[u]) {
                  if (x1[u] == hb) l[u] = i0[u] + 1;
                  else for (int i = i0[u] + 2; i < i1[u]; i++) if (ah[i] == hb) { l[u] = i; break; }
                }
              }
              found[u] = l[u] >= 0;
              anyf |= found[u];
            }
          } else {
            // no table (its 8 KB would cost resident waves, and this kernel's time is inversely proportional to them): lower bound of every hash among the query's by binary search, l = last index whose hash
            // is smaller (-1: none).  Fixed probe sequence for a sorted array of any length (the first probe splits [0, nA) into two
            // overlapping halves of p2 entries); the OJ_U searches' dependent LDS reads overlap each other
            const int p2 = 1 << (31 - __builtin_clz((unsigned)nA));   // largest power of two <= nA
#pragma unroll
            for (int u = 0; u < OJ_U; u++) l[u] = (ah[p2 - 1] < (int)e[u].x) ? nA - p2 : -1;
            for (int q = p2 >> 1; q > 0; q >>= 1) {
#pragma unroll
              for (int u = 0; u < OJ_U; u++) l[u] = (ah[l[u] + q] < (int)e[u].x) ? l[u] + q : l[u];
            }
#pragma unroll
            for (int u = 0; u < OJ_U; u++) {
              const int j = j0 + u * 64 + lane;
              l[u] += 1;
              found[u] = j < nB && l[u] < nA && ah[l[u] < nA ? l[u] : 0] == (int)e[u].x;
              anyf |= found[u];
            }
          }
#ifdef MH_OJ_NO_FOUND
          { int acc = 0;   // (timing experiment: the lookups are done, what they find is dropped; results are wrong)
#pragma unroll
            for (int u = 0; u < OJ_U; u++) acc ^= l[u] + (found[u] ? 7 : 0);
            if (acc == 0x12345678) bad = true;
            anyf = false; }
#endif
          if (__any(anyf)) {
#pragma unroll
            for (int u = 0; u < OJ_U; u++) {
              if (!bad && __any(found[u])) {
                const int j = j0 + u * 64 + lane;
                const int hb = (int)e[u].x;
                int hprev = __shfl_up(hb, 1);
                if (lane == 0) hprev = u ? __builtin_amdgcn_readlane((int)e[u ? u - 1 : 0].x, 63) : carry;
                // the first entry of a run of equal hashes in the other sketch speaks for the run
                const bool leader = found[u] && !(j > 0 && hprev == hb);
                int hnext = __shfl_down(hb, 1);
                if (lane == 63 && leader && j + 1 < nB) hnext = (int)brow[j + 1].x;
                bool grp = false;
                if (leader) grp = (l[u] + 1 < nA && ah[l[u] + 1] == hb) || (j + 1 < nB && hnext == hb);
                const bool reg = leader && !grp;
                const unsigned long long balr = __ballot(reg), balg = __ballot(grp);
                if (nj + __popcll(balr) > OJ_JCAP || ng + __popcll(balg) > OJ_GCAP) { bad = true; bad_groups = bad_groups || ng + __popcll(balg) > OJ_GCAP; }
                else {
                  if (reg) {
                    const int idx = nj + oj_mbcnt(balr);
                    jp2[idx] = (int)e[u].y;
                    jij[idx] = (uint32_t)l[u] | ((uint32_t)j << 16);
                  }
                  if (grp) {
                    const int idx = ng + oj_mbcnt(balg);
                    gi[idx * 6 + 0] = l[u]; gi[idx * 6 + 1] = j;
                  }
                  nj += __popcll(balr);
                  ng += __popcll(balg);
                }
              }
            }
          }
          carry = __builtin_amdgcn_readlane((int)e[OJ_U - 1].x, 63);
        };
        {
          constexpr int TPV = 12 / OJ_U;
          int j0 = jstart;
          for (int t = 0; t < TPV && j0 < nB && !bad; t++, j0 += 64 * OJ_U) trip(j0, std::integral_constant<int, 0>());
          for (int t = 0; t < TPV && j0 < nB && !bad; t++, j0 += 64 * OJ_U) trip(j0, std::integral_constant<int, 1>());
          for (; j0 < nB && !bad; j0 += 64 * OJ_U) trip(j0, std::integral_constant<int, 2>());
        }
        }
      }
      oj_lds_sync();
#ifdef MH_OJ_NO_GROUPS
      ng = 0;   // (timing experiment: the duplicated-hash groups are dropped; results are wrong)
#endif
      int gtot = 0;
      if constexpr (WAVES == 4) {
      // collect the groups' entries, THREE groups per round (each round waits for a load from the other sketch's row): lanes 20 t .. 20 t + 8
      // read the query's entries of the round's t-th group, lanes 20 t + 10 .. 20 t + 18 the other sketch's
      for (int g0 = 0; g0 < ng && !bad; g0 += 3) {
        const int t = lane / 20, r = lane - 20 * t;           // lanes 60..63: t = 3, idle
        const int g = g0 + t;
        const bool live = t < 3 && g < ng;
        const int x = r < 10 ? r : r - 10;
        const int lo = live ? gi[g * 6 + 0] : 0, j = live ? gi[g * 6 + 1] : 0;
        const int h = ah[lo];
        const bool a_ok = live && r <= OJ_GLEN && lo + x < nA && ah[lo + x] == h;
        uint2 be = make_uint2(0u, 0u);
        const bool b_in = live && r >= 10 && r <= 10 + OJ_GLEN && j + x < nB;
        if (b_in) be = brow[j + x];
        const bool b_ok = b_in && (int)be.x == h;
        const unsigned long long bala = __ballot(a_ok), balb = __ballot(b_ok);
        const int sh = 20 * (t < 3 ? t : 0);
        const int m = __popcll((bala >> sh) & 0x3FFULL), nn = __popcll((balb >> (sh + 10)) & 0x3FFULL);
        if (__any(live && (m > OJ_GLEN || nn > OJ_GLEN))) { bad = true; bad_groups = true; OJ_STAT(16, 1); break; }
        if (a_ok) gpa[g * OJ_GLEN + x] = APOS ? ap[lo + x] : qrow[2 * (lo + x) + 1];
        if (b_ok) gpb[g * OJ_GLEN + x] = (int)be.y;
        int sz[3];   // entries of the round's groups (wave-uniform)
#pragma unroll
        for (int tt = 0; tt < 3; tt++)
          sz[tt] = g0 + tt < ng ? __popcll((bala >> (20 * tt)) & 0x3FFULL) + __popcll((balb >> (20 * tt + 10)) & 0x3FFULL) : 0;
        // (gi[4]: where the group's records go — behind the joined k-mers, every group as many words as it has entries: oj_pass)
        if (live && r == 0) { gi[g * 6 + 2] = m; gi[g * 6 + 3] = nn; gi[g * 6 + 4] = nj + gtot + (t >= 1 ? sz[0] : 0) + (t >= 2 ? sz[1] : 0); }
        gtot += sz[0] + sz[1] + sz[2];
      }
      } else {
      for (int g = 0; g < ng && !bad; g++) {   // collect the groups' entries: lanes 0..8 the query's, lanes 16..24 the other sketch's
        const int lo = gi[g * 6 + 0], j = gi[g * 6 + 1];
        const int h = ah[lo];
        const int x = lane & 15;
        const bool a_ok = lane <= OJ_GLEN && lo + x < nA && ah[lo + x] == h;
        const bool b_ok = lane >= 16 && lane <= 16 + OJ_GLEN && j + x < nB && (int)brow[j + x].x == h;
        const int m = __popcll(__ballot(a_ok)), nn = __popcll(__ballot(b_ok));
        if (m > OJ_GLEN || nn > OJ_GLEN) { bad = true; bad_groups = true; break; }
        if (a_ok) gpa[g * OJ_GLEN + x] = APOS ? ap[lo + x] : qrow[2 * (lo + x) + 1];
        if (b_ok) gpb[g * OJ_GLEN + x] = (int)brow[j + x].y;
        if (lane == 0) { gi[g * 6 + 2] = m; gi[g * 6 + 3] = nn; gi[g * 6 + 4] = nj + gtot; }
        gtot += m + nn;
      }
      }
      if (!bad && nj + gtot > OJ_JCAP) { bad = true; OJ_STAT(17, 1); }
      if (bad) {
        // (slow_count[8]: how many of the pairs handed over were handed over for the group caps — what the host decides a wider pass by)
        if (lane == 0) { const unsigned long long slot = atomicAdd(slow_count, 1ULL); slow[slot] = cd; if (bad_groups) atomicAdd(slow_count + 8, 1ULL); }
        return;
      }
      mine++;
      // OverlapInfo.EMPTY (score 0, all zero) unless the pair gets through every stage below
      double score = 0.0;
      int valid = 0, a1 = 0, a2 = 0, b1 = 0, b2 = 0;
      do {
#ifdef MH_OJ_JOIN_ONLY
        if (nj >= 0) break;   // (timing experiment: everything after the join skipped; results are wrong)
#endif
        OJ_STAT(6, nj); OJ_STAT(8, ng); OJ_STAT(9, ng > 0 ? 1 : 0); OJ_STAT(10, ng >= 3 ? 1 : 0); OJ_STAT(11, gtot);
        if (ng == 0 && nj < 3) { OJ_STAT(0, 1); break; }   // computeEdges needs three valid records (:126): fewer joined k-mers can only end EMPTY
        int iA[OJ_R], jB[OJ_R];   // the joined k-mers' entry indices move to registers, their LDS words become `sh`
#pragma unroll
        for (int r = 0; r < OJ_R; r++) {
          const int t = r * 64 + lane;
          const uint32_t ij = t < nj ? jij[t] : 0xffffffffu;
          iA[r] = (int)(ij & 0xffffu); jB[r] = (int)(ij >> 16);
          if (t < nj) jp1[t] = APOS ? ap[iA[r]] : qrow[2 * iA[r] + 1];
        }
        oj_lds_sync();
        // ---- recordMatchingKmers twice (:600-606), median shift after each ----
        const int shift_lb = 32 - __builtin_clz((unsigned)((len1 > len2 ? len1 : len2) | 1));   // (2^shift_lb > either length)
        int count = 0;
        const int nx = gtot;   // words behind the joined k-mers that the groups' records may take
        ShiftStats st = oj_shift_stats(false, 0, len1, len2, sp.max_shift);
        uint32_t fl = oj_pass<true>(jp1, jp2, nj, ng, nx, gi, gpa, gpb, len1, len2, st, lane, count);
        if (count <= 0) { OJ_STAT(1, 1); break; }
        st = oj_shift_stats(true, oj_median_shift(jp1, jp2, sh, fl, nj + nx, count, lane, shift_lb), len1, len2, sp.max_shift);
        fl = oj_pass<false>(jp1, jp2, nj, ng, nx, gi, gpa, gpb, len1, len2, st, lane, count);
        if (count <= 0) { OJ_STAT(2, 1); break; }
        st = oj_shift_stats(true, oj_median_shift(jp1, jp2, sh, fl, nj + nx, count, lane, shift_lb), len1, len2, sp.max_shift);
        // optimizeShifts (:156-189): neighbouring records of one query position exist only inside a group
        // (lane g walks group g's records and marks the dropped ones, then every lane looks at its own entries)
        int removed = 0;
        if (ng) {
          int rem = 0;
          if (lane < ng) {
            const int start = gi[lane * 6 + 4], k = gi[lane * 6 + 5];
            int red = -1, rp1 = 0, rp2 = 0;
            for (int x = 0; x < k; x++) {
              const int t = start + x;
              const int p1 = jp1[t], p2 = jp2[t];
              if (red >= 0 && rp1 == p1) {
                if (iabs32((rp2 - rp1) - st.med) > iabs32((p2 - p1) - st.med)) { jp1[red] = INT32_MIN; red = t; rp1 = p1; rp2 = p2; }
                else jp1[t] = INT32_MIN;
                rem++;
              } else { red = t; rp1 = p1; rp2 = p2; }
            }
          }
          if (__builtin_amdgcn_ballot_w64(rem != 0)) {
            oj_lds_sync();
#pragma unroll
            for (int r = 0; r < OJ_R; r++) {
              if (r * 64 < nj + nx) {
                const int t = r * 64 + lane;
                const bool gone = ((fl >> r) & 1u) && t >= nj && jp1[t] == INT32_MIN;
                if (gone) fl &= ~(1u << r);
                removed += __popcll(__builtin_amdgcn_ballot_w64(gone));
              }
            }
          }
        }
        OJ_STAT(12, nx); OJ_STAT(13, removed);
        if (removed) {
          count -= removed;
          st = oj_shift_stats(true, oj_median_shift(jp1, jp2, sh, fl, nj + nx, count, lane, shift_lb), len1, len2, sp.max_shift);
        }
        // computeEdges (:90-137)
        int le1 = INT32_MAX, le2 = INT32_MAX, re1 = INT32_MIN, re2 = INT32_MIN, nvalid = 0;
#pragma unroll
        for (int r = 0; r < OJ_R; r++) {
          if (r * 64 < nj + nx) {
            bool ok = (fl >> r) & 1u;
            if (ok) {
              const int t = r * 64 + lane;
              const int p1 = jp1[t], p2 = jp2[t];
              ok = !(iabs32((p2 - p1) - st.med) > st.absmax);
              if (ok) {
                le1 = p1 < le1 ? p1 : le1; le2 = p2 < le2 ? p2 : le2;
                re1 = p1 > re1 ? p1 : re1; re2 = p2 > re2 ? p2 : re2;
              }
            }
            nvalid += __popcll(__ballot(ok));
          }
        }
        if (nvalid < 3) { OJ_STAT(3, 1); break; }
        le1 = oj_wave_min(le1); le2 = oj_wave_min(le2); re1 = oj_wave_max(re1); re2 = oj_wave_max(re2);
        const double den = (double)(nvalid - 1);
        const int32_t na1 = (int32_t)((uint32_t)nvalid * (uint32_t)le1 - (uint32_t)re1);   // int products wrap like Java's (:131-134)
        const int32_t na2 = (int32_t)((uint32_t)nvalid * (uint32_t)re1 - (uint32_t)le1);
        const int32_t nb1 = (int32_t)((uint32_t)nvalid * (uint32_t)le2 - (uint32_t)re2);
        const int32_t nb2 = (int32_t)((uint32_t)nvalid * (uint32_t)re2 - (uint32_t)le2);
        a1 = (int)java_round((double)na1 / den); if (a1 < 0) a1 = 0;
        a2 = (int)java_round((double)na2 / den); if (a2 > len1) a2 = len1;
        b1 = (int)java_round((double)nb1 / den); if (b1 < 0) b1 = 0;
        b2 = (int)java_round((double)nb2 / den); if (b2 > len2) b2 = len2;
        valid = nvalid;
        if (ph != nullptr) {
          // the early "below the threshold" (poshist_kernel above): J = joined k-mers inside both windows + what the groups can add
          int J = 0;
#pragma unroll
          for (int r = 0; r < OJ_R; r++) {
            if (r * 64 < nj) {
              const int t = r * 64 + lane;
              bool in = false;
              if (t < nj) { const int p1 = jp1[t], p2 = jp2[t]; in = p1 >= a1 && p1 <= a2 && p2 >= b1 && p2 <= b2; }
              J += __popcll(__ballot(in));
            }
          }
          for (int g = 0; g < ng; g++) { const int m = gi[g * 6 + 2], nn = gi[g * 6 + 3]; J += m < nn ? m : nn; }
          const int s1lb = ph_count_lb(qph + (int64_t)cd.q * PH_BINS, len1, a1, a2), s2lb = ph_count_lb(ph + (int64_t)cd.m * PH_BINS, len2, b1, b2);
          const int kklb = s1lb < s2lb ? s1lb : s2lb;
          if (J < pass_min[kklb]) { OJ_STAT(4, 1); break; }     // score stays 0: below any threshold that was asked for
        }
        // ---- computeKBottomSketchJaccard (:304-364): in-window counts, and for every joined k-mer (and every group's first
        // entries) its rank among the in-window entries of either sketch: prefix counts over 64-entry blocks, the lane that
        // holds entry i of the block hands the rank to the lane that holds the joined k-mer ----
        int rA[OJ_R], rB[OJ_R];
#pragma unroll
        for (int r = 0; r < OJ_R; r++) { rA[r] = 0; rB[r] = 0; }
        // lane g < ng speaks for group g
        int giA = 0xffff, gjB = 0xffff, grA = 0, grB = 0, gmin = 0;
        if (lane < ng) {
          giA = gi[lane * 6 + 0]; gjB = gi[lane * 6 + 1];
          const int32_t *pa = gpa + lane * OJ_GLEN, *pb = gpb + lane * OJ_GLEN;
          const int ca = __popc(oj_mask8(pa, gi[lane * 6 + 2], a1, a2 + 1)), cb = __popc(oj_mask8(pb, gi[lane * 6 + 3], b1, b2 + 1));
          gmin = ca < cb ? ca : cb;   // equal hashes pair up one to one in the union walk
        }
        const int jrounds = (nj + 63) >> 6;
        int s1 = 0, s2 = 0;
        // One pass over the positions of either sketch: a block of 64 entries costs its in-window test, the ballot and three
        // v_writelane — lane b of the wave collects block b's mask and the count ahead of it — and the joined k-mers fetch
        // their block's words from that lane afterwards (three shuffles per round of 64 joined k-mers and sketch).  Round 3
        // handed every block's ranks to the joined k-mers as it went: two or three shuffles and ten more instructions per BLOCK.
        int pan[OJ_U], pbn[OJ_U];   // next OJ_U blocks of positions of either sketch, in flight while the current ones are tested
#pragma unroll
        for (int u = 0; u < OJ_U; u++) {
          const int i = u * 64 + lane;
          pan[u] = (!APOS && i < nA) ? qrow[2 * i + 1] : INT32_MIN;
          pbn[u] = (!keepb && i < nB) ? (int)brow[i].y : INT32_MIN;
        }
        for (int cb = 0; cb < nA; cb += 64 * OJ_RCH) {
          uint32_t mlo = 0u, mhi = 0u;
          int mpre = 0;
          const int cend = cb + 64 * OJ_RCH < nA ? cb + 64 * OJ_RCH : nA;
          for (int ib = cb; ib < cend; ib += 64 * OJ_U) {
            int posv[OJ_U];
#pragma unroll
            for (int u = 0; u < OJ_U; u++) {
              if (APOS) posv[u] = ap[ib + u * 64 + lane];   // (past the sketch: whatever follows in LDS — the last block's ballot is trimmed below)
              else {
                posv[u] = pan[u];
                const int i = ib + (u + OJ_U) * 64 + lane;
                pan[u] = i < nA ? qrow[2 * i + 1] : INT32_MIN;
              }
            }
#pragma unroll
            for (int u = 0; u < OJ_U; u++) {
              const int i0 = ib + u * 64;
              if (i0 < nA) {
                // (read from memory, entries past the sketch hold INT32_MIN, and a1 >= 0; a ballot per comparison: the compiler turns a ballot of their conjunction into
                // two more vector instructions)
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(posv[u] >= a1) & __builtin_amdgcn_ballot_w64(posv[u] <= a2);
                const int blk = (i0 - cb) >> 6;
                mlo = (uint32_t)oj_writelane((int)(uint32_t)bal, blk, (int)mlo);
                mhi = (uint32_t)oj_writelane((int)(uint32_t)(bal >> 32), blk, (int)mhi);
                mpre = oj_writelane(s1, blk, mpre);
                s1 += __popcll(bal);
              }
            }
          }
          if (APOS && cend == nA && (nA & 63)) oj_trim_last_block(mlo, mhi, s1, (nA - 1 - cb) >> 6, nA & 63);   // (what the LDS words behind the positions happened to hold)
#pragma unroll
          for (int r = 0; r < OJ_R; r++) {
            if (r < jrounds) {
              const int v = oj_rank_from(mlo, mhi, mpre, iA[r] - cb);
              if (iA[r] >= cb && iA[r] < cend) rA[r] = v;
            }
          }
          if (ng) { const int v = oj_rank_from(mlo, mhi, mpre, giA - cb); if (giA >= cb && giA < cend) grA = v; }
        }
        for (int cb = 0, kit2 = 0; cb < nB; cb += 64 * OJ_RCH) {
          uint32_t mlo = 0u, mhi = 0u;
          int mpre = 0;
          const int cend = cb + 64 * OJ_RCH < nB ? cb + 64 * OJ_RCH : nB;
          for (int jb = cb; jb < cend; jb += 64 * OJ_U, kit2++) {
            int posv[OJ_U];
            if (keepb) { OJ_KEEP_LOAD(kit2, pbk, posv) }
            else {
#pragma unroll
              for (int u = 0; u < OJ_U; u++) {
                posv[u] = pbn[u];
                const int j = jb + (u + OJ_U) * 64 + lane;
                pbn[u] = j < nB ? (int)brow[j].y : INT32_MIN;
              }
            }
#pragma unroll
            for (int u = 0; u < OJ_U; u++) {
              const int j0 = jb + u * 64;
              if (j0 < nB) {
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(posv[u] >= b1) & __builtin_amdgcn_ballot_w64(posv[u] <= b2);
                const int blk = (j0 - cb) >> 6;
                mlo = (uint32_t)oj_writelane((int)(uint32_t)bal, blk, (int)mlo);
                mhi = (uint32_t)oj_writelane((int)(uint32_t)(bal >> 32), blk, (int)mhi);
                mpre = oj_writelane(s2, blk, mpre);
                s2 += __popcll(bal);
              }
            }
          }
          if (keepb && cend == nB && (nB & 63)) oj_trim_last_block(mlo, mhi, s2, (nB - 1 - cb) >> 6, nB & 63);   // (the kept registers of lanes past the sketch's end)
#pragma unroll
          for (int r = 0; r < OJ_R; r++) {
            if (r < jrounds) {
              const int v = oj_rank_from(mlo, mhi, mpre, jB[r] - cb);
              if (jB[r] >= cb && jB[r] < cend) rB[r] = v;
            }
          }
          if (ng) { const int v = oj_rank_from(mlo, mhi, mpre, gjB - cb); if (gjB >= cb && gjB < cend) grB = v; }
        }
        const int kk = s1 < s2 ? s1 : s2;
        // a joined k-mer counts if its index in the merged union (in-window entries of both, joined ones once) is below k:
        // index = rank in the query + rank in the other sketch - joined in-window k-mers ahead of it
        int inter = 0, before = 0;
        bool both[OJ_R];
#pragma unroll
        for (int r = 0; r < OJ_R; r++) {
          both[r] = false;
          if (r < jrounds) {
            const int t = r * 64 + lane;
            if (t < nj) { const int p1 = jp1[t], p2 = jp2[t]; both[r] = p1 >= a1 && p1 <= a2 && p2 >= b1 && p2 <= b2; }
            const unsigned long long bal = __ballot(both[r]);
            int m = before + oj_mbcnt(bal);
            for (int g = 0; g < ng; g++)   // pairs of the groups ahead of it
              m += (__builtin_amdgcn_readlane(giA, g) < iA[r]) ? __builtin_amdgcn_readlane(gmin, g) : 0;
            inter += __popcll(__ballot(both[r] && rA[r] + rB[r] - m < kk));
            before += __popcll(bal);
          }
        }
        int gacc = 0;   // pairs of the groups ahead of group g
        for (int g = 0; g < ng; g++) {
          const int glo = __builtin_amdgcn_readlane(giA, g), gm = __builtin_amdgcn_readlane(gmin, g);
          int ahead = gacc;
#pragma unroll
          for (int r = 0; r < OJ_R; r++)
            if (r < jrounds) ahead += __popcll(__ballot(both[r] && iA[r] < glo));
          const int base = __builtin_amdgcn_readlane(grA, g) + __builtin_amdgcn_readlane(grB, g) - ahead;
          int take = kk - base;                    // the group's pairs sit at union indices base, base+1, ...
          take = take < 0 ? 0 : (take > gm ? gm : take);
          inter += take;
          gacc += gm;
        }
        score = score_table[score_index(inter, kk)];
        OJ_STAT(score >= sp.threshold ? 5 : 4, 1);
        OJ_STAT(7, before);
      } while (0);
      if (score >= sp.threshold && lane == 0) {                                                  // MinHashSearch.java:229
        const unsigned long long slot = atomicAdd(rec_count, 1ULL);
        if (slot < rec_cap) {
          DevRecord d;
          d.q = cd.q; d.m = cd.m; d.score = score; d.raw = valid; d.a1 = a1; d.a2 = a2; d.b1 = b1; d.b2 = b2; d.pad = 0;
          recs[slot] = d;
        }
      }
  };
  // chunks of consecutive candidates (one query's candidates are contiguous) are pulled from a counter: a static split makes the
  // launch's duration depend on every workgroup of the grid being resident at once (one more per CU than fit = a second round)
  const unsigned long long step = SHARED ? (unsigned long long)chunk * WAVES : (unsigned long long)chunk;
  for (;;) {
    unsigned long long c0 = 0;
    if (SHARED) {
      __syncthreads();
      if (threadIdx.x == 0) {
        const unsigned long long v = atomicAdd(work, step);
        svar[2] = (int32_t)(uint32_t)v; svar[3] = (int32_t)(uint32_t)(v >> 32);
      }
      __syncthreads();
      c0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(svar[3]) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(svar[2]);
    } else {
      if (lane == 0) c0 = atomicAdd(work, step);
      c0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(c0 >> 32)) << 32) |
           (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)c0);
    }
    if (c0 >= n) break;
    const unsigned long long c1 = c0 + step < n ? c0 + step : n;
    unsigned long long c = c0, rend = c0;
    while (c < c1) {
      Candidate cd = cand[c];   // wave-uniform values are pinned to SGPRs: loop bounds and branches below become scalar
      cd.q = __builtin_amdgcn_readfirstlane(cd.q); cd.m = __builtin_amdgcn_readfirstlane(cd.m);
      if (SHARED || cd.q != curq) {   // candidates of one query are contiguous: its hashes are staged once per run
        curq = cd.q;
        const int32_t* qm = qmeta + (int64_t)cd.q * META_W;
        nA = __builtin_amdgcn_readfirstlane(qm[0]); len1 = __builtin_amdgcn_readfirstlane(qm[1]);
        qrow = qordered + (int64_t)cd.q * qord_stride;
        if (SHARED) {
          __syncthreads();               // the previous run's waves are done with the hashes and the table
          if (threadIdx.x == 0) svar[1] = 0;
          rend = c1;                     // the run ends at the chunk's first candidate of another query (every wave finds it for itself)
          for (unsigned long long t0 = c + 1; t0 < c1; t0 += 64) {
            const unsigned long long bal = __ballot(t0 + lane < c1 && cand[t0 + lane].q != curq);
            if (bal) { rend = t0 + (unsigned long long)__builtin_ctzll(bal); break; }
          }
        } else __builtin_amdgcn_wave_barrier();
        if (APOS) for (int i = tid; i < nA; i += NT) ap[i] = qrow[2 * i + 1];
        if (FILTER) {
          for (int i = tid; i < tabw; i += NT) bm[i] = 0u;
          __syncthreads();
        }
        if (!TABLE) {
          for (int i = tid; i < nA; i += NT) {
            const int h = qrow[2 * i];
            ah[i] = h;
            if (FILTER) { const uint32_t fb = oj_filter_bit((uint32_t)h, (uint32_t)ts); atomicOr(&bm[fb >> 5], 1u << (fb & 31u)); }
          }
        } else if (nA > 0) {
          bk = oj_buckets(__builtin_amdgcn_readfirstlane(qrow[0]), __builtin_amdgcn_readfirstlane(qrow[2 * (nA - 1)]), ts);
          for (int i = tid; i < nA; i += NT) {   // entry i starts every bucket after its predecessor's up to its own
            const int h = qrow[2 * i];
            ah[i] = h;
            const int b1 = oj_bucket_of(bk, h), b0 = i ? oj_bucket_of(bk, qrow[2 * (i - 1)]) + 1 : 0;
            for (int b = b0; b <= b1; b++) st[b] = (uint16_t)i;
            if (i == nA - 1) for (int b = b1 + 1; b <= ts; b++) st[b] = (uint16_t)nA;
          }
        }
        if (SHARED) __syncthreads(); else oj_lds_sync();
      }
      if (SHARED) {                      // this wave's next candidate of the run
        int ci = 0;
        if (lane == 0) ci = atomicAdd(&svar[1], 1);
        ci = __builtin_amdgcn_readfirstlane(ci);
        // (the loop below keeps `c` at the run's start and walks `cw`; it leaves with c = the run's end)
        for (unsigned long long cw = c + (unsigned long long)(uint32_t)ci; cw < rend;) {
          Candidate cx = cand[cw];
          cx.q = __builtin_amdgcn_readfirstlane(cx.q); cx.m = __builtin_amdgcn_readfirstlane(cx.m);
          one_candidate(cx);
          int cn = 0;
          if (lane == 0) cn = atomicAdd(&svar[1], 1);
          cw = c + (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(cn);
        }
        c = rend;
      } else {
        one_candidate(cd);
        c++;
      }
    }
  }
  if (mine && lane == 0) atomicAdd(compared, mine);
}

// The three shapes of the join kernel.  Its time is inversely proportional to the waves a CU holds (padding the LDS of the ALONE
// shape by 4 / 12 KB per wave: 4.97 -> 6.66 / 12.5 ms at C2, i.e. 18 -> 12 / 7 waves per CU), and what bounds those is LDS:
//   ALONE  every wave stages its own query (6 KB of hashes at S = 1536) and searches them by bisection: 8.7 KB per wave, 18 waves per CU.
//          For a candidate or fewer per query (a rank of a multi-GPU job: every query against an eighth of the reads).
//   PAIR   two waves share one staged query and take its candidates in turn: 5.6 KB per wave, 28 waves per CU (the VGPR limit) —
//          but a wave now waits for its partner at every run's end, which takes most of that back (C2, 4.5 candidates per query:
//          4.83 ms against 4.94 alone).  For a few candidates per query.
//   TEAM   four waves share the query and a bucket table over its hashes (two LDS round trips per lookup instead of eleven).
//          For tens of candidates per query (repeat-rich reads).
#ifndef MH_OJ_FILTER
#define MH_OJ_FILTER 1   // the shared shapes filter the other sketch's entries through a bitmap of the query's hashes (0: round 3's lookups of every entry)
#endif
enum { OJ_ALONE = 0, OJ_PAIR = 1, OJ_TEAM = 2 };
constexpr int OJ_SHAPE_WAVES[3] = {2, 2, OJ_WAVES};
int overlap_join_waves_per_block(int shape) { return OJ_SHAPE_WAVES[shape]; }
// LDS bytes of one workgroup: the hashes (and the table) once or per wave, the join scratch per wave
size_t overlap_join_lds_bytes(int S, int shape) {
  const size_t sp = (size_t)((S + 3) & ~3), w = (size_t)OJ_SHAPE_WAVES[shape];
  if (shape == OJ_ALONE) return w * (sp + OJ_LDS_EXTRA) * 4;
  const size_t aid = MH_OJ_FILTER ? (size_t)overlap_join_filter_bits(S, (int)w) / 32 : (shape == OJ_TEAM ? (size_t)((overlap_join_table_slots(S) / 2 + 4) & ~3) : 0);
  return (sp + (MH_OJ_KEEP ? sp : 0) + aid + 4 + w * OJ_LDS_EXTRA) * 4;
}
template <class F> static auto oj_dispatch(int shape, F f) {
#ifndef MH_OJ_WIDE_UNIT
  if (shape == OJ_TEAM) return f(overlap_join_kernel<true, OJ_SHAPE_WAVES[OJ_TEAM], !MH_OJ_FILTER, MH_OJ_FILTER != 0>);
  if (shape == OJ_PAIR) return f(overlap_join_kernel<true, OJ_SHAPE_WAVES[OJ_PAIR], false, MH_OJ_FILTER != 0>);
#endif
  (void)shape;   // (the wider passes run every wave alone)
  return f(overlap_join_kernel<false, OJ_SHAPE_WAVES[OJ_ALONE], false, false>);
}
// workgroups of the join kernel one CU holds at this sketch size
int overlap_join_blocks_per_cu(int S, int shape) {
  int n = 0;
  const hipError_t e = oj_dispatch(shape, [&](auto kern) { return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 64 * OJ_SHAPE_WAVES[shape], overlap_join_lds_bytes(S, shape)); });
  if (e != hipSuccess || n < 1) n = 1;
  return n;
}

void launch_overlap_join(hipStream_t st, int shape, int nblocks, int chunk, const Candidate* cand, const unsigned long long* cand_count,
                         unsigned long long cand_cap, const int32_t* ordered, int64_t ord_stride, const int32_t* meta, const int32_t* qordered,
                         int64_t qord_stride, const int32_t* qmeta, const SearchParams& sp, const double* score_table, DevRecord* recs,
                         unsigned long long* rec_count, unsigned long long rec_cap, unsigned long long* compared, Candidate* slow,
                         unsigned long long* slow_count, unsigned long long* work, const uint16_t* ph, const uint16_t* qph, const int32_t* pass_min) {
  const int ts = (MH_OJ_FILTER && shape != OJ_ALONE) ? overlap_join_filter_bits(sp.S, OJ_SHAPE_WAVES[shape]) : overlap_join_table_slots(sp.S);
  oj_dispatch(shape, [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64 * OJ_SHAPE_WAVES[shape]), overlap_join_lds_bytes(sp.S, shape), st, cand, cand_count, cand_cap, ordered,
                       ord_stride, meta, qordered, qord_stride, qmeta, sp, score_table, recs, rec_count, rec_cap, compared, slow, slow_count, chunk, work, ts,
                       ph, qph, pass_min);
    return 0;
  });
}

#ifndef MH_OJ_WIDE_UNIT
void oj_stats_dump() {
#ifdef MH_OJ_STATS
  unsigned long long h[20];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_oj_stats), sizeof h) != hipSuccess) return;
  const double np = (double)(h[0] + h[1] + h[2] + h[3] + h[4] + h[5] + 1);
  fprintf(stderr, "[oj stats] pairs: nj<3 %llu, no record in pass 1 %llu, in pass 2 %llu, <3 valid %llu, below threshold %llu, accepted %llu; mean nj %.2f, mean in-window joined of scored %.2f\n",
          h[0], h[1], h[2], h[3], h[4], h[5], (double)h[6] / np, (double)h[7] / (double)(h[4] + h[5] + 1));
  fprintf(stderr, "[oj stats] groups: per pair %.2f, pairs with groups %llu, with >= 3 %llu, entries in groups per pair %.2f, words reserved for group records per pair %.2f, removed by optimizeShifts per pair %.2f\n",
          (double)h[8] / np, h[9], h[10], (double)h[11] / np, (double)h[12] / np, (double)h[13] / np);
  fprintf(stderr, "[oj stats] handed to the per-lane kernel: more than %d joined k-mers %llu, more than %d groups %llu, a group of more than %d entries %llu, joined k-mers + group entries > %d: %llu\n",
          OJ_JCAP, h[14], OJ_GCAP, h[15], OJ_GLEN, h[16], OJ_JCAP, h[17]);
  memset(h, 0, sizeof h);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_oj_stats), h, sizeof h);
#endif
}

void launch_overlap(hipStream_t st, int nblocks, const Candidate* cand, const unsigned long long* cand_count, unsigned long long cand_cap,
                    const int32_t* ordered, int64_t ord_stride, const int32_t* meta, const int32_t* qordered, int64_t qord_stride,
                    const int32_t* qmeta, const SearchParams& sp, const double* score_table, int32_t* scratch, int64_t scratch_per_lane,
                    DevRecord* recs, unsigned long long* rec_count, unsigned long long rec_cap, unsigned long long* compared, int spread) {
  hipLaunchKernelGGL(overlap_kernel, dim3(nblocks), dim3(OVL_THREADS), 0, st, cand, cand_count, cand_cap, ordered, ord_stride, meta,
                     qordered, qord_stride, qmeta, sp, score_table, scratch, scratch_per_lane, recs, rec_count, rec_cap, compared, spread);
}
#endif   // MH_OJ_WIDE_UNIT

}  // namespace mhap
