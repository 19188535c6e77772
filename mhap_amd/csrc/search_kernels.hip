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
//   overlap_kernel   : BottomOverlapSketch.getOverlapInfo per candidate, one lane each (overlap_lane.hpp).
#include "kernels.hpp"
#include "overlap_lane.hpp"

namespace mhap {

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
// Inverted index on the GPU (the reference's own structure, J/impl/MinHashSearch.java:100-147,161-181):
// one open-addressing table per MinHash slot holding (value, entry+1) words; entries with equal values sit in
// one probe run.  A query does H probes and counts hits per stored entry in an LDS count table; the hit count
// of a pair equals the number of equal slots, so the candidate set is identical to the brute-force count.
// Work ~ N*H probes + hits instead of N*2N*H/2 compares.
// =============================================================================================
__device__ inline uint32_t inv_hash(uint32_t v) { return fmix32(v); }

__global__ __launch_bounds__(256) void index_build_kernel(const int32_t* __restrict__ minhash, int64_t row_stride, const int32_t* __restrict__ meta,
                                                          int ne, int H, unsigned long long* __restrict__ table, uint32_t cmask) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)ne * H;
  if (idx >= total) return;
  const int e = (int)(idx / H), s = (int)(idx % H);
  if (meta[(int64_t)e * META_W + 3] != 0) return;                       // skipped strands are not stored (addSequence never sees them)
  const uint32_t v = (uint32_t)minhash[(int64_t)e * row_stride + s];
  unsigned long long* T = table + (size_t)s * ((size_t)cmask + 1);
  const unsigned long long word = ((unsigned long long)v << 32) | (unsigned long long)(uint32_t)(e + 1);
  uint32_t pos = inv_hash(v) & cmask;
  for (;;) {
    const unsigned long long old = atomicCAS(&T[pos], 0ULL, word);
    if (old == 0ULL) break;
    pos = (pos + 1) & cmask;
  }
}

void launch_index_build(hipStream_t st, const int32_t* minhash, int64_t row_stride, const int32_t* meta, int ne, int H,
                        unsigned long long* table, uint32_t cmask) {
  const int64_t total = (int64_t)ne * H;
  if (total <= 0) return;
  hipLaunchKernelGGL(index_build_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, minhash, row_stride, meta, ne, H, table, cmask);
}

// One workgroup per query.  LDS: keys[CT] (entry+1), cnts[CT].  A query whose distinct-hit set outgrows the table is
// appended to `overflow` (the caller re-runs those through candidate_kernel).
__global__ __launch_bounds__(256) void index_query_kernel(const unsigned long long* __restrict__ table, uint32_t cmask,
                                                          const int32_t* __restrict__ qminhash, int64_t qrow_stride,
                                                          const int32_t* __restrict__ qlist, int nq, const int64_t* __restrict__ ids,
                                                          const int64_t* __restrict__ qids, const int32_t* __restrict__ meta,
                                                          const int32_t* __restrict__ qmeta, SearchParams sp,
                                                          Candidate* __restrict__ cand, unsigned long long* __restrict__ cand_count,
                                                          unsigned long long cand_cap, int32_t* __restrict__ overflow,
                                                          unsigned long long* __restrict__ overflow_count,
                                                          unsigned long long* __restrict__ elements) {
  __shared__ uint32_t keys[INV_CT];
  __shared__ uint32_t cnts[INV_CT];
  __shared__ uint32_t s_distinct, s_over;
  const int qi = blockIdx.x;
  if (qi >= nq) return;
  const int qe = qlist[qi];
  for (int j = threadIdx.x; j < INV_CT; j += 256) { keys[j] = 0; cnts[j] = 0; }
  if (threadIdx.x == 0) { s_distinct = 0; s_over = 0; }
  __syncthreads();
  const int32_t* qm = qmeta + (int64_t)qe * META_W;
  const int64_t qid = qids[qe];
  const int qlen = qm[2];
  unsigned long long mine = 0;
  for (int s = threadIdx.x; s < sp.H; s += 256) {
    const uint32_t v = (uint32_t)qminhash[(int64_t)qe * qrow_stride + s];
    const unsigned long long* T = table + (size_t)s * ((size_t)cmask + 1);
    uint32_t pos = inv_hash(v) & cmask;
    for (;;) {
      const unsigned long long w = T[pos];
      if (w == 0ULL) break;
      pos = (pos + 1) & cmask;
      if ((uint32_t)(w >> 32) != v) continue;
      mine++;                                                              // "table elements processed" (:173)
      const int me = (int)(uint32_t)w - 1;
      if (!pair_passes(sp, qid, ids[me], qlen, meta[(int64_t)me * META_W + 2])) continue;   // id/length rules do not depend on the count
      // count the hit
      uint32_t slot = inv_hash((uint32_t)me) & (INV_CT - 1);
      for (int tries = 0; tries < INV_CT; tries++) {
        uint32_t k = *(volatile uint32_t*)&keys[slot];
        if (k == 0) {
          if (*(volatile uint32_t*)&s_distinct >= (INV_CT * 3) / 4) { s_over = 1; break; }
          const uint32_t old = atomicCAS(&keys[slot], 0u, (uint32_t)me + 1u);
          if (old == 0) { atomicAdd(&s_distinct, 1u); k = (uint32_t)me + 1u; } else k = old;
        }
        if (k == (uint32_t)me + 1u) { atomicAdd(&cnts[slot], 1u); break; }
        slot = (slot + 1) & (INV_CT - 1);
      }
    }
  }
  if (mine) atomicAdd(elements, mine);
  __syncthreads();
  if (s_over) {
    if (threadIdx.x == 0) { const unsigned long long o = atomicAdd(overflow_count, 1ULL); overflow[o] = qe; }
    return;
  }
  // emit this query's candidates as ONE contiguous block (one global atomic per query): the second stage then finds
  // the lanes of a wave sharing the query's ordered-sketch row
  uint32_t mymask = 0;   // bit t set -> table slot threadIdx.x + 256*t is a candidate
  int mycount = 0;
#pragma unroll
  for (int t = 0; t < INV_CT / 256; t++) {
    const int j = threadIdx.x + 256 * t;
    if (keys[j] != 0 && (int)cnts[j] >= sp.num_min_matches) { mymask |= 1u << t; mycount++; }   // MinHashSearch.java:204
  }
  __syncthreads();            // s_distinct is dead from here on: reuse it as the block's emit counter
  if (threadIdx.x == 0) s_distinct = 0;
  __syncthreads();
  uint32_t local = 0;
  if (mycount) local = atomicAdd(&s_distinct, (uint32_t)mycount);
  __syncthreads();
  __shared__ unsigned long long s_base;
  if (threadIdx.x == 0) s_base = s_distinct ? atomicAdd(cand_count, (unsigned long long)s_distinct) : 0ULL;
  __syncthreads();
  unsigned long long slot = s_base + local;
#pragma unroll
  for (int t = 0; t < INV_CT / 256; t++) {
    if (mymask & (1u << t)) {
      if (slot < cand_cap) { cand[slot].q = qe; cand[slot].m = (int)keys[threadIdx.x + 256 * t] - 1; }
      slot++;
    }
  }
}

void launch_index_query(hipStream_t st, const unsigned long long* table, uint32_t cmask, const int32_t* qminhash, int64_t qrow_stride,
                        const int32_t* qlist, int nq, const int64_t* ids, const int64_t* qids, const int32_t* meta, const int32_t* qmeta,
                        const SearchParams& sp, Candidate* cand, unsigned long long* cand_count, unsigned long long cand_cap,
                        int32_t* overflow, unsigned long long* overflow_count, unsigned long long* elements) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(index_query_kernel, dim3((unsigned)nq), dim3(256), 0, st, table, cmask, qminhash, qrow_stride, qlist, nq, ids, qids,
                     meta, qmeta, sp, cand, cand_count, cand_cap, overflow, overflow_count, elements);
}

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
                                                              unsigned long long* __restrict__ compared) {
  __shared__ uint2 lines[2][8 * OVL_THREADS];
  unsigned long long n = *cand_count;
  if (n > cand_cap) n = cand_cap;
  const int64_t G = (int64_t)gridDim.x * OVL_THREADS;
  const int64_t g = (int64_t)blockIdx.x * OVL_THREADS + threadIdx.x;
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

void launch_overlap(hipStream_t st, int nblocks, const Candidate* cand, const unsigned long long* cand_count, unsigned long long cand_cap,
                    const int32_t* ordered, int64_t ord_stride, const int32_t* meta, const int32_t* qordered, int64_t qord_stride,
                    const int32_t* qmeta, const SearchParams& sp, const double* score_table, int32_t* scratch, int64_t scratch_per_lane,
                    DevRecord* recs, unsigned long long* rec_count, unsigned long long rec_cap, unsigned long long* compared) {
  hipLaunchKernelGGL(overlap_kernel, dim3(nblocks), dim3(OVL_THREADS), 0, st, cand, cand_count, cand_cap, ordered, ord_stride, meta,
                     qordered, qord_stride, qmeta, sp, score_table, scratch, scratch_per_lane, recs, rec_count, rec_cap, compared);
}

}  // namespace mhap
