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

constexpr int CAND_LD = CAND_KS + 4;  // padded LDS row (ints): 36 -> conflict-free ds_read_b128 across 16 rows

__device__ inline bool pair_passes(const SearchParams& sp, int64_t qid, int64_t mid, int qlen, int mlen) {
  if (sp.to_self && mid == qid) return false;                                               // MinHashSearch.java:200-201
  if (mlen < sp.min_store_length && qlen < sp.min_store_length) return false;               // :211-212
  if (sp.to_self && mid > qid && mlen >= sp.min_store_length && qlen >= sp.min_store_length) return false;  // :215-219
  if (sp.to_self && mlen < sp.min_store_length && qlen >= sp.min_store_length) return false;                // :222-225
  return true;
}

__global__ __launch_bounds__(256) void candidate_kernel(const int32_t* __restrict__ minhash, int64_t row_stride,
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
  for (int s0 = 0; s0 < H; s0 += CAND_KS) {
    // stage [128 x 32] slots of both tiles; 1024 int4 per tile -> 4 per thread
    const bool full = vec_ok && (s0 + CAND_KS <= H);
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
      const int idx = tid + rep * 256;
      const int row = idx >> 3, c4 = (idx & 7) * 4;
      int4 qv = make_int4(0, 0, 0, 0), mv = make_int4(1, 1, 1, 1);
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
      *(int4*)&qs[row * CAND_LD + c4] = qv;
      *(int4*)&ms[row * CAND_LD + c4] = mv;
    }
    __syncthreads();
#pragma unroll 2
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
          cnt[i][j] += (qv[i].x == mv[j].x) ? 1 : 0;
          cnt[i][j] += (qv[i].y == mv[j].y) ? 1 : 0;
          cnt[i][j] += (qv[i].z == mv[j].z) ? 1 : 0;
          cnt[i][j] += (qv[i].w == mv[j].w) ? 1 : 0;
        }
    }
    __syncthreads();
  }
  // emit
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int qe = qent[i * 16 + tq];
    if (qe < 0) continue;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (cnt[i][j] < sp.num_min_matches) continue;                                          // :204
      const int me = m0 + j * 16 + tm;
      if (me >= ne) continue;
      const int32_t* qm = qmeta + (int64_t)qe * META_W;
      const int32_t* mm = meta + (int64_t)me * META_W;
      if (qm[3] != 0 || mm[3] != 0) continue;   // placeholder entries (skipped strands) are not in the index
      if (!pair_passes(sp, qids[qe], ids[me], qm[2], mm[2])) continue;
      const unsigned long long slot = atomicAdd(cand_count, 1ULL);
      if (slot < cand_cap) { cand[slot].q = qe; cand[slot].m = me; }
    }
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
// Second stage.  Persistent lanes: lane g handles candidates g, g+G, ...  Scratch (3 int arrays of
// maxrec entries per lane) is interleaved across lanes so that lanes of a wave touch adjacent words.
// =============================================================================================
__global__ __launch_bounds__(OVL_THREADS) void overlap_kernel(const Candidate* __restrict__ cand, const unsigned long long* __restrict__ cand_count,
                                                              unsigned long long cand_cap, const int32_t* __restrict__ ordered,
                                                              int64_t ord_stride, const int32_t* __restrict__ meta,
                                                              const int32_t* __restrict__ qordered, int64_t qord_stride,
                                                              const int32_t* __restrict__ qmeta, SearchParams sp,
                                                              const double* __restrict__ score_table, int32_t* __restrict__ scratch,
                                                              int64_t scratch_per_lane, DevRecord* __restrict__ recs,
                                                              unsigned long long* __restrict__ rec_count, unsigned long long rec_cap,
                                                              unsigned long long* __restrict__ compared) {
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
    const int32_t* A = qordered + (int64_t)cd.q * qord_stride;
    const int32_t* B = ordered + (int64_t)cd.m * ord_stride;
    const LaneOverlap r = lane_overlap(A, qm[0], qm[1], B, mm[0], mm[1], sp.max_shift, sc);   // MinHashSearch.java:228
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
