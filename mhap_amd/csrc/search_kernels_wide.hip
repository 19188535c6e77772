// search_kernels_wide.hip — the second-stage join kernel compiled once more with room for 512 joined k-mers per pair.
//
// overlap_join_kernel (search_kernels.hip) keeps a pair's joined k-mers in OJ_JCAP words of LDS and two rounds of one entry per lane;
// a pair with more than 128 of them is handed to the per-lane kernel, which is exact and about two hundred times slower per pair.
// 128 is plenty for the reads MHAP was built for — at 15 % error two overlapping 10-kb reads join about 40 of their 1 536 bottom
// 12-mers — and not for better reads: a 12-mer survives in both reads with probability (1 - e)^24, so at 8 % error the average true
// overlap joins about a hundred, at 4 % about three hundred (round 5, 20 000 reads x 10 kb: 57 % / 83 % of the pairs handed over, second
// stage 54 / 149 ms where the join kernel alone would take 3).  Widening the kernel for everybody costs LDS and registers on the path
// BASELINE measures (four rounds per lane: 110 VGPRs in the PAIR shape; eight: 128 + scratch), so the SAME source is compiled a second
// time with MH_OJ_JCAP = 512 into a namespace of its own, and the search runs it as a second pass over the pairs the first pass hands over;
// only what it hands over in turn (more than 512 joined k-mers, the group caps) goes to the per-lane kernel.
//
// MH_OJ_WIDE_UNIT: only the join kernel (its ALONE shape) of search_kernels.hip is compiled again under the other namespace (round 6).
#define MH_OJ_WIDE_UNIT 1
#define mhap mhap_wide
#define MH_OJ_JCAP 512
#include "search_kernels.hip"
#undef mhap

#include "kernels.hpp"   // (already seen above under the other name: #pragma once makes this a no-op; the wrappers below spell the types out)

namespace mhap {
size_t overlap_join_wide_lds_bytes(int S, int shape) { return mhap_wide::overlap_join_lds_bytes(S, shape); }
int overlap_join_wide_blocks_per_cu(int S, int shape) { return mhap_wide::overlap_join_blocks_per_cu(S, shape); }
int overlap_join_wide_waves_per_block(int shape) { return mhap_wide::overlap_join_waves_per_block(shape); }
int overlap_join_wide_capacity() { return MH_OJ_JCAP; }
// (pointer types as void: the two namespaces' Candidate / DevRecord / SearchParams are the same structs under different names)
void launch_overlap_join_wide(hipStream_t st, int shape, int nblocks, int chunk, const void* cand, const unsigned long long* cand_count,
                              unsigned long long cand_cap, const int32_t* ordered, int64_t ord_stride, const int32_t* meta, const int32_t* qordered,
                              int64_t qord_stride, const int32_t* qmeta, const void* sp, const double* score_table, void* recs,
                              unsigned long long* rec_count, unsigned long long rec_cap, unsigned long long* compared, void* slow,
                              unsigned long long* slow_count, unsigned long long* work, const uint16_t* ph, const uint16_t* qph, const int32_t* pass_min) {
  mhap_wide::launch_overlap_join(st, shape, nblocks, chunk, (const mhap_wide::Candidate*)cand, cand_count, cand_cap, ordered, ord_stride, meta, qordered,
                                 qord_stride, qmeta, *(const mhap_wide::SearchParams*)sp, score_table, (mhap_wide::DevRecord*)recs, rec_count, rec_cap, compared,
                                 (mhap_wide::Candidate*)slow, slow_count, work, ph, qph, pass_min);
}
}  // namespace mhap
