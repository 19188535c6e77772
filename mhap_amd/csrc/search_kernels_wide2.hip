// search_kernels_wide2.hip — and a third time with room for 1 536 (the whole of a default ordered sketch): see search_kernels_wide.hip.
// Eight times the first pass's cost per pair, a tenth of the per-lane kernel's: the pass for reads of a few per cent error and better.
#define MH_OJ_WIDE_UNIT 1
#define mhap mhap_wide2
#define MH_OJ_JCAP 1536
#include "search_kernels.hip"
#undef mhap

#include "kernels.hpp"   // (already seen above under the other name: #pragma once makes this a no-op; the wrappers below spell the types out)

namespace mhap {
size_t overlap_join_wide2_lds_bytes(int S, int shape) { return mhap_wide2::overlap_join_lds_bytes(S, shape); }
int overlap_join_wide2_blocks_per_cu(int S, int shape) { return mhap_wide2::overlap_join_blocks_per_cu(S, shape); }
int overlap_join_wide2_waves_per_block(int shape) { return mhap_wide2::overlap_join_waves_per_block(shape); }
int overlap_join_wide2_capacity() { return MH_OJ_JCAP; }
// (pointer types as void: the two namespaces' Candidate / DevRecord / SearchParams are the same structs under different names)
void launch_overlap_join_wide2(hipStream_t st, int shape, int nblocks, int chunk, const void* cand, const unsigned long long* cand_count,
                              unsigned long long cand_cap, const int32_t* ordered, int64_t ord_stride, const int32_t* meta, const int32_t* qordered,
                              int64_t qord_stride, const int32_t* qmeta, const void* sp, const double* score_table, void* recs,
                              unsigned long long* rec_count, unsigned long long rec_cap, unsigned long long* compared, void* slow,
                              unsigned long long* slow_count, unsigned long long* work, const uint16_t* ph, const uint16_t* qph, const int32_t* pass_min) {
  mhap_wide2::launch_overlap_join(st, shape, nblocks, chunk, (const mhap_wide2::Candidate*)cand, cand_count, cand_cap, ordered, ord_stride, meta, qordered,
                                 qord_stride, qmeta, *(const mhap_wide2::SearchParams*)sp, score_table, (mhap_wide2::DevRecord*)recs, rec_count, rec_cap, compared,
                                 (mhap_wide2::Candidate*)slow, slow_count, work, ph, qph, pass_min);
}
}  // namespace mhap
