// overlap_lane.hpp — second-stage overlap estimate for ONE candidate pair, written for one GPU lane.
// Exact integer logic of BottomOverlapSketch.getOverlapInfo (J/sketch/BottomOverlapSketch.java:592-630)
// incl. recordMatchingKmers (:397-516), MatchData (:64-298), computeKBottomSketchJaccard (:304-364)
// and Utils.quickSelect (J/utils/Utils.java:445-494).  The final log/exp is NOT done here: the lane
// returns (inter, k) and the caller looks the score up in a host-built table (one place for libm).
//
// __host__ __device__ so the same code is unit-tested on the CPU (tests/test_host_logic.py) and run
// per lane by overlap_kernel.  Scratch is strided so that lanes of a wavefront interleave.
#pragma once
#include "device_common.hpp"

namespace mhap {

struct LaneOverlap {
  int32_t empty;          // 1 -> OverlapInfo.EMPTY
  int32_t valid;          // rawScore
  int32_t a1, a2, b1, b2;
  int32_t inter, kk;      // bottom-k intersection / k
};

// Strided scratch view: element i of array `a` lives at base[(a*maxrec + i) * stride].
struct LaneScratch {
  int32_t* base;
  int64_t stride;
  int32_t maxrec;
  __host__ __device__ inline int32_t& at(int a, int i) const { return base[((int64_t)a * maxrec + i) * stride]; }
};

struct ShiftStats { int32_t med, absmax; };

// Read-only view of one ordered sketch: entry i = (hash, pos).  The algorithms below only ever move FORWARD through a
// view between two reset() calls, which lets the device view keep one 64-byte line per lane in LDS and prefetch the
// next line in registers (search_kernels.hip: CachedView).
struct PlainView {
  const int32_t* p;
  int n;
  __host__ __device__ inline void reset() {}
  __host__ __device__ inline void get(int i, int& h, int& pos) { h = p[2 * i]; pos = p[2 * i + 1]; }
};

// Utils.quickSelect (J/utils/Utils.java:445-494) on scratch array 2: its control flow verbatim — while it behaves.  What it returns is the
// k-th smallest value (0-based) of the array whatever the input (restated and checked against the sorted order statistic, duplicates
// included: tests/test_oracle_kat.py), but its partition moves every element EQUAL to the pivot to the right, so on an array of equal
// values — the shifts of two low-error reads that overlap: hundreds of joined k-mers, one shift — it narrows the range by one element per
// pass: 885 000 steps for 1 536 equal values (20 000 error-free reads: 24 s of second stage, round 5).  So the literal loop runs on a
// budget of steps (random input needs ~3 n, the worst seen on 20 000 duplicate-rich arrays 14 n) and, when that is spent, the same order
// statistic is selected bit by bit from the array as the loop has left it (a permutation of the input): 32 counting passes, O(n).
__host__ __device__ inline int32_t lane_select_bits(const LaneScratch& sc, int k, int length) {
  uint32_t prefix = 0u, mask = 0u;
  int kk = k;
  for (int b = 31; b >= 0; b--) {
    const uint32_t bit = 1u << b;
    int zeros = 0;
    for (int i = 0; i < length; i++) {
      const uint32_t u = (uint32_t)sc.at(2, i) ^ 0x80000000u;     // signed order as unsigned order
      zeros += ((u & mask) == prefix && !(u & bit)) ? 1 : 0;
    }
    if (kk >= zeros) { kk -= zeros; prefix |= bit; }
    mask |= bit;
  }
  return (int32_t)(prefix ^ 0x80000000u);
}
__host__ __device__ inline int32_t lane_quickselect(const LaneScratch& sc, int k, int length) {
  if (length <= k) return INT32_MAX;
  int from = 0, to = length - 1;
  long long budget = 24LL * length + 64;
  while (from < to) {
    int r = from, w = to;
    const int32_t mid = sc.at(2, (r + w) / 2);
    budget -= (long long)(w - r);
    if (budget < 0) return lane_select_bits(sc, k, length);
    while (r < w) {
      const int32_t ar = sc.at(2, r);
      if (ar >= mid) { const int32_t tmp = sc.at(2, w); sc.at(2, w) = ar; sc.at(2, r) = tmp; w--; }
      else r++;
    }
    if (sc.at(2, r) > mid) r--;
    if (k <= r) to = r; else from = r + 1;
  }
  return sc.at(2, k);
}

// MatchData.performUpdate (:191-215) over the current `count` records (shift = p2 - p1).
__host__ __device__ inline ShiftStats lane_stats(const LaneScratch& sc, int count, int len1, int len2, double max_shift) {
  ShiftStats st;
  if (count > 0) {
    for (int i = 0; i < count; i++) sc.at(2, i) = sc.at(1, i) - sc.at(0, i);
    st.med = lane_quickselect(sc, count / 2, count);
    const int left = 0 > -st.med ? 0 : -st.med;
    const int right = len1 < len2 - st.med ? len1 : len2 - st.med;
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

// recordMatchingKmers (:397-516): returns the new record count (records in scratch arrays 0/1).
template <class VA, class VB>
__host__ __device__ inline int lane_merge(const LaneScratch& sc, VA& A, VB& B, int len1, int len2, ShiftStats st) {
  const int nA = A.n, nB = B.n;
  const int med = st.med, absmax = st.absmax;
  const int t1 = -med - absmax, t2 = len2 - med + absmax, t3 = med - absmax, t4 = len1 + med + absmax;
  const int v1lo = 0 > t1 ? 0 : t1;            // valid1Lower :246-252
  const int v1hi = len1 < t2 ? len1 : t2;      // valid1Upper :254-260
  const int v2lo = 0 > t3 ? 0 : t3;            // valid2Lower :262-268
  const int v2hi = len2 < t4 ? len2 : t4;      // valid2Upper :270-276
  int i1 = 0, i2 = 0, count = 0;
  A.reset(); B.reset();
  while (i1 < nA && i2 < nB) {
    int h1, p1, h2, p2;
    A.get(i1, h1, p1);
    B.get(i2, h2, p2);
    if (h1 < h2 || p1 < v1lo || p1 >= v1hi) i1++;
    else if (h2 < h1 || p2 < v2lo || p2 >= v2hi) i2++;
    else {
      const int cur = p2 - p1;
      const int diff = cur - med;
      if (diff > absmax) i1++;
      else if (diff < -absmax) i2++;
      else {
        if (count < sc.maxrec) { sc.at(0, count) = p1; sc.at(1, count) = p2; }
        count++;
        // last entry of the run with the same hash and an in-window position (:460-496); positions of the run ends
        // are carried along so that the views are never read backwards
        int i1Last = i1, i1Try = i1 + 1, p1Last = p1;
        if (i1Try < nA) {
          int ht, pt;
          A.get(i1Try, ht, pt);
          while (ht == h1 && pt >= v1lo && pt < v1hi) {
            i1Last = i1Try; p1Last = pt; i1Try++;
            if (i1Try >= nA) break;
            A.get(i1Try, ht, pt);
          }
        }
        int i2Last = i2, i2Try = i2 + 1, p2Last = p2;
        if (i2Try < nB) {
          int ht, pt;
          B.get(i2Try, ht, pt);
          while (ht == h2 && pt >= v2lo && pt < v2hi) {
            i2Last = i2Try; p2Last = pt; i2Try++;
            if (i2Try >= nB) break;
            B.get(i2Try, ht, pt);
          }
        }
        if (i1 != i1Last || i2 != i2Last) {
          if (count < sc.maxrec) { sc.at(0, count) = p1Last; sc.at(1, count) = p2Last; }
          count++;
          i1 = i1Last + 1; i2 = i2Last + 1;
        } else { i1++; i2++; }
      }
    }
  }
  return count;
}

__host__ __device__ inline int iabs32(int v) { return v < 0 ? -v : v; }

template <class VA, class VB>
__host__ __device__ inline LaneOverlap lane_overlap(VA& A, int len1, VB& B, int len2, double max_shift, const LaneScratch& sc) {
  const int nA = A.n, nB = B.n;
  LaneOverlap r;
  r.empty = 1; r.valid = 0; r.a1 = r.a2 = r.b1 = r.b2 = 0; r.inter = 0; r.kk = 0;
  ShiftStats st = lane_stats(sc, 0, len1, len2, max_shift);
  int count = lane_merge(sc, A, B, len1, len2, st);                   // pass 1 (:600)
  if (count <= 0) return r;
  st = lane_stats(sc, count, len1, len2, max_shift);
  count = lane_merge(sc, A, B, len1, len2, st);                       // pass 2 (:606)
  if (count <= 0) return r;
  // optimizeShifts (:156-189)
  st = lane_stats(sc, count, len1, len2, max_shift);
  {
    int red = -1;
    const int med = st.med;
    for (int it = 0; it < count; it++) {
      const int p1 = sc.at(0, it), p2 = sc.at(1, it);
      if (red >= 0 && sc.at(0, red) == p1) {
        const int sr = sc.at(1, red) - sc.at(0, red);
        if (iabs32(sr - med) > iabs32((p2 - p1) - med)) { sc.at(0, red) = p1; sc.at(1, red) = p2; }
      } else { red++; sc.at(0, red) = p1; sc.at(1, red) = p2; }
    }
    count = red + 1;
  }
  if (count <= 0) return r;
  // computeEdges (:90-137)
  st = lane_stats(sc, count, len1, len2, max_shift);
  int le1 = INT32_MAX, le2 = INT32_MAX, re1 = INT32_MIN, re2 = INT32_MIN, valid = 0;
  for (int it = 0; it < count; it++) {
    const int p1 = sc.at(0, it), p2 = sc.at(1, it);
    if (iabs32((p2 - p1) - st.med) > st.absmax) continue;
    if (p1 < le1) le1 = p1;
    if (p2 < le2) le2 = p2;
    if (p1 > re1) re1 = p1;
    if (p2 > re2) re2 = p2;
    valid++;
  }
  if (valid < 3) return r;
  const double den = (double)(valid - 1);
  // int products wrap like Java's (:131-134)
  const int32_t na1 = (int32_t)((uint32_t)valid * (uint32_t)le1 - (uint32_t)re1);
  const int32_t na2 = (int32_t)((uint32_t)valid * (uint32_t)re1 - (uint32_t)le1);
  const int32_t nb1 = (int32_t)((uint32_t)valid * (uint32_t)le2 - (uint32_t)re2);
  const int32_t nb2 = (int32_t)((uint32_t)valid * (uint32_t)re2 - (uint32_t)le2);
  int a1 = (int)java_round((double)na1 / den); if (a1 < 0) a1 = 0;
  int a2 = (int)java_round((double)na2 / den); if (a2 > len1) a2 = len1;
  int b1 = (int)java_round((double)nb1 / den); if (b1 < 0) b1 = 0;
  int b2 = (int)java_round((double)nb2 / den); if (b2 > len2) b2 = len2;
  // computeKBottomSketchJaccard (:304-364) without materialising the filtered arrays
  int s1 = 0, s2 = 0;
  A.reset(); B.reset();
  for (int i = 0; i < nA; i++) { int h, pos; A.get(i, h, pos); s1 += (pos >= a1 && pos <= a2) ? 1 : 0; }
  for (int j = 0; j < nB; j++) { int h, pos; B.get(j, h, pos); s2 += (pos >= b1 && pos <= b2) ? 1 : 0; }
  const int kk = s1 < s2 ? s1 : s2;
  int inter = 0;
  if (kk > 0) {
    int i = 0, j = 0, uni = 0;
    A.reset(); B.reset();
    while (uni < kk) {
      int ha, pa, hb, pb;
      A.get(i, ha, pa);
      while (!(pa >= a1 && pa <= a2)) { i++; A.get(i, ha, pa); }
      B.get(j, hb, pb);
      while (!(pb >= b1 && pb <= b2)) { j++; B.get(j, hb, pb); }
      if (ha < hb) i++;
      else if (ha > hb) j++;
      else { inter++; i++; j++; }
      uni++;
    }
  }
  r.empty = 0; r.valid = valid; r.a1 = a1; r.a2 = a2; r.b1 = b1; r.b2 = b2; r.inter = inter; r.kk = kk;
  return r;
}

// Index of (inter, kk) in the host-built identity-score table: row kk holds inter = 0..kk.
__host__ __device__ inline int64_t score_index(int inter, int kk) { return (int64_t)kk * (kk + 1) / 2 + inter; }

}  // namespace mhap
