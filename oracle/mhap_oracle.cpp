// ============================================================================
// mhap_oracle.cpp — CPU restatement of MHAP's MinHash overlap hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//   * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
//     load liboracle.so.  The product (libmhaphip.so, mhap_amd/) never links,
//     imports or falls back to anything in this directory.
//   * PARITY UNPINNED: the reference (marbl/MHAP, Java) ships no tests, no golden
//     vectors, and cannot be compiled or run in this image (no JVM/javac, and its
//     hash arithmetic lives in un-vendored jars: Guava 19.0 murmur3, fastutil
//     7.0.12).  What pins this file instead: (a) the published MurmurHash3
//     known-answer vectors / SMHasher verification constants
//     (tests/test_oracle_kat.py), (b) line-by-line restatement of the cited Java,
//     (c) self-consistency properties.  Any JVM-derived vector added later goes to
//     tests/golden/.
//
// Citation convention: J/ = /root/reference/src/main/java/edu/umd/marbl/mhap/
//
// Types follow Java: int32/int64 two's complement with wraparound, >>> logical,
// all hash comparisons SIGNED.  Sequences are handled as bytes; each byte is one
// Java `char` (UTF-16 code unit, high byte 0), which is exact for ASCII FASTA.
// ============================================================================
#include <algorithm>
#include <atomic>
#include <chrono>
#include <malloc.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// MurmurHash3 (Austin Appleby, public domain algorithm), as used by Guava 19.0
// Hashing.murmur3_128(seed) / murmur3_32(seed)  (J/sketch/HashUtils.java:215,239).
// Guava: h1=h2=seed; putUnencodedChars feeds each char little-endian (2 bytes);
// HashCode.asLong() = first 8 digest bytes little-endian = h1; asInt() = h.
// ---------------------------------------------------------------------------
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33; return k;
}
inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bU; h ^= h >> 13; h *= 0xc2b2ae35U; h ^= h >> 16; return h;
}

void murmur3_x64_128(const uint8_t* data, size_t len, uint32_t seed, uint64_t out[2]) {
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = seed, h2 = seed;
  const size_t nblocks = len / 16;
  for (size_t i = 0; i < nblocks; i++) {
    uint64_t k1, k2;
    memcpy(&k1, data + 16 * i, 8); memcpy(&k2, data + 16 * i + 8, 8);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t* tail = data + nblocks * 16;
  uint64_t k1 = 0, k2 = 0;
  switch (len & 15) {
    case 15: k2 ^= (uint64_t)tail[14] << 48; [[fallthrough]];
    case 14: k2 ^= (uint64_t)tail[13] << 40; [[fallthrough]];
    case 13: k2 ^= (uint64_t)tail[12] << 32; [[fallthrough]];
    case 12: k2 ^= (uint64_t)tail[11] << 24; [[fallthrough]];
    case 11: k2 ^= (uint64_t)tail[10] << 16; [[fallthrough]];
    case 10: k2 ^= (uint64_t)tail[9] << 8; [[fallthrough]];
    case 9:  k2 ^= (uint64_t)tail[8];
             k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; [[fallthrough]];
    case 8:  k1 ^= (uint64_t)tail[7] << 56; [[fallthrough]];
    case 7:  k1 ^= (uint64_t)tail[6] << 48; [[fallthrough]];
    case 6:  k1 ^= (uint64_t)tail[5] << 40; [[fallthrough]];
    case 5:  k1 ^= (uint64_t)tail[4] << 32; [[fallthrough]];
    case 4:  k1 ^= (uint64_t)tail[3] << 24; [[fallthrough]];
    case 3:  k1 ^= (uint64_t)tail[2] << 16; [[fallthrough]];
    case 2:  k1 ^= (uint64_t)tail[1] << 8; [[fallthrough]];
    case 1:  k1 ^= (uint64_t)tail[0];
             k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
  }
  h1 ^= len; h2 ^= len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2; h2 += h1;
  out[0] = h1; out[1] = h2;
}

uint32_t murmur3_x86_32(const uint8_t* data, size_t len, uint32_t seed) {
  const uint32_t c1 = 0xcc9e2d51U, c2 = 0x1b873593U;
  uint32_t h1 = seed;
  const size_t nblocks = len / 4;
  for (size_t i = 0; i < nblocks; i++) {
    uint32_t k1; memcpy(&k1, data + 4 * i, 4);
    k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2;
    h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5 + 0xe6546b64U;
  }
  const uint8_t* tail = data + nblocks * 4;
  uint32_t k1 = 0;
  switch (len & 3) {
    case 3: k1 ^= (uint32_t)tail[2] << 16; [[fallthrough]];
    case 2: k1 ^= (uint32_t)tail[1] << 8; [[fallthrough]];
    case 1: k1 ^= tail[0]; k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1;
  }
  h1 ^= (uint32_t)len;
  return fmix32(h1);
}

// Bytes of a k-mer as Guava's putUnencodedChars sees them: every char -> (lo, hi) LE.
inline void utf16le(const char* s, int n, uint8_t* out) {
  for (int i = 0; i < n; i++) { out[2 * i] = (uint8_t)s[i]; out[2 * i + 1] = 0; }
}

// J/utils/Utils.java:84-117,496-507  (rc + Translate table; unknown chars unchanged, upper-cased)
inline char rc_char(char c) {
  if (c >= 'a' && c <= 'z') c = (char)(c - 'a' + 'A');
  switch (c) {
    case 'A': return 'T'; case 'B': return 'V'; case 'C': return 'G'; case 'D': return 'H';
    case 'G': return 'C'; case 'H': return 'D'; case 'K': return 'M'; case 'M': return 'K';
    case 'N': return 'N'; case 'R': return 'Y'; case 'S': return 'S'; case 'T': return 'A';
    case 'V': return 'B'; case 'W': return 'W'; case 'Y': return 'R';
    default: return c;
  }
}
std::string rc(const std::string& s) {
  std::string r(s.size(), 'N');
  for (size_t i = 0; i < s.size(); i++) r[i] = rc_char(s[s.size() - 1 - i]);
  return r;
}

// J/sketch/HashUtils.java:237-258 computeSequenceHashesLong (seed 0)
std::vector<int64_t> kmer_hashes64(const char* seq, int L, int k, bool doRC) {
  std::vector<int64_t> out;
  int n = L - k + 1;
  if (n < 1) return out;
  out.resize(n);
  std::vector<uint8_t> buf(2 * k);
  std::string tmp, r;
  for (int i = 0; i < n; i++) {
    const char* p = seq + i;
    if (doRC) {  // canonical form: rc if it compares lower (String.compareTo), HashUtils.java:246-251
      tmp.assign(seq + i, k); r = rc(tmp);
      if (r.compare(tmp) < 0) p = r.data();
    }
    utf16le(p, k, buf.data());
    uint64_t h[2]; murmur3_x64_128(buf.data(), buf.size(), 0, h);
    out[i] = (int64_t)h[0];
  }
  return out;
}

// J/sketch/HashUtils.java:213-235 computeSequenceHashes (murmur3_32 seed 0)
std::vector<int32_t> kmer_hashes32(const char* seq, int L, int k, bool doRC) {
  std::vector<int32_t> out;
  int n = L - k + 1;
  if (n < 1) return out;
  out.resize(n);
  std::vector<uint8_t> buf(2 * k);
  std::string tmp, r;
  for (int i = 0; i < n; i++) {
    const char* p = seq + i;
    if (doRC) { tmp.assign(seq + i, k); r = rc(tmp); if (r.compare(tmp) < 0) p = r.data(); }
    utf16le(p, k, buf.data());
    out[i] = (int32_t)murmur3_x86_32(buf.data(), buf.size(), 0);
  }
  return out;
}

// Java Math.round(double) (JDK 8): closest long, ties toward +inf.
inline int64_t java_round(double x) {
  if (std::isnan(x)) return 0;
  double f = std::floor(x);
  double d = x - f;  // exact
  if (d >= 0.5) f += 1.0;
  if (f >= 9.2233720368547758e18) return INT64_MAX;
  if (f <= -9.2233720368547758e18) return INT64_MIN;
  return (int64_t)f;
}

// ---------------------------------------------------------------------------
// J/sketch/FrequencyCounts.java:63-319 (removeUnique==0 only; the Guava BloomFilter
// whitelist used by --supress-noise 1|2 is not restated).
// ---------------------------------------------------------------------------
// Guava 19.0 BloomFilter<Long> as FrequencyCounts builds it (J/sketch/FrequencyCounts.java:137):
//   BloomFilter.create((value, sink) -> sink.putLong(value), sizeBloom, 1.0e-5)  — strategy MURMUR128_MITZ_64:
//   numBits = (long)(-n ln p / (ln 2)^2), k = max(1, round(numBits / n * ln 2)), bit array of ceil(numBits / 64) longs,
//   (h1, h2) = murmur3_x64_128(seed 0) of the 8 little-endian bytes of the value; bit i = ((h1 + i h2) & Long.MAX_VALUE) % bitSize.
// (third-party algorithm, restated from its published source; the reference only calls create / put / mightContain)
struct Bloom {
  std::vector<uint64_t> words; uint64_t bitSize = 0; int k = 0;
  void create(int64_t n, double p) {
    if (n < 1) n = 1;
    int64_t m = (int64_t)(-(double)n * std::log(p) / (std::log(2.0) * std::log(2.0)));
    k = std::max(1, (int)java_round((double)m / (double)n * std::log(2.0)));
    words.assign((size_t)((m + 63) / 64), 0ULL); bitSize = (uint64_t)words.size() * 64;
  }
  static void hash(int64_t v, uint64_t& h1, uint64_t& h2) {
    uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)((uint64_t)v >> (8 * i));
    uint64_t out[2]; murmur3_x64_128(b, 8, 0, out); h1 = out[0]; h2 = out[1];
  }
  void put(int64_t v) {
    uint64_t h1, h2; hash(v, h1, h2); uint64_t c = h1;
    for (int i = 0; i < k; i++) { const uint64_t bit = (c & 0x7fffffffffffffffULL) % bitSize; words[bit >> 6] |= 1ULL << (bit & 63); c += h2; }
  }
  bool mightContain(int64_t v) const {
    uint64_t h1, h2; hash(v, h1, h2); uint64_t c = h1;
    for (int i = 0; i < k; i++) { const uint64_t bit = (c & 0x7fffffffffffffffULL) % bitSize; if (!((words[bit >> 6] >> (bit & 63)) & 1ULL)) return false; c += h2; }
    return true;
  }
};

struct Filter {
  std::unordered_map<int64_t, double> frac;
  double filterCutoff = 0, offset = 0, range = 3.0, maxValue = 0, minValue = 0;
  double minIdf = 0, maxIdf = 0;
  bool noTf = false;
  int removeUnique = 0;     // --supress-noise: 1 = k-mers absent from the file are dropped, 2 = they get idf 1 (:66-68)
  Bloom validMers;
  bool keepKmer(int64_t h) const { return removeUnique == 1 ? validMers.mightContain(h) : true; }   // :272-278
  double idf_of(double freq) const { return std::log(maxValue / freq - offset); }  // :250-254
  void finish() {
    minValue = filterCutoff;          // :224
    minIdf = idf_of(maxValue);        // :226
    maxIdf = idf_of(minValue);        // :227
  }
  bool isPopular(int64_t h) const { return frac.count(h) != 0; }                    // :267-270
  double scaledIdf(int64_t h) const {                                               // :290-311
    if (removeUnique == 2 && !validMers.mightContain(h)) return 1.0;                // :297-298
    auto it = frac.find(h);
    if (it == frac.end()) return range;
    double idf = idf_of(it->second);
    double scale = (maxIdf - minIdf) / (double)(range - 1.0);
    return 1.0 + (idf - minIdf) / scale;
  }
  double tfWeight(int w) const { return noTf ? 1.0 : (double)w; }                   // :313-319
};

// ---------------------------------------------------------------------------
// J/sketch/MinHashSketch.java:51-179 computeNgramMinHashesWeighted
// returns false on ZeroNGramsFoundException
// ---------------------------------------------------------------------------
bool minhash_sketch(const char* seq, int L, int k, int H, const Filter* filter, double repeatWeight,
                    int32_t* out /* max(1,H) */) {
  int n = L - k + 1;
  if (n < 1) return false;                                            // :56-57
  std::vector<int64_t> keys = kmer_hashes64(seq, L, k, false);        // :63
  // insertion-ordered dedup with counts (Long2ObjectLinkedOpenHashMap)  :66-81
  // (flat open-addressing table instead of a node-based map: same insertion order and counts, but no per-k-mer
  //  heap allocation, so the multithreaded CPU baseline is not throttled by the allocator)
  std::vector<int64_t> order; std::vector<int> cnt;
  order.reserve(keys.size()); cnt.reserve(keys.size());
  {
    size_t ts = 64; while (ts < keys.size() * 2) ts <<= 1;
    std::vector<int32_t> slot_of(ts, -1);
    for (int64_t key : keys) {
      if (filter && !filter->keepKmer(key)) continue;                 // :72-73
      size_t s = (size_t)fmix64((uint64_t)key) & (ts - 1);
      for (;;) {
        const int32_t e = slot_of[s];
        if (e < 0) { slot_of[s] = (int32_t)order.size(); order.push_back(key); cnt.push_back(1); break; }
        if (order[(size_t)e] == key) { cnt[(size_t)e]++; break; }
        s = (s + 1) & (ts - 1);
      }
    }
  }
  if (order.empty()) return false;                                    // :84-85
  int outn = std::max(1, H);
  for (int i = 0; i < outn; i++) out[i] = 0;                          // :88
  std::vector<int64_t> best(H, INT64_MAX);                            // :89-90
  int numberValid = 0;
  for (size_t e = 0; e < order.size(); e++) {
    int64_t key = order[e];
    int weight = cnt[e];
    if (repeatWeight < 0.0) {                                         // :101-107
      weight = 1;
      if (filter && filter->isPopular(key)) weight = 0;
    } else if (filter) {                                              // :109-124
      if (repeatWeight >= 0.0 && repeatWeight < 1.0) {
        double tf = filter->tfWeight(weight);
        double idf = filter->scaledIdf(key);
        weight = (int)java_round(tf * idf);
        if (weight < 1) weight = 1;
      }
    }
    if (weight <= 0) continue;                                        // :127-128
    numberValid++;
    uint64_t x = (uint64_t)key;                                       // :134
    for (int word = 0; word < H; word++) {
      for (int c = 0; c < weight; c++) {
        x ^= x << 21; x ^= x >> 35; x ^= x << 4;                      // :140-142
        if ((int64_t)x < best[word]) {                                // :144 signed, strict
          best[word] = (int64_t)x;
          out[word] = (word % 2 == 0) ? (int32_t)(uint32_t)key : (int32_t)(uint32_t)((uint64_t)key >> 32);
        }
      }
    }
  }
  return numberValid > 0;                                             // :156-157
}

// ---------------------------------------------------------------------------
// J/sketch/BottomOverlapSketch.java:525-559 — ordered bottom-k sketch.
// IntArrays.radixSortIndirect(perm, hashes, stable=true): ascending signed, ties by index.
// ---------------------------------------------------------------------------
struct Ordered { int seqLength = 0; std::vector<int32_t> hp; /* (hash,pos) pairs */ int size() const { return (int)hp.size() / 2; } };

bool ordered_sketch(const char* seq, int L, int k2, int S, Ordered& o) {
  o.seqLength = L - k2 + 1;                                           // :528
  o.hp.clear();
  if (o.seqLength <= 0) return false;                                 // :530-531
  std::vector<int32_t> h = kmer_hashes32(seq, L, k2, false);          // :534
  std::vector<int32_t> perm(h.size());
  for (size_t i = 0; i < h.size(); i++) perm[i] = (int32_t)i;
  std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return h[a] < h[b]; });  // :543
  int kk = std::min(S, (int)h.size());                                // :548
  if (kk < 0) kk = 0;
  o.hp.resize((size_t)kk * 2);
  for (int i = 0; i < kk; i++) { o.hp[2 * i] = h[perm[i]]; o.hp[2 * i + 1] = perm[i]; }           // :553-558
  return true;
}

// J/utils/Utils.java:445-494 quickSelect — verbatim (value pivot, in place)
int quick_select(int32_t* array, int k, int length) {
  if (array == nullptr || length <= k) return INT32_MAX;
  int from = 0, to = length - 1;
  while (from < to) {
    int r = from, w = to;
    int mid = array[(r + w) / 2];
    while (r < w) {
      if (array[r] >= mid) { int tmp = array[w]; array[w] = array[r]; array[r] = tmp; w--; }
      else r++;
    }
    if (array[r] > mid) r--;
    if (k <= r) to = r; else from = r + 1;
  }
  return array[k];
}

struct OverlapInfo { double score = 0, rawScore = 0; int a1 = 0, a2 = 0, b1 = 0, b2 = 0; int inter = 0, kk = 0; int empty = 1; };

// J/sketch/BottomOverlapSketch.java:64-298 MatchData
struct MatchData {
  int absMax = 0, count = 0; double maxShiftPercent; int med = 0; bool need = true;
  std::vector<int32_t> p1, p2, sh; int len1, len2;
  MatchData(int l1, int l2, double ms) : maxShiftPercent(ms), len1(l1), len2(l2) { reset(); }
  void reset() { count = 0; need = true; }                                           // :236-240
  void update() {                                                                    // :191-215
    if (need) {
      if (count > 0) {
        std::vector<int32_t> cp(sh.begin(), sh.begin() + count);
        med = quick_select(cp.data(), count / 2, count);
        int left = std::max(0, -med);
        int right = std::min(len1, len2 - med);
        int ov = std::max(10, right - left);
        absMax = std::min(std::max(len1, len2), (int)((double)ov * maxShiftPercent));
      } else { med = 0; absMax = std::max(len1, len2) + 1; }
    }
    need = false;
  }
  int getMed() { update(); return med; }
  int getAbs() { update(); return absMax; }
  void record(int a, int b, int s) {                                                 // :217-234
    if ((int)sh.size() <= count) { sh.resize(count + 1); p1.resize(count + 1); p2.resize(count + 1); }
    sh[count] = s; p1[count] = a; p2[count] = b; count++; need = true;
  }
  int v1lo() { return std::max(0, -getMed() - getAbs()); }                           // :246-252
  int v1hi() { return std::min(len1, len2 - getMed() + getAbs()); }                  // :254-260
  int v2lo() { return std::max(0, getMed() - getAbs()); }                            // :262-268
  int v2hi() { return std::min(len2, len1 + getMed() + getAbs()); }                  // :270-276
  void optimizeShifts() {                                                            // :156-189
    if (count <= 0) return;
    int red = -1; int m = getMed();
    for (int it = 0; it < count; it++) {
      if (red >= 0 && p1[red] == p1[it]) {
        if (std::abs(sh[red] - m) > std::abs(sh[it] - m)) { p1[red] = p1[it]; p2[red] = p2[it]; sh[red] = sh[it]; }
      } else { red++; p1[red] = p1[it]; p2[red] = p2[it]; sh[red] = sh[it]; }
    }
    count = red + 1; need = true;
  }
};

// J/sketch/BottomOverlapSketch.java:397-516 recordMatchingKmers
void record_matching(MatchData& md, const int32_t* A, int nA, const int32_t* B, int nB) {
  int med = md.getMed(), absMax = md.getAbs();
  int v1lo = md.v1lo(), v2lo = md.v2lo(), v1hi = md.v1hi(), v2hi = md.v2hi();
  int i1 = 0, i2 = 0;
  md.reset();
  while (true) {
    if (i1 >= nA) break;
    if (i2 >= nB) break;
    int h1 = A[2 * i1], p1 = A[2 * i1 + 1], h2 = B[2 * i2], p2 = B[2 * i2 + 1];
    if (h1 < h2 || p1 < v1lo || p1 >= v1hi) i1++;
    else if (h2 < h1 || p2 < v2lo || p2 >= v2hi) i2++;
    else {
      int cur = p2 - p1; int diff = cur - med;
      if (diff > absMax) i1++;
      else if (diff < -absMax) i2++;
      else {
        md.record(p1, p2, cur);
        int i1Last = i1, i1Try = i1 + 1;
        if (i1Try < nA) {
          int ht = A[2 * i1Try], pt = A[2 * i1Try + 1];
          while (ht == h1 && pt >= v1lo && pt < v1hi) {
            i1Last = i1Try; i1Try++;
            if (i1Try >= nA) break;
            ht = A[2 * i1Try]; pt = A[2 * i1Try + 1];
          }
        }
        int i2Last = i2, i2Try = i2 + 1;
        if (i2Try < nB) {
          int ht = B[2 * i2Try], pt = B[2 * i2Try + 1];
          while (ht == h2 && pt >= v2lo && pt < v2hi) {
            i2Last = i2Try; i2Try++;
            if (i2Try >= nB) break;
            ht = B[2 * i2Try]; pt = B[2 * i2Try + 1];
          }
        }
        if (i1 != i1Last || i2 != i2Last) {
          int p1n = A[2 * i1Last + 1], p2n = B[2 * i2Last + 1];
          md.record(p1n, p2n, p2n - p1n);
          i1 = i1Last + 1; i2 = i2Last + 1;
        } else { i1++; i2++; }
      }
    }
  }
}

// J/sketch/BottomOverlapSketch.java:304-364 computeKBottomSketchJaccard -> (inter,k)
void kbottom(const int32_t* A, int nA, const int32_t* B, int nB, int a1, int a2, int b1, int b2, int& inter, int& kk) {
  std::vector<int32_t> x, y; x.reserve(nA); y.reserve(nB);
  for (int i = 0; i < nA; i++) { int pos = A[2 * i + 1]; if (pos >= a1 && pos <= a2) x.push_back(A[2 * i]); }
  for (int j = 0; j < nB; j++) { int pos = B[2 * j + 1]; if (pos >= b1 && pos <= b2) y.push_back(B[2 * j]); }
  kk = (int)std::min(x.size(), y.size()); inter = 0;
  if (kk == 0) return;
  int i = 0, j = 0, uni = 0;
  while (uni < kk) {
    if (x[i] < y[j]) i++; else if (x[i] > y[j]) j++; else { inter++; i++; j++; }
    uni++;
  }
}

// J/sketch/BottomOverlapSketch.java:391-395 jaccardToIdentity
inline double jaccard_to_identity(double score, int kmerSize) {
  double d = -1.0 / (double)kmerSize * std::log(2.0 * score / (1.0 + score));
  return std::exp(-d);
}

// J/sketch/BottomOverlapSketch.java:592-630 getOverlapInfo
OverlapInfo overlap_info(const int32_t* A, int nA, int lenA, const int32_t* B, int nB, int lenB, int k2, double maxShift) {
  OverlapInfo r;  // EMPTY (J/impl/OverlapInfo.java:40)
  MatchData md(lenA, lenB, maxShift);
  record_matching(md, A, nA, B, nB);
  if (md.count <= 0) return r;
  record_matching(md, A, nA, B, nB);
  if (md.count <= 0) return r;
  md.optimizeShifts();
  if (md.count <= 0) return r;
  // computeEdges :90-137
  int le1 = INT32_MAX, le2 = INT32_MAX, re1 = INT32_MIN, re2 = INT32_MIN, valid = 0;
  int med = md.getMed(), absMax = md.getAbs();
  for (int it = 0; it < md.count; it++) {
    int p1 = md.p1[it], p2 = md.p2[it];
    if (std::abs(md.sh[it] - med) > absMax) continue;
    if (p1 < le1) le1 = p1; if (p2 < le2) le2 = p2; if (p1 > re1) re1 = p1; if (p2 > re2) re2 = p2;
    valid++;
  }
  if (valid < 3) return r;
  auto wrapmul_sub = [](int n, int a, int b) -> int32_t {  // int arithmetic with Java wraparound
    return (int32_t)((uint32_t)n * (uint32_t)a - (uint32_t)b);
  };
  double den = (double)(valid - 1);
  int a1 = std::max(0, (int)java_round((double)wrapmul_sub(valid, le1, re1) / den));
  int a2 = std::min(lenA, (int)java_round((double)wrapmul_sub(valid, re1, le1) / den));
  int b1 = std::max(0, (int)java_round((double)wrapmul_sub(valid, le2, re2) / den));
  int b2 = std::min(lenB, (int)java_round((double)wrapmul_sub(valid, re2, le2) / den));
  int inter, kk; kbottom(A, nA, B, nB, a1, a2, b1, b2, inter, kk);
  double j = (kk == 0) ? 0.0 : ((double)inter) / (double)kk;
  r.score = jaccard_to_identity(j, k2);
  r.rawScore = (double)valid; r.a1 = a1; r.a2 = a2; r.b1 = b1; r.b2 = b2; r.inter = inter; r.kk = kk; r.empty = 0;
  return r;
}

// ---------------------------------------------------------------------------
// Java String.format("%.6f", x) for finite x (J/impl/MatchResult.java:100):
// JDK 8 Formatter takes the shortest-repr decimal digits (FloatingDecimal) and
// rounds them HALF_UP at the requested precision.
// ---------------------------------------------------------------------------
std::string java_fmt6(double v) {
  if (std::isnan(v)) return "NaN";
  if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity";
  bool neg = std::signbit(v); double a = std::fabs(v);
  char buf[64]; int prec = 1;
  for (; prec <= 17; prec++) { snprintf(buf, sizeof buf, "%.*e", prec - 1, a); if (strtod(buf, nullptr) == a) break; }
  // buf = d.ddddde[+-]XX
  std::string digits; int exp10 = 0;
  { const char* e = strchr(buf, 'e'); exp10 = atoi(e + 1); for (const char* p = buf; p < e; p++) if (*p >= '0' && *p <= '9') digits.push_back(*p); }
  // value = 0.d1d2d3... * 10^(exp10+1)
  int pointPos = exp10 + 1;  // digits before the decimal point (may be <= 0)
  std::string ip, fp;
  if (pointPos <= 0) { ip = "0"; fp = std::string((size_t)(-pointPos), '0') + digits; }
  else if ((size_t)pointPos >= digits.size()) { ip = digits + std::string((size_t)pointPos - digits.size(), '0'); fp = ""; }
  else { ip = digits.substr(0, pointPos); fp = digits.substr(pointPos); }
  bool up = false;
  if (fp.size() > 6) { up = fp[6] >= '5'; fp.resize(6); }
  while (fp.size() < 6) fp.push_back('0');
  if (up) {
    std::string all = ip + fp; int i = (int)all.size() - 1;
    while (i >= 0) { if (all[i] == '9') { all[i] = '0'; i--; } else { all[i]++; break; } }
    if (i < 0) all = "1" + all;
    ip = all.substr(0, all.size() - 6); fp = all.substr(all.size() - 6);
  }
  return (neg ? "-" : "") + ip + "." + fp;
}

struct Params {
  int k = 16, H = 512, k2 = 12, S = 1536, numMinMatches = 3, minStoreLength = 0, minOlapLength = 116;
  double threshold = 0.78, maxShift = 0.2, repeatWeight = 0.9;
};

struct Entry { int64_t id; int fwd; int L; std::vector<int32_t> mh; Ordered ord; };

}  // namespace

// ===========================================================================
// C API (ctypes-callable).  Test infrastructure only.
// ===========================================================================
extern "C" {

struct orc_record { int64_t from_id, to_id; double score, raw; int32_t a1, a2, alen, b1, b2, blen; int32_t to_rc, pad; };

void orc_murmur3_x64_128(const uint8_t* d, int64_t len, uint32_t seed, uint64_t* out2) { murmur3_x64_128(d, (size_t)len, seed, out2); }
uint32_t orc_murmur3_x86_32(const uint8_t* d, int64_t len, uint32_t seed) { return murmur3_x86_32(d, (size_t)len, seed); }

int orc_kmer_hashes64(const char* seq, int L, int k, int doRC, int64_t* out) {
  auto v = kmer_hashes64(seq, L, k, doRC != 0); if (!v.empty()) memcpy(out, v.data(), v.size() * 8); return (int)v.size();
}
int orc_kmer_hashes32(const char* seq, int L, int k, int doRC, int32_t* out) {
  auto v = kmer_hashes32(seq, L, k, doRC != 0); if (!v.empty()) memcpy(out, v.data(), v.size() * 4); return (int)v.size();
}
void orc_rc(const char* seq, int L, char* out) { std::string r = rc(std::string(seq, L)); memcpy(out, r.data(), L); }

// filter handle
void* orc_filter_create(const int64_t* hashes, const double* fractions, int64_t n, double filterCutoff, double offset,
                        double range, int noTf) {
  Filter* f = new Filter(); f->filterCutoff = filterCutoff; f->offset = offset; f->range = range; f->noTf = noTf != 0;
  double mx = -std::numeric_limits<double>::infinity();
  for (int64_t i = 0; i < n; i++) if (fractions[i] >= filterCutoff) { mx = std::max(mx, fractions[i]); f->frac[hashes[i]] = fractions[i]; }
  f->maxValue = mx; f->finish(); return f;
}
// FrequencyCounts with --supress-noise (removeUnique 1|2): every line's k-mer hash goes into the Bloom filter (:192), sized by
// the file's first number (sizeBloom, :102-137)
void* orc_filter_create2(const int64_t* hashes, const double* fractions, int64_t n, double filterCutoff, double offset,
                         double range, int noTf, int removeUnique, int64_t sizeBloom) {
  Filter* f = (Filter*)orc_filter_create(hashes, fractions, n, filterCutoff, offset, range, noTf);
  f->removeUnique = removeUnique;
  if (removeUnique > 0) {
    f->validMers.create(sizeBloom == 0 ? 1 : sizeBloom, 1.0e-5);
    for (int64_t i = 0; i < n; i++) f->validMers.put(hashes[i]);
  }
  return f;
}
void orc_bloom_params(int64_t n, double p, int64_t* bits, int32_t* k) { Bloom b; b.create(n, p); *bits = (int64_t)b.bitSize; *k = b.k; }
int orc_filter_might_contain(void* f, int64_t h) { return ((Filter*)f)->validMers.mightContain(h) ? 1 : 0; }
void orc_filter_destroy(void* f) { delete (Filter*)f; }
double orc_filter_scaled_idf(void* f, int64_t h) { return ((Filter*)f)->scaledIdf(h); }

// 0 = ok, 1 = ZeroNGramsFoundException
int orc_minhash(const char* seq, int L, int k, int H, void* filter, double repeatWeight, int32_t* out) {
  return minhash_sketch(seq, L, k, H, (const Filter*)filter, repeatWeight, out) ? 0 : 1;
}
// out_pairs must hold 2*min(S, L-k2+1) ints; returns 0 ok / 1 ZeroNGrams
int orc_ordered(const char* seq, int L, int k2, int S, int32_t* out_pairs, int32_t* out_size, int32_t* out_seqlen) {
  Ordered o; bool ok = ordered_sketch(seq, L, k2, S, o);
  *out_seqlen = o.seqLength; *out_size = o.size();
  if (!ok) return 1;
  if (!o.hp.empty()) memcpy(out_pairs, o.hp.data(), o.hp.size() * 4);
  return 0;
}
int orc_quickselect(int32_t* arr, int k, int len) { return quick_select(arr, k, len); }
int64_t orc_java_round(double x) { return java_round(x); }
void orc_java_fmt6(double v, char* out, int cap) { std::string s = java_fmt6(v); snprintf(out, cap, "%s", s.c_str()); }
double orc_jaccard_to_identity(double j, int k2) { return jaccard_to_identity(j, k2); }

// out: score, raw, a1,a2,b1,b2, inter, k, empty
void orc_overlap(const int32_t* A, int nA, int lenA, const int32_t* B, int nB, int lenB, int k2, double maxShift,
                 double* out_score, double* out_raw, int32_t* out6 /* a1 a2 b1 b2 inter k */, int32_t* out_empty) {
  OverlapInfo r = overlap_info(A, nA, lenA, B, nB, lenB, k2, maxShift);
  *out_score = r.score; *out_raw = r.rawScore; out6[0] = r.a1; out6[1] = r.a2; out6[2] = r.b1; out6[3] = r.b2; out6[4] = r.inter; out6[5] = r.kk;
  *out_empty = r.empty;
}

// MatchResult ctor + toString (J/impl/MatchResult.java:46-65,98-113) for numeric headers
void orc_format_record(const orc_record* r, char* out, int cap) {
  double score = r->score > 1.0 ? 1.0 : r->score;
  snprintf(out, cap, "%lld %lld %s %s %d %d %d %d %d %d %d %d", (long long)r->from_id, (long long)r->to_id,
           java_fmt6(1.0 - score).c_str(), java_fmt6(r->raw).c_str(), 0, r->a1, r->a2, r->alen, r->to_rc, r->b1, r->b2, r->blen);
}

// ---------------------------------------------------------------------------
// Whole self-overlap pipeline (J/main/MhapMain.java:377-476 -s mode):
//   reads (already upper-cased) given as concatenated bytes + offsets; read i has
//   id = ids[i] (1-based FASTA order incl. skipped-short reads, FastaData.java:180-181).
//   J/impl/SequenceSketchStreamer.java:123-156: skip L < minOlapLength; sketch fwd and rc.
//   J/impl/MinHashSearch.java:100-251: inverted index (restated as per-slot sorted
//   postings; hit count per stored entry is identical), filters, second stage.
// Returns number of records; writes up to cap into out. Records are in (query order,
// entry order) — the reference's order is nondeterministic.
// timings[0]=sketch seconds, timings[1]=search seconds. stats[0]=strands, [1]=candidates
// fully compared, [2]=table elements processed.
// ---------------------------------------------------------------------------
int64_t orc_run_self(const char* bases, const int64_t* offsets, const int32_t* lengths, const int64_t* ids, int64_t nreads,
                     int k, int H, int k2, int S, int numMinMatches, int minStoreLength, int minOlapLength,
                     double threshold, double maxShift, double repeatWeight, void* filter, int nthreads,
                     orc_record* out, int64_t cap, double* timings, int64_t* stats,
                     int32_t* out_minhash /* optional: 2*nreads*max(1,H), kept strands only in order */,
                     int32_t* out_status /* optional per strand 2*nreads: 0 ok,1 zero-ngrams,2 skipped short */) {
  // keep the per-strand work buffers (80-130 KB each) in the malloc arenas: with the default 128 KB mmap threshold every
  // strand pays mmap/munmap + page faults, which serialises a many-core baseline in the kernel
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  auto t0 = std::chrono::steady_clock::now();
  if (nthreads < 1) nthreads = 1;
  std::vector<Entry> all((size_t)nreads * 2);
  std::vector<int> status((size_t)nreads * 2, 0);
  std::atomic<int64_t> next(0);
  auto sk = [&]() {
    for (;;) {
      int64_t i = next.fetch_add(1); if (i >= nreads) break;
      int L = lengths[i]; const char* s = bases + offsets[i];
      Entry& f = all[2 * i]; Entry& r = all[2 * i + 1];
      f.id = r.id = ids[i]; f.fwd = 1; r.fwd = 0; f.L = r.L = L;
      if (L < minOlapLength) { status[2 * i] = status[2 * i + 1] = 2; continue; }
      f.mh.assign(std::max(1, H), 0);
      bool ok = minhash_sketch(s, L, k, H, (const Filter*)filter, repeatWeight, f.mh.data()) && ordered_sketch(s, L, k2, S, f.ord);
      if (!ok) { status[2 * i] = status[2 * i + 1] = 1; continue; }   // SequenceSketchStreamer.java:235-238 (read skipped)
      std::string rs = rc(std::string(s, L));
      r.mh.assign(std::max(1, H), 0);
      bool ok2 = minhash_sketch(rs.data(), L, k, H, (const Filter*)filter, repeatWeight, r.mh.data()) && ordered_sketch(rs.data(), L, k2, S, r.ord);
      if (!ok2) status[2 * i + 1] = 1;
    }
  };
  { std::vector<std::thread> th; for (int t = 0; t < nthreads; t++) th.emplace_back(sk); for (auto& t : th) t.join(); }
  std::vector<int> kept;  // entry indices stored in the index
  for (size_t e = 0; e < all.size(); e++) if (status[e] == 0) kept.push_back((int)e);
  if (out_status) for (size_t e = 0; e < all.size(); e++) out_status[e] = status[e];
  if (out_minhash) { size_t w = std::max(1, H); for (size_t e = 0; e < all.size(); e++) if (status[e] == 0) memcpy(out_minhash + e * w, all[e].mh.data(), w * 4); }
  auto t1 = std::chrono::steady_clock::now();

  // inverted index: per slot, postings sorted by value (MinHashSearch.java:100-147)
  std::vector<std::vector<std::pair<int32_t, int>>> post(H);
  {
    std::atomic<int> ns(0);
    auto build = [&]() { for (;;) { int s = ns.fetch_add(1); if (s >= H) break; auto& p = post[s]; p.reserve(kept.size());
        for (int e : kept) p.emplace_back(all[e].mh[s], e); std::sort(p.begin(), p.end()); } };
    std::vector<std::thread> th; for (int t = 0; t < nthreads; t++) th.emplace_back(build); for (auto& t : th) t.join();
  }
  std::vector<int> fwdq; for (int e : kept) if (all[e].fwd) fwdq.push_back(e);
  std::vector<std::vector<orc_record>> per((size_t)nthreads);
  std::atomic<int64_t> qn(0), ncmp(0), nelem(0);
  auto search = [&](int tid) {
    std::unordered_map<int, int> hits;
    for (;;) {
      int64_t qi = qn.fetch_add(1); if (qi >= (int64_t)fwdq.size()) break;
      const Entry& q = all[fwdq[qi]];
      hits.clear(); int64_t proc = 0;
      for (int s = 0; s < H; s++) {                                   // MinHashSearch.java:166-181
        auto& p = post[s]; int32_t v = q.mh[s];
        auto lo = std::lower_bound(p.begin(), p.end(), std::make_pair(v, INT32_MIN));
        for (; lo != p.end() && lo->first == v; ++lo) { hits[lo->second]++; proc++; }
      }
      nelem += proc;
      std::vector<std::pair<int, int>> hv(hits.begin(), hits.end()); std::sort(hv.begin(), hv.end());
      for (auto& kv : hv) {
        const Entry& m = all[kv.first];
        if (m.id == q.id) continue;                                   // :200-201 (toSelf)
        if (kv.second < numMinMatches) continue;                      // :204
        if (m.L < minStoreLength && q.L < minStoreLength) continue;   // :211-212
        if (m.id > q.id && m.L >= minStoreLength && q.L >= minStoreLength) continue;  // :215-219
        if (m.L < minStoreLength && q.L >= minStoreLength) continue;  // :222-225
        OverlapInfo r = overlap_info(q.ord.hp.data(), q.ord.size(), q.ord.seqLength, m.ord.hp.data(), m.ord.size(), m.ord.seqLength, k2, maxShift);  // :228
        ncmp++;
        if (r.score >= threshold) {                                   // :229
          orc_record rec{}; rec.from_id = q.id; rec.to_id = m.id; rec.score = r.score; rec.raw = r.rawScore;
          rec.a1 = r.a1; rec.a2 = r.a2; rec.alen = q.L; rec.blen = m.L; rec.to_rc = m.fwd ? 0 : 1;
          if (m.fwd) { rec.b1 = r.b1; rec.b2 = r.b2; } else { rec.b1 = m.L - r.b2 - 1; rec.b2 = m.L - r.b1 - 1; }  // MatchResult.java:56-57
          per[tid].push_back(rec);
        }
      }
    }
  };
  { std::vector<std::thread> th; for (int t = 0; t < nthreads; t++) th.emplace_back(search, t); for (auto& t : th) t.join(); }
  auto t2 = std::chrono::steady_clock::now();
  int64_t n = 0;
  for (auto& v : per) for (auto& r : v) { if (n < cap && out) out[n] = r; n++; }
  if (timings) { timings[0] = std::chrono::duration<double>(t1 - t0).count(); timings[1] = std::chrono::duration<double>(t2 - t1).count(); }
  if (stats) { stats[0] = (int64_t)kept.size(); stats[1] = ncmp.load(); stats[2] = nelem.load(); }
  return n;
}

}  // extern "C"
