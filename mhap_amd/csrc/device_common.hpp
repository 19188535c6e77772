// device_common.hpp — shared device/host helpers for the gfx950 MinHash overlap kernels.
// Hash arithmetic follows Guava 19.0's MurmurHash3 as called from
// J/sketch/HashUtils.java:213-258 (J/ = reference src/main/java/edu/umd/marbl/mhap/).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MHAP_WAVE 64

namespace mhap {

// ---- strand descriptors (host builds them, kernels consume them) -----------------------------
// One descriptor per READ; both strands share the base storage.
struct ReadDesc {
  int64_t base_off;   // packed: byte offset into packed2 (4 bases/byte, 4-byte aligned); raw: byte offset into raw bytes
  int64_t key_off;    // MHAP_RD_MAT reads: element offset of the forward strand's k-mer key array (rc strand: + key_stride)
  int64_t h2_off;     // MHAP_RD_MAT reads: element offset of the forward strand's 32-bit hash array (rc: + h2_stride)
  int64_t w_off;      // element offset of the forward strand's weight / class-list arrays (rc: + key_stride)
  int32_t length;     // bases
  int32_t key_stride; // elements between fwd and rc key / weight arrays (aligned nk)
  int32_t h2_stride;  // elements between fwd and rc h32 arrays (aligned nk2)
  int32_t flags;      // MHAP_RD_* bits
};
#define MHAP_RD_RAW 1       // raw bytes (read has non-ACGT chars)
#define MHAP_RD_SKIP 2      // skipped (too short)
#define MHAP_RD_FWDONLY 4   // -q mode: only the forward strand is sketched (AbstractMatchSearch.java:225)
#define MHAP_RD_MAT 8       // k-mer hashes are materialised in HBM by hash_kmers_kernel (raw bytes, k != 16 / k2 != 12, very long reads);
                            // every other strand's hashes are recomputed from its 2-bit codes wherever they are consumed

__host__ __device__ inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__host__ __device__ inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__host__ __device__ inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}
__host__ __device__ inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bU; h ^= h >> 13; h *= 0xc2b2ae35U; h ^= h >> 16;
  return h;
}

// 4 ASCII chars packed in a dword (c0 lowest byte) -> the 8 UTF-16LE message bytes as a u64
// (Guava Hasher.putUnencodedChars: each char little-endian, high byte 0 for ASCII).
__host__ __device__ inline uint64_t widen4(uint32_t d) {
  uint32_t lo = (d & 0xFFu) | ((d & 0xFF00u) << 8);
  uint32_t hi = ((d >> 16) & 0xFFu) | ((d >> 8) & 0xFF0000u);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

// Unaligned 4-char fetch from a byte string stored as aligned dwords: chars [o, o+4).
__host__ __device__ inline uint32_t fetch4(const uint32_t* W, int o) {
  int wi = o >> 2, sh = (o & 3) * 8;
  uint32_t a = W[wi];
  if (sh == 0) return a;
  uint32_t b = W[wi + 1];
  return (a >> sh) | (b << (32 - sh));
}

// murmur3_x64_128(seed 0).h1 of the k chars starting at char offset p of W (bytes in dwords).
// Message length = 2k bytes.  KT > 0 fixes k at compile time.
template <int KT>
__host__ __device__ inline uint64_t murmur128_h1_chars(const uint32_t* W, int p, int k_rt) {
  const int k = KT > 0 ? KT : k_rt;
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = 0, h2 = 0;
  const int nblocks = k >> 3;  // 8 chars = 16 bytes per block
#pragma unroll
  for (int b = 0; b < nblocks; b++) {
    uint64_t k1 = widen4(fetch4(W, p + 8 * b));
    uint64_t k2 = widen4(fetch4(W, p + 8 * b + 4));
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const int r = k & 7;  // tail chars (2r tail bytes)
  if (r > 0) {
    const int tp = p + 8 * nblocks;
    uint64_t k1 = 0, k2 = 0;
    // tail bytes 0..7 -> k1 (chars 0..3), bytes 8..15 -> k2 (chars 4..7)
    uint64_t lo = widen4(fetch4(W, tp));
    if (r >= 4) k1 = lo; else k1 = lo & ((1ULL << (16 * r)) - 1);
    if (r > 4) {
      uint64_t hi = widen4(fetch4(W, tp + 4));
      k2 = hi & ((1ULL << (16 * (r - 4))) - 1);
      k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    }
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
  }
  const uint64_t len = 2ULL * (uint64_t)k;
  h1 ^= len; h2 ^= len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2;
  return h1;
}

// murmur3_x86_32(seed 0) of the k chars starting at char offset p (2k message bytes).
template <int KT>
__host__ __device__ inline uint32_t murmur32_chars(const uint32_t* W, int p, int k_rt) {
  const int k = KT > 0 ? KT : k_rt;
  const uint32_t c1 = 0xcc9e2d51U, c2 = 0x1b873593U;
  uint32_t h1 = 0;
  const int nquads = k >> 2;  // 4 chars = two 4-byte blocks
#pragma unroll
  for (int q = 0; q < nquads; q++) {
    uint64_t w = widen4(fetch4(W, p + 4 * q));
    uint32_t k1 = (uint32_t)w;
    k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5 + 0xe6546b64U;
    k1 = (uint32_t)(w >> 32);
    k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5 + 0xe6546b64U;
  }
  const int r = k & 3;  // remaining chars: r>=2 -> one more full block; odd -> 2-byte tail
  if (r > 0) {
    uint64_t w = widen4(fetch4(W, p + 4 * nquads));
    if (r >= 2) {
      uint32_t k1 = (uint32_t)w;
      k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5 + 0xe6546b64U;
      w >>= 32;
    }
    if (r & 1) {
      uint32_t k1 = (uint32_t)w & 0xFFFFu;
      k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1;
    }
  }
  h1 ^= (uint32_t)(2 * k);
  return fmix32(h1);
}

// murmur3_x64_128(seed 0) of the 8 little-endian bytes of a long — what Guava's BloomFilter<Long> with the funnel
// `sink.putLong(value)` hashes (J/sketch/FrequencyCounts.java:137; BloomFilterStrategies.MURMUR128_MITZ_64): an 8-byte tail, no block.
__host__ __device__ inline void murmur128_long(uint64_t v, uint64_t& h1, uint64_t& h2) {
  uint64_t k1 = v * 0x87c37b91114253d5ULL; k1 = rotl64(k1, 31); k1 *= 0x4cf5ad432745937fULL;
  h1 = k1; h2 = 0;
  h1 ^= 8ULL; h2 ^= 8ULL;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2; h2 += h1;
}
// Guava 19.0 BloomFilter.mightContain for that strategy: bit i = ((h1 + i h2) & Long.MAX_VALUE) % bitSize, i = 0..k-1
__host__ __device__ inline bool bloom_might_contain(const unsigned long long* words, uint64_t bit_size, int k, uint64_t v) {
  uint64_t h1, h2;
  murmur128_long(v, h1, h2);
  uint64_t c = h1;
  for (int i = 0; i < k; i++) {
    const uint64_t bit = (c & 0x7fffffffffffffffULL) % bit_size;
    if (!((words[bit >> 6] >> (bit & 63)) & 1ULL)) return false;
    c += h2;
  }
  return true;
}

// The --num-hashes hash family: one xorshift64 step (J/sketch/MinHashSketch.java:140-142).
__host__ __device__ inline uint64_t xorshift_step(uint64_t x) {
  x ^= x << 21; x ^= x >> 35; x ^= x << 4;
  return x;
}

// The step undone: (I + L^4)^-1 = (I + L^4)(I + L^8)(I + L^16)(I + L^32), (I + R^35)^-1 = I + R^35, (I + L^21)^-1 = (I + L^21)(I + L^42).
// The step is a bijection, so the k-mer key behind a slot's minimal chain value x after n steps is unstep^n(x): the MinHash
// kernel of the weight-1 strands keeps only the minima and recovers the winning keys at the end (inverse jump tables).
__host__ __device__ inline uint64_t xorshift_unstep(uint64_t x) {
  x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32;
  x ^= x >> 35;
  x ^= x << 21; x ^= x << 42;
  return x;
}

// In-place transpose of a 32x32 bit matrix held as 32 dwords: afterwards bit j of a[b] == bit b of the old a[j].
// Five butterfly stages with compile-time indices (registers on the device).  Used to turn 32 k-mer keys per lane
// into 64 bit-planes for the bit-sliced xorshift rows of minhash_kernel.
__host__ __device__ inline void transpose32(uint32_t (&a)[32]) {
#define MHAP_TR_STAGE(J, M)                                                    \
  _Pragma("unroll") for (int k = 0; k < 32; k++) {                             \
    if ((k & (J)) == 0) {                                                      \
      const uint32_t t = ((a[k] >> (J)) ^ a[k + (J)]) & (M);                   \
      a[k] ^= t << (J);                                                        \
      a[k + (J)] ^= t;                                                         \
    }                                                                          \
  }
  MHAP_TR_STAGE(16, 0x0000FFFFu)
  MHAP_TR_STAGE(8, 0x00FF00FFu)
  MHAP_TR_STAGE(4, 0x0F0F0F0Fu)
  MHAP_TR_STAGE(2, 0x33333333u)
  MHAP_TR_STAGE(1, 0x55555555u)
#undef MHAP_TR_STAGE
}

// Java Math.round(double) for the tf-idf weight (J/sketch/MinHashSketch.java:120): ties toward +inf.
__host__ __device__ inline int64_t java_round(double x) {
  if (x != x) return 0;
  double f = floor(x);
  if (x - f >= 0.5) f += 1.0;
  if (f >= 9.2233720368547758e18) return INT64_MAX;
  if (f <= -9.2233720368547758e18) return INT64_MIN;
  return (int64_t)f;
}

// Utils.rc translate table (J/utils/Utils.java:84-117,496-507): upper-case, map, unknown unchanged.
__host__ __device__ inline uint32_t rc_char(uint32_t c) {
  if (c >= 'a' && c <= 'z') c = c - 'a' + 'A';
  switch (c) {
    case 'A': return 'T'; case 'B': return 'V'; case 'C': return 'G'; case 'D': return 'H';
    case 'G': return 'C'; case 'H': return 'D'; case 'K': return 'M'; case 'M': return 'K';
    case 'N': return 'N'; case 'R': return 'Y'; case 'S': return 'S'; case 'T': return 'A';
    case 'V': return 'B'; case 'W': return 'W'; case 'Y': return 'R';
    default: return c;
  }
}

__host__ __device__ inline bool strand_skipped(const ReadDesc& rd, int rcstrand) {
  return (rd.flags & MHAP_RD_SKIP) || (rcstrand && (rd.flags & MHAP_RD_FWDONLY));
}

// Char at strand position i. packed: 2 bits/base, A=0 C=1 G=2 T=3, base j in bits 2*(j&3) of byte j>>2.
__device__ inline uint32_t strand_char(const uint8_t* __restrict__ store, const ReadDesc& rd, int rcstrand, int i) {
  const int f = rcstrand ? (rd.length - 1 - i) : i;
  if (rd.flags & MHAP_RD_RAW) {
    uint32_t c = store[rd.base_off + f];
    return rcstrand ? rc_char(c) : c;
  }
  uint32_t code = (store[rd.base_off + (f >> 2)] >> (2 * (f & 3))) & 3u;
  if (rcstrand) code = 3u - code;
  return (0x54474341u >> (8 * code)) & 0xFFu;  // "ACGT"
}

}  // namespace mhap
