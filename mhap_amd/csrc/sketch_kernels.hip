// sketch_kernels.hip — gfx950 kernels for MHAP sketch construction (hot loops A-D of SURVEY §3.1).
//
//   hash_kmers_kernel   : murmur3_x64_128 of every k-mer + murmur3_x86_32 of every k2-mer
//                         (J/sketch/HashUtils.java:213-258), both strands, chars staged in LDS.
//   kmer_weight_kernel  : per-strand k-mer multiplicity / tf-idf weight, first-occurrence flag
//                         (J/sketch/MinHashSketch.java:66-81,98-128).
//   minhash_kernel      : weighted xorshift MinHash, one wavefront per strand, bit-sliced chains with a
//                         deferred candidate queue (J/sketch/MinHashSketch.java:130-154).
//   ordered_kernel      : bottom-S (signed hash, pos) select + sort
//                         (J/sketch/BottomOverlapSketch.java:525-559).
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"
#include <algorithm>

namespace mhap {

// =============================================================================================
// K-mer hashing.  grid = (nstrands, max tiles), block = 256.  A tile is HASH_TILE consecutive
// window starts of one strand; the chars it needs are decoded once into LDS (packed 2-bit or raw
// bytes -> ASCII, reverse-complemented on the fly for odd strands).
// =============================================================================================
// Fast path for k = 16, k2 = 12 on 2-bit packed strands (reads of pure ACGT): murmur3 mixes every 8-byte block
// (4 UTF-16 chars) on its own before folding it into the state, so the mixed value of a block is a function of 8 bits of
// base codes.  Three host-built 256-entry tables (k1-type and k2-type mixes of murmur3_x64_128, the two 4-byte mixes of
// murmur3_x86_32) replace 8 of the 12 64-bit multiplies and all 12 block multiplies of the 32-bit hash by 7 ds_read_b64.
constexpr int HASH_SEG = 4096;            // window starts per workgroup on the table path (4 tiles of the generic path's grid)
constexpr int HASH_LUT_WORDS = 3 * 256;   // uint64 words: [0,256) k1 mix, [256,512) k2 mix, [512,768) murmur32 {lo: chars 0-1, hi: chars 2-3}

void build_kmer_hash_luts(uint64_t* out) {
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  const uint32_t d1 = 0xcc9e2d51U, d2 = 0x1b873593U;
  const char* acgt = "ACGT";
  for (int idx = 0; idx < 256; idx++) {
    uint64_t block = 0;
    for (int c = 0; c < 4; c++) block |= (uint64_t)(uint8_t)acgt[(idx >> (2 * c)) & 3] << (16 * c);   // UTF-16LE, high bytes 0
    uint64_t k1 = block * c1; k1 = rotl64(k1, 31); k1 *= c2;
    uint64_t k2 = block * c2; k2 = rotl64(k2, 33); k2 *= c1;
    uint32_t a = (uint32_t)block, b = (uint32_t)(block >> 32);
    a *= d1; a = rotl32(a, 15); a *= d2;
    b *= d1; b = rotl32(b, 15); b *= d2;
    out[idx] = k1; out[256 + idx] = k2; out[512 + idx] = (uint64_t)a | ((uint64_t)b << 32);
  }
}

// Base codes of strand positions i0 .. i0+15 of a packed read (A=0 C=1 G=2 T=3, position i0+b in bits 2b), i0 a multiple
// of 16.  Forward strand: a 32-bit window of the packed bytes; reverse complement: the window that ends at base L-1-i0,
// 2-bit groups reversed, complemented (3 - code = ~code).  Bases outside the read come back as arbitrary codes (callers
// never hash a window that touches them); the loads stay inside the read's bytes.
__device__ __forceinline__ uint32_t strand_codes16(const uint8_t* __restrict__ pk, int L, int rcs, int i0) {
  // dword loads (reads are 4-byte aligned and padded to whole dwords): two aligned loads + a funnel shift instead of five byte loads
  const uint32_t* W = (const uint32_t*)pk;
  const int nd = (((L + 3) >> 2) + 3) >> 2;
  int f0 = rcs ? (L - 16 - i0) : i0;          // lowest forward base of the window (may be negative / beyond L)
  if (f0 <= -16 || f0 >= 16 * nd) return 0u;
  int sh = 0;
  if (f0 < 0) { sh = -2 * f0; f0 = 0; }
  const int d = f0 >> 4;
  const uint32_t d0 = W[d], d1 = W[d + 1 < nd ? d + 1 : nd - 1];
  uint32_t x = __builtin_amdgcn_alignbit(d1, d0, (uint32_t)(2 * (f0 & 15))) << sh;
  if (rcs) {
    x = __builtin_bswap32(x);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ~x;
  }
  return x;
}

// 16 base codes from strand position p on, out of a code stream held 16 per dword
__device__ __forceinline__ uint32_t codes_at(const uint32_t* codes, int p) {
  return __builtin_amdgcn_alignbit(codes[(p >> 4) + 1], codes[p >> 4], (uint32_t)(2 * (p & 15)));
}
// 16 base codes from strand position p on (position p+b in bits 2b) straight from the packed read in HBM/L2: W = the read's
// packed bytes as dwords (16 bases each), nd dwords.  Lanes of a wavefront ask for neighbouring positions, so the two dword
// loads hit the same few cache lines.  Valid for 0 <= p <= L-12; codes beyond the read come back arbitrary.
__device__ __forceinline__ uint32_t strand_window16(const uint32_t* __restrict__ W, int nd, int L, int rcs, int p) {
  int f0 = rcs ? (L - 16 - p) : p;              // lowest forward base of the window
  int sh = 0;
  if (f0 < 0) { sh = -2 * f0; f0 = 0; }         // (12-mers at the end of the reverse strand)
  const int i0 = f0 >> 4;
  const uint32_t d0 = W[i0 < nd ? i0 : nd - 1], d1 = W[i0 + 1 < nd ? i0 + 1 : nd - 1];
  uint32_t x = __builtin_amdgcn_alignbit(d1, d0, (uint32_t)(2 * (f0 & 15))) << sh;
  if (rcs) {
    x = __builtin_bswap32(x);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ~x;
  }
  return x;
}
// murmur3_x64_128(seed 0).h1 of the 16-mer / murmur3_x86_32(seed 0) of the 12-mer that start the 16 codes cw
__device__ __forceinline__ uint64_t mul5(uint64_t x) {
  uint64_t r;
  asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ uint64_t lut_key16(const uint64_t* lut, uint32_t cw) {
  uint64_t h1 = 0, h2 = 0;
  // (x * 5 as shift + add: v_lshl_add_u64 runs at full rate, the 64-bit multiply is three quarter-rate ops)
  h1 ^= lut[cw & 255u];                 h1 = rotl64(h1, 27); h1 += h2; h1 = mul5(h1) + 0x52dce729;
  h2 ^= lut[256 + ((cw >> 8) & 255u)];  h2 = rotl64(h2, 31); h2 += h1; h2 = mul5(h2) + 0x38495ab5;
  h1 ^= lut[(cw >> 16) & 255u];         h1 = rotl64(h1, 27); h1 += h2; h1 = mul5(h1) + 0x52dce729;
  h2 ^= lut[256 + (cw >> 24)];          h2 = rotl64(h2, 31); h2 += h1; h2 = mul5(h2) + 0x38495ab5;
  h1 ^= 32ULL; h2 ^= 32ULL;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  return h1 + h2;
}
__device__ __forceinline__ uint32_t lut_hash12(const uint64_t* lut, uint32_t cw) {
  uint32_t h = 0;
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const uint64_t kk = lut[512 + ((cw >> (8 * q)) & 255u)];
    h ^= (uint32_t)kk;         h = rotl32(h, 13); h = h * 5 + 0xe6546b64U;
    h ^= (uint32_t)(kk >> 32); h = rotl32(h, 13); h = h * 5 + 0xe6546b64U;
  }
  h ^= 24u;
  return fmix32(h);
}

template <int KT, int K2T>
__global__ __launch_bounds__(256) void hash_kmers_kernel(const ReadDesc* __restrict__ descs, const uint8_t* __restrict__ store,
                                                         int64_t* __restrict__ keys, int32_t* __restrict__ h32, int k_rt,
                                                         int k2_rt, const uint64_t* __restrict__ luts, int only_mat) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_chars[];
  const int k = KT > 0 ? KT : k_rt, k2 = K2T > 0 ? K2T : k2_rt;
  const int strand = blockIdx.x;
  const ReadDesc rd = descs[strand >> 1];
  const int rcs = strand & 1;
  if (strand_skipped(rd, rcs)) return;
  if (only_mat && !(rd.flags & MHAP_RD_MAT)) return;   // the other strands' hashes are recomputed from their 2-bit codes where they are used
  const int L = rd.length;
  const int nk = L - k + 1, nk2 = L - k2 + 1;
  const int nmax = nk > nk2 ? nk : nk2;
  int64_t* kout = keys + rd.key_off + (rcs ? rd.key_stride : 0);
  int32_t* hout = h32 + rd.h2_off + (rcs ? rd.h2_stride : 0);
  if (KT == 16 && K2T == 12 && !(rd.flags & MHAP_RD_RAW)) {
    // ---- table path: LDS = three tables + the segment's base codes, 16 per dword ----
    const int t0 = blockIdx.y * HASH_SEG;          // this workgroup's segment of window starts
    if (t0 >= nmax) return;
    uint64_t* lut = (uint64_t*)lds_chars;
    uint32_t* codes = lds_chars + 2 * HASH_LUT_WORDS;
    for (int i = threadIdx.x; i < HASH_LUT_WORDS; i += 256) lut[i] = luts[i];
    const int ncw = (HASH_SEG + 15 + 15) / 16 + 1;
    for (int wj = threadIdx.x; wj < ncw; wj += 256) codes[wj] = strand_codes16(store + rd.base_off, L, rcs, t0 + 16 * wj);
    __syncthreads();
    {
      for (int pp = threadIdx.x; pp < HASH_SEG; pp += 256) {
        const int p = t0 + pp;
        if (p >= nmax) break;
        const uint32_t cw = codes_at(codes, pp);
        if (p < nk) kout[p] = (int64_t)lut_key16(lut, cw);
        if (p < nk2) hout[p] = (int32_t)lut_hash12(lut, cw);
      }
    }
    return;
  }
  const int t0 = blockIdx.y * HASH_TILE;
  if (t0 >= nmax) return;
  const int halo = (k > k2 ? k : k2) - 1;
  const int nchars = HASH_TILE + halo;
  const int nwords = (nchars + 3) / 4 + 1;
  for (int wj = threadIdx.x; wj < nwords; wj += 256) {
    uint32_t d = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      int i = t0 + 4 * wj + b;
      uint32_t c = (i < L) ? strand_char(store, rd, rcs, i) : 0u;
      d |= c << (8 * b);
    }
    lds_chars[wj] = d;
  }
  __syncthreads();
  for (int pp = threadIdx.x; pp < HASH_TILE; pp += 256) {
    const int p = t0 + pp;
    if (p < nk) kout[p] = (int64_t)murmur128_h1_chars<KT>(lds_chars, pp, k);
    if (p < nk2) hout[p] = (int32_t)murmur32_chars<K2T>(lds_chars, pp, k2);
  }
}

void launch_hash_kmers(hipStream_t st, const ReadDesc* descs, int64_t nstrands, int max_len, const uint8_t* store,
                       int64_t* keys, int32_t* h32, int k, int k2, const uint64_t* luts, int only_mat) {
  if (nstrands <= 0) return;
  int kmin = k < k2 ? k : k2;
  int nmax = max_len - kmin + 1;
  if (nmax < 1) return;
  dim3 grid((unsigned)nstrands, (unsigned)((nmax + HASH_TILE - 1) / HASH_TILE));
  int halo = (k > k2 ? k : k2) - 1;
  size_t lds = (size_t)(((HASH_TILE + halo + 3) / 4 + 2) * 4);
  if (k == 16 && k2 == 12) {
    const size_t lut_lds = (size_t)HASH_LUT_WORDS * 8 + (size_t)((HASH_SEG + 30) / 16 + 2) * 4;
    if (lut_lds > lds) lds = lut_lds;
    // packed strands: the workgroups with blockIdx.y < ceil(windows / HASH_SEG) hash one segment each, the others exit
    // (raw-byte strands of the same launch take the generic path tile by tile, so the grid keeps all tiles)
    hipLaunchKernelGGL((hash_kmers_kernel<16, 12>), grid, dim3(256), lds, st, descs, store, keys, h32, k, k2, luts, only_mat);
  } else
    hipLaunchKernelGGL((hash_kmers_kernel<0, 0>), grid, dim3(256), lds, st, descs, store, keys, h32, k, k2, luts, only_mat);
}

// =============================================================================================
// K-mer multiplicity -> weight (the tf part of MHAP's tf-idf MinHash, MinHashSketch.java:66-81,98-128).
// Persistent workgroups pull strands from an atomic counter.  Each strand gets an open-addressing table of
// "first position + 1" entries keyed by the 64-bit k-mer hash:
//   * LDS path (nk <= 65534 and the table fits the launch's LDS budget): entry = fp16<<16 | (pos+1); the 16-bit
//     fingerprint filters probes so the full 64-bit key is only gathered from HBM/L2 on a fingerprint match
//     (true duplicate or 2^-16 collision).  ds_cmpst/ds_min atomics, no HBM traffic besides the key stream.
//   * HBM path (long reads): entry = pos+1 in a per-workgroup slab (stays in L2/MALL), global atomics.
// Multiplicities are accumulated directly in the output array: wts[first] += 1 per later occurrence.
// Output wts[i] = weight of k-mer i if i is the first occurrence of its key, else 0 — written only for strands whose k-mers
// do not all carry the same weight; those strands also get their first occurrences listed by weight class (class_list),
// which is what lets the MinHash kernel keep weighted k-mers (tf repeats, tf-idf under -f) on its bit-sliced rows.
// =============================================================================================
// Ordering between the waves of ONE workgroup over data in global memory: every cross-thread read in this kernel is an L2 access
// (atomics, sc1 loads through ld_agent) and the vector L1 is write-through, so the producer only has to wait until its stores
// and atomics have left the CU — a workgroup-scope release (s_waitcnt) before the barrier.  An agent-scope fence
// (__threadfence) would write back the whole XCD's dirty L2 lines each time: with every workgroup writing weights (tf-idf under
// -f) that cost ~0.9 ms per strand.
__device__ inline void wg_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ inline uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline bool filter_lookup(const FilterTable& ft, int64_t key, double& val) {
  uint64_t h = fmix64((uint64_t)key);
  uint32_t slot = (uint32_t)h & ft.mask;
  for (;;) {
    double v = ft.vals[slot];
    if (v == 0.0) return false;
    if (ft.keys[slot] == key) { val = v; return true; }
    slot = (slot + 1) & ft.mask;
  }
}

constexpr uint32_t DUP_MARK = 0x80000000u;

template <bool PACKED>
__device__ inline uint32_t wt_encode(int64_t key, int i) {
  return PACKED ? ((((uint32_t)((uint64_t)key >> 32)) << 16) | (uint32_t)(i + 1)) : (uint32_t)(i + 1);
}
template <bool PACKED>
__device__ inline bool wt_maybe(uint32_t e, uint32_t mine) { return PACKED ? ((e ^ mine) >> 16) == 0 : true; }
template <bool PACKED>
__device__ inline int wt_pos(uint32_t e) { return (int)((PACKED ? (e & 0xFFFFu) : e) - 1u); }

// One strand, whole workgroup.  tab: ts entries (LDS when PACKED, HBM slab otherwise), zeroed here.
template <bool PACKED>
__device__ inline void weight_strand(uint32_t* tab, uint32_t ts, const int64_t* __restrict__ kp, uint32_t* __restrict__ wp, int nk,
                                     unsigned int* s_heavy) {
  const uint32_t mask = ts - 1;
  for (uint32_t j = threadIdx.x; j < ts; j += WEIGHT_THREADS) tab[j] = 0;
  if (!PACKED) wg_release();
  __syncthreads();
  for (int i = threadIdx.x; i < nk; i += WEIGHT_THREADS) {
    const int64_t key = kp[i];
    const uint32_t mine = wt_encode<PACKED>(key, i);
    uint32_t slot = (uint32_t)(uint64_t)key & mask;
    for (;;) {
      uint32_t e = PACKED ? *(volatile uint32_t*)&tab[slot] : ld_agent(&tab[slot]);
      if (e == 0) {
        const uint32_t old = atomicCAS(&tab[slot], 0u, mine);
        if (old == 0) break;
        e = old;
      }
      if (wt_maybe<PACKED>(e, mine) && kp[wt_pos<PACKED>(e)] == key) { atomicMin(&tab[slot], mine); break; }
      slot = (slot + 1) & mask;
    }
  }
  if (!PACKED) wg_release();
  __syncthreads();
  // pass A: first occurrences get 1, later occurrences remember their first position
  bool anydup = false;
  for (int i = threadIdx.x; i < nk; i += WEIGHT_THREADS) {
    const int64_t key = kp[i];
    const uint32_t mine = wt_encode<PACKED>(key, i);
    uint32_t slot = (uint32_t)(uint64_t)key & mask;
    uint32_t w;
    for (;;) {
      const uint32_t e = PACKED ? *(volatile uint32_t*)&tab[slot] : ld_agent(&tab[slot]);
      if (e == mine) { w = 1u; break; }
      if (e != 0 && wt_maybe<PACKED>(e, mine)) {
        const int p = wt_pos<PACKED>(e);
        if (kp[p] == key) { w = DUP_MARK | (uint32_t)p; anydup = true; break; }
      }
      slot = (slot + 1) & mask;
    }
    wp[i] = w;
  }
  if (anydup) atomicOr(s_heavy, 1u);
  wg_release();
  __syncthreads();
  // pass B (only strands with duplicates): fold multiplicities into the first occurrence
  if (*(volatile unsigned int*)s_heavy) {
    for (int i = threadIdx.x; i < nk; i += WEIGHT_THREADS) {
      const uint32_t w = wp[i];
      if (w & DUP_MARK) { atomicAdd(&wp[w & ~DUP_MARK], 1u); wp[i] = 0u; }
    }
    wg_release();
    __syncthreads();
  }
}

// LDS fast path: the whole strand's probe state lives in registers (one dword per k-mer: fp16<<16 | slot), inserts go straight
// to ds_cmpst (no read-before-CAS), and the post-barrier pass is ONE ds_read per k-mer: the remembered slot holds the smallest
// position of that key, so `entry == mine` <=> first occurrence.  Requires nk <= MAXIT*WEIGHT_THREADS, ts <= 32768.
// FUSED (k = 16, k2 = 12, packed strand): nothing is read from memory but the strand's base codes staged in LDS (fz.codes), and
// the table is keyed by the 32 bits of codes themselves (code_mix) — equal codes <=> equal k-mers, so no hash of the reference is
// needed to find repeats (round 2 computed murmur3_x64_128 per k-mer here: the kernel was bound by its 64-bit multiplies).
// Materialised strands (raw bytes, k != 16) are keyed by the 64-bit keys hash_kmers_kernel stored, fetched CH at a time.
struct FusedHash { const uint64_t* lut; const uint32_t* codes; bool on; };
// Dedupe key of a packed 16-mer: its 32 bits of base codes ARE the k-mer, so equal k-mers are found by comparing codes — no
// murmur3 in this kernel unless a k-mer filter needs the hash.  Two multiplies mix the codes so that table slot (low bits),
// probe stride (middle bits) and fingerprint (top 16 bits) are independent.
__device__ __forceinline__ uint32_t code_mix(uint32_t cw) {
  uint32_t h = cw * 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
  return h;
}
template <int MAXIT, bool FUSED>
__device__ inline void weight_strand_lds(uint32_t* tab, uint32_t ts, const int64_t* __restrict__ kp, uint32_t* __restrict__ wp, int nk,
                                         unsigned int* s_heavy, const FusedHash& fz, bool may_skip) {
  const uint32_t mask = ts - 1;
  // The lane's k-mer positions (tx + it * 1024) must be recomputed inside the strand loop: left to the compiler they are hoisted
  // out of it — two dozen loop-invariant values per lane — and, at 64 VGPRs, spilled to scratch and reloaded for every strand
  // (18 GB of HBM traffic per step at C2).  The empty asm makes tx opaque, so nothing derived from it can be hoisted.
  int tx = (int)threadIdx.x;
  asm volatile("" : "+v"(tx));
  for (uint32_t j = (uint32_t)tx * 4; j < ts; j += WEIGHT_THREADS * 4) *(uint4*)&tab[j] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  uint32_t st[MAXIT];
  const bool codekey = FUSED && fz.on;
  constexpr int CH = FUSED ? 2 : ((MAXIT >= 24) ? 8 : 4);   // keys in flight per lane (register budget: two workgroups per CU need <= 64 VGPRs)
#pragma unroll
  for (int c = 0; c < MAXIT; c += CH) {
    if (c * WEIGHT_THREADS < nk) {
      int64_t key[CH];     // materialised keys (!codekey)
      uint32_t cw[CH];     // base codes of the 16-mer (codekey)
#pragma unroll
      for (int u = 0; u < CH; u++) {
        const int i = tx + (c + u) * WEIGHT_THREADS;
        if (codekey) { cw[u] = (i < nk) ? codes_at(fz.codes, i) : 0u; key[u] = 0; }
        else { key[u] = (i < nk) ? kp[i] : 0; cw[u] = 0u; }
      }
#pragma unroll
      for (int u = 0; u < CH; u++) {
        const int i = tx + (c + u) * WEIGHT_THREADS;
        st[c + u] = 0;
        if (i < nk) {
          const uint32_t hm = codekey ? code_mix(cw[u]) : 0u;
          const uint32_t fp = codekey ? (hm & 0xFFFF0000u) : (((uint32_t)((uint64_t)key[u] >> 32)) << 16);
          const uint32_t mine = fp | (uint32_t)(i + 1);
          uint32_t slot = (codekey ? hm : (uint32_t)(uint64_t)key[u]) & mask;
          const uint32_t stride = ((codekey ? (hm >> 7) : (uint32_t)((uint64_t)key[u] >> 20)) | 1u) & mask;   // double hashing (odd stride, power-of-two table): no primary clustering
          for (;;) {
            const uint32_t old = atomicCAS(&tab[slot], 0u, mine);
            if (old == 0) break;
            if (((old ^ mine) >> 16) == 0) {   // fingerprint match: compare the k-mers themselves (their codes, or the stored keys)
              const int op = (int)((old & 0xFFFFu) - 1u);
              const bool same = codekey ? (codes_at(fz.codes, op) == cw[u]) : (kp[op] == key[u]);
              if (same) { atomicMin(&tab[slot], mine); break; }
            }
            slot = (slot + stride) & mask;
          }
          st[c + u] = fp | slot;
        }
      }
    }
  }
  __syncthreads();
  // a strand without a repeated k-mer (nearly all of them) has weight 1 everywhere: when the caller allows it, that is
  // reported through *s_heavy = 2 and the weight array is not written at all (the MinHash kernel then does not read it)
  bool anydup = false;
#pragma unroll
  for (int it = 0; it < MAXIT; it++) {
    const int i = tx + it * WEIGHT_THREADS;
    if (i < nk) {
      const uint32_t e = tab[st[it] & 0xFFFFu];
      const uint32_t mine = (st[it] & 0xFFFF0000u) | (uint32_t)(i + 1);
      uint32_t w = 1u;
      if (e != mine) { w = DUP_MARK | ((e & 0xFFFFu) - 1u); anydup = true; }
      if (!may_skip) wp[i] = w;
    }
  }
  if (anydup) atomicOr(s_heavy, 1u);
  __syncthreads();
  if (may_skip) {
    if (*(volatile unsigned int*)s_heavy == 0u) {
      __syncthreads();
      if (tx == 0) *s_heavy = 2u;
      return;
    }
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
      const int i = tx + it * WEIGHT_THREADS;
      if (i < nk) {
        const uint32_t e = tab[st[it] & 0xFFFFu];
        const uint32_t mine = (st[it] & 0xFFFF0000u) | (uint32_t)(i + 1);
        wp[i] = (e != mine) ? (DUP_MARK | ((e & 0xFFFFu) - 1u)) : 1u;
      }
    }
  }
  if (*(volatile unsigned int*)s_heavy) {
    wg_release();
    __syncthreads();
    for (int i = tx; i < nk; i += WEIGHT_THREADS) {
      const uint32_t w = wp[i];
      if (w & DUP_MARK) { atomicAdd(&wp[w & ~DUP_MARK], 1u); wp[i] = 0u; }
    }
    wg_release();
    __syncthreads();
  }
}

// Long reads (more k-mers than the LDS table or the per-lane register state can hold): the k-mers are split by the top
// bits of their hash into `npart` partitions and each partition gets its own pass through the LDS table (equal keys
// always share a partition).  Entry = fingerprint << posbits | (pos+1).  Returns false (wave-uniformly) if a partition
// ever fills the table (pathological skew); the caller then falls back to the HBM-slab path.
__device__ inline bool weight_strand_lds_part(uint32_t* tab, uint32_t ts, const int64_t* __restrict__ kp, uint32_t* __restrict__ wp, int nk,
                                              int npart_log2, int posbits, unsigned int* s_heavy, unsigned int* s_fill) {
  const uint32_t mask = ts - 1, posmask = (posbits >= 32) ? 0xFFFFFFFFu : ((1u << posbits) - 1u);
  const uint32_t limit = ts - ts / 8;
  bool anydup = false;
  for (int part = 0; part < (1 << npart_log2); part++) {
    for (uint32_t j = threadIdx.x * 4; j < ts; j += WEIGHT_THREADS * 4) *(uint4*)&tab[j] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x == 0) *s_fill = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nk; i += WEIGHT_THREADS) {
      const int64_t key = kp[i];
      if (npart_log2 && (int)((uint64_t)key >> (64 - npart_log2)) != part) continue;
      const uint32_t fp = (posbits >= 32) ? 0u : (((uint32_t)((uint64_t)key >> 32)) << posbits);
      const uint32_t mine = fp | (uint32_t)(i + 1);
      uint32_t slot = (uint32_t)(uint64_t)key & mask;
      for (uint32_t tries = 0; tries < ts; tries++) {
        const uint32_t old = atomicCAS(&tab[slot], 0u, mine);
        if (old == 0) { atomicAdd(s_fill, 1u); break; }
        if (((old ^ mine) & ~posmask) == 0 && kp[(old & posmask) - 1u] == key) { atomicMin(&tab[slot], mine); break; }
        if (*(volatile unsigned int*)s_fill >= limit) break;    // table (nearly) full: give up, see below
        slot = (slot + 1) & mask;
      }
    }
    __syncthreads();
    if (*(volatile unsigned int*)s_fill >= limit) return false;
    for (int i = threadIdx.x; i < nk; i += WEIGHT_THREADS) {
      const int64_t key = kp[i];
      if (npart_log2 && (int)((uint64_t)key >> (64 - npart_log2)) != part) continue;
      const uint32_t fp = (posbits >= 32) ? 0u : (((uint32_t)((uint64_t)key >> 32)) << posbits);
      const uint32_t mine = fp | (uint32_t)(i + 1);
      uint32_t slot = (uint32_t)(uint64_t)key & mask;
      uint32_t w;
      for (;;) {
        const uint32_t e = tab[slot];
        if (e == mine) { w = 1u; break; }
        if (e != 0 && ((e ^ mine) & ~posmask) == 0) {
          const uint32_t p = (e & posmask) - 1u;
          if (kp[p] == key) { w = DUP_MARK | p; anydup = true; break; }
        }
        slot = (slot + 1) & mask;
      }
      wp[i] = w;
    }
    __syncthreads();
  }
  if (anydup) atomicOr(s_heavy, 1u);
  __syncthreads();
  if (*(volatile unsigned int*)s_heavy) {
    wg_release();
    __syncthreads();
    for (int i = threadIdx.x; i < nk; i += WEIGHT_THREADS) {
      const uint32_t w = wp[i];
      if (w & DUP_MARK) { atomicAdd(&wp[w & ~DUP_MARK], 1u); wp[i] = 0u; }
    }
    wg_release();
    __syncthreads();
  }
  return true;
}

// First occurrences of a strand listed by weight class (see StrandInfo): two passes over the final weights in wp[] — count
// per class, then scatter — with one LDS atomic per wavefront and class present (ballot + mbcnt ranks).  sv: 3 x 8 words of
// LDS (counts, segment starts, fill counters), zero on entry; order inside a class is arbitrary (the MinHash update resolves
// ties by position, not by processing order).
__device__ inline int weight_class(uint32_t w) { return w == 0u ? -1 : (int)((w > (uint32_t)BS_WCLASSES ? (uint32_t)BS_WCLASSES + 1u : w) - 1u); }
__device__ inline void class_list(const uint32_t* __restrict__ wp, uint32_t* __restrict__ perm, int nk, uint32_t* sv, int32_t* cnt_out) {
  const int lane = threadIdx.x & 63;
  for (int i0 = 0; i0 < nk; i0 += WEIGHT_THREADS) {           // block-uniform trip count (ballots inside)
    const int i = i0 + (int)threadIdx.x;
    int c = i < nk ? weight_class(ld_agent(&wp[i])) : -1;
    unsigned long long rest = __ballot(c >= 0);
    while (rest) {
      const int cc = __shfl(c, __builtin_ctzll(rest));
      const unsigned long long m = __ballot(c == cc);
      if (lane == (int)__builtin_ctzll(m)) atomicAdd(&sv[cc], (uint32_t)__popcll(m));
      rest &= ~m;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int c = 0; c <= BS_WCLASSES; c++) { sv[8 + c] = run; cnt_out[c] = (int32_t)sv[c]; run += sv[c]; }
  }
  __syncthreads();
  for (int i0 = 0; i0 < nk; i0 += WEIGHT_THREADS) {
    const int i = i0 + (int)threadIdx.x;
    int c = i < nk ? weight_class(ld_agent(&wp[i])) : -1;
    unsigned long long rest = __ballot(c >= 0);
    while (rest) {
      const int cc = __shfl(c, __builtin_ctzll(rest));
      const unsigned long long m = __ballot(c == cc);
      uint32_t base = 0;
      if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&sv[16 + cc], (uint32_t)__popcll(m));
      base = __shfl(base, __builtin_ctzll(m));
      if (c == cc) perm[sv[8 + cc] + base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint32_t)i;
      rest &= ~m;
    }
  }
}

constexpr int WEIGHT_SVARS = 32;   // LDS scalars of kmer_weight_kernel: [0..1] strand index, [2] valid / partition fill, [3] heavy, [4] min weight,
                                   // [5] max weight, [8..31] class_list scratch
template <int MAXIT, int WAVES_PER_SIMD, bool FUSED, bool REWEIGH>
__global__ __launch_bounds__(WEIGHT_THREADS, WAVES_PER_SIMD) void kmer_weight_kernel(const ReadDesc* __restrict__ descs, int64_t nstrands,
                                                                     const int64_t* __restrict__ keys, uint32_t* __restrict__ wts,
                                                                     uint32_t* __restrict__ perms,
                                                                     uint32_t* __restrict__ slabs, int64_t slab_entries, uint32_t lds_entries,
                                                                     unsigned long long* __restrict__ counter, int k,
                                                                     FilterTable ft, double repeat_weight,
                                                                     StrandInfo* __restrict__ info, const uint8_t* __restrict__ store,
                                                                     const uint64_t* __restrict__ luts,
                                                                     const int32_t* __restrict__ order, int32_t* __restrict__ slist) {
  // slist: the two work lists of the MinHash launches — [0, nstrands) strands of weight 1 + strands without a sketch,
  // [nstrands, 2 nstrands) weighted strands; their fill counts are counter[4] and counter[5]
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_tab[];
  uint32_t* svars = lds_tab + lds_entries;
  uint64_t* lut = (uint64_t*)(svars + WEIGHT_SVARS);            // FUSED: block-mix tables, then the strand's base codes
  uint32_t* codes = (uint32_t*)(lut + HASH_LUT_WORDS);
  if (FUSED) for (int i = threadIdx.x; i < HASH_LUT_WORDS; i += WEIGHT_THREADS) lut[i] = luts[i];
  uint32_t* slab = slabs + (size_t)blockIdx.x * (size_t)slab_entries;
  // REWEIGH (decided by the host: v1.0 mode, tf-idf under -f, --supress-noise 1): weights are not plain multiplicities.  A template
  // parameter, so that the common instantiation carries none of the filter's state in its (scarce, at 8 waves per SIMD) scalar registers
  const bool reweigh = REWEIGH;
  // Without a reweighing rule a k-mer's weight is its multiplicity, and a strand and its reverse complement repeat the same
  // k-mers (reverse-complementing is a bijection on k-mers): the work item is then a READ — its forward strand is examined, and a
  // packed read without a repeated k-mer (nearly all of them) settles both strands at once; only a read with repeats has its
  // reverse strand examined too (its first occurrences are other positions).  With a filter the weights depend on each k-mer's
  // own hash, so every strand is an item.
  const bool pairs = !reweigh;
  const int64_t nitems = pairs ? (nstrands >> 1) : nstrands;
  // (round 6: the NEXT item is pulled while the current one is worked on — the atomic's answer used to be waited for by all 1 024
  //  lanes at the top of every item, ~200 round trips to the counter per workgroup and launch.  The counter ends nblocks past the items.)
  unsigned long long sx_next = 0;
  if (threadIdx.x == 0) sx_next = atomicAdd(counter, 1ULL);
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long sx = sx_next;
      svars[0] = (uint32_t)sx; svars[1] = (uint32_t)(sx >> 32);
      if ((int64_t)sx < nitems) sx_next = atomicAdd(counter, 1ULL);
    }
    __syncthreads();
    // readfirstlane: the item — and with it the descriptor, every pointer and loop bound below — is provably uniform (scalar
    // registers and scalar loads instead of a copy per lane)
    const int64_t item = (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)svars[1]) << 32) |
                                   (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)svars[0]));
    if (item >= nitems) break;
    int64_t read = pairs ? item : (item >> 1);
    if (order) read = order[read];   // longest reads first
    const ReadDesc rd = descs[read];
    const int nk = rd.length - k + 1;
    const int rcs_first = pairs ? 0 : (int)(item & 1), rcs_last = pairs ? 1 : rcs_first;
    for (int rcs = rcs_first; rcs <= rcs_last; rcs++) {
    const int64_t strand = 2 * read + rcs;
    __syncthreads();
    if (threadIdx.x == 0) { svars[2] = 0; svars[3] = 0; svars[4] = 0xFFFFFFFFu; svars[5] = 0; }
    if (threadIdx.x >= 8 && threadIdx.x < WEIGHT_SVARS) svars[threadIdx.x] = 0;
    __syncthreads();
    if (strand_skipped(rd, rcs) || nk < 1) {
      if (threadIdx.x == 0) { info[strand].valid = 0; info[strand].mode = 1; slist[atomicAdd(counter + 4, 1ULL)] = (int32_t)strand; }
      continue;
    }
    const bool mat = !FUSED || (rd.flags & MHAP_RD_MAT);           // keys materialised by hash_kmers_kernel
    const int64_t* kp = keys + rd.key_off + (rcs ? rd.key_stride : 0);
    uint32_t* wp = wts + rd.w_off + (rcs ? rd.key_stride : 0);
    uint32_t* perm = perms + rd.w_off + (rcs ? rd.key_stride : 0);
    uint32_t ts = 64;
    while (3ull * ts < 4ull * (uint32_t)nk) ts <<= 1;          // load factor <= 0.75
    FusedHash fz;
    fz.lut = lut; fz.codes = codes; fz.on = false;
    if (!mat) {
      // strands without MHAP_RD_MAT always fit the LDS path (strand_hashes_from_codes)
      fz.on = true;
      const int ncw = (rd.length + 15) / 16 + 2;
      for (int wj = threadIdx.x; wj < ncw; wj += WEIGHT_THREADS) codes[wj] = strand_codes16(store + rd.base_off, rd.length, rcs, 16 * wj);
      // (the table-zeroing barrier inside weight_strand_lds orders these stores before the first hash)
      weight_strand_lds<MAXIT, FUSED>(lds_tab, ts, kp, wp, nk, &svars[3], fz, true);
    } else if (ts <= lds_entries && nk <= MAXIT * WEIGHT_THREADS) {
      weight_strand_lds<MAXIT, FUSED>(lds_tab, ts, kp, wp, nk, &svars[3], fz, true);
    } else {
      // long read: hash-partitioned passes through the LDS table; HBM slab only if a partition overflows it
      int plog = 0;
      while (plog < 10 && 3ull * lds_entries < (4ull * (uint32_t)nk * 5 / 4) >> plog) plog++;   // 25 % head-room for uneven partitions
      int posbits = 1;
      while (posbits < 32 && (1ull << posbits) <= (unsigned long long)nk) posbits++;
      if (!weight_strand_lds_part(lds_tab, lds_entries, kp, wp, nk, plog, posbits, &svars[3], &svars[2])) {
        __syncthreads();
        if (threadIdx.x == 0) svars[3] = 0;
        __syncthreads();
        weight_strand<false>(slab, ts, kp, wp, nk, &svars[3]);
      }
      if (threadIdx.x == 0) svars[2] = 0;
    }
    __syncthreads();
    const bool dups = svars[3] == 1u;
    const bool unwritten = svars[3] == 2u;   // no k-mer repeats and wp[] was not written (every multiplicity is 1)
    if (reweigh) {
      // weights differ from plain multiplicity: v1.0 mode or tf-idf (MinHashSketch.java:101-124)
      uint32_t mymin = 0xFFFFFFFFu, mymax = 0;
      for (int i = threadIdx.x; i < nk; i += WEIGHT_THREADS) {
        const int count = unwritten ? 1 : (int)ld_agent(&wp[i]);
        if (count == 0) continue;
        const int64_t key = fz.on ? (int64_t)lut_key16(lut, codes_at(codes, i)) : kp[i];
        int weight;
        double v;
        const bool listed = ft.bloom_mode == 0 || bloom_might_contain(ft.bloom, ft.bloom_bits, ft.bloom_k, (uint64_t)key);
        if (ft.bloom_mode == 1 && !listed) {
          weight = 0;                                                          // keepKmer false: the k-mer never enters the map (MinHashSketch.java:72-73)
        } else if (repeat_weight < 0.0) {
          weight = (ft.size > 0 && filter_lookup(ft, key, v)) ? 0 : 1;
        } else if (ft.enabled && repeat_weight < 1.0) {
          double idf = ft.range;
          if (ft.bloom_mode == 2 && !listed) idf = 1.0;                        // FrequencyCounts.java:297-298
          else if (ft.size > 0 && filter_lookup(ft, key, v)) idf = v;
          const double tf = ft.no_tf ? 1.0 : (double)count;
          weight = (int)java_round(tf * idf);
          if (weight < 1) weight = 1;
        } else weight = count;                                                 // tf only (repeat weight >= 1) behind a whitelist
        wp[i] = (uint32_t)weight;
        mymin = (uint32_t)weight < mymin ? (uint32_t)weight : mymin;
        mymax = (uint32_t)weight > mymax ? (uint32_t)weight : mymax;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const uint32_t a = __shfl_xor(mymin, off), b = __shfl_xor(mymax, off);
        mymin = a < mymin ? a : mymin; mymax = b > mymax ? b : mymax;
      }
      if ((threadIdx.x & 63) == 0) { atomicMin(&svars[4], mymin); atomicMax(&svars[5], mymax); }
      wg_release();
      __syncthreads();
    }
    // weight classes: one common weight and no repeats -> mode = that weight, nothing else is read downstream
    int mode = 0, valid = 1;
    if (reweigh) {
      valid = svars[5] > 0u;
      if (!dups && svars[4] == svars[5] && svars[4] > 0u) mode = (int)svars[4];
    } else if (!dups) {
      mode = 1;
      // (partitioned / slab paths write wp[] even without repeats; it is simply not read)
    }
    if (valid && mode == 0) {
      class_list(wp, perm, nk, svars + 8, info[strand].cnt);
      wg_release();
    }
    // a packed read whose forward strand repeats no k-mer: neither does its reverse complement
    const bool settle_rc = pairs && rcs == 0 && mode == 1 && !(rd.flags & MHAP_RD_RAW);
    if (threadIdx.x == 0) {
      info[strand].valid = valid; info[strand].mode = mode;
      if (!valid || mode == 1) slist[atomicAdd(counter + 4, 1ULL)] = (int32_t)strand;
      else slist[nstrands + (int64_t)atomicAdd(counter + 5, 1ULL)] = (int32_t)strand;
      if (settle_rc) {
        const bool rc_skipped = strand_skipped(rd, 1);
        info[strand + 1].valid = rc_skipped ? 0 : 1; info[strand + 1].mode = 1;
        slist[atomicAdd(counter + 4, 1ULL)] = (int32_t)(strand + 1);
      }
    }
    if (settle_rc) break;
    }
  }
}

// A read's k-mer hashes are recomputed from its 2-bit codes (no MHAP_RD_MAT) when the table path applies and its k-mers fit the
// weight kernel's LDS path; the host flags everything else for hash_kmers_kernel.
bool strand_hashes_from_codes(int length, int k, int k2) { return k == 16 && k2 == 12 && length - k + 1 <= WEIGHT_MAXIT * WEIGHT_THREADS; }

// fused = k = 16 and k2 = 12: strands without MHAP_RD_MAT are hashed here from their base codes (the others must have been
// hashed by hash_kmers_kernel before).
void launch_kmer_weights(hipStream_t st, int num_cus, const ReadDesc* descs, int64_t nstrands, int max_len, const int64_t* keys,
                         uint32_t* wts, uint32_t* perm, uint32_t* slabs, int64_t slab_entries, unsigned long long* counter, int k,
                         const FilterTable& ft, double repeat_weight, StrandInfo* info, bool fused, const uint8_t* store,
                         const uint64_t* luts, const int32_t* order, int32_t* slist) {
  if (nstrands <= 0) return;
  uint32_t need = 64;
  const uint32_t nkmax = (uint32_t)(max_len - k + 1 > 1 ? max_len - k + 1 : 1);
  while (3ull * need < 4ull * nkmax) need <<= 1;
  const uint32_t lds_entries = need > 32768u ? 32768u : need;           // <= 128 KiB of the CU's 160 KiB LDS
  size_t lds = (size_t)lds_entries * 4 + (size_t)WEIGHT_SVARS * 4;
  if (fused) {   // block-mix tables + base codes of one strand (strands longer than the LDS path are MHAP_RD_MAT)
    const int code_len = max_len < WEIGHT_MAXIT * WEIGHT_THREADS + k ? max_len : WEIGHT_MAXIT * WEIGHT_THREADS + k;
    lds += (size_t)HASH_LUT_WORDS * 8 + (size_t)((code_len + 15) / 16 + 4) * 4;
  }
  const int nblocks = weight_grid(num_cus, nstrands, max_len, k);
  const dim3 g(nblocks), b(WEIGHT_THREADS);
  const bool reweigh = (repeat_weight < 0.0) || (ft.enabled && repeat_weight < 1.0) || (ft.enabled && ft.bloom_mode == 1);
#define MHAP_LAUNCH_WEIGHT(MI, WPS, FU, RW)                                                                                              \
  hipLaunchKernelGGL((kmer_weight_kernel<MI, WPS, FU, RW>), g, b, lds, st, descs, nstrands, keys, wts, perm, slabs, slab_entries, lds_entries, \
                     counter, k, ft, repeat_weight, info, store, luts, order, slist)
  if (lds_entries <= 16384u) {   // reads up to 12288 k-mers: 64 KiB table, 12 k-mers per lane in registers, two workgroups per CU
    if (fused) { if (reweigh) MHAP_LAUNCH_WEIGHT(WEIGHT_MAXIT / 2, 8, true, true); else MHAP_LAUNCH_WEIGHT(WEIGHT_MAXIT / 2, 8, true, false); }
    else { if (reweigh) MHAP_LAUNCH_WEIGHT(WEIGHT_MAXIT / 2, 8, false, true); else MHAP_LAUNCH_WEIGHT(WEIGHT_MAXIT / 2, 8, false, false); }
  } else {
    if (fused) { if (reweigh) MHAP_LAUNCH_WEIGHT(WEIGHT_MAXIT, 4, true, true); else MHAP_LAUNCH_WEIGHT(WEIGHT_MAXIT, 4, true, false); }
    else { if (reweigh) MHAP_LAUNCH_WEIGHT(WEIGHT_MAXIT, 4, false, true); else MHAP_LAUNCH_WEIGHT(WEIGHT_MAXIT, 4, false, false); }
  }
#undef MHAP_LAUNCH_WEIGHT
}

// number of persistent workgroups (= HBM slabs the caller must provide)
int weight_grid(int num_cus, int64_t nstrands, int max_len, int k) {
  uint32_t need = 64;
  const uint32_t nkmax = (uint32_t)(max_len - k + 1 > 1 ? max_len - k + 1 : 1);
  while (3ull * need < 4ull * nkmax) need <<= 1;
  const size_t lds = (size_t)(need > 32768u ? 32768u : need) * 4 + 16;
  const int per_cu = lds <= 72 * 1024 ? 2 : 1;
  const int64_t g = (int64_t)num_cus * per_cu;
  return (int)(nstrands < g ? nstrands : g);
}

// =============================================================================================
// Weighted MinHash (J/sketch/MinHashSketch.java:130-154).  One wavefront per strand (strands pulled from an atomic
// counter); the per-slot running minimum (best[s], bpos[s] = first position of the winning k-mer) is wave-private in
// LDS.  Ties (equal chain values of different k-mers; 2^-64) resolve to the earlier first-occurrence position, which
// is the reference's insertion order.
//
// A k-mer of weight w takes w xorshift steps per slot and is compared after every step (:137-152).  The strand's distinct
// k-mers arrive grouped by weight (StrandInfo: one common weight, or the class list written by kmer_weight_kernel), and
// two row formats walk their chains through all H slots:
//   * bit-sliced rows (default, >= BS_MINREM k-mers of one weight left): 32 chains per lane as 64 bit-planes, w steps
//     and w candidate filters per slot;
//   * per-chain rows (what is left of every class, weights above BS_WCLASSES, MHAP_MINHASH=perchain): MH_U chains per
//     lane in 64-bit registers with their own weights, the hot loop compares the high dword of the chain's minimum over
//     its steps with the slot threshold (uniform ds_read, prefetched) and a ballot sends the wave into the exact update.
// K-mer keys are not read from memory for packed strands: every consumer (row load, candidate drain, final slot values)
// recomputes murmur3_x64_128 from the read's 2-bit codes (two dword loads + the LDS block-mix tables), so the sketch
// phase no longer writes or reads 8 B per k-mer; raw-byte strands and k != 16 read the keys hash_kmers_kernel stored.
// =============================================================================================
struct KeySrc {
  const int64_t* kp;          // materialised keys (MHAP_RD_MAT strands) or null
  const uint32_t* W;          // packed read as dwords
  int nd, L, rcs;
  const uint64_t* lut;        // LDS: k1 / k2 block-mix tables of murmur3_x64_128
  const uint32_t* perm;       // class list (positions) or null = identity
};
__device__ __forceinline__ int ks_pos(const KeySrc& ks, int idx) { return ks.perm ? (int)ks.perm[idx] : idx; }
__device__ __forceinline__ uint64_t ks_key(const KeySrc& ks, int pos) {
  if (ks.kp) return (uint64_t)ks.kp[pos];
  return lut_key16(ks.lut, strand_window16(ks.W, ks.nd, ks.L, ks.rcs, pos));
}

// Exact update of slot s from the wave's N candidate values per lane.  Wave-uniform control flow; the winner is
// moved with v_readlane (SGPR lane index from the ballot), no LDS permutes.
template <int N>
__device__ __forceinline__ void minhash_update(int64_t* best, int32_t* bpos, int s, const int64_t (&xv)[N], const int (&pv)[N],
                                               const bool (&act)[N], int lane) {
  int64_t cur = best[s];
  int32_t curpos = bpos[s];
  bool changed = false;
#pragma unroll
  for (int u = 0; u < N; u++) {
    const int64_t x = xv[u];
    const int p = pv[u];
    bool c = act[u] && (x < cur || (x == cur && p < curpos));
    unsigned long long m = __ballot(c);
    while (m) {
      const int l = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)x, l);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)x >> 32), l);
      cur = (int64_t)(((uint64_t)hi << 32) | lo);
      curpos = __builtin_amdgcn_readlane(p, l);
      changed = true;
      c = act[u] && (x < cur || (x == cur && p < curpos));
      m = __ballot(c);
    }
  }
  if (changed && lane == 0) { best[s] = cur; bpos[s] = curpos; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- bit-sliced rows ---------------------------------------------------------------------------------------------
// 32 k-mers per lane are held as 64 bit-planes (P[b] bit j = bit b of k-mer j's chain value), so one xorshift64 step
// of all 32 chains is 107 full-rate ops (the shifts become register renaming) = 3.3 issue slots per chain step
// instead of ~10 (2 v_lshlrev_b64 + 6 ops).  A slot's candidates are the chains whose value is negative with at least
// as many leading zero magnitude bits as the slot's current minimum (necessary for x <= min: 1 + z more VALU ops).
constexpr int BS_MINREM = 512;   // remaining k-mers of a weight class needed to start another 2048-chain bit-sliced row
#ifndef MH_ZMAX
#define MH_ZMAX 12
#endif
constexpr int BS_ZMAX = MH_ZMAX;   // filter depth cap.  Every plane costs one VALU op per step, every false candidate a queue entry: measured at C2 with
                                   // caps 10 / 12 / 13 / 16 / 24: 91.4 / 89.5 / 89.6 / 91.6 / 101.7 ms (2048 x 2^-13 = a quarter of the steps pass a false candidate at 12)
#ifndef MH_QCAP
#define MH_QCAP 4096
#endif
constexpr int BS_QCAP = MH_QCAP;     // deferred-candidate queue entries per wave.  The queue lives in global memory (16 KB per resident wave, written
                                     // once and read once: it stays in L2): a row's candidates (~1000 in a strand's first row, fewer later) always fit, so the
                                     // queue is drained at the END of a row only — where the 64 plane registers are dead — and the slot loop carries no
                                     // drain code.  Round 2 kept 640 entries in LDS and drained mid-row: the drain's state stayed live across the slot loop
                                     // (122 VGPRs) and its LDS kept a fifth workgroup off the CU.
                                     // (Round 3 also tried the first 1024 entries of a row in LDS with the global queue as overflow, to keep the entries'
                                     // 4-byte stores — each a 32-byte write on the memory side, 4-7 GB per C2 step in the PMC pass next to 0.4 GB of
                                     // output — off the fabric: 16 KB more LDS per workgroup cost the fourth resident workgroup per CU, 88.6 -> 100.8 ms.)
                                     // (And a trimmed bs_defer — the trigger's ballot reused, the slot index from mbcnt on top of the fill count, a clamp
                                     // instead of the bounds branch: in the first-row loop alone 1385 -> 1244 clocks per step there, but 1387 -> 1432 in
                                     // the later rows it did not touch, 88.6 -> 91.5 ms; in the later-row loop 12 spill accesses per trigger, 107 ms.
                                     // The kernel sits at its 128-VGPR budget; whatever adds a live value to the trigger path pays in spills.)
constexpr int MH_LUT_WORDS = 512;   // k1 / k2 block-mix tables of the key hash (the murmur3_x86_32 part of the tables is not needed here)

// One xorshift64 step of the 32 chains.  With A = x ^ (x << 21) the result is C = (I + L^4)(I + R^35) A; plane by plane:
//   C[b] = A[b] ^ A[b-4]                      b = 33..63
//   C[b] = A[b] ^ A[b-4] ^ A[b+31]            b = 29..32
//   C[b] = A[b] ^ A[b-4] ^ C[b+35]            b =  4..28   (A[b+35] ^ A[b+31] is C[b+35])
//   C[b] = A[b] ^ A[b+35]                     b =  0..3
// = 43 + 64 = 107 full-rate ops (v_xor_b32 / three-input v_bitop3_b32) instead of the 132 two-input xors of the three
// shifts done one after the other.  Every plane is updated IN PLACE: "C[b] may overwrite A[b] once the other readers of A[b]
// are done" has one cycle through all 64 planes, which a copy of plane 35 breaks; the order below is the topological sort
// (tools/gen_bs_step.py derives it and checks it against the 64-bit step).  Eight temporaries and, at the back edge of the
// slot loop, four v_mov_b32 per step are gone with it.
__device__ __forceinline__ uint32_t bs_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
#ifndef MH_BS_STEP
#define MH_BS_STEP 92
#endif
#if MH_BS_STEP == 92
// 92 operations instead of 107.  With A[b] = x[b] ^ x[b-21] the outputs are C[b] = A[b] ^ A[b-4] (b >= 33: a three-input xor with one input
// free), C[b] = A[b] ^ A[b-4] ^ A[b+31] (29..32), C[b] = A[b] ^ A[b-4] ^ C[b+35] (4..28), C[b] = A[b] ^ A[b+35] (0..3: one free).  An A[b]
// whose readers all have a free input is never formed — the reader takes x[b] and x[b-21] itself: every other A along each chain b, b+4, ...
// inside 33..59, fifteen of the forty-three.  tools/gen_bs_step92.py derives the list, finds an order that works IN PLACE (five A's are
// written to temporaries instead of their planes, nothing is copied) and checks it against the 64-bit step.
__device__ __forceinline__ void bs_step(uint32_t (&P)[64]) {
  // 92 operations (61 three-input), in place, five temporaries and no copy: tools/gen_bs_step92.py
  P[63] ^= P[42];
  const uint32_t t62 = P[62] ^ P[41];
  P[62] = bs_xor3(t62, P[58], P[37]);
  P[61] ^= P[40];
  P[60] ^= P[39];
  P[56] ^= P[35];
  P[55] ^= P[34];
  P[54] ^= P[33];
  P[58] = bs_xor3(P[58], P[37], P[54]);
  P[54] = bs_xor3(P[54], P[50], P[29]);
  const uint32_t t53 = P[53] ^ P[32];
  P[53] = bs_xor3(t53, P[49], P[28]);
  P[48] ^= P[27];
  P[47] ^= P[26];
  P[46] ^= P[25];
  P[50] = bs_xor3(P[50], P[29], P[46]);
  P[46] = bs_xor3(P[46], P[42], P[21]);
  P[45] ^= P[24];
  P[49] = bs_xor3(P[49], P[28], P[45]);
  P[45] = bs_xor3(P[45], P[41], P[20]);
  P[40] ^= P[19];
  P[39] ^= P[18];
  const uint32_t t37 = P[37] ^ P[16];
  P[41] = bs_xor3(P[41], P[20], t37);
  P[37] = bs_xor3(t37, P[33], P[12]);
  const uint32_t t32 = P[32] ^ P[11];
  P[29] ^= P[8];
  P[33] = bs_xor3(P[33], P[12], P[29]);
  P[28] ^= P[7];
  P[32] = bs_xor3(t32, P[28], P[63]);
  P[63] = bs_xor3(P[63], P[59], P[38]);
  P[59] = bs_xor3(P[59], P[38], P[55]);
  P[55] = bs_xor3(P[55], P[51], P[30]);
  P[51] = bs_xor3(P[51], P[30], P[47]);
  P[47] = bs_xor3(P[47], P[43], P[22]);
  P[43] = bs_xor3(P[43], P[22], P[39]);
  P[39] = bs_xor3(P[39], P[35], P[14]);
  const uint32_t t38 = P[38] ^ P[17];
  P[42] = bs_xor3(P[42], P[21], t38);
  P[38] = bs_xor3(t38, P[34], P[13]);
  P[30] ^= P[9];
  P[34] = bs_xor3(P[34], P[13], P[30]);
  P[27] ^= P[6];
  P[26] ^= P[5];
  P[30] = bs_xor3(P[30], P[26], P[61]);
  P[61] = bs_xor3(P[61], P[57], P[36]);
  P[57] = bs_xor3(P[57], P[36], t53);
  P[25] ^= P[4];
  P[29] = bs_xor3(P[29], P[25], P[60]);
  P[60] ^= P[56];
  P[56] = bs_xor3(P[56], P[52], P[31]);
  P[52] = bs_xor3(P[52], P[31], P[48]);
  P[48] = bs_xor3(P[48], P[44], P[23]);
  P[44] = bs_xor3(P[44], P[23], P[40]);
  P[40] = bs_xor3(P[40], P[36], P[15]);
  P[31] ^= P[10];
  P[24] ^= P[3];
  P[28] = bs_xor3(P[28], P[24], P[63]);
  P[24] = bs_xor3(P[24], P[20], P[59]);
  P[23] ^= P[2];
  P[22] ^= P[1];
  P[26] = bs_xor3(P[26], P[22], P[61]);
  P[22] = bs_xor3(P[22], P[18], P[57]);
  P[21] ^= P[0];
  P[25] = bs_xor3(P[25], P[21], P[60]);
  P[21] = bs_xor3(P[21], P[17], P[56]);
  P[20] = bs_xor3(P[20], P[16], P[55]);
  P[18] = bs_xor3(P[18], P[14], P[53]);
  P[17] = bs_xor3(P[17], P[13], P[52]);
  P[16] = bs_xor3(P[16], P[12], P[51]);
  P[13] = bs_xor3(P[13], P[9], P[48]);
  P[12] = bs_xor3(P[12], P[8], P[47]);
  P[9] = bs_xor3(P[9], P[5], P[44]);
  P[8] = bs_xor3(P[8], P[4], P[43]);
  P[5] = bs_xor3(P[5], P[1], P[40]);
  P[4] = bs_xor3(P[4], P[0], P[39]);
  P[1] = bs_xor3(P[1], P[36], P[15]);
  P[36] = bs_xor3(P[36], P[15], t32);
  P[0] = bs_xor3(P[0], P[35], P[14]);
  P[35] = bs_xor3(P[35], P[14], P[31]);
  P[31] = bs_xor3(P[31], P[27], t62);
  P[27] = bs_xor3(P[27], P[23], P[62]);
  P[23] = bs_xor3(P[23], P[19], P[58]);
  P[19] = bs_xor3(P[19], P[15], P[54]);
  P[15] = bs_xor3(P[15], P[11], P[50]);
  P[14] = bs_xor3(P[14], P[10], P[49]);
  P[11] = bs_xor3(P[11], P[7], P[46]);
  P[10] = bs_xor3(P[10], P[6], P[45]);
  P[7] = bs_xor3(P[7], P[3], P[42]);
  P[6] = bs_xor3(P[6], P[2], P[41]);
  P[3] ^= t38;
  P[2] ^= t37;
}
#else
__device__ __forceinline__ void bs_step(uint32_t (&P)[64]) {
#pragma unroll
  for (int b = 63; b >= 21; b--) P[b] ^= P[b - 21];                       // A
  const uint32_t T = P[35];
  P[35] ^= P[31];
  P[31] = bs_xor3(P[31], P[27], P[62]);
  P[62] ^= P[58];
  P[58] ^= P[54];
  P[54] ^= P[50];
  P[50] ^= P[46];
  P[46] ^= P[42];
  P[42] ^= P[38];
  P[27] = bs_xor3(P[27], P[23], P[62]);
  P[23] = bs_xor3(P[23], P[19], P[58]);
  P[19] = bs_xor3(P[19], P[15], P[54]);
  P[15] = bs_xor3(P[15], P[11], P[50]);
  P[11] = bs_xor3(P[11], P[7], P[46]);
  P[7] = bs_xor3(P[7], P[3], P[42]);
  P[3] ^= P[38];
  P[38] ^= P[34];
  P[34] ^= P[30];
  P[30] = bs_xor3(P[30], P[26], P[61]);
  P[61] ^= P[57];
  P[57] ^= P[53];
  P[53] ^= P[49];
  P[49] ^= P[45];
  P[45] ^= P[41];
  P[41] ^= P[37];
  P[26] = bs_xor3(P[26], P[22], P[61]);
  P[22] = bs_xor3(P[22], P[18], P[57]);
  P[18] = bs_xor3(P[18], P[14], P[53]);
  P[14] = bs_xor3(P[14], P[10], P[49]);
  P[10] = bs_xor3(P[10], P[6], P[45]);
  P[6] = bs_xor3(P[6], P[2], P[41]);
  P[2] ^= P[37];
  P[37] ^= P[33];
  P[33] ^= P[29];
  P[29] = bs_xor3(P[29], P[25], P[60]);
  P[60] ^= P[56];
  P[56] ^= P[52];
  P[52] ^= P[48];
  P[48] ^= P[44];
  P[44] ^= P[40];
  P[40] ^= P[36];
  P[25] = bs_xor3(P[25], P[21], P[60]);
  P[21] = bs_xor3(P[21], P[17], P[56]);
  P[17] = bs_xor3(P[17], P[13], P[52]);
  P[13] = bs_xor3(P[13], P[9], P[48]);
  P[9] = bs_xor3(P[9], P[5], P[44]);
  P[5] = bs_xor3(P[5], P[1], P[40]);
  P[1] ^= P[36];
  P[36] ^= P[32];
  P[32] = bs_xor3(P[32], P[28], P[63]);
  P[63] ^= P[59];
  P[59] ^= P[55];
  P[55] ^= P[51];
  P[51] ^= P[47];
  P[47] ^= P[43];
  P[43] ^= P[39];
  P[39] ^= T;
  P[28] = bs_xor3(P[28], P[24], P[63]);
  P[24] = bs_xor3(P[24], P[20], P[59]);
  P[20] = bs_xor3(P[20], P[16], P[55]);
  P[16] = bs_xor3(P[16], P[12], P[51]);
  P[12] = bs_xor3(P[12], P[8], P[47]);
  P[8] = bs_xor3(P[8], P[4], P[43]);
  P[4] = bs_xor3(P[4], P[0], P[39]);
  P[0] ^= T;
}
#endif

// filter depth of a slot whose minimum has the high dword bhs: -1 = no negative minimum yet (every active chain is a candidate),
// else the leading zero magnitude bits of the minimum, capped at BS_ZMAX
#ifndef MH_ZDEEP
#define MH_ZDEEP 20
#endif
constexpr int BS_ZDEEP = MH_ZDEEP;   // planes the second look of a triggered step goes down to (>= BS_ZMAX)
__device__ __forceinline__ int bs_depth(int32_t bhs) {
  if (bhs >= 0) return -1;
  const uint32_t mag = (uint32_t)bhs & 0x7fffffffu;
  const int z = mag ? (__builtin_clz(mag) - 1) : 31;
  return z > BS_ZDEEP ? BS_ZDEEP : z;
}

// The candidate filter of a step.  A slot's candidates are the chains that are negative with the magnitude planes 62 .. 63-z
// all zero (z = the depth of the slot's minimum).  The depths of 64 slots are held one per lane as MASK words
// m = (2 << z) - 1 (bit 0: sign plane, bit b: magnitude plane 63-b; 0 = no negative minimum yet); a slot's word comes out
// with one v_readlane and its bits become scalar enable words (s_bfe_i32).
//
// Hot form (every step, bs_hot_filter): ten full-rate three-input ops and no branch —
//   planes 1..8 OR-ed unconditionally (once a slot has a minimum of >= 2048 chain values its depth is practically always >= 8),
//   planes 9..12 through their enable words ((plane & enable) | acc),
//   and a slot whose depth is below 8 (incl. "no minimum yet") is routed THROUGH THE TRIGGER: its `keep` word is 0, every chain
//   looks like a candidate, and the triggered path recomputes the exact masked filter for it.
// Every op is v_bitop3_b32: v_or3_b32 / v_and_or_b32, which round 2 used here, issue at half rate on gfx950
// (profiles/r01_valu_microbench.txt: 3.5e13 against 5.8e13 lane-ops/s), and so did the depth handling's ten scalar ops per slot
// (tools/bs_loop_probe.hip: filter + depth cost 40 % of the 107-op step they guard; in this form 9 %).
constexpr uint32_t BS_TT_OR3 = 0xFE, BS_TT_ANDOR = 0xEA /* (a & b) | c */, BS_TT_ORAND = 0xA8 /* (a | b) & c */;
__device__ __forceinline__ uint32_t bs_mask_of(int32_t bhs) {
  const int z = bs_depth(bhs);
  return z < 0 ? 0u : ((2u << z) - 1u);
}
__device__ __forceinline__ uint32_t bs_sbit(uint32_t m, int b) {   // bit b of a wave-uniform word as a scalar mask word
  uint32_t r = (uint32_t)((int)(m << (31 - b)) >> 31);
  asm volatile("" : "+s"(r));   // keep it a scalar mask word (otherwise it is turned back into 64-bit select conditions)
  return r;
}
static_assert(BS_ZMAX == 12, "bs_hot_filter is written for a first look of 12 planes");
__device__ __forceinline__ uint32_t bs_hot_filter(const uint32_t (&P)[64], uint32_t nACT, uint32_t sm) {
  const uint32_t keep = bs_sbit(sm, 8), e9 = bs_sbit(sm, 9), e10 = bs_sbit(sm, 10), e11 = bs_sbit(sm, 11), e12 = bs_sbit(sm, 12);
  uint32_t n0 = __builtin_amdgcn_bitop3_b32(P[62], P[61], P[60], BS_TT_OR3);
  const uint32_t t1 = __builtin_amdgcn_bitop3_b32(P[59], P[58], P[57], BS_TT_OR3);
  n0 = __builtin_amdgcn_bitop3_b32(n0, P[56], P[55], BS_TT_OR3);
  uint32_t n1 = __builtin_amdgcn_bitop3_b32(t1, P[63], nACT, 0xFB);          // t1 | ~sign | inactive
  n0 = __builtin_amdgcn_bitop3_b32(P[54], e9, n0, BS_TT_ANDOR);
  n1 = __builtin_amdgcn_bitop3_b32(P[53], e10, n1, BS_TT_ANDOR);
  n0 = __builtin_amdgcn_bitop3_b32(P[52], e11, n0, BS_TT_ANDOR);
  n1 = __builtin_amdgcn_bitop3_b32(P[51], e12, n1, BS_TT_ANDOR);
  return __builtin_amdgcn_bitop3_b32(n0, n1, keep, BS_TT_ORAND);
}
// Triggered steps only.  Exact first look for a shallow slot (every plane through its enable bit; sm = 0: every active chain
// is a candidate), then the second look: the planes down to the slot's real depth (<= BS_ZDEEP) are checked before anything is
// queued, which removes nearly all false candidates of the 12-plane first look — and with them most of the queueing and draining.
__device__ __forceinline__ uint32_t bs_cold_filter(const uint32_t (&P)[64], uint32_t nacc, uint32_t nACT, uint32_t sm) {
  // (every plane is folded in with ONE three-input op, (plane & enable) | acc, so that no temporaries pile up next to the 64 planes)
  if (!(sm & 0x100u)) {
    nacc = __builtin_amdgcn_bitop3_b32(P[63], bs_sbit(sm, 0), nACT, 0xBA);   // (~sign & enable) | inactive   (table index = 4a + 2b + c)
#pragma unroll
    for (int b = 1; b <= BS_ZMAX; b++) nacc = __builtin_amdgcn_bitop3_b32(P[63 - b], bs_sbit(sm, b), nacc, BS_TT_ANDOR);
  }
  if (sm >> (BS_ZMAX + 1)) {
#pragma unroll
    for (int b = BS_ZMAX + 1; b <= BS_ZDEEP; b++) nacc = __builtin_amdgcn_bitop3_b32(P[63 - b], bs_sbit(sm, b), nacc, BS_TT_ANDOR);
  }
  return nacc;
}

// First bit-sliced row of a strand (no slot minimum exists yet): bit-serial narrowing towards the row's arg-min.
// Walking the planes from the sign bit down, the candidate set is narrowed to the chains that have the "smaller" bit
// whenever at least one does (wave-uniform decision).  The row's minimum always survives; after BS_ARGMIN_PLANES planes
// about 2048 / 2^BS_ARGMIN_PLANES other chains still share its prefix, and the deferred exact update sorts those out,
// so the walk stops there instead of testing for a single survivor.
constexpr int BS_ARGMIN_PLANES = 11;   // measured 9 / 10 / 11 / 12 / 14 planes: 86.9 / 85.9 / 85.8 / 86.9 / 87.2 ms at C2
__device__ __forceinline__ uint32_t bs_argmin_walk(const uint32_t (&P)[64], uint32_t ACT) {
  uint32_t cand = ACT & P[63];                 // negative values first (signed compare)
  if (!__any(cand != 0u)) cand = ACT;
#pragma unroll
  for (int b = 62; b > 62 - BS_ARGMIN_PLANES; b--) {
    const uint32_t m = cand & ~P[b];
    if (__any(m != 0u)) cand = m;
  }
  return cand;
}
// the shortcut alone (the callers that test its ballot themselves)
__device__ __forceinline__ uint32_t bs_argmin_pre(const uint32_t (&P)[64], uint32_t ACT) {
  uint32_t o = __builtin_amdgcn_bitop3_b32(P[62], P[61], P[60], BS_TT_OR3);
  o = __builtin_amdgcn_bitop3_b32(o, P[59], P[58], BS_TT_OR3);
  o = __builtin_amdgcn_bitop3_b32(o, P[57], P[56], BS_TT_OR3);
  o = __builtin_amdgcn_bitop3_b32(o, P[55], P[54], BS_TT_OR3);
  return __builtin_amdgcn_bitop3_b32(ACT, P[63], o, 0x40);   // ACT & sign & ~o
}
__device__ __forceinline__ uint32_t bs_argmin(const uint32_t (&P)[64], uint32_t ACT) {
  // shortcut: the chains that are negative with nine leading zero magnitude bits (two are expected among 2048, and if there is
  // any the minimum is among them); the plane-by-plane walk below only runs for the ~14 % of slots without one.  Measured with
  // 8 / 9 / 10 planes: 83.6 / 77.8 / 79.7 ms at C2 (more planes = more walks, fewer = more candidates to drain).
  // (three-input ORs as v_bitop3_b32: the compiler's v_or3_b32 is half rate on gfx950)
  uint32_t o = __builtin_amdgcn_bitop3_b32(P[62], P[61], P[60], BS_TT_OR3);
  o = __builtin_amdgcn_bitop3_b32(o, P[59], P[58], BS_TT_OR3);
  o = __builtin_amdgcn_bitop3_b32(o, P[57], P[56], BS_TT_OR3);
  o = __builtin_amdgcn_bitop3_b32(o, P[55], P[54], BS_TT_OR3);
  const uint32_t pre = __builtin_amdgcn_bitop3_b32(ACT, P[63], o, 0x40);   // ACT & sign & ~o   (table index = 4a + 2b + c)
  if (__any(pre != 0u)) return pre;
  return bs_argmin_walk(P, ACT);
}
// Deferred candidates: pulling a candidate's 64-bit value out of the planes would cost 64 v_readlane + ~250 scalar ops
// (~1600 issue cycles).  Instead a trigger only appends (slot, sub-step, lane, chain) words to the wave's queue (global memory,
// fire-and-forget stores); slots are independent within a row, so the queue is drained at the end of the row, 64 entries at a
// time, one per lane: each lane re-derives its chain value from the key (GF(2) jump-ahead tables for the multiple of 4 steps
// + <= 3 single steps), then ds_min_rtn_i64 lowers the slot minimum and the lane that ends up owning the minimum records its
// k-mer position.  Queue entry (32 bits): bits 0-4 chain j of the lane, 5-10 lane, 11-16 sub-step c, 17-29 slot s.  Chain value
// = (s w + c + 1) steps from the key.  A row that overflows the queue (qn > BS_QCAP at the drain; never seen) makes the strand
// fall back to the per-chain rows.
#define MHAP_TICK() (PROF ? (unsigned long long)clock64() : 0ULL)
template <bool PROF = false>
__device__ __forceinline__ void bs_flush(int64_t* best, int32_t* bpos, const uint32_t* q, int& qn_ref, int qcap, int rb, int w, const KeySrc& ks,
                                         const uint64_t* __restrict__ jump, int na, int lane, unsigned long long* tf = nullptr) {
  const unsigned long long t0 = MHAP_TICK();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the queue stores of this wave have left it (vmcnt(0)) before its lanes read each other's entries
  __builtin_amdgcn_wave_barrier();
  const int qn = qn_ref < qcap ? qn_ref : qcap;
  for (int b0 = 0; b0 < qn; b0 += 64) {
    const bool valid = b0 + lane < qn;
    const uint32_t e = valid ? q[b0 + lane] : 0u;
    const int j = (int)(e & 31u), l = (int)((e >> 5) & 63u), c = (int)((e >> 11) & 63u), s = (int)(e >> 17);
    const int nsteps = s * w + c + 1;
    int a = nsteps >> XS_JUMP_LOG2;
    const int r = valid ? (nsteps & ((1 << XS_JUMP_LOG2) - 1)) : 0;
    int qa = 0;                                  // weighted chains run past the fine tables: one coarse jump of qa * na tables first
    if (a > na) { qa = (a - 1) / na; a -= qa * na; }
    const int pos = valid ? ks_pos(ks, rb + j * 64 + l) : 0;
    uint64_t x = valid ? ks_key(ks, pos) : 0ULL;
    if (valid && qa > 0) {
      const uint32_t tb = (uint32_t)(na + qa - 1) * 2048u;
      uint64_t y = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) y ^= jump[tb + (uint32_t)(i * 256) + ((uint32_t)(x >> (8 * i)) & 255u)];
      x = y;
    }
    if (valid && a > 0) {
      const uint32_t tb = (uint32_t)(a - 1) * 2048u;
      uint64_t y = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) y ^= jump[tb + (uint32_t)(i * 256) + ((uint32_t)(x >> (8 * i)) & 255u)];
      x = y;
    }
#pragma unroll
    for (int t = 0; t < (1 << XS_JUMP_LOG2) - 1; t++) {
      const uint64_t nx = xorshift_step(x);
      x = (t < r) ? nx : x;
    }
    // exact update, all lanes at once: the lane whose value is the slot's final minimum and that strictly undercut
    // what it saw owns the slot (chain values of distinct k-mers are distinct)
    long long old = INT64_MAX;
    if (valid) old = atomicMin((long long*)&best[s], (long long)x);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (valid && (long long)x < old && best[s] == (int64_t)x) bpos[s] = pos;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  qn_ref = 0;
  if (PROF) { tf[0] += MHAP_TICK() - t0; tf[1] += (unsigned long long)((qn + 63) >> 6); }
}

// append this trigger's candidates: one entry per candidate chain (a lane's mask nearly always has a single bit; the loop
// runs while any lane has bits left).  The fill count lives in a wave-uniform register and queue slots are handed out with
// ballot + mbcnt (no atomic, no read-back).  Entries beyond the capacity are dropped; the count keeps running, so the drain sees it.
__device__ __forceinline__ void bs_defer(uint32_t* __restrict__ q, int& qn, int qcap, int s, int c, uint32_t cand) {
  // Both MinHash kernels sit at their VGPR limit in the slot loop, and whatever this path keeps alive is spilled: the lane id, hoisted
  // out of the loop by the compiler (the mbcnt builtins are pure), and the queue's base as a VGPR pair were reloaded from scratch —
  // a memory round trip and an s_waitcnt vmcnt(0) each — in front of every queue store (later rows of the weight-1 kernel: 1387 ->
  // 1162 clocks per step without them).  So: the lane id from volatile asm right where the entry is put together, the base as a
  // scalar pair (the callers' readfirstlane), and nothing computed ahead of the loop.
  const uint32_t shead = ((uint32_t)s << 17) | ((uint32_t)c << 11);   // (wave-uniform: scalar)
  unsigned long long m = __ballot(cand != 0u);
  do {
    if (cand) {
      const int idx = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (idx < qcap) {
        uint32_t lane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(lane));
        ((__attribute__((address_space(1))) uint32_t*)q)[idx] = shead | (lane << 5) | (uint32_t)__builtin_ctz(cand);   // (global_store with the scalar base)
      }
      cand &= cand - 1u;
    }
    qn += __popcll(m);
    m = __ballot(cand != 0u);
  } while (m);
}

// One per-chain row: U chains per lane with their own weights (0 = idle lane slot, x = 0 is a fixed point of the chain).
// A chain of weight w takes w steps per slot; the minimum over those steps is what the slot sees (:137-152).
template <int U>
__device__ __forceinline__ void perchain_row(int64_t* best, int32_t* bpos, const int32_t* besthi, int H, uint64_t (&x)[U], const int (&pv)[U],
                                             const uint32_t (&wt)[U], int lane) {
  bool act[U];
  uint32_t wmax = 0;
#pragma unroll
  for (int u = 0; u < U; u++) { act[u] = wt[u] > 0u; wmax = wt[u] > wmax ? wt[u] : wmax; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(wmax, off); wmax = o > wmax ? o : wmax; }
  if (wmax == 0u) return;
  int32_t bh_next = besthi[1];
  if (wmax == 1u) {
    for (int s = 0; s < H; s++) {
      const int32_t bh = bh_next;
      bh_next = besthi[2 * (s + 1 < H ? s + 1 : s) + 1];   // prefetch the next slot's threshold (uniform ds_read)
      bool hit = false;
#pragma unroll
      for (int u = 0; u < U; u++) {
        x[u] = xorshift_step(x[u]);
        hit |= ((int32_t)(x[u] >> 32) <= bh);
      }
      if (__any(hit)) {
        int64_t xs[U];
#pragma unroll
        for (int u = 0; u < U; u++) xs[u] = (int64_t)x[u];
        minhash_update<U>(best, bpos, s, xs, pv, act, lane);
      }
    }
    return;
  }
  for (int s = 0; s < H; s++) {
    const int32_t bh = bh_next;
    bh_next = besthi[2 * (s + 1 < H ? s + 1 : s) + 1];
    int64_t mn[U];
#pragma unroll
    for (int u = 0; u < U; u++) mn[u] = INT64_MAX;
    for (uint32_t c = 0; c < wmax; c++) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (c < wt[u]) { x[u] = xorshift_step(x[u]); mn[u] = (int64_t)x[u] < mn[u] ? (int64_t)x[u] : mn[u]; }
      }
    }
    bool hit = false;
#pragma unroll
    for (int u = 0; u < U; u++) hit |= act[u] && ((int32_t)(mn[u] >> 32) <= bh);
    if (__any(hit)) minhash_update<U>(best, bpos, s, mn, pv, act, lane);
  }
}

// amdgpu_waves_per_eu(4, 4): the slot loop needs 64 plane registers + the filter ladder's scalar conditions; left alone the
// allocator keeps loop-invariant table addresses of the (cold) drain in VGPRs and ends at ~150 (3 waves per SIMD).  Capped at
// 128 it spills those ~25 dwords to scratch and reloads them inside the drain only; the slot loop has no scratch access.
// WEIGHTED = false: the launch takes the strands whose k-mers all have weight 1 (StrandInfo.mode == 1: nearly every strand
// without -f) — one step per slot, no class list, no inner step loop (control flow inside the slot loop is what this kernel
// pays most for) — plus the strands without a sketch (their zero rows / status).  WEIGHTED = true: a second launch takes
// the other strands (class lists, a common weight > 1).  kmer_weight_kernel sorts the strands into the two work lists.
// (amdgpu_waves_per_eu: the weighted instantiation needs 129 VGPRs left alone — one over four waves per SIMD.)
// ---- depth classes of a slot (round 5, the weight-1 kernel): the filter a step needs is selected as CODE by the slot's class, not through
// scalar enable operands (an SGPR source makes a v_bitop3 1.5 times as dear: profiles/r05_issue_probe.txt).  (The same selection in the
// general kernel — seven copies of its step loop, one per class, switched per slot — ran 52 % SLOWER on the C5 slice, 181 against 119 ms:
// the copies with their trigger paths no longer fit the instruction cache the slot loop lives in.) ----
constexpr int W1_CLS_BASE = 8;            // class c >= 1 filters down to depth W1_CLS_BASE + c - 1; class 0 = the exact masked filter
constexpr int W1_CLS_MAX = 6;             // depth 13: 2048 x 2^-14 = one false candidate per eight steps at worst

// class of a slot from the high dword of its minimum
__device__ __forceinline__ int w1_class_of(int32_t bhs) {
  const int z = bs_depth(bhs);
  if (z < W1_CLS_BASE) return 0;
  const int c = z - W1_CLS_BASE + 1;
  return c > W1_CLS_MAX ? W1_CLS_MAX : c;
}

// the planes 1..8 of the magnitude and the sign: a chain that is not negative or has a one among them is out (bit = 1)
__device__ __forceinline__ uint32_t w1_base_filter(const uint32_t (&P)[64]) {
  const uint32_t t0 = __builtin_amdgcn_bitop3_b32(P[62], P[61], P[60], BS_TT_OR3);
  const uint32_t t1 = __builtin_amdgcn_bitop3_b32(P[59], P[58], P[57], BS_TT_OR3);
  const uint32_t t2 = __builtin_amdgcn_bitop3_b32(P[56], P[55], P[63], 0xFD);     // a | b | ~c   (table index = 4a + 2b + c)
  return __builtin_amdgcn_bitop3_b32(t0, t1, t2, BS_TT_OR3);
}

#ifndef MH_WAVES_EU
#define MH_WAVES_EU 4
#endif
template <int U, bool BITSLICED, bool WEIGHTED, bool PROF = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MH_WAVES_EU, 8))) void minhash_kernel(const ReadDesc* __restrict__ descs, int64_t nstrands,
                                                      const int64_t* __restrict__ keys, const uint32_t* __restrict__ wts,
                                                      const uint32_t* __restrict__ perms, const StrandInfo* __restrict__ info,
                                                      const uint8_t* __restrict__ store, const uint64_t* __restrict__ luts,
                                                      int k, int k2, int H,
                                                      unsigned long long* __restrict__ counter, int32_t* __restrict__ out_rows,
                                                      int64_t out_stride, int32_t* __restrict__ out_status, int64_t status_stride,
                                                      const uint64_t* __restrict__ jump, int jump_na, const int32_t* __restrict__ slist,
                                                      const unsigned long long* __restrict__ slist_count,
                                                      uint32_t* __restrict__ qbuf, int qcap, unsigned long long* __restrict__ prof = nullptr, int split_arg = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform: scalar LDS / queue bases)
  // PROF: wave-clock attribution {strand total, first-row total, first-row argmin, first-row defer, later-row defer, key load+transpose,
  // flush (nested in the defers / row ends), flush batches, strands}
  unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, tf[2] = {0, 0}, nst = 0;
  uint64_t* lut = (uint64_t*)smem;                                   // shared by the workgroup's waves
  for (int i = threadIdx.x; i < MH_LUT_WORDS; i += blockDim.x) lut[i] = luts[i];
  __syncthreads();
  const size_t per_wave = (size_t)H * 12 + 8;
  char* wbase = smem + (size_t)MH_LUT_WORDS * 8 + (size_t)wv * ((per_wave + 15) & ~(size_t)15);
  int64_t* best = (int64_t*)wbase;
  int32_t* bpos = (int32_t*)(best + H);
  uint32_t* bsq = qbuf + ((size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)wv) * (size_t)qcap;   // this wave's deferred-candidate queue (global memory)
  {   // the window's base as a scalar pair: the queue stores then take it as their SGPR base + a 32-bit lane offset (as a VGPR pair it
      // was spilled, and reloaded from scratch in front of every queue store)
    const unsigned long long qa = (unsigned long long)(uintptr_t)bsq;
    const uint32_t qlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)qa), qhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(qa >> 32));
    bsq = (uint32_t*)(uintptr_t)(((unsigned long long)qhi << 32) | qlo);
  }
  const int32_t* besthi = (const int32_t*)best;   // high dword of best[s] = the hot loops' threshold
  // SPLIT (a launch of a few strands only — a small job's handful of strands with repeated k-mers, which would otherwise hold the add for one
  // strand's time on one wave, 0.63 ms of C1's 1.2): a work item is a strand for the WORKGROUP, its rows — bit-sliced and per-chain, numbered
  // in one wave-uniform sequence — go round the four waves, each wave keeps minima of its own and the four tables meet in LDS at the end.
  const bool split = WEIGHTED && BITSLICED && !PROF && split_arg != 0 && blockDim.x == 256;
  const size_t wave_stride = (per_wave + 15) & ~(size_t)15;
  unsigned long long* wgslot = (unsigned long long*)(smem + (size_t)MH_LUT_WORDS * 8 + 4 * wave_stride);   // (SPLIT: the workgroup's work item)
  for (;;) {
    unsigned long long tk = 0;
    if (!split) { if (lane == 0) tk = atomicAdd(counter, 1ULL); }
    else {
      __syncthreads();   // (the previous strand's tables have been read)
      if (threadIdx.x == 0) *wgslot = atomicAdd(counter, 1ULL);
      __syncthreads();
      tk = *wgslot;
    }
    // readfirstlane: the strand index — and with it every descriptor field, pointer, count and loop bound below — is provably
    // wave-uniform, so the strand's control flow compiles to scalar branches instead of EXEC-mask bookkeeping
    long long sidx = (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tk >> 32)) << 32) |
                                 (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tk));
    if (sidx >= (long long)*slist_count) break;
    sidx = slist[sidx];   // this launch's strands, in the order kmer_weight_kernel finished them (longest reads first when lengths vary)
    const ReadDesc rd = descs[sidx >> 1];
    const int rcs = (int)(sidx & 1);
    const int nk = rd.length - k + 1;
    int32_t* orow = out_rows + sidx * out_stride;
    const StrandInfo* sip = info + sidx;
    const int si_valid = sip->valid, si_mode = sip->mode;
    const bool nosketch = strand_skipped(rd, rcs) || nk < 1 || rd.length - k2 + 1 < 1 || !si_valid;
    if (nosketch) {
      // too short (status 2) or ZeroNGramsFoundException from either sketch (status 1)
      if (!split || wv == 0) {
        for (int s = lane; s < H; s += 64) orow[s] = 0;
        if (lane == 0) out_status[sidx * status_stride] = strand_skipped(rd, rcs) ? 2 : 1;
      }
    } else {   // (no `continue` above: the strand loop keeps a single back edge)
    const bool listed = WEIGHTED && si_mode == 0;
    KeySrc ks;
    ks.kp = (rd.flags & MHAP_RD_MAT) ? keys + rd.key_off + (rcs ? rd.key_stride : 0) : nullptr;
    ks.W = (const uint32_t*)(store + rd.base_off); ks.nd = (((rd.length + 3) >> 2) + 3) >> 2; ks.L = rd.length; ks.rcs = rcs;
    ks.lut = lut;
    const uint32_t* plist = perms + rd.w_off + (rcs ? rd.key_stride : 0);
    const uint32_t* wp = wts + rd.w_off + (rcs ? rd.key_stride : 0);
    for (int s = lane; s < H; s += 64) { best[s] = INT64_MAX; bpos[s] = INT32_MIN; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const unsigned long long ts0 = MHAP_TICK();
    if (!split || wv == 0) nst++;
    int rowi = 0;              // rows of the strand so far (SPLIT: row r is wave r & 3's)
    const int ncls = listed ? BS_WCLASSES : 1;
    const int w_uniform = WEIGHTED ? si_mode : 1;
    // ---- bit-sliced rows: every weight class (or the whole strand at its common weight), 2048 chains per row ----
    bool bs_ok = true;         // false: a row's candidates overflowed the queue -> the strand is redone on the per-chain rows
    if (BITSLICED) {
      int bsqn = 0;            // deferred-candidate queue fill (wave-uniform)
      bool first = true;       // no slot minimum exists yet
      int off = 0;
      for (int cl = 0; cl < ncls; cl++) {
        const int w = listed ? cl + 1 : w_uniform;
        const int count = listed ? sip->cnt[cl] : nk;
        ks.perm = listed ? plist + off : nullptr;
        off += count;
        if (w > BS_WMAX) continue;
        for (int base = 0; count - base >= BS_MINREM; base += 2048) {
          if (split && ((rowi++) & 3) != wv) continue;
          uint32_t P[64];
          uint32_t ACT = 0;
          const unsigned long long tr0 = MHAP_TICK();
#pragma unroll
          for (int j = 0; j < 32; j++) {
            const int i = base + j * 64 + lane;
            uint64_t key = 0;
            if (i < count) { key = ks_key(ks, ks_pos(ks, i)); ACT |= 1u << j; }
            P[j] = (uint32_t)key;
            P[32 + j] = (uint32_t)(key >> 32);
            if ((j & 7) == 7) asm volatile("" ::: "memory");   // 8 keys in flight at a time
          }
          transpose32(*reinterpret_cast<uint32_t(*)[32]>(&P[0]));
          transpose32(*reinterpret_cast<uint32_t(*)[32]>(&P[32]));
          if (PROF) tp[5] += MHAP_TICK() - tr0;
          if (first) {
            // nothing seen yet: every step's exact row arg-min becomes the slot's first entry
            for (int s = 0; s < H; s++)
              for (int c = 0; c < w; c++) {
                bs_step(P);
                const unsigned long long ta = MHAP_TICK();
                const uint32_t cand = bs_argmin(P, ACT);
                const unsigned long long tb = MHAP_TICK();
                bs_defer(bsq, bsqn, qcap, s, c, cand);
                if (PROF) { tp[2] += tb - ta; tp[3] += MHAP_TICK() - tb; }
              }
          } else {
            const uint32_t nACT = ~ACT;
            for (int s0 = 0; s0 < H; s0 += 64) {
              // depth masks of 64 slots, one per lane (a drain in between only makes them conservative)
              const uint32_t vm = bs_mask_of(besthi[2 * (s0 + lane < H ? s0 + lane : H - 1) + 1]);
              const int tn = H - s0 < 64 ? H - s0 : 64;
              for (int t = 0; t < tn; t++) {
                const uint32_t sm = (uint32_t)__builtin_amdgcn_readlane((int)vm, t);
                for (int c = 0; c < (WEIGHTED ? w : 1); c++) {
                  bs_step(P);
                  uint32_t nacc = bs_hot_filter(P, nACT, sm);
                  if (__builtin_expect(__any(nacc != 0xFFFFFFFFu), 0)) {
                    const unsigned long long ta = MHAP_TICK();
                    nacc = bs_cold_filter(P, nacc, nACT, sm);
                    if (__any(nacc != 0xFFFFFFFFu)) bs_defer(bsq, bsqn, qcap, s0 + t, c, ~nacc);
                    if (PROF) tp[4] += MHAP_TICK() - ta;
                  }
                }
              }
            }
          }
          if (bsqn > qcap) bs_ok = false;
          bs_flush<PROF>(best, bpos, bsq, bsqn, qcap, base, w, ks, jump, jump_na, lane, tf);
          if (PROF && first) tp[1] += MHAP_TICK() - tr0;
          first = false;
        }
      }
      if (split) { bs_ok = __syncthreads_or(bs_ok ? 0 : 1) == 0; if (!bs_ok) rowi = 0; }   // (one wave's overflow: the whole strand again, per chain, on all four)
      if (!bs_ok) {
        for (int s = lane; s < H; s += 64) { best[s] = INT64_MAX; bpos[s] = INT32_MIN; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    // ---- per-chain rows: what the bit-sliced rows left of every class + the weights above BS_WCLASSES, U k-mers per lane ----
    {
      uint64_t x[U];
      int pv[U];
      uint32_t wt[U];
#pragma unroll
      for (int u = 0; u < U; u++) { x[u] = 0; pv[u] = 0; wt[u] = 0; }
      int pfill = 0;           // chains placed in the pending row (wave-uniform): chain n sits in lane n & 63, register n >> 6
      int off = 0;
      for (int cl = 0; cl <= ncls; cl++) {
        if (cl == ncls && !listed) break;
        const int w = (cl == ncls) ? 0 : (listed ? cl + 1 : w_uniform);        // 0: weights come from wp[]
        const int count = (cl == ncls) ? sip->cnt[BS_WCLASSES] : (listed ? sip->cnt[cl] : nk);
        ks.perm = listed ? plist + off : nullptr;
        off += count;
        int e0 = 0;
        if (BITSLICED && bs_ok && cl < ncls && w <= BS_WMAX) while (count - e0 >= BS_MINREM) e0 += 2048;   // done above
        while (e0 < count) {
          const int take = (64 * U - pfill) < (count - e0) ? (64 * U - pfill) : (count - e0);
          const bool mine = !split || (rowi & 3) == wv;
#pragma unroll
          for (int u = 0; u < U; u++) {
            const int e = u * 64 + lane - pfill;
            if (mine && e >= 0 && e < take) {
              const int pos = ks_pos(ks, e0 + e);
              pv[u] = pos;
              x[u] = ks_key(ks, pos);
              wt[u] = w > 0 ? (uint32_t)w : wp[pos];
            }
          }
          pfill += take; e0 += take;
          if (pfill == 64 * U) {
            if (mine) perchain_row<U>(best, bpos, besthi, H, x, pv, wt, lane);
#pragma unroll
            for (int u = 0; u < U; u++) { x[u] = 0; wt[u] = 0; }
            pfill = 0;
            rowi++;
          }
        }
      }
      if (pfill > 0 && (!split || (rowi & 3) == wv)) perchain_row<U>(best, bpos, besthi, H, x, pv, wt, lane);
    }
    __builtin_amdgcn_wave_barrier();
    ks.perm = nullptr;
    if (split) __syncthreads();   // the four waves' tables are complete
    for (int s = split ? (int)threadIdx.x : lane; s < H; s += split ? 256 : 64) {
      int32_t p = bpos[s];
      if (split) {   // the smallest of the four minima (equal values are one k-mer: the step is a bijection and the lists hold first occurrences)
        const char* t0 = smem + (size_t)MH_LUT_WORDS * 8;
        int64_t bv = INT64_MAX; p = INT32_MIN;
        for (int w4 = 0; w4 < 4; w4++) {
          const int64_t* b4 = (const int64_t*)(t0 + (size_t)w4 * wave_stride);
          const int32_t pp = ((const int32_t*)(b4 + H))[s];
          if (pp != INT32_MIN && (p == INT32_MIN || b4[s] < bv)) { bv = b4[s]; p = pp; }
        }
      }
      int32_t v = 0;
      if (p != INT32_MIN) {
        const uint64_t key = ks_key(ks, p);
        v = (s & 1) ? (int32_t)(uint32_t)(key >> 32) : (int32_t)(uint32_t)key;   // MinHashSketch.java:147-150
      }
      orow[s] = v;
    }
    if (lane == 0 && (!split || wv == 0)) out_status[sidx * status_stride] = 0;
    if (PROF) tp[0] += MHAP_TICK() - ts0;
    }
  }
  if (lane == 0) atomicAdd(counter + 8 + (WEIGHTED ? 1 : 0), nst);   // strands sketched by this launch (statistics words next to the work counters)
  if (PROF && lane == 0) {
    for (int i = 0; i < 6; i++) atomicAdd(&prof[i], tp[i]);
    atomicAdd(&prof[6], tf[0]); atomicAdd(&prof[7], tf[1]); atomicAdd(&prof[8], nst);
  }
}

// =============================================================================================
// The weight-1 strands' own kernel (every strand of a run without -f whose k-mers do not repeat: all of C2).  Same rows, step
// and filters as minhash_kernel above, built around what round 2's profile showed the slot loop losing: a wave-step cost the
// SIMD 176 ns where the 107-op step alone costs 121-134 (tools/bs_loop_probe.hip).
//   * Only the slot MINIMA are kept (8 B per slot in LDS, no winner positions): the step is a bijection, so the key behind a
//     minimum after n steps is unstep^n(min) — recovered once per strand through inverse jump tables.  Half the LDS, and the
//     minima of different waves merge with a plain 64-bit atomic min, which is what makes the next point possible.
//   * Work items are whole strands while there are plenty, then single ROWS: the last n_tail strands of the list are cut into
//     their 2048-k-mer rows, any wave takes any row (as a strand's first row: no thresholds, the row's arg-min per slot), and the
//     rows' minima meet in a global table (atomic min of the sign-flipped value; minhash_w1_finish_kernel turns it into the
//     output rows).  A persistent launch then ends within one row's time (0.35 ms) on every wave instead of one strand's
//     (1.75 ms), and a launch with fewer strands than waves (one rank's share of a small job, a batch of -q reads) still fills the GPU.
//   * The deferred-candidate queue is in global memory and drained at the end of a row only; nothing of the drain, the key
//     hashing or the per-chain rows of the general kernel is alive in the slot loop.  (MH_W1_WAVES_EU: 5 or 6 waves per SIMD fit the
//     LDS now, but at 96 / 80 registers the allocator spills around the key load and the chip clocks lower: 4 / 5 / 6 waves measured
//     88.1 / 92.6 / 94.6 ms at C2 on one box, round 2's kernel 89.3.)
// =============================================================================================
#ifndef MH_W1_WAVES_EU
#define MH_W1_WAVES_EU 4
#endif
struct W1Args {
  const ReadDesc* descs; const int64_t* keys; const uint8_t* store; const uint64_t* luts; const StrandInfo* info;
  const int32_t* slist;              // the launch's strands: [0, n_whole) whole, [n_whole, n_whole + n_tail) row by row
  long long n_whole, n_tail; int rmax;   // rmax: rows of the longest strand (row items: n_tail x rmax, row-major)
  int k, k2, H;
  unsigned long long* counter;       // work counter
  unsigned long long* stat;          // strands sketched (statistics)
  int32_t* out_rows; long long out_stride; int32_t* out_status; long long status_stride;
  const uint64_t* jump; const uint64_t* unjump; int jump_na;
  uint32_t* qbuf; int qcap;          // the waves' candidate queues: qcap words each (minhash_queue_words)
  unsigned long long* merge;         // n_tail x H words, filled with 0xFF: minima of the row items, sign bit flipped (unsigned order)
  unsigned long long* prof;          // MHAP_MINHASH_PROF: wave-clock sums {key load + transpose, first-row slots, later-row slots, drains, rows, first rows, candidates}
  int stagger;                       // clocks a wave waits before its first item, times its slot number on its SIMD (mod 4): minhash_w1_kernel
};

// the key behind chain value x after n >= 1 steps: inverse tables for the next multiple of 4 (coarse, then fine), then up to 3 steps forward
__device__ __forceinline__ uint64_t w1_key_of(uint64_t x, int n, const uint64_t* __restrict__ unjump) {
  int a = (n + (1 << XS_JUMP_LOG2) - 1) >> XS_JUMP_LOG2;
  const int r = (a << XS_JUMP_LOG2) - n;
  const int qa = a > W1_JUMP_NA ? (a - 1) / W1_JUMP_NA : 0;      // two levels, like the forward set (w1_flush)
  a -= qa * W1_JUMP_NA;
  uint64_t y = x;
  if (qa > 0) {
    const uint32_t tb = (uint32_t)(W1_JUMP_NA + qa - 1) * 2048u;
    uint64_t z = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) z ^= unjump[tb + (uint32_t)(i * 256) + ((uint32_t)(y >> (8 * i)) & 255u)];
    y = z;
  }
  {
    const uint32_t tb = (uint32_t)(a - 1) * 2048u;
    uint64_t z = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) z ^= unjump[tb + (uint32_t)(i * 256) + ((uint32_t)(y >> (8 * i)) & 255u)];
    y = z;
  }
#pragma unroll
  for (int t = 0; t < (1 << XS_JUMP_LOG2) - 1; t++) { const uint64_t ny = xorshift_step(y); y = (t < r) ? ny : y; }
  return y;
}

// ---- round 5: the row loop priced by what its instructions COST (tools/issue_probe.hip, profiles/r05_issue_probe.txt) -----------
// A VALU-dense kernel on this chip runs at a constant number of lane-operations per second per instruction kind, whatever the
// occupancy and whatever clock results (more waves per SIMD = fewer clocks per instruction AND a lower shader clock): what it pays
// is the SUM of its instructions' prices.  In units of one v_xor_b32_e32: a three-VGPR v_bitop3_b32 1.2, ANY VALU instruction with
// an SGPR source 1.7, v_readlane / v_mbcnt / v_ffbl / ds_write 3.2-3.9, a global store 6.6, a scalar instruction 0.65 and more.
// The xorshift step is 114 of those units; the round-3 loop around it cost 19 (filter: five of its ten v_bitop3 took scalar enable
// words, + v_readlane + five s_bfe) and a triggered step 40 more (second ballot, four v_mbcnt, v_ffbl, bounds compare, the
// per-bit loop) plus up to eight scalar-operand v_bitop3 of the second look — on two steps in three.  This version:
//   * the slot's filter depth selects CODE, not operands: the planes 1..8 are OR-ed unconditionally (four three-VGPR ops; inactive
//     chains carry key 0 = chain value 0 for ever = never negative, so no mask of active chains is needed), and a scalar switch on the
//     slot's depth class (one v_readlane per step) adds the planes 9..13 its depth allows with one to three more VGPR-only ops;
//     a slot without a usable minimum (fewer than eight leading zero bits) takes the exact masked filter — rare;
//   * a triggered step appends ONE 8-byte entry per lane that has candidates — its whole 32-chain candidate mask + (slot, lane) —
//     at the position the trigger's own ballot gives it (two v_mbcnt on the scalar fill count); no per-bit loop, no v_ffbl, no
//     second ballot, no bounds branch (the index is clamped to a spare last entry and the running count says "overflow");
//   * the drain takes the lowest chain of every entry and re-queues what is left of a multi-chain mask (one entry in a hundred).
// (The round-3/4 loop this replaced is in the history of this file: commit e70836d keeps both.)
// (8-byte entries in the wave's window of qcap words: wcap = qcap / 2 - 1 of them, entry wcap — the window's last — is the spare one overflowing appends land in)
// append the lanes' candidate masks (cand != 0 somewhere in the wave, m = its ballot)
__device__ __forceinline__ void w1_enqueue(uint2* __restrict__ q, int& qn, int wcap, uint32_t head, uint32_t cand, unsigned long long m) {
  if (cand) {
    uint32_t idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)qn));
    idx = idx < (uint32_t)wcap ? idx : (uint32_t)wcap;
    uint32_t lanehi;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 16, %0" : "=&v"(lanehi));   // (computed here: a value kept across the slot loop is a spill)
    ((__attribute__((address_space(1))) unsigned long long*)q)[idx] = (unsigned long long)cand | ((unsigned long long)(head | lanehi) << 32);
  }
  qn += __popcll(m);
}

// drain: entry = (candidate mask of one lane, slot | lane << 16); chain value recomputed from the key, slot minimum lowered
// Returns false when the re-queued rests of multi-chain masks did not fit the queue: the row's own count fitted (the caller checked it), but a
// rest is appended BEHIND it — with the count close to the capacity (a first row at --num-hashes 1024 queues about 2 048 entries) rests were
// dropped silently; found by the fuzz sweep on the 31-entry variant build (round 5).  The caller redoes the strand exactly.
__device__ __forceinline__ bool w1_flush2(int64_t* best, uint2* __restrict__ q, int qn, int wcap, int rb, const KeySrc& ks,
                                          const uint64_t* __restrict__ jump, int lane) {
  bool fits = true;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the queue stores have left the wave before its lanes read each other's entries
  __builtin_amdgcn_wave_barrier();
  int qend = qn < wcap ? qn : wcap;
  for (int b0 = 0; b0 < qend; b0 += 64) {
    const bool valid = b0 + lane < qend;
    uint32_t rest = 0u, ehi = 0u;
    if (valid) {
      const uint2 e = q[b0 + lane];
      ehi = e.y;
      const int j = __builtin_ctz(e.x), l = (int)(e.y >> 16), s = (int)(e.y & 0xFFFFu);
      rest = e.x & (e.x - 1u);
      const int nsteps = s + 1;
      int a = nsteps >> XS_JUMP_LOG2;
      const int r = nsteps & ((1 << XS_JUMP_LOG2) - 1);
      const int qa = a > W1_JUMP_NA ? (a - 1) / W1_JUMP_NA : 0;      // two levels of tables, as in w1_flush
      a -= qa * W1_JUMP_NA;
      uint64_t x = ks_key(ks, rb + j * 64 + l);
      if (qa > 0) {
        const uint32_t tb = (uint32_t)(W1_JUMP_NA + qa - 1) * 2048u;
        uint64_t y = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) y ^= jump[tb + (uint32_t)(i * 256) + ((uint32_t)(x >> (8 * i)) & 255u)];
        x = y;
      }
      if (a > 0) {
        const uint32_t tb = (uint32_t)(a - 1) * 2048u;
        uint64_t y = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) y ^= jump[tb + (uint32_t)(i * 256) + ((uint32_t)(x >> (8 * i)) & 255u)];
        x = y;
      }
#pragma unroll
      for (int t = 0; t < (1 << XS_JUMP_LOG2) - 1; t++) { const uint64_t nx = xorshift_step(x); x = (t < r) ? nx : x; }
      atomicMin((long long*)&best[s], (long long)x);
    }
    // what is left of a multi-chain mask goes to the end of the queue (rare: two candidates of one step in one lane)
    const unsigned long long m2 = __ballot(rest != 0u);
    if (m2) {
      // (behind the entries this trip covers: an append into the last, partly filled trip would land on lanes that have already run)
      const int base = b0 + 64 > qend ? b0 + 64 : qend;
      if (rest) {
        const uint32_t idx = (uint32_t)base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
        if (idx < (uint32_t)wcap) q[idx] = make_uint2(rest, ehi);
      }
      qend = base + __popcll(m2);
      if (qend > wcap) { qend = wcap; fits = false; }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  return fits;
}

#ifndef MH_W1_CLS
#define MH_W1_CLS 2     // how a step selects its class's filter code: 0 = `switch`, 1 = if-chain in C, 2 = the whole chain in assembly, 3 = classes 3..6 in assembly behind one C branch (fewer scalar instructions than 2 but more TAKEN branches: 1 ms slower)
#endif
// a slot without a usable minimum (fewer than eight leading zero bits, or none at all: every active chain is a candidate): the exact masked
// filter of its depth.  left = k-mers of the strand from this row's first on (chains j with 64 j + lane < left are active)
__device__ __forceinline__ uint32_t w1_exact_filter(const uint32_t (&P)[64], int32_t bhs, int left_row) {
  const uint32_t sm = bs_mask_of(bhs);
  uint32_t ln;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(ln));
  const int left = left_row - (int)ln;
  const int na_ = left <= 0 ? 0 : (left + 63) >> 6;
  const uint32_t act = na_ >= 32 ? 0xFFFFFFFFu : ((1u << na_) - 1u);
  uint32_t n = __builtin_amdgcn_bitop3_b32(P[63], bs_sbit(sm, 0), ~act, 0xBA);   // (~sign & enable) | inactive
#pragma unroll
  for (int b = 1; b < W1_CLS_BASE; b++) n = __builtin_amdgcn_bitop3_b32(P[63 - b], bs_sbit(sm, b), n, BS_TT_ANDOR);
  return n;
}

template <bool PROF>
__device__ __forceinline__ void w1_row2(int64_t* best, uint2* __restrict__ q, int wcap, int rb, int nk, bool first, int H, const KeySrc& ks,
                                        const uint64_t* __restrict__ jump, int lane, bool& ok, unsigned long long* tp) {
  const unsigned long long t0 = MHAP_TICK();
  uint32_t P[64];
  uint32_t ACT = 0;
#pragma unroll
  for (int j = 0; j < 32; j++) {
    const int i = rb + j * 64 + lane;
    uint64_t key = 0;                                      // (an inactive chain stays 0 for ever: never negative, never a candidate)
    if (i < nk) { key = ks_key(ks, i); ACT |= 1u << j; }
    P[j] = (uint32_t)key;
    P[32 + j] = (uint32_t)(key >> 32);
    if ((j & 7) == 7) asm volatile("" ::: "memory");   // 8 keys in flight at a time
  }
  transpose32(*reinterpret_cast<uint32_t(*)[32]>(&P[0]));
  transpose32(*reinterpret_cast<uint32_t(*)[32]>(&P[32]));
  int qn = 0;   // queue fill (wave-uniform)
  const unsigned long long t1 = MHAP_TICK();
  if (first) {
    for (int s = 0; s < H; s++) {
      bs_step(P);
      // the nine-plane shortcut of bs_argmin in line, its ballot serving both the "anyone?" test and the enqueue; the plane walk (one
      // step in seven) is the out-of-line side
      uint32_t cand = bs_argmin_pre(P, ACT);
      unsigned long long m = __ballot(cand != 0u);
      if (__builtin_expect(m == 0ULL, 0)) { cand = bs_argmin_walk(P, ACT); m = __ballot(cand != 0u); }
      w1_enqueue(q, qn, wcap, (uint32_t)s, cand, m);
    }
  } else {
    const int32_t* besthi = (const int32_t*)best;
    for (int s0 = 0; s0 < H; s0 += 64) {
      int vcls;
      {
        uint32_t ln;   // (the lane id where it is used: a value kept across the slot loop is one more register the allocator spills around)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(ln));
        vcls = w1_class_of(besthi[2 * (s0 + (int)ln < H ? s0 + (int)ln : H - 1) + 1]);   // classes of 64 slots, one per lane
      }
      int send = s0 + 64;   // (scalar loop bound: written as a select it became v_med3 and a VALU compare per step)
      if (send > H) send = H;
      send = __builtin_amdgcn_readfirstlane(send);
      for (int sl = s0; sl < send; sl++) {
        const int t = sl - s0;
        const int cls = __builtin_amdgcn_readlane(vcls, t);
        bs_step(P);
        uint32_t n = w1_base_filter(P);
#if MH_W1_CLS == 0
        switch (cls) {
          case 1: break;
          case 2: n |= P[54]; break;
          case 3: n = __builtin_amdgcn_bitop3_b32(n, P[54], P[53], BS_TT_OR3); break;
          case 4: n = __builtin_amdgcn_bitop3_b32(n, P[54], P[53], BS_TT_OR3) | P[52]; break;
          case 5: n = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(n, P[54], P[53], BS_TT_OR3), P[52], P[51], BS_TT_OR3); break;
          case 6: n = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(n, P[54], P[53], BS_TT_OR3), P[52], P[51], BS_TT_OR3) | P[50]; break;
          default: n = w1_exact_filter(P, besthi[2 * sl + 1], nk - rb); break;
        }
#elif MH_W1_CLS == 1
        if (__builtin_expect(cls >= 3, 1)) {
          n = __builtin_amdgcn_bitop3_b32(n, P[54], P[53], BS_TT_OR3);
          if (cls >= 5) {
            n = __builtin_amdgcn_bitop3_b32(n, P[52], P[51], BS_TT_OR3);
            if (cls >= 6) n |= P[50];
          } else if (cls == 4) n |= P[52];
        } else if (cls == 2) n |= P[54];
        else if (__builtin_expect(cls == 0, 0)) n = w1_exact_filter(P, besthi[2 * sl + 1], nk - rb);   // no usable minimum yet: the exact filter of the slot's depth
#elif MH_W1_CLS == 2
        // The class selects code through a short chain of scalar compares, deepest classes first (the common ones from the third row on),
        // written out in assembly: the compiler lowers the equivalent `switch` / if-chain through its control-flow structurizer — a compare
        // tree with "which way did I come" flags in SGPR pairs, 11 to 16 scalar instructions a step (and selects in place of the
        // innermost branches); this is 6 or 7.  (One asm statement, branches local to it; classes 1..6 add the planes 9..13.)
        asm volatile(
            "s_cmp_lt_i32 %[c], 3\n\t"
            "s_cbranch_scc1 1f\n\t"
            "v_bitop3_b32 %[n], %[n], %[p54], %[p53] bitop3:0xfe\n\t"
            "s_cmp_lt_i32 %[c], 5\n\t"
            "s_cbranch_scc1 2f\n\t"
            "v_bitop3_b32 %[n], %[n], %[p52], %[p51] bitop3:0xfe\n\t"
            "s_cmp_lt_i32 %[c], 6\n\t"
            "s_cbranch_scc1 3f\n\t"
            "v_or_b32 %[n], %[n], %[p50]\n\t"
            "s_branch 3f\n"
            "2:\n\t"
            "s_cmp_lt_i32 %[c], 4\n\t"
            "s_cbranch_scc1 3f\n\t"
            "v_or_b32 %[n], %[n], %[p52]\n\t"
            "s_branch 3f\n"
            "1:\n\t"
            "s_cmp_lt_i32 %[c], 2\n\t"
            "s_cbranch_scc1 3f\n\t"
            "v_or_b32 %[n], %[n], %[p54]\n"
            "3:\n"
            : [n] "+v"(n)
            : [c] "s"(cls), [p54] "v"(P[54]), [p53] "v"(P[53]), [p52] "v"(P[52]), [p51] "v"(P[51]), [p50] "v"(P[50])
            : "scc");
        if (__builtin_expect(cls == 0, 0)) n = w1_exact_filter(P, besthi[2 * sl + 1], nk - rb);   // no usable minimum yet: the exact filter of the slot's depth
#else
        // as 2, with the deepest classes (from the third row on the most frequent) on the straightest path: one C branch sends the
        // classes 0..2 out of line, the assembly takes 3..6 in four scalar instructions
        if (__builtin_expect(cls < 3, 0)) {
          if (cls == 2) n |= P[54];
          else if (cls == 0) {   // no usable minimum yet: the exact filter of the slot's depth
            int so;   // (the slot index through an opaque copy: otherwise the address of its minimum is strength-reduced into one more
                      //  scalar add in EVERY step of the loop, for a path one step in a thousand takes)
            asm volatile("s_mov_b32 %0, %1" : "=s"(so) : "s"(sl));
            n = w1_exact_filter(P, besthi[2 * so + 1], nk - rb);
          }
        } else {
          uint32_t o;   // (a result register of its own: tied to n the compiler copies n first, for the other path's sake)
          asm volatile(
              "v_bitop3_b32 %[o], %[n], %[p54], %[p53] bitop3:0xfe\n\t"
              "s_cmp_lt_i32 %[c], 5\n\t"
              "s_cbranch_scc0 4f\n\t"
              "s_cmp_lt_i32 %[c], 4\n\t"
              "s_cbranch_scc1 3f\n\t"
              "v_or_b32 %[o], %[o], %[p52]\n\t"
              "s_branch 3f\n"
              "4:\n\t"
              "v_bitop3_b32 %[o], %[o], %[p52], %[p51] bitop3:0xfe\n\t"
              "s_cmp_lt_i32 %[c], 6\n\t"
              "s_cbranch_scc1 3f\n\t"
              "v_or_b32 %[o], %[o], %[p50]\n"
              "3:\n"
              : [o] "=&v"(o)
              : [n] "v"(n), [c] "s"(cls), [p54] "v"(P[54]), [p53] "v"(P[53]), [p52] "v"(P[52]), [p51] "v"(P[51]), [p50] "v"(P[50])
              : "scc");
          n = o;
        }
#endif
        const unsigned long long m = __ballot(n != 0xFFFFFFFFu);
        if (__builtin_expect(m != 0ULL, 1)) w1_enqueue(q, qn, wcap, (uint32_t)sl, ~n, m);   // (two steps in three trigger: the enqueue is the fall-through side)
      }
    }
  }
  if (qn > wcap) ok = false;
  const unsigned long long t2 = MHAP_TICK();
  if (!w1_flush2(best, q, qn, wcap, rb, ks, jump, lane)) ok = false;
  if (PROF) { tp[0] += t1 - t0; tp[first ? 1 : 2] += t2 - t1; tp[3] += MHAP_TICK() - t2; tp[4]++; tp[5] += first ? 1 : 0; tp[6] += (unsigned long long)qn; }
}

// FIXQ: the queue window has the default 4 096 words (every run up to --num-hashes 512): its capacity is then a literal in the enqueue and
// the drain — as a kernel argument it cost the slot loops one more live scalar and 1.3 % (78.8 -> 79.8 ms at C2)
#ifndef MH_W1_SEED_ROWS
#define MH_W1_SEED_ROWS 1
#endif
constexpr bool W1_SEED_ROWS = MH_W1_SEED_ROWS != 0;
template <bool PROF, bool FIXQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MH_W1_WAVES_EU, 8))) void minhash_w1_kernel(W1Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int H = a.H;
  uint64_t* lut = (uint64_t*)smem;                                   // shared by the workgroup's waves
  for (int i = threadIdx.x; i < MH_LUT_WORDS; i += blockDim.x) lut[i] = a.luts[i];
  __syncthreads();
  int64_t* best = (int64_t*)(smem + (size_t)MH_LUT_WORDS * 8 + (size_t)wv * (size_t)H * 8);
  const int qcap = FIXQ ? BS_QCAP : a.qcap;
  uint32_t* q = a.qbuf + ((size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)wv) * (size_t)qcap;
  {   // the window's base as a scalar pair: the queue stores then take it as their SGPR base + a 32-bit lane offset (as a VGPR pair it
      // was spilled, and reloaded from scratch in front of every queue store)
    const unsigned long long qa = (unsigned long long)(uintptr_t)q;
    const uint32_t qlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)qa), qhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(qa >> 32));
    q = (uint32_t*)(uintptr_t)(((unsigned long long)qhi << 32) | qlo);
  }
  const long long nitems = a.n_whole + a.n_tail * (long long)a.rmax;
  unsigned long long nst = 0;
  unsigned long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long tk0 = MHAP_TICK();
  // Round 6: the grid's waves all start a first row at the same moment, and a row is three phases — key load + transposes (memory
  // and LDS latency), the slot loop (VALU), the drain (L2 tables, LDS atomics) — so the four waves of a SIMD sit in the SAME phase
  // together, row after row: the VALU idles through everybody's loads and drains.  A long launch drifts apart by itself; a rank of an
  // 8-GPU job has six strands per wave and stays in step to the end (MHAP_MINHASH_PROF: 679 k wave-clocks per row against 618 k, all of
  // it in the two memory phases).  So the waves start a quarter of a row apart: slot number on the SIMD (HW_ID.WAVE_ID) x a.stagger clocks.
  if (a.stagger > 0) {
    const int slot = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 4) & 3u);   // HW_REG_HW_ID (4), bits 0..3 = WAVE_ID
    const long long until = (long long)slot * a.stagger;
    for (long long c = 0; c < until; c += 127 * 64) __builtin_amdgcn_s_sleep(127);
  }
  for (;;) {
    unsigned long long tk = 0;
    if (lane == 0) tk = atomicAdd(a.counter, 1ULL);
    const long long idx = (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tk >> 32)) << 32) |
                                      (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tk));
    if (idx >= nitems) break;
    const bool whole = idx < a.n_whole;
    long long ti = 0; int row = 0;
    if (!whole) { const long long t = idx - a.n_whole; ti = t % a.n_tail; row = (int)(t / a.n_tail); }
    const long long sidx = a.slist[whole ? idx : a.n_whole + ti];
    const ReadDesc rd = a.descs[sidx >> 1];
    const int rcs = (int)(sidx & 1);
    const int nk = rd.length - a.k + 1;
    int32_t* orow = a.out_rows + sidx * a.out_stride;
    const bool nosketch = strand_skipped(rd, rcs) || nk < 1 || rd.length - a.k2 + 1 < 1 || !a.info[sidx].valid;
    if (nosketch) {
      // too short (status 2) or ZeroNGramsFoundException from either sketch (status 1)
      if (whole || row == 0) {
        for (int s = lane; s < H; s += 64) orow[s] = 0;
        if (lane == 0) a.out_status[sidx * a.status_stride] = strand_skipped(rd, rcs) ? 2 : 1;
      }
    } else {
      const int nrows = (nk + 2047) >> 11;
      if (whole || row < nrows) {
        KeySrc ks;
        ks.kp = (rd.flags & MHAP_RD_MAT) ? a.keys + rd.key_off + (rcs ? rd.key_stride : 0) : nullptr;
        ks.W = (const uint32_t*)(a.store + rd.base_off); ks.nd = (((rd.length + 3) >> 2) + 3) >> 2; ks.L = rd.length; ks.rcs = rcs;
        ks.lut = lut; ks.perm = nullptr;
        // A row item of a strand whose earlier rows other waves have finished (the items are dealt row-major: every row 0 first) starts from
        // THEIR minima, read from the merge buffer: any chain value of the strand's own k-mers is a valid upper bound of the slot's
        // minimum, the result is the same minimum, and the row runs as a "later" row — class filter, a third of a first row's candidates
        // (round 6: a rank of an 8-GPU job cuts a sixth of its strands into row items, every one of them was a first row)
        bool seeded = !whole && row > 0 && W1_SEED_ROWS;
        if (seeded) {
          const unsigned long long* g = a.merge + (size_t)ti * (size_t)H;
          bool all = true;
          for (int s = lane; s < H; s += 64) {
            const unsigned long long v = __hip_atomic_load(&g[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            best[s] = v == ~0ULL ? INT64_MAX : (int64_t)(v ^ 0x8000000000000000ULL);
            all = all && v != ~0ULL;
          }
          seeded = __all(all);
        } else
          for (int s = lane; s < H; s += 64) best[s] = INT64_MAX;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool ok = true;
        const int r0 = whole ? 0 : row, r1 = whole ? nrows : row + 1;
        for (int r = r0; r < r1; r++) w1_row2<PROF>(best, (uint2*)q, qcap / 2 - 1, r << 11, nk, r == r0 && !seeded, H, ks, a.jump, lane, ok, tp);
        // (a row whose candidates overflowed the queue — never seen — is redone one k-mer at a time: exact, slow)
        if (!ok) {
          for (int s = lane; s < H; s += 64) best[s] = INT64_MAX;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const int i1 = (r1 << 11) < nk ? (r1 << 11) : nk;
          for (int i = (r0 << 11) + lane; i < i1; i += 64) {
            uint64_t x = ks_key(ks, i);
            for (int s = 0; s < H; s++) { x = xorshift_step(x); atomicMin((long long*)&best[s], (long long)x); }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          __builtin_amdgcn_wave_barrier();
        }
        if (whole) {
          for (int s = lane; s < H; s += 64) {
            const int64_t x = best[s];
            int32_t v = 0;
            if (x != INT64_MAX) {
              const uint64_t key = w1_key_of((uint64_t)x, s + 1, a.unjump);
              v = (s & 1) ? (int32_t)(uint32_t)(key >> 32) : (int32_t)(uint32_t)key;   // MinHashSketch.java:147-150
            }
            orow[s] = v;
          }
          if (lane == 0) a.out_status[sidx * a.status_stride] = 0;
          nst++;
        } else {
          unsigned long long* g = a.merge + (size_t)ti * (size_t)H;
          for (int s = lane; s < H; s += 64) {
            const int64_t x = best[s];
            if (x != INT64_MAX) atomicMin(&g[s], (unsigned long long)x ^ 0x8000000000000000ULL);
          }
          if (row == 0) { if (lane == 0) a.out_status[sidx * a.status_stride] = 0; nst++; }
        }
      }
    }
  }
  if (lane == 0) atomicAdd(a.stat, nst);
  if (PROF && lane == 0) { tp[7] = MHAP_TICK() - tk0; for (int i = 0; i < 8; i++) atomicAdd(&a.prof[i], tp[i]); }
}

// output rows of the strands that were sketched row by row: the merged minima back to keys
__global__ __launch_bounds__(256) void minhash_w1_finish_kernel(W1Args a) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.n_tail * (long long)a.H) return;
  const long long ti = t / a.H;
  const int s = (int)(t % a.H);
  const long long sidx = a.slist[a.n_whole + ti];
  const unsigned long long g = a.merge[t];
  int32_t v = 0;
  if (g != ~0ULL) {
    const uint64_t key = w1_key_of(g ^ 0x8000000000000000ULL, s + 1, a.unjump);
    v = (s & 1) ? (int32_t)(uint32_t)(key >> 32) : (int32_t)(uint32_t)key;
  }
  a.out_rows[sidx * a.out_stride + s] = v;
}

// Jump-ahead tables for the xorshift64 chain: the step is linear over GF(2), so M^(g a) x (g = 2^XS_JUMP_LOG2) is the XOR of
// eight byte-indexed table entries.  out[(a-1)*2048 + i*256 + v] = M^(g a) applied to (v << 8i), a = 1..na; then nq coarse
// tables out[(na+q-1)*2048 + ...] = M^(g na q), q = 1..nq, for chains that run past g na steps (weighted k-mers).
void build_xorshift_jump_tables(int na, int nq, uint64_t* out) {
  uint64_t col[64], nxt[64];
  auto apply = [](const uint64_t* c, uint64_t x) { uint64_t y = 0; while (x) { const int b = __builtin_ctzll(x); x &= x - 1; y ^= c[b]; } return y; };
  auto emit = [&](const uint64_t* c, uint64_t* T) {   // (one xor per entry: the entry without v's lowest set bit, plus that bit's column)
    for (int i = 0; i < 8; i++) {
      T[i * 256] = 0;
      for (int v = 1; v < 256; v++) T[i * 256 + v] = T[i * 256 + (v & (v - 1))] ^ c[8 * i + __builtin_ctz((unsigned)v)];
    }
  };
  for (int j = 0; j < 64; j++) { uint64_t x = 1ULL << j; for (int t = 0; t < (1 << XS_JUMP_LOG2); t++) x = xorshift_step(x); col[j] = x; }   // M^g
  uint64_t base[64], big[64];
  for (int j = 0; j < 64; j++) base[j] = col[j];
  for (int a = 1; a <= na; a++) {
    emit(col, out + (size_t)(a - 1) * 2048);
    if (a == na) for (int j = 0; j < 64; j++) big[j] = col[j];   // M^(g na)
    for (int j = 0; j < 64; j++) nxt[j] = apply(base, col[j]);   // M^(g(a+1)) = M^g o M^(g a)
    for (int j = 0; j < 64; j++) col[j] = nxt[j];
  }
  for (int j = 0; j < 64; j++) col[j] = big[j];
  for (int q = 1; q <= nq; q++) {
    emit(col, out + (size_t)(na + q - 1) * 2048);
    for (int j = 0; j < 64; j++) nxt[j] = apply(big, col[j]);
    for (int j = 0; j < 64; j++) col[j] = nxt[j];
  }
}
// the same for the inverse map: out[(a-1)*2048 + i*256 + v] = M^-(g a) applied to (v << 8i), a = 1..na
void build_xorshift_unjump_tables(int na, int nq, uint64_t* out) {
  uint64_t col[64], nxt[64], base[64], big[64];
  auto apply = [](const uint64_t* c, uint64_t x) { uint64_t y = 0; while (x) { const int b = __builtin_ctzll(x); x &= x - 1; y ^= c[b]; } return y; };
  auto emit = [&](const uint64_t* c, uint64_t* T) {   // (one xor per entry: the entry without v's lowest set bit, plus that bit's column)
    for (int i = 0; i < 8; i++) {
      T[i * 256] = 0;
      for (int v = 1; v < 256; v++) T[i * 256 + v] = T[i * 256 + (v & (v - 1))] ^ c[8 * i + __builtin_ctz((unsigned)v)];
    }
  };
  for (int j = 0; j < 64; j++) { uint64_t x = 1ULL << j; for (int t = 0; t < (1 << XS_JUMP_LOG2); t++) x = xorshift_unstep(x); col[j] = base[j] = x; }
  for (int a = 1; a <= na; a++) {
    emit(col, out + (size_t)(a - 1) * 2048);
    if (a == na) for (int j = 0; j < 64; j++) big[j] = col[j];   // M^-(g na)
    for (int j = 0; j < 64; j++) nxt[j] = apply(base, col[j]);
    for (int j = 0; j < 64; j++) col[j] = nxt[j];
  }
  for (int j = 0; j < 64; j++) col[j] = big[j];
  for (int q = 1; q <= nq; q++) {                                 // coarse: M^-(g na q)
    emit(col, out + (size_t)(na + q - 1) * 2048);
    for (int j = 0; j < 64; j++) nxt[j] = apply(big, col[j]);
    for (int j = 0; j < 64; j++) col[j] = nxt[j];
  }
}

// Resident workgroups per CU of the MinHash launches (4 waves = 4 strands each) and the bytes of queue memory their waves need
int minhash_wgs_per_cu(int H) {
  // the weight-1 kernel keeps 8 B per slot and wave in LDS (the general kernel 12 B: its launch simply gets fewer workgroups resident)
  const size_t lds = (size_t)H * 8 * 4 + (size_t)MH_LUT_WORDS * 8;
  int n = (int)((160 * 1024) / lds);
  const int by_regs = MH_W1_WAVES_EU > MH_WAVES_EU ? MH_W1_WAVES_EU : MH_WAVES_EU;   // waves per SIMD = workgroups of four waves per CU
  if (n > by_regs) n = by_regs;
  return n < 1 ? 1 : n;
}
// Words of one wave's candidate queue.  A strand's FIRST row queues about two entries per slot (no minima to filter by yet): 4 096 words —
// 2 047 eight-byte entries of the weight-1 kernel, 4 096 four-byte ones of the general kernel — hold that up to --num-hashes 512 with a
// factor of two to spare; beyond, the window grows with H (8 H words), so that a run at --num-hashes 1024 or 2048 does not send every strand
// through the exact one-k-mer-at-a-time redo (round 5).  A variant build with another MH_QCAP keeps its fixed size (tests).
int minhash_queue_words(int H) {
  if (BS_QCAP != 4096) return BS_QCAP;
  const int want = ((8 * H + 63) / 64) * 64;
  return want > BS_QCAP ? want : BS_QCAP;
}
size_t minhash_queue_bytes(int nblocks_total, int H) { return (size_t)nblocks_total * 4 * (size_t)minhash_queue_words(H) * 4; }
static_assert(MHAP_MAX_NUM_HASHES < 65536, "a weight-1 queue entry is slot | lane << 16 (w1_enqueue)");
// waves (= strands in flight) of one MinHash workgroup: launch_minhash's own rule, for whoever sizes per-wave scratch
int minhash_waves_per_workgroup(int H) {
  const size_t per_wave = (((size_t)H * 12 + 8) + 15) & ~(size_t)15, lut_bytes = (size_t)MH_LUT_WORDS * 8;
  int waves = 4;
  while (waves > 1 && per_wave * waves + lut_bytes > 150 * 1024) waves >>= 1;
  return waves;
}

// Strands of a weight-1 launch that are cut into row items (the others are taken whole): one strand's worth of rows per resident
// wave at the end of the list evens the waves' finish times out to one row; a list shorter than that is all rows.
int64_t minhash_tail_strands(int nblocks, int64_t n_unweighted) {
  static int div = 0;   // MHAP_W1_TAIL_DIV: 1 / 2 / 4 ... = a strand's worth of rows for every / every second / fourth resident wave (experiments)
  if (div == 0) { const char* e = getenv("MHAP_W1_TAIL_DIV"); div = e && atoi(e) > 0 ? atoi(e) : 1; }
  const int64_t waves = (int64_t)nblocks * 4 / div;
  return n_unweighted < waves ? n_unweighted : waves;
}
size_t minhash_merge_bytes(int nblocks, int H) { return (size_t)nblocks * 4 * (size_t)H * 8; }

// MHAP_MINHASH=perchain selects the kernel without bit-sliced rows, MHAP_MINHASH=classic round 2's kernel for the weight-1 strands too (A/B measurements)
bool launch_minhash(hipStream_t st, hipStream_t st_weighted, int nblocks, int64_t n_unweighted, int64_t n_weighted, const ReadDesc* descs, int64_t nstrands, const int64_t* keys, const uint32_t* wts,
                    const uint32_t* perm, const StrandInfo* info, const uint8_t* store, const uint64_t* luts, int k, int k2, int H,
                    unsigned long long* counter, int32_t* out_rows, int64_t out_stride, int32_t* out_status, int64_t status_stride,
                    const uint64_t* jump, int jump_na, const int32_t* slist, uint32_t* qbuf, const uint64_t* unjump, const uint64_t* jump_w1, unsigned long long* merge, int max_nk) {
  // counter: the sketch phase's counter block — [1] / [2] work counters of the two launches, [4] / [5] lengths of their strand lists
  // (written by kmer_weight_kernel), [9] / [11] strands sketched.  qbuf: minhash_queue_bytes(2 * nblocks) (one half per launch);
  // merge: minhash_merge_bytes(nblocks, H); max_nk: k-mers of the launch's longest strand.  Returns whether anything was put on st_weighted
  // (the caller joins that stream only then)
  if (nstrands <= 0) return false;
  static int perchain = -1, classic = 0;
  if (perchain < 0) { const char* e = getenv("MHAP_MINHASH"); perchain = (e && strcmp(e, "perchain") == 0) ? 1 : 0; classic = (e && strcmp(e, "classic") == 0) ? 1 : 0; }
  size_t per_wave = (((size_t)H * 12 + 8) + 15) & ~(size_t)15;
  const size_t lut_bytes = (size_t)MH_LUT_WORDS * 8;
  const int waves = minhash_waves_per_workgroup(H);   // waves (= strands in flight) per workgroup; fewer when --num-hashes is huge
  const size_t lds = per_wave * waves + lut_bytes;
  const dim3 block(64 * waves);
  nblocks = (int)(((int64_t)nblocks * 4 + waves - 1) / waves);
  const int qcap = minhash_queue_words(H);
  uint32_t* qbuf_w = qbuf + (size_t)nblocks * (size_t)waves * (size_t)qcap;
  static int profmode = -1;
  if (profmode < 0) { const char* e = getenv("MHAP_MINHASH_PROF"); profmode = (e && atoi(e)) ? 1 : 0; }
  unsigned long long* counter_u = counter + 1; unsigned long long* counter_w = counter + 2;
  const int32_t* slist_w = slist + nstrands;
  if (profmode && !perchain && classic) {   // wave-clock attribution of round 2's kernels (diagnostics; printed per launch)
    static unsigned long long* dprof = nullptr;
    if (!dprof) (void)hipMalloc(&dprof, 16 * sizeof(unsigned long long));
    for (int pass = 0; pass < 2; pass++) {
      (void)hipMemsetAsync(dprof, 0, 16 * sizeof(unsigned long long), st);
      if (pass == 0)
        hipLaunchKernelGGL((minhash_kernel<MH_U, true, false, true>), dim3(nblocks), block, lds, st, descs, nstrands, keys, wts, perm, info, store, luts, k, k2, H,
                           counter_u, out_rows, out_stride, out_status, status_stride, jump, jump_na, slist, counter + 4, qbuf, qcap, dprof);
      else
        hipLaunchKernelGGL((minhash_kernel<MH_U, true, true, true>), dim3(nblocks), block, lds, st, descs, nstrands, keys, wts, perm, info, store, luts, k, k2, H,
                           counter_w, out_rows, out_stride, out_status, status_stride, jump, jump_na, slist_w, counter + 5, qbuf_w, qcap, dprof);
      unsigned long long hp[16];
      (void)hipMemcpyAsync(hp, dprof, sizeof(hp), hipMemcpyDeviceToHost, st);
      (void)hipStreamSynchronize(st);
      const double tot = (double)hp[0];
      if (hp[8])
        fprintf(stderr, "[minhash prof %s] strands %llu  wave-clocks/strand %.0f  row0 %.1f%% (argmin %.1f%%, defer %.1f%%)  later defers %.1f%%  load+transpose %.1f%%  "
                        "flush %.1f%% (%.1f batches/strand, %.0f clocks/batch)\n", pass ? "weighted" : "weight-1",
                hp[8], tot / (double)hp[8], 100.0 * hp[1] / tot, 100.0 * hp[2] / tot, 100.0 * hp[3] / tot, 100.0 * hp[4] / tot, 100.0 * hp[5] / tot,
                100.0 * hp[6] / tot, (double)hp[7] / (double)hp[8], hp[7] ? (double)hp[6] / (double)hp[7] : 0.0);
    }
    return false;
  }
  if (perchain) {
    hipLaunchKernelGGL((minhash_kernel<MH_U, false, true>), dim3(nblocks), block, lds, st_weighted, descs, nstrands, keys, wts, perm, info, store, luts, k, k2, H,
                       counter_w, out_rows, out_stride, out_status, status_stride, jump, jump_na, slist_w, counter + 5, qbuf_w, qcap);
    hipLaunchKernelGGL((minhash_kernel<MH_U, false, false>), dim3(nblocks), block, lds, st, descs, nstrands, keys, wts, perm, info, store, luts, k, k2, H,
                       counter_u, out_rows, out_stride, out_status, status_stride, jump, jump_na, slist, counter + 4, qbuf, qcap);
    return true;
  } else {
    // n_unweighted / n_weighted >= 0: the lengths of the two work lists (the caller read them back): each launch gets only the
    // workgroups its list can feed, the weighted one first, on its own stream — its few workgroups take their slots, the weight-1
    // launch fills the rest of the GPU, and nothing is left to run alone at the end.  < 0: unknown, full grids.
    const int waves_wg = (int)block.x / 64;
    // a few strands only: one WORKGROUP per strand, its rows going round the four waves (SPLIT in minhash_kernel); MHAP_MINHASH_SPLIT=0 turns it off
    const char* split_env = getenv("MHAP_MINHASH_SPLIT");
    const bool split_ok = !(split_env && atoi(split_env) == 0);
    const bool split = split_ok && waves_wg == 4 && n_weighted > 0 && n_weighted * 4 <= nblocks;
    const int nb_w = n_weighted < 0 ? nblocks : (int)std::min<int64_t>(nblocks, split ? n_weighted : (n_weighted + waves_wg - 1) / waves_wg);
    const int nb_u = n_unweighted < 0 ? nblocks : (int)std::min<int64_t>(nblocks, (n_unweighted + waves_wg - 1) / waves_wg);
    if (nb_w > 0)
      hipLaunchKernelGGL((minhash_kernel<MH_U, true, true>), dim3(nb_w), block, lds + 16, st_weighted, descs, nstrands, keys, wts, perm, info, store, luts, k, k2, H,
                         counter_w, out_rows, out_stride, out_status, status_stride, jump, jump_na, slist_w, counter + 5, qbuf_w, qcap, nullptr, split ? 1 : 0);
    if (nb_u > 0 && (classic || n_unweighted < 0 || waves_wg != 4)) {
      hipLaunchKernelGGL((minhash_kernel<MH_U, true, false>), dim3(nb_u), block, lds, st, descs, nstrands, keys, wts, perm, info, store, luts, k, k2, H,
                         counter_u, out_rows, out_stride, out_status, status_stride, jump, jump_na, slist, counter + 4, qbuf, qcap);
    } else if (nb_u > 0) {
      // the weight-1 strands: whole strands first, the last minhash_tail_strands() of the list row by row (see minhash_w1_kernel)
      W1Args a;
      a.descs = descs; a.keys = keys; a.store = store; a.luts = luts; a.info = info; a.slist = slist;
      a.n_tail = minhash_tail_strands(nblocks, n_unweighted); a.n_whole = n_unweighted - a.n_tail;
      a.rmax = max_nk > 0 ? (max_nk + 2047) >> 11 : 1;
      a.k = k; a.k2 = k2; a.H = H; a.counter = counter_u; a.stat = counter_u + 8;
      a.out_rows = out_rows; a.out_stride = out_stride; a.out_status = out_status; a.status_stride = status_stride;
      a.jump = jump_w1; a.unjump = unjump; a.jump_na = W1_JUMP_NA; a.qbuf = qbuf; a.qcap = qcap; a.merge = merge;   // (its own two-level table set)
      {   // a quarter of a row's ~1 300 clocks per slot (MHAP_W1_STAGGER: clocks per SIMD slot number; 0 = all waves start together)
        static int stag = -2;
        if (stag == -2) { const char* e = getenv("MHAP_W1_STAGGER"); stag = e ? atoi(e) : -1; }
        a.stagger = stag >= 0 ? stag : 0;   // (default off: measured without effect on a rank's launch — 10.83 / 10.70 ms with, 10.83 / 10.69 without: EXPERIMENTS round 6)
      }
      const long long items = a.n_whole + a.n_tail * (long long)a.rmax;
      const int nb = (int)std::min<long long>(nblocks, (items + 3) / 4);
      const size_t lds1 = (size_t)H * 8 * 4 + lut_bytes;
      if (a.n_tail > 0) (void)hipMemsetAsync(merge, 0xFF, (size_t)a.n_tail * (size_t)H * 8, st);
      a.prof = nullptr;
      if (profmode) {
        static unsigned long long* dprof1 = nullptr;
        if (!dprof1) (void)hipMalloc(&dprof1, 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dprof1, 0, 8 * sizeof(unsigned long long), st);
        a.prof = dprof1;
        static hipEvent_t pe0 = nullptr, pe1 = nullptr;
        if (!pe0) { (void)hipEventCreate(&pe0); (void)hipEventCreate(&pe1); }
        (void)hipEventRecord(pe0, st);
        if (qcap == BS_QCAP) hipLaunchKernelGGL((minhash_w1_kernel<true, true>), dim3(nb), dim3(256), lds1, st, a);
        else hipLaunchKernelGGL((minhash_w1_kernel<true, false>), dim3(nb), dim3(256), lds1, st, a);
        (void)hipEventRecord(pe1, st);
        unsigned long long hp[8];
        (void)hipMemcpyAsync(hp, dprof1, sizeof(hp), hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        const double tot = (double)hp[7];
        float pms = 0.f; (void)hipEventElapsedTime(&pms, pe0, pe1);
        // shader clock the chip held: every resident wave counts its own s_memtime ticks from start to end of the persistent loop
        fprintf(stderr, "[minhash w1 prof] launch %.3f ms, %d waves: mean shader clock %.0f MHz (wave-clocks / waves / time)\n", pms, nb * 4,
                pms > 0.f ? tot / ((double)nb * 4.0) / ((double)pms * 1e3) : 0.0);
        fprintf(stderr, "[minhash w1 prof] rows %llu (first-type %llu)  wave-clocks: total %.3g  key load+transpose %.1f%%  first-row slots %.1f%% (%.0f clocks/step)  "
                        "later-row slots %.1f%% (%.0f clocks/step)  drains %.1f%%  candidates/row %.0f\n", hp[4], hp[5], tot, 100.0 * hp[0] / tot, 100.0 * hp[1] / tot,
                hp[5] ? (double)hp[1] / ((double)hp[5] * H) : 0.0, 100.0 * hp[2] / tot, hp[4] > hp[5] ? (double)hp[2] / ((double)(hp[4] - hp[5]) * H) : 0.0,
                100.0 * hp[3] / tot, hp[4] ? (double)hp[6] / (double)hp[4] : 0.0);
      } else
        if (qcap == BS_QCAP) hipLaunchKernelGGL((minhash_w1_kernel<false, true>), dim3(nb), dim3(256), lds1, st, a);
        else hipLaunchKernelGGL((minhash_w1_kernel<false, false>), dim3(nb), dim3(256), lds1, st, a);
      if (a.n_tail > 0)
        hipLaunchKernelGGL(minhash_w1_finish_kernel, dim3((unsigned)((a.n_tail * (long long)H + 255) / 256)), dim3(256), 0, st, a);
    }
    return nb_w > 0;
  }
}


// =============================================================================================
// Ordered bottom-S sketch.  One workgroup per strand.  Composite key = (hash ^ 0x80000000) << 32 | pos
// orders exactly like (signed hash asc, pos asc) = fastutil's stable radixSortIndirect.  A multi-level
// radix select (11/11/10 bits of hash, then 11/11/10 bits of pos) narrows the S smallest keys down to
// at most CAP candidates, which are then bitonic-sorted in LDS.
// =============================================================================================
#ifdef MH_ORD_PROF
__device__ unsigned long long g_ord_prof[8];
#define ORD_TICK(k) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_ord_prof[k], t_ - ord_t0); ord_t0 = t_; } } while (0)
#else
#define ORD_TICK(k) do { } while (0)
#endif
// h * 5 + 0xe6546b64 of the murmur3_x86_32 body as a shift-add and an add: the compiler turns the multiply-add into v_mad_u64_u32
// (quarter rate; 48 of them per eight hashes in the pass below)
__device__ __forceinline__ uint32_t mm3_mul5c(uint32_t h) {
  uint32_t t;
  asm("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(t) : "v"(h));
  return t + 0xe6546b64U;
}
constexpr uint32_t ORD_BUCKET_MAX = 32;   // keys per first-level bin the bucket path sorts by insertion
__device__ inline uint64_t okey(int32_t h, int pos) { return ((uint64_t)((uint32_t)h ^ 0x80000000u) << 32) | (uint32_t)pos; }

__global__ __launch_bounds__(ORD_THREADS) void ordered_kernel(const ReadDesc* __restrict__ descs, int64_t nstrands,
                                                              const int32_t* __restrict__ h32, const uint8_t* __restrict__ store,
                                                              const uint64_t* __restrict__ luts, int code_words, int stage_wide, int k2, int S, int cap,
                                                              int32_t* __restrict__ out_rows, int64_t out_stride,
                                                              int32_t* __restrict__ out_meta, int64_t meta_stride, int64_t first) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* hist = (uint32_t*)smem;                       // ORD_BINS
  uint32_t* part = hist + ORD_BINS;                       // ORD_THREADS partial sums
  uint32_t* svars = part + ORD_THREADS;                   // 4 scalars (kept in the dynamic region: 16-B aligned base)
  uint64_t* buf = (uint64_t*)(svars + 4);                 // cap keys
  uint16_t* bstart = (uint16_t*)(buf + cap);              // bucket path: first buffer slot of every first-level bin (+1 end marker)
  uint32_t* stage = (uint32_t*)(bstart + ORD_BINS + 2);   // one-pass path: positions of the keys below the guessed cut (16-bit unless the
  uint16_t* stage16 = (uint16_t*)stage;                   // launch has a strand with more than 65535 k-mers: 4 KB instead of 8 at S = 1536)
  // murmur3_x86_32 block-mix table (256 words), then the strand's base codes.  The offset is computed as a NUMBER and added to the
  // shared-memory base: rounding the pointer itself up through uintptr_t (round 1-4) cost the compiler the address space — every read of
  // the table and of the codes was a flat_load (85 of them in the kernel, 7.2e7 wave-level VMEM reads per C2 launch in the PMC pass),
  // which goes through the vector-memory path and its address check instead of a ds_read
  const size_t lut_off = ((size_t)ORD_BINS * 4 + (size_t)ORD_THREADS * 4 + 16 + (size_t)cap * 8 + (size_t)(ORD_BINS + 2) * 2 + (size_t)cap * (stage_wide ? 4 : 2) + 7) & ~(size_t)7;
  uint64_t* lut = (uint64_t*)(smem + lut_off);
  uint32_t* codes = (uint32_t*)(lut + 256);
  uint32_t& s_bin = svars[0]; uint32_t& s_below = svars[1]; uint32_t& s_cnt = svars[2]; uint32_t& s_fill = svars[3];
  const int64_t strand = first + (int64_t)blockIdx.x;   // (a launch covers the strands [first, first + gridDim.x): sketch_staged runs the kernel in two parts)
  if (strand >= nstrands) return;
#ifdef MH_ORD_PROF
  unsigned long long ord_t0 = wall_clock64();
#endif
  const ReadDesc rd = descs[strand >> 1];
  const int rcs = (int)(strand & 1);
  const int n = rd.length - k2 + 1;
  int32_t* meta = out_meta + strand * meta_stride;
  if (strand_skipped(rd, rcs) || n < 1) {
    if (threadIdx.x == 0) { meta[0] = 0; meta[1] = n; meta[2] = rd.length; }
    return;
  }
  // 32-bit hashes of the strand's 12-mers: stored by hash_kmers_kernel for MHAP_RD_MAT strands, otherwise recomputed from the
  // strand's 2-bit codes (staged once in LDS, reverse strand already complemented) wherever a pass needs them
  const bool mat = (rd.flags & MHAP_RD_MAT) != 0;
  const int32_t* hp = h32 + rd.h2_off + (rcs ? rd.h2_stride : 0);
  if (!mat) {
    for (int i = threadIdx.x; i < 256; i += ORD_THREADS) lut[i] = luts[512 + i];
    const int ncw = (rd.length + 15) / 16 + 2;
    for (int wj = threadIdx.x; wj < ncw && wj < code_words; wj += ORD_THREADS) codes[wj] = strand_codes16(store + rd.base_off, rd.length, rcs, 16 * wj);
    __syncthreads();
  }
  ORD_TICK(0);
  auto hget = [&](int i) -> int32_t {
    if (mat) return hp[i];
    const uint32_t cw = codes_at(codes, i);
    uint32_t h = 0;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const uint64_t kk = lut[(cw >> (8 * q)) & 255u];
      h ^= (uint32_t)kk;         h = rotl32(h, 13); h = mm3_mul5c(h);
      h ^= (uint32_t)(kk >> 32); h = rotl32(h, 13); h = mm3_mul5c(h);
    }
    h ^= 24u;
    return (int32_t)fmix32(h);
  };
  const int K = S < n ? S : n;   // BottomOverlapSketch.java:548
  if (threadIdx.x == 0) { meta[0] = K; meta[1] = n; meta[2] = rd.length; }
  if (K <= 0) return;

  // ---- one-pass path.  murmur3 hashes are uniform, so the K-th smallest key is close to the K/n quantile of the hash range:
  // one pass keeps the positions of the keys below a cut placed a few standard deviations above that quantile (and their
  // first-level histogram); if the guess holds (at least K and at most cap keys kept, no crowded bin) the kept keys are
  // bucket-sorted straight away, everything else about the strand's other ~85 % of keys is never looked at again.  If
  // it does not hold (skewed hashes, e.g. low-complexity sequence) the exact multi-level selection below runs instead.
  // The number of keys below a cut is binomial, sigma < sqrt(expected): aim at K + 7 sigma, but stay 5 sigma under cap.
  const double ksig = sqrt((double)K), csig = sqrt((double)cap);
  const double target = fmin((double)K + 7.0 * ksig, (double)cap - 5.0 * csig);
  if (n > cap && target >= (double)K + 3.0 * ksig) {
    const double quant = target / (double)n;
    const uint32_t cut_u = quant >= 1.0 ? 0xFFFFFFFFu : (uint32_t)(quant * 4294967296.0);   // on (hash ^ 0x80000000) = unsigned rank order
    const uint32_t cutbin = cut_u >> 21;
    for (uint32_t j = threadIdx.x; j < ORD_BINS; j += ORD_THREADS) hist[j] = 0;
    if (threadIdx.x == 0) { s_fill = 0; s_cnt = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // A lane hashes the eight 12-mers p0, p0+4, .., p0+28 (p0 = 32 (thread / 4) + thread % 4 inside the 2048 positions of one
    // trip): they are made of ten consecutive 4-base groups, and a group's block-mix table entry serves the three 12-mers it
    // belongs to — 10 table reads + 4 code dwords per eight hashes instead of 24 + 16 (the pass is bound by LDS reads).
    for (int ib = 0; ib < n; ib += 8 * ORD_THREADS) {   // wave-uniform trip count: ballots inside
      const int p0 = ib + 32 * (int)(threadIdx.x >> 2) + (int)(threadIdx.x & 3);
      int32_t hv[8];
      if (mat) {
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = p0 + 4 * u; hv[u] = i < n ? hp[i] : 0; }
      } else if (p0 < n) {
        const int d = p0 >> 4;
        const uint32_t sh = (uint32_t)(2 * (p0 & 15));
        const uint32_t w0 = codes[d], w1 = codes[d + 1], w2 = codes[d + 2], w3 = codes[d + 3];
        const uint32_t a[3] = {__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh), __builtin_amdgcn_alignbit(w3, w2, sh)};
        uint64_t g[10];
#pragma unroll
        for (int j = 0; j < 10; j++) g[j] = lut[(a[j >> 2] >> (8 * (j & 3))) & 255u];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          uint32_t h = 0;
#pragma unroll
          for (int q = 0; q < 3; q++) {
            const uint64_t kk = g[u + q];
            h ^= (uint32_t)kk;         h = rotl32(h, 13); h = mm3_mul5c(h);
            h ^= (uint32_t)(kk >> 32); h = rotl32(h, 13); h = mm3_mul5c(h);
          }
          h ^= 24u;
          hv[u] = (int32_t)fmix32(h);
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; u++) hv[u] = 0;
      }
      unsigned long long bal[8];
      uint32_t total = 0;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        bal[u] = __ballot(p0 + 4 * u < n && ((uint32_t)hv[u] ^ 0x80000000u) < cut_u);
        total += (uint32_t)__popcll(bal[u]);
      }
      if (total) {   // one queue reservation per wavefront and group of eight
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_fill, total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if ((bal[u] >> lane) & 1ULL) {
            const uint32_t idx = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[u], 0u));
            if (idx < (uint32_t)cap) { if (stage_wide) stage[idx] = (uint32_t)(p0 + 4 * u); else stage16[idx] = (uint16_t)(p0 + 4 * u); }
            atomicAdd(&hist[((uint32_t)hv[u] ^ 0x80000000u) >> 21], 1u);
          }
          base += (uint32_t)__popcll(bal[u]);
        }
      }
    }
    __syncthreads();
    ORD_TICK(1);
    const uint32_t m = s_fill;
    bool ok = m >= (uint32_t)K && m <= (uint32_t)cap;
    if (ok) {
      // exclusive prefix over the bins up to the cut: per-lane sums, shuffle scan inside a wavefront, wavefront totals through LDS
      const int per = ORD_BINS / ORD_THREADS;
      uint32_t loc = 0, mx = 0;
      for (int j = 0; j < per; j++) { const uint32_t c = hist[threadIdx.x * per + j]; loc += c; mx = c > mx ? c : mx; }
      uint32_t incl = loc;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off); if (lane >= off) incl += v; }
      if (lane == 63) part[threadIdx.x >> 6] = incl;
      if (mx > ORD_BUCKET_MAX) atomicMax(&s_cnt, mx);
      __syncthreads();
      for (unsigned w = 0; w < (threadIdx.x >> 6); w++) incl += part[w];
      uint32_t run = incl - loc;
      for (int j = 0; j < per; j++) { const uint32_t b = threadIdx.x * per + j; bstart[b] = (uint16_t)run; run += hist[b]; }
      ok = s_cnt == 0;
      __syncthreads();
    }
    ORD_TICK(2);
    if (ok) {
      for (uint32_t b = threadIdx.x; b <= cutbin; b += ORD_THREADS) hist[b] = 0;   // becomes the bins' fill counters
      __syncthreads();
      // (four kept keys per lane and trip: a lane has about seven of them, and one at a time the chain position -> codes -> table ->
      //  hash -> bin counter -> store ran seven times back to back with nothing to overlap it)
      for (uint32_t t0 = threadIdx.x; t0 < m; t0 += 4 * ORD_THREADS) {
        int ii[4]; uint64_t key[4]; uint32_t slot[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t t = t0 + (uint32_t)u * ORD_THREADS; ii[u] = t < m ? (stage_wide ? (int)stage[t] : (int)stage16[t]) : 0; }
#pragma unroll
        for (int u = 0; u < 4; u++) key[u] = okey(hget(ii[u]), ii[u]);
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t b = (uint32_t)(key[u] >> 53); slot[u] = t0 + (uint32_t)u * ORD_THREADS < m ? (uint32_t)bstart[b] + atomicAdd(&hist[b], 1u) : 0xFFFFFFFFu; }
#pragma unroll
        for (int u = 0; u < 4; u++) if (slot[u] != 0xFFFFFFFFu) buf[slot[u]] = key[u];
      }
      __syncthreads();
      ORD_TICK(3);
      // every key finds its rank inside its bin (<= ORD_BUCKET_MAX keys, all reads independent) and goes straight to its place in
      // the output row: one key per lane instead of one bin per lane walking a serial insertion sort
      int32_t* orow = out_rows + strand * out_stride;
      // (two keys per lane and trip, their bins walked side by side: eight independent reads in flight instead of four)
      for (uint32_t t0 = threadIdx.x; t0 < m; t0 += 2 * ORD_THREADS) {
        uint64_t key[2]; uint32_t s0[2], c[2], rank[2] = {0u, 0u};
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const uint32_t t = t0 + (uint32_t)u * ORD_THREADS;
          key[u] = buf[t < m ? t : t0];
          const uint32_t b = (uint32_t)(key[u] >> 53);
          s0[u] = bstart[b]; c[u] = t < m ? hist[b] : 0u;
        }
        const uint32_t cmax = c[0] > c[1] ? c[0] : c[1];
        for (uint32_t q = 0; q < cmax; q += 4) {   // (a bin's slice is followed by valid buffer words)
          uint64_t o[2][4];
#pragma unroll
          for (int u = 0; u < 2; u++)
#pragma unroll
            for (int x = 0; x < 4; x++) o[u][x] = buf[s0[u] + q + (uint32_t)x < (uint32_t)cap ? s0[u] + q + (uint32_t)x : s0[u]];
#pragma unroll
          for (int u = 0; u < 2; u++)
#pragma unroll
            for (int x = 0; x < 4; x++) rank[u] += (q + (uint32_t)x < c[u] && o[u][x] < key[u]) ? 1u : 0u;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const uint32_t j = s0[u] + rank[u];
          if (t0 + (uint32_t)u * ORD_THREADS < m && j < (uint32_t)K) { orow[2 * j] = (int32_t)((uint32_t)(key[u] >> 32) ^ 0x80000000u); orow[2 * j + 1] = (int32_t)(uint32_t)key[u]; }
        }
      }
      ORD_TICK(5);
#ifdef MH_ORD_PROF
      if (threadIdx.x == 0) atomicAdd(&g_ord_prof[7], 1ULL);
#endif
      return;
    }
    __syncthreads();
  }

  uint64_t bound = ~0ULL;  // select all keys <= bound
  bool buckets = false;    // the first-level histogram already is a bucket sort (see below)
  uint32_t bk_bin = 0, bk_total = 0;
  if (threadIdx.x == 0) s_fill = 0;
  if (n > cap) {
    const int shifts[6] = {53, 42, 32, 21, 10, 0};
    const int widths[6] = {11, 11, 10, 11, 11, 10};
    uint64_t prefix = 0;      // high bits fixed so far
    uint64_t prefmask = 0;
    uint32_t below = 0;       // keys strictly below the current prefix bin
    for (int lv = 0; lv < 6; lv++) {
      const int sh = shifts[lv];
      const uint32_t nb = 1u << widths[lv];
      for (uint32_t j = threadIdx.x; j < ORD_BINS; j += ORD_THREADS) hist[j] = 0;
      __syncthreads();
      for (int i0 = threadIdx.x; i0 < n; i0 += 8 * ORD_THREADS) {   // eight loads in flight per lane
        int32_t hv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * ORD_THREADS; hv[u] = i < n ? hget(i) : 0; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int i = i0 + u * ORD_THREADS;
          const uint64_t key = okey(hv[u], i);
          if (i < n && (key & prefmask) == prefix) atomicAdd(&hist[(uint32_t)(key >> sh) & (nb - 1)], 1u);
        }
      }
      __syncthreads();
      // find the bin holding rank (K-1-below) among keys matching the prefix
      const uint32_t target = (uint32_t)(K - 1) - below;
      const int per = ORD_BINS / ORD_THREADS;
      uint32_t loc = 0;
      for (int j = 0; j < per; j++) loc += hist[threadIdx.x * per + j];
      // inclusive scan of the per-thread sums: shuffles inside a wavefront, the wavefront totals through LDS
      uint32_t incl = loc;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if ((threadIdx.x & 63) >= (unsigned)off) incl += v;
      }
      if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = incl;
      __syncthreads();
      for (unsigned w = 0; w < (threadIdx.x >> 6); w++) incl += part[w];
      const uint32_t excl = incl - loc;
      if (target >= excl && target < incl) {
        uint32_t run = excl;
        for (int j = 0; j < per; j++) {
          const uint32_t c = hist[threadIdx.x * per + j];
          if (target < run + c) { s_bin = threadIdx.x * per + j; s_below = run; s_cnt = c; break; }
          run += c;
        }
      }
      __syncthreads();
      const uint32_t bin = s_bin, binbelow = s_below, bincnt = s_cnt;
      __syncthreads();
      below += binbelow;
      prefix |= (uint64_t)bin << sh;
      prefmask |= (uint64_t)(nb - 1) << sh;
      bound = prefix | (sh > 0 ? ((1ULL << sh) - 1) : 0ULL);
      if (below + bincnt <= (uint32_t)cap) {   // candidates (<= bound) fit the sort buffer
        if (lv == 0) {
          // Hashes are close to uniform, so the 2048 first-level bins hold a handful of keys each: give every bin up to the
          // cut its own slice of the buffer (exclusive prefix of the counts) and sort inside the bins only — O(n) instead
          // of the bitonic network.  Skewed inputs (a bin with more than ORD_BUCKET_MAX keys) keep the network.
          uint32_t run = excl, mx = 0;
          for (int j = 0; j < per; j++) {
            const uint32_t b = threadIdx.x * per + j, c = hist[b];
            if (b <= bin) { bstart[b] = (uint16_t)run; mx = c > mx ? c : mx; }
            run += c;
          }
          if (mx > ORD_BUCKET_MAX) atomicMax(&s_fill, mx);
          __syncthreads();
          buckets = s_fill == 0;
          bk_bin = bin; bk_total = below + bincnt;
          __syncthreads();
          if (threadIdx.x == 0) s_fill = 0;
        }
        break;
      }
    }
  }
  if (buckets) {
    for (uint32_t b = threadIdx.x; b <= bk_bin; b += ORD_THREADS) hist[b] = 0;   // becomes the bins' fill counters
    __syncthreads();
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * ORD_THREADS) {
      int32_t hv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u * ORD_THREADS; hv[u] = i < n ? hget(i) : 0; }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + u * ORD_THREADS;
        const uint64_t key = okey(hv[u], i);
        const uint32_t b = (uint32_t)(key >> 53);
        if (i < n && b <= bk_bin) buf[(uint32_t)bstart[b] + atomicAdd(&hist[b], 1u)] = key;
      }
    }
    __syncthreads();
    int32_t* orow = out_rows + strand * out_stride;
    for (uint32_t t = threadIdx.x; t < bk_total; t += ORD_THREADS) {   // rank inside the bin = place in the row (see the one-pass path)
      const uint64_t key = buf[t];
      const uint32_t b = (uint32_t)(key >> 53), s0 = bstart[b], c = hist[b];
      uint32_t rank = 0;
      for (uint32_t q = 0; q < c; q += 4) {   // four independent reads per trip (the bin's slice is followed by valid buffer words)
        uint64_t o[4];
#pragma unroll
        for (int x = 0; x < 4; x++) o[x] = buf[s0 + q + (uint32_t)x < (uint32_t)cap ? s0 + q + (uint32_t)x : s0];
#pragma unroll
        for (int x = 0; x < 4; x++) rank += (q + (uint32_t)x < c && o[x] < key) ? 1u : 0u;
      }
      const uint32_t j = s0 + rank;
      if (j < (uint32_t)K) { orow[2 * j] = (int32_t)((uint32_t)(key >> 32) ^ 0x80000000u); orow[2 * j + 1] = (int32_t)(uint32_t)key; }
    }
    return;
  }
  // compact candidates into LDS, pad, sort
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += ORD_THREADS) {
    const uint64_t key = okey(hget(i), i);
    if (key <= bound) { uint32_t slot = atomicAdd(&s_fill, 1u); if (slot < (uint32_t)cap) buf[slot] = key; }
  }
  __syncthreads();
  const uint32_t m = s_fill < (uint32_t)cap ? s_fill : (uint32_t)cap;
  uint32_t np2 = 1;
  while (np2 < m) np2 <<= 1;
  for (uint32_t j = m + threadIdx.x; j < np2; j += ORD_THREADS) buf[j] = ~0ULL;
  __syncthreads();
  for (uint32_t size = 2; size <= np2; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t t = threadIdx.x; t < (np2 >> 1); t += ORD_THREADS) {
        const uint32_t lo = 2 * t - (t & (stride - 1));
        const uint32_t hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const uint64_t a = buf[lo], b = buf[hi];
        if ((a > b) == up) { buf[lo] = b; buf[hi] = a; }
      }
      __syncthreads();
    }
  }
  int32_t* orow = out_rows + strand * out_stride;
  for (int j = threadIdx.x; j < K; j += ORD_THREADS) {
    const uint64_t key = buf[j];
    orow[2 * j] = (int32_t)((uint32_t)(key >> 32) ^ 0x80000000u);
    orow[2 * j + 1] = (int32_t)(uint32_t)key;
  }
}

// code_words = dwords of base codes staged per strand (0: every strand of the launch is MHAP_RD_MAT)
#ifndef MH_ORD_PAD
#define MH_ORD_PAD 0   // unused LDS bytes per workgroup (occupancy experiments)
#endif
size_t ordered_lds_bytes(int cap, int code_words, int stage_wide) {
  return (size_t)ORD_BINS * 4 + (size_t)ORD_THREADS * 4 + 16 + (size_t)cap * 8 + (size_t)(ORD_BINS + 2) * 2 + (size_t)cap * (stage_wide ? 4 : 2) +
         (code_words > 0 ? (size_t)256 * 8 + (size_t)code_words * 4 : 0) + 8 + MH_ORD_PAD;
}

// A read whose forward sketch throws ZeroNGramsFoundException is dropped entirely
// (J/impl/SequenceSketchStreamer.java:123-156,235-238): propagate the forward status to the rc entry.
__global__ void fix_status_kernel(int32_t* __restrict__ meta, int64_t nreads) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nreads) return;
  const int32_t fs = meta[(2 * i) * META_W + 3];
  if (fs != 0 && meta[(2 * i + 1) * META_W + 3] == 0) meta[(2 * i + 1) * META_W + 3] = fs;
}

void launch_fix_status(hipStream_t st, int32_t* meta, int64_t nreads) {
  if (nreads <= 0) return;
  hipLaunchKernelGGL(fix_status_kernel, dim3((unsigned)((nreads + 255) / 256)), dim3(256), 0, st, meta, nreads);
}

// max_len_codes = longest read of the launch that is not MHAP_RD_MAT (0 if there is none): sizes the LDS code stream;
// max_len = longest read of the launch
void launch_ordered(hipStream_t st, const ReadDesc* descs, int64_t nstrands, int max_len_codes, int max_len, const int32_t* h32,
                    const uint8_t* store, const uint64_t* luts, int k2, int S, int cap, int32_t* out_rows, int64_t out_stride,
                    int32_t* out_meta, int64_t meta_stride, int64_t first, int64_t count) {
  // strands [first, first + count) of the launch's nstrands (count < 0: all of them from `first` on)
  if (count < 0 || first + count > nstrands) count = nstrands - first;
  if (nstrands <= 0 || count <= 0) return;
  const int code_words = max_len_codes > 0 ? (max_len_codes + 15) / 16 + 4 : 0;
  const int stage_wide = max_len - k2 + 1 > 65535 ? 1 : 0;
  hipLaunchKernelGGL(ordered_kernel, dim3((unsigned)count), dim3(ORD_THREADS), ordered_lds_bytes(cap, code_words, stage_wide), st, descs, nstrands, h32,
                     store, luts, code_words, stage_wide, k2, S, cap, out_rows, out_stride, out_meta, meta_stride, first);
#ifdef MH_ORD_PROF
  (void)hipStreamSynchronize(st);
  unsigned long long hp[8];
  (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_ord_prof), sizeof(hp));
  fprintf(stderr, "[ordered prof] cap %d lds %zu strands(one-pass) %llu  100MHz ticks/strand: stage %.1f pass1 %.1f scan %.1f fill %.1f isort %.1f out %.1f\n", cap,
          ordered_lds_bytes(cap, code_words, stage_wide), hp[7], (double)hp[0] / hp[7], (double)hp[1] / hp[7], (double)hp[2] / hp[7], (double)hp[3] / hp[7],
          (double)hp[4] / hp[7], (double)hp[5] / hp[7]);
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ord_prof), z, sizeof(z));
#endif
}

}  // namespace mhap
