// How fast can wavefronts stream randomly chosen 12-KB rows (the second stage's access pattern: one ordered-sketch row per candidate pair)?
// One wave per row at a time, rows picked by a hash of the pair number, 20 waves per CU resident; 8-byte loads per lane (what the join
// kernel issues: entry j = block * 64 + lane) against 16-byte loads (two entries per lane), with 3 or 6 loads in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/row_gather_probe.hip -o tools/bin/row_gather_probe && tools/bin/row_gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ROW_BYTES = 12288;
__device__ __forceinline__ unsigned mixu(unsigned x) { x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x; }
template <typename V, int INFL>
__global__ __launch_bounds__(256) void gather(const char* __restrict__ rows, unsigned nrows, unsigned long long npairs, unsigned long long* work, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  constexpr int PER = ROW_BYTES / (64 * (int)sizeof(V));   // loads per lane and row
  unsigned acc = 0;
  for (;;) {
    unsigned long long c = 0;
    if (lane == 0) c = atomicAdd(work, 8ULL);
    c = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(c >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)c);
    if (c >= npairs) break;
    for (int k = 0; k < 8 && c + k < npairs; k++) {
      const unsigned r = mixu((unsigned)(c + k) * 2654435761u) % nrows;
      const V* p = (const V*)(rows + (size_t)r * ROW_BYTES);
      V v[INFL];
#pragma unroll
      for (int u = 0; u < INFL; u++) v[u] = p[u * 64 + lane];
      for (int j0 = 0; j0 < PER; j0 += INFL) {
        V e[INFL];
#pragma unroll
        for (int u = 0; u < INFL; u++) { e[u] = v[u]; const int jn = j0 + INFL + u; if (jn < PER) v[u] = p[jn * 64 + lane]; }
#pragma unroll
        for (int u = 0; u < INFL; u++) { const unsigned* w = (const unsigned*)&e[u]; for (int q = 0; q < (int)(sizeof(V) / 4); q++) acc ^= w[q]; }
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <typename V, int INFL> static void run(const char* name, const char* rows, unsigned nrows, unsigned long long npairs, unsigned long long* work, unsigned* sink, int blocks) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int it = 0; it < 4; it++) {
    CK(hipMemset(work, 0, 8));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<V, INFL>), dim3(blocks), dim3(256), 0, 0, rows, nrows, npairs, work, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); if (it && ms < best) best = ms;
  }
  printf("%-34s rows %8u (%.2f GB)  %llu rows read in %.3f ms = %.2f TB/s\n", name, nrows, nrows * (double)ROW_BYTES / 1e9, npairs, best, npairs * (double)ROW_BYTES / best / 1e9);
}
int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int blocks = pr.multiProcessorCount * 5;   // 20 waves per CU
  unsigned long long* work; unsigned* sink; CK(hipMalloc(&work, 8)); CK(hipMalloc(&sink, 4));
  for (unsigned nrows : {10000u, 80000u, 200000u}) {   // 0.12 GB (cache-resident), 0.98 GB (the C5 slice), 2.46 GB (C2)
    char* rows; CK(hipMalloc(&rows, (size_t)nrows * ROW_BYTES)); CK(hipMemset(rows, 1, (size_t)nrows * ROW_BYTES));
    const unsigned long long npairs = 2000000ULL;
    run<uint2, 3>("8-byte loads, 3 + 3 in flight", rows, nrows, npairs, work, sink, blocks);
    run<uint2, 6>("8-byte loads, 6 + 6 in flight", rows, nrows, npairs, work, sink, blocks);
    run<uint4, 2>("16-byte loads, 2 + 2 in flight", rows, nrows, npairs, work, sink, blocks);
    run<uint4, 3>("16-byte loads, 3 + 3 in flight", rows, nrows, npairs, work, sink, blocks);
    CK(hipFree(rows));
  }
  return 0;
}
