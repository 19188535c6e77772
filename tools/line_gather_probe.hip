// What does the chip do on the first query tier's access pattern — one random 64-byte line per lookup out of a table of 0.13 / 1.1 / 17 GB
// (one rank of eight at C2, one GPU at C2, one GPU at C4) — when nothing else is in the way?  Round 6: the line table took index_query from
// two dependent random accesses per lookup to one and bought 10 %; this probe says where the floor is and which way of issuing the loads
// reaches it.  PER-LANE: a lane reads its own line with four 16-byte loads (what index_query_kernel<..., LINES> does); QUAD: four lanes
// share a line, one 16-byte load each (a wave instruction touches 16 lines instead of 64); PAIR: two lanes, two loads each.
//   hipcc --offload-arch=gfx950 -O3 tools/line_gather_probe.hip -o tools/bin/line_gather_probe && tools/bin/line_gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ unsigned mixu(unsigned x) { x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x; }
// MODE 0 per-lane (4 x uint4), 1 quad (1 x uint4 per lane), 2 pair (2 x uint4 per lane); INFL = lookups (mode 0) / loads (1, 2) in flight per lane
template <int MODE, int INFL>
__global__ __launch_bounds__(64) void gather(const uint4* __restrict__ lines, unsigned nlines_mask, unsigned per_wave, unsigned* sink) {
  const unsigned lane = threadIdx.x, wave = blockIdx.x;
  unsigned acc = 0;
  constexpr int LPL = MODE == 0 ? 1 : (MODE == 1 ? 4 : 2);          // lanes per line
  const unsigned sub = lane / LPL, part = lane % LPL;
  const unsigned per_instr = 64 / LPL;                                  // lookups one wave instruction (group) covers
  for (unsigned i0 = 0; i0 < per_wave; i0 += per_instr * INFL) {
    uint4 v[INFL][4 / LPL];
#pragma unroll
    for (int u = 0; u < INFL; u++) {
      const unsigned q = wave * per_wave + i0 + u * per_instr + sub;
      const uint4* p = lines + (size_t)(mixu(q * 2654435761u) & nlines_mask) * 4;
#pragma unroll
      for (int w = 0; w < 4 / LPL; w++) v[u][w] = p[part * (4 / LPL) + w];
    }
#pragma unroll
    for (int u = 0; u < INFL; u++)
#pragma unroll
      for (int w = 0; w < 4 / LPL; w++) acc ^= v[u][w].x ^ v[u][w].y ^ v[u][w].z ^ v[u][w].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int MODE, int INFL> static void run(const char* name, const uint4* lines, unsigned nlines, unsigned* sink) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const unsigned waves = 100000, per_wave = 512;   // = the queries and --num-hashes of C2
  float best = 1e30f;
  for (int it = 0; it < 4; it++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<MODE, INFL>), dim3(waves), dim3(64), 0, 0, lines, nlines - 1, per_wave, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); if (it && ms < best) best = ms;
  }
  const double n = (double)waves * per_wave;
  printf("  %-44s %.3f ms  %.1f G lookups/s  %.2f TB/s of lines\n", name, best, n / best / 1e6, n * 64 / best / 1e9);
}
int main() {
  unsigned* sink; CK(hipMalloc(&sink, 4));
  for (unsigned lg : {21u, 24u, 28u}) {   // 2^21 lines = 134 MB, 2^24 = 1.07 GB, 2^28 = 17 GB
    const unsigned nlines = 1u << lg;
    uint4* lines; CK(hipMalloc(&lines, (size_t)nlines * 64)); CK(hipMemset(lines, 1, (size_t)nlines * 64));
    printf("table of %u lines = %.2f GB, 51.2 M lookups by 100 000 one-wave workgroups:\n", nlines, nlines * 64.0 / 1e9);
    run<0, 1>("per lane, 1 line in flight", lines, nlines, sink);
    run<0, 2>("per lane, 2 lines in flight", lines, nlines, sink);
    run<0, 4>("per lane, 4 lines in flight", lines, nlines, sink);
    run<0, 8>("per lane, 8 lines in flight", lines, nlines, sink);
    run<2, 4>("pair of lanes per line, 4 groups in flight", lines, nlines, sink);
    run<2, 8>("pair of lanes per line, 8 groups in flight", lines, nlines, sink);
    run<1, 4>("quad of lanes per line, 4 groups in flight", lines, nlines, sink);
    run<1, 8>("quad of lanes per line, 8 groups in flight", lines, nlines, sink);
    run<1, 16>("quad of lanes per line, 16 groups in flight", lines, nlines, sink);
    CK(hipFree(lines));
  }
  return 0;
}
