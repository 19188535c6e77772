// Integer-VALU issue-rate microbenchmark for gfx950: how many 32-bit integer lane-ops/s does the chip sustain?
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak ; run: /tmp/valu_peak
// Used to calibrate the ceiling the MinHash kernel's xorshift steps are priced against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
  uint32_t a[8];
#pragma unroll
  for (int j = 0; j < 8; j++) a[j] = threadIdx.x * 2654435761u + j;
  uint64_t x[4];
#pragma unroll
  for (int j = 0; j < 4; j++) x[j] = ((uint64_t)a[j] << 32) | a[j + 4];
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = (a[j] ^ (a[(j + 1) & 7] >> 3));   // v_lshrrev + v_xor : 2 ops
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = __builtin_amdgcn_alignbit(a[j], a[(j + 1) & 7], 11) ^ a[(j + 3) & 7];   // 2 ops
    } else if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) { x[j] ^= x[j] << 21; x[j] ^= x[j] >> 35; x[j] ^= x[j] << 4; }   // one xorshift64 step
    } else if (MODE == 4) {
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          uint32_t t;
          asm("v_lshrrev_b32 %2, 11, %0\n\tv_lshl_or_b32 %2, %1, 21, %2\n\tv_xor_b32 %1, %1, %2\n\t"
              "v_lshlrev_b32 %2, 21, %0\n\tv_xor_b32 %0, %0, %2\n\t"
              "v_lshrrev_b32 %2, 3, %1\n\tv_xor_b32 %0, %0, %2\n\t"
              "v_lshrrev_b32 %2, 28, %0\n\tv_lshl_or_b32 %2, %1, 4, %2\n\tv_xor_b32 %1, %1, %2\n\t"
              "v_lshlrev_b32 %2, 4, %0\n\tv_xor_b32 %0, %0, %2"
              : "+v"(a[j]), "+v"(a[j + 4]), "=&v"(t));
        }
    } else if (MODE == 5) {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) asm("v_lshl_or_b32 %0, %1, 21, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
    } else {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) { float f = __uint_as_float(a[j]); f = fmaf(f, 1.0001f, 0.5f); a[j] = __float_as_uint(f); }   // v_fma_f32
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) s ^= a[j];
#pragma unroll
  for (int j = 0; j < 4; j++) s ^= (uint32_t)x[j] ^ (uint32_t)(x[j] >> 32);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
double run(const char* name, double ops_per_iter_lane) {
  const int blocks = 256 * 8, iters = 4096;
  uint32_t* d; hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 16);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double lane_ops = (double)blocks * 256 * iters * ops_per_iter_lane;
  printf("%-28s %8.3f ms  %.3e lane-ops/s\n", name, ms, lane_ops / (ms * 1e-3));
  hipFree(d);
  return lane_ops / (ms * 1e-3);
}

int main() {
  run<0>("v_lshrrev+v_xor (2 ops)", 8 * 8 * 2);
  run<1>("v_alignbit+v_xor (2 ops)", 8 * 8 * 2);
  double s = run<2>("xorshift64 step (as 1 op)", 16 * 4);
  printf("  -> %.3e xorshift64 steps/s chip-wide (compiler's 64-bit shift lowering)\n", s);
  run<3>("v_fma_f32 (1 op)", 8 * 8);
  double s4 = run<4>("xorshift64 12-op asm (1 op)", 16 * 4);
  printf("  -> %.3e xorshift64 steps/s chip-wide (12 full-rate 32-bit ops per step)\n", s4);
  run<5>("v_lshl_or_b32 (1 op)", 8 * 8);
  return 0;
}
