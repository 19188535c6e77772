// Full-rate VALU issue vs waves per SIMD on gfx950: the same v_xor_b32 stream (8 independent chains per lane) with the
// workgroups per CU limited through dynamic LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k_xor(uint32_t* out, int iters) {
  extern __shared__ uint32_t pad[];
  uint32_t a[8];
#pragma unroll
  for (int j = 0; j < 8; j++) a[j] = threadIdx.x * 2654435761u + j;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int j = 0; j < 8; j++) asm("v_xor_b32 %0, %0, %1" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
  }
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) s ^= a[j];
  if (s == 0x12345678u) pad[threadIdx.x] = s;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  const int iters = 4096;
  uint32_t* d; hipMalloc(&d, 256 * 64 * 256 * 4);
  hipFuncSetAttribute((const void*)k_xor, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int wgs_per_cu : {8, 4, 3, 2, 1}) {
    const size_t lds = wgs_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / wgs_per_cu) - 1024;
    const int blocks = 256 * wgs_per_cu * 4;   // four rounds of resident workgroups
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_xor, dim3(blocks), dim3(256), lds, 0, d, 16); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k_xor, dim3(blocks), dim3(256), lds, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("waves/SIMD %d  %8.3f ms  %.3e lane-ops/s\n", wgs_per_cu, ms, (double)blocks * 256 * iters * 64 / (ms * 1e-3));
  }
  return 0;
}
