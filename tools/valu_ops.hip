// Per-instruction issue-rate probe for gfx950 integer VALU ops (companion of valu_peak.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define DEFK(NAME, ASM)                                                                  \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters) {                 \
    uint32_t a[8];                                                                        \
    _Pragma("unroll") for (int j = 0; j < 8; j++) a[j] = threadIdx.x * 2654435761u + j;   \
    for (int i = 0; i < iters; i++) {                                                     \
      _Pragma("unroll") for (int r = 0; r < 8; r++)                                       \
      _Pragma("unroll") for (int j = 0; j < 8; j++) asm(ASM : "+v"(a[j]) : "v"(a[(j + 1) & 7]), "v"(a[(j + 3) & 7])); \
    }                                                                                     \
    uint32_t s = 0;                                                                       \
    _Pragma("unroll") for (int j = 0; j < 8; j++) s ^= a[j];                              \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                              \
  }
DEFK(k_xor, "v_xor_b32 %0, %0, %1")
DEFK(k_mad24, "v_mad_u32_u24 %0, %1, %2, %0")
DEFK(k_lshladd, "v_lshl_add_u32 %0, %1, 21, %0")
DEFK(k_lshlor, "v_lshl_or_b32 %0, %1, 21, %0")
DEFK(k_perm, "v_perm_b32 %0, %1, %0, %2")
DEFK(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
DEFK(k_andor, "v_and_or_b32 %0, %1, %2, %0")
DEFK(k_xad, "v_xad_u32 %0, %1, %2, %0")
DEFK(k_or3, "v_or3_b32 %0, %1, %2, %0")
DEFK(k_add3, "v_add3_u32 %0, %1, %2, %0")
DEFK(k_mul24, "v_mul_u32_u24 %0, %0, %1")
DEFK(k_alignbit, "v_alignbit_b32 %0, %1, %0, 11")
DEFK(k_fma, "v_fma_f32 %0, %1, %2, %0")
DEFK(k_min3, "v_min3_i32 %0, %1, %2, %0")
DEFK(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFK(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEFK(k_shl, "v_lshlrev_b32 %0, 21, %1")
DEFK(k_shr, "v_lshrrev_b32 %0, 11, %1")
DEFK(k_or, "v_or_b32 %0, %0, %1")
DEFK(k_mullo, "v_mul_lo_u32 %0, %0, %1")
DEFK(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
DEFK(k_sdwa, "v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
#define DEFK64(NAME, ASM)                                                                \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters) {                 \
    uint64_t a[8];                                                                        \
    _Pragma("unroll") for (int j = 0; j < 8; j++) a[j] = (threadIdx.x * 2654435761ull + j) * 0x9E3779B97F4A7C15ull;   \
    for (int i = 0; i < iters; i++) {                                                     \
      _Pragma("unroll") for (int r = 0; r < 8; r++)                                       \
      _Pragma("unroll") for (int j = 0; j < 8; j++) asm(ASM : "+v"(a[j]) : "v"(a[(j + 1) & 7])); \
    }                                                                                     \
    uint64_t s = 0;                                                                       \
    _Pragma("unroll") for (int j = 0; j < 8; j++) s ^= a[j];                              \
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);              \
  }
__global__ __launch_bounds__(256) void k_mad64(uint32_t* out, int iters) {
  uint64_t a[8]; uint32_t b[8];
  _Pragma("unroll") for (int j = 0; j < 8; j++) { a[j] = (threadIdx.x * 2654435761ull + j) * 0x9E3779B97F4A7C15ull; b[j] = threadIdx.x * 40503u + j; }
  for (int i = 0; i < iters; i++) {
    _Pragma("unroll") for (int r = 0; r < 8; r++)
    _Pragma("unroll") for (int j = 0; j < 8; j++) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[j]) : "v"(b[j]), "v"(b[(j + 1) & 7]) : "vcc");
  }
  uint64_t s = 0;
  _Pragma("unroll") for (int j = 0; j < 8; j++) s ^= a[j];
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
DEFK64(k_shl64, "v_lshlrev_b64 %0, 21, %1")
DEFK64(k_shr64, "v_lshrrev_b64 %0, 21, %1")
DEFK64(k_ashr64, "v_ashrrev_i64 %0, 21, %1")
template <class K> void run(const char* name, K kern) {
  const int blocks = 256 * 8, iters = 2048;
  uint32_t* d; hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 16); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-16s %8.3f ms  %.3e lane-ops/s\n", name, ms, (double)blocks * 256 * iters * 64 / (ms * 1e-3));
  hipFree(d);
}
int main() {
  run("v_xor_b32", k_xor); run("v_mad_u32_u24", k_mad24); run("v_lshl_add_u32", k_lshladd); run("v_lshl_or_b32", k_lshlor);
  run("v_perm_b32", k_perm); run("v_bfi_b32", k_bfi); run("v_and_or_b32", k_andor); run("v_xad_u32", k_xad); run("v_or3_b32", k_or3);
  run("v_add3_u32", k_add3); run("v_mul_u32_u24", k_mul24); run("v_alignbit_b32", k_alignbit); run("v_fma_f32", k_fma);
  run("v_min3_i32", k_min3); run("v_lshlrev_b64", k_shl64); run("v_lshrrev_b64", k_shr64); run("v_ashrrev_i64", k_ashr64); run("v_bitop3_b32", k_bitop3); run("v_lshlrev_b32", k_shl); run("v_lshrrev_b32", k_shr); run("v_or_b32", k_or); run("v_cndmask_b32", k_cndmask); run("v_xor_sdwa", k_sdwa);
  run("v_mul_lo_u32", k_mullo); run("v_mul_hi_u32", k_mulhi); run("v_mad_u64_u32", k_mad64);
  return 0;
}
