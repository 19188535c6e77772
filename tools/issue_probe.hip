// What does a gfx950 SIMD really issue, and what does an instruction COST?  Streams of full-rate VALU instructions in several
// encodings, every wave resident for the whole launch (N workgroups of 256 lanes per CU), timed twice: wall clock (events) and the
// shader clock every wave reads itself (s_memtime).  Finding of round 5 (profiles/r05_issue_probe.txt): the chip sustains a constant
// number of lane-operations per second for a given instruction whatever the occupancy — with more waves per SIMD the clocks per
// instruction go down and the shader clock goes down with them — so a VALU-dense kernel is bounded by the sum of its instructions'
// costs (1 / that rate), not by issue cycles.  The second table prices other instructions by mixing them into a v_xor_b32 stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/issue_probe.hip -o tools/bin/issue_probe && tools/bin/issue_probe [iters]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define REP8(x) x x x x x x x x

// 16 registers a0..a15; one "round" = 16 instructions; no read of a value written fewer than ~10 instructions ago unless stated
#define X32(d, s)      asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(a[d]) : "v"(a[s]));
#define X64(d, s)      asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a[d]) : "v"(a[s]));
#define XS0(d, s)      asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[d]) : "s"(sc));
#define B3(d, s, t)    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[d]) : "v"(a[s]), "v"(a[t]));
#define B3S0(d, s, t)  asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0xEA" : "+v"(a[d]) : "s"(sc), "v"(a[t]));
#define B3S1(d, s, t)  asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xEA" : "+v"(a[d]) : "s"(sc), "v"(a[t]));
#define B3S2(d, s, t)  asm volatile("v_bitop3_b32 %0, %0, %2, %1 bitop3:0xEA" : "+v"(a[d]) : "s"(sc), "v"(a[t]));
#define B3SAME(d, s, t) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a[d]) : "v"(a[s]));
#define MOV(d, s)      asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[d]) : "v"(a[s]));
#define CMP(d, s)      asm volatile("v_cmp_ne_u32_e32 vcc, %0, %1" : : "v"(a[d]), "v"(a[s]) : "vcc");
#define RDL(d, s)      asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(sc) : "v"(a[d]));
#define AND32(d, s)    asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(a[d]) : "v"(a[s]));
#define ALB(d, s, t)   asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[d]) : "v"(a[s]));
#define SBFE           asm volatile("s_bfe_i32 %0, %1, 0x10005" : "=s"(s2) : "s"(sc) : "scc");
#define SADD           asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(sc) : "scc");
#define MOVREL(d)      asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC0)\n\tv_mov_b32_e32 %0, %1\n\ts_set_gpr_idx_off" : "=v"(a[d]) : "v"(a[(d + 7) & 15]), "s"(szero) : "m0");
#define CMP64(d, s)    asm volatile("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(s64) : "v"(a[d]), "v"(a[s]));
#define SAVEEXEC       asm volatile("s_and_saveexec_b64 %0, vcc\n\ts_mov_b64 exec, %0" : "=s"(s64) : : "scc");
#define FFBL(d, s)     asm volatile("v_ffbl_b32_e32 %0, %1" : "=v"(a[d]) : "v"(a[s]));
#define ADDI(d)        asm volatile("v_add_u32_e32 %0, 32, %0" : "+v"(a[d]));
#define BCNT(d, s)     asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %1" : "=v"(a[d]) : "v"(a[s]));
#define DSW            asm volatile("ds_write_b32 %0, %1" : : "v"(ldsaddr), "v"(a[3]) : "memory");
#define GST            asm volatile("global_store_dword %0, %1, %2" : : "v"(goff), "v"(a[5]), "s"(gbase) : "memory");

#define R16(M, ...) M(0,1,6) M(1,2,7) M(2,3,8) M(3,4,9) M(4,5,10) M(5,6,11) M(6,7,12) M(7,8,13) M(8,9,14) M(9,10,15) M(10,11,0) M(11,12,1) M(12,13,2) M(13,14,3) M(14,15,4) M(15,0,5)
#define X32_3(d, s, t) X32(d, s)
#define X64_3(d, s, t) X64(d, s)
#define XS0_3(d, s, t) XS0(d, s)
#define MOV_3(d, s, t) MOV(d, t)
#define AND_3(d, s, t) AND32(d, s)
#define CMP_3(d, s, t) CMP(d, s)
#define ROUND_MIX X32(0,1) X32(1,2) B3(2,3,8) X32(3,4) X32(4,5) B3(5,6,11) X32(6,7) X32(7,8) B3(8,9,14) X32(9,10) X32(10,11) B3(11,12,1) X32(12,13) X32(13,14) B3(14,15,4) X32(15,0)
#define ROUND_DEP X32(0,15) X32(1,0) X32(2,1) X32(3,2) X32(4,3) X32(5,4) X32(6,5) X32(7,6) X32(8,7) X32(9,8) X32(10,9) X32(11,10) X32(12,11) X32(13,12) X32(14,13) X32(15,14)
// 15 x e32 + ONE of something else (its price = what the round costs more than 15/16 of an all-e32 round)
#define R15 X32(0,1) X32(1,2) X32(2,3) X32(3,4) X32(4,5) X32(5,6) X32(6,7) X32(7,8) X32(8,9) X32(9,10) X32(10,11) X32(11,12) X32(12,13) X32(13,14) X32(14,15)

__device__ unsigned long long g_clk[2];

template <int MODE>
__global__ __launch_bounds__(256) void k_issue(uint32_t* out, int iters, uint32_t* sink) {
  extern __shared__ uint32_t pad[];
  uint32_t a[16];
#pragma unroll
  for (int j = 0; j < 16; j++) a[j] = (threadIdx.x + 17u * blockIdx.x + 1u) * 2654435761u * (uint32_t)(j + 3);
  uint32_t sc = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 2654435761u)), s2 = sc;
  unsigned long long s64 = 0;
  uint32_t szero = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x >> 30));
  const uint32_t ldsaddr = threadIdx.x * 4u;
  const uint32_t goff = threadIdx.x * 4u;
  const uint64_t gbase = (uint64_t)(uintptr_t)(sink + (size_t)blockIdx.x * 256);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {   // 128 instructions per trip (+ loop control)
    if (MODE == 0) { REP8(R16(X32_3)) }
    if (MODE == 1) { REP8(R16(X64_3)) }
    if (MODE == 2) { REP8(R16(B3)) }
    if (MODE == 3) { REP8(ROUND_MIX) }
    if (MODE == 4) { REP8(ROUND_DEP) }
    if (MODE == 5) { REP8(R16(B3S0)) }
    if (MODE == 6) { REP8(R16(B3S1)) }
    if (MODE == 7) { REP8(R16(B3S2)) }
    if (MODE == 8) { REP8(R16(XS0_3)) }
    if (MODE == 9) { REP8(R16(MOV_3)) }
    if (MODE == 10) { REP8(R16(AND_3)) }
    if (MODE == 11) { REP8(R16(CMP_3)) }
    if (MODE == 12) { REP8(R16(ALB)) }
    if (MODE == 13) { REP8(R16(B3SAME)) }
    if (MODE == 20) { REP8(R15 RDL(15, 0)) }
    if (MODE == 21) { REP8(R15 SBFE) }
    if (MODE == 22) { REP8(R15 SBFE SADD SBFE SADD) }
    if (MODE == 23) { REP8(R15 DSW) }
    if (MODE == 24) { REP8(R15 GST) }
    if (MODE == 25) { REP8(R15 CMP(15, 0)) }
    if (MODE == 26) { REP8(R15) }
    if (MODE == 27) { REP8(R15 MOVREL(15)) }
    if (MODE == 28) { REP8(R15 CMP64(15, 0)) }
    if (MODE == 29) { REP8(R15 SAVEEXEC) }
    if (MODE == 30) { REP8(R15 FFBL(15, 0)) }
    if (MODE == 31) { REP8(R15 ADDI(15)) }
    if (MODE == 32) { REP8(R15 BCNT(15, 0)) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t s = s2 + (uint32_t)s64;
#pragma unroll
  for (int j = 0; j < 16; j++) s ^= a[j];
  if (s == 0x12345678u) pad[threadIdx.x] = s;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { atomicAdd(&g_clk[0], t1 - t0); atomicAdd(&g_clk[1], 1ULL); }
}

static double g_base = 0.0;   // seconds per wave-instruction-unit of the all-e32 stream (set by the first run)
template <int MODE> void run(const char* name, uint32_t* d, uint32_t* sink, int wgs_per_cu, int iters, int valu_per_round = 16, int extra_per_round = 0) {
  const size_t lds = wgs_per_cu >= 8 ? 4096 : (size_t)(160 * 1024 / wgs_per_cu) - 1024;
  const int blocks = 256 * wgs_per_cu;
  (void)hipFuncSetAttribute((const void*)k_issue<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_issue<MODE>, dim3(blocks), dim3(256), lds, 0, d, 64, sink); (void)hipDeviceSynchronize();
  unsigned long long z[2] = {0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof z);
  (void)hipEventRecord(e0); hipLaunchKernelGGL(k_issue<MODE>, dim3(blocks), dim3(256), lds, 0, d, iters, sink); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_clk), sizeof z);
  const double wave_clk = (double)z[0] / (double)z[1];              // shader clocks one wave spent in the loop
  const double rounds = (double)iters * 8.0 * blocks * 4.0;          // rounds of all waves
  const double per_round = ms * 1e-3 / rounds;                       // chip seconds per round
  if (MODE == 0 && wgs_per_cu == 4) g_base = per_round / 16.0;
  printf("%-46s %d w/SIMD %8.2f ms  clock %4.0f MHz  %.3f clk/VALU/SIMD  %.3e VALU lane-ops/s", name, wgs_per_cu, ms, wave_clk / (ms * 1e3),
         wave_clk / ((double)iters * 8.0 * valu_per_round * wgs_per_cu), rounds * valu_per_round * 64.0 / (ms * 1e-3));
  if (g_base > 0.0) {
    if (extra_per_round > 0) printf("  | price of the extra: %.2f e32 each", (per_round / g_base - 15.0) / extra_per_round);
    else printf("  | price: %.2f e32", per_round / g_base / valu_per_round);
  }
  printf("\n");
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int iters = argc > 1 ? atoi(argv[1]) : 100000;   // ~50 ms per launch at 4 waves per SIMD: long enough for the clock to settle
  uint32_t* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  uint32_t* sink; (void)hipMalloc(&sink, 256 * 8 * 256 * 4);
  for (int w : {4, 4, 8, 2}) {
    run<0>("v_xor_b32_e32 (4 bytes)", d, sink, w, iters);
    run<1>("v_xor_b32_e64 (8 bytes)", d, sink, w, iters);
    run<2>("v_bitop3_b32, 3 VGPR sources", d, sink, w, iters);
    run<3>("mix: 11 x e32 + 5 x bitop3", d, sink, w, iters);
    run<4>("e32, every one reads the last", d, sink, w, iters);
  }
  const int w = 4;
  run<5>("v_bitop3_b32, SGPR src0 + 2 VGPR", d, sink, w, iters);
  run<6>("v_bitop3_b32, SGPR src1 + 2 VGPR", d, sink, w, iters);
  run<7>("v_bitop3_b32, SGPR src2 + 2 VGPR", d, sink, w, iters);
  run<13>("v_bitop3_b32, 2 distinct VGPR sources", d, sink, w, iters);
  run<8>("v_xor_b32_e32 v, s, v", d, sink, w, iters);
  run<9>("v_mov_b32_e32", d, sink, w, iters);
  run<10>("v_and_b32_e32", d, sink, w, iters);
  run<11>("v_cmp_ne_u32_e32 vcc", d, sink, w, iters);
  run<12>("v_alignbit_b32", d, sink, w, iters);
  run<26>("15 x e32 alone", d, sink, w, iters, 15, 0);
  run<20>("15 x e32 + v_readlane_b32", d, sink, w, iters, 15, 1);
  run<21>("15 x e32 + s_bfe_i32", d, sink, w, iters, 15, 1);
  run<22>("15 x e32 + 4 SALU", d, sink, w, iters, 15, 4);
  run<23>("15 x e32 + ds_write_b32", d, sink, w, iters, 15, 1);
  run<24>("15 x e32 + global_store_dword", d, sink, w, iters / 16, 15, 1);
  run<25>("15 x e32 + v_cmp_ne_u32 vcc", d, sink, w, iters, 15, 1);
  run<27>("15 x e32 + gpr_idx on / v_mov / off", d, sink, w, iters, 15, 1);
  run<28>("15 x e32 + v_cmp_ne_u32_e64 -> SGPR pair", d, sink, w, iters, 15, 1);
  run<29>("15 x e32 + s_and_saveexec + s_mov exec", d, sink, w, iters, 15, 1);
  run<30>("15 x e32 + v_ffbl_b32", d, sink, w, iters, 15, 1);
  run<31>("15 x e32 + v_add_u32 v, 32, v", d, sink, w, iters, 15, 1);
  run<32>("15 x e32 + v_mbcnt_lo", d, sink, w, iters, 15, 1);
  return 0;
}
