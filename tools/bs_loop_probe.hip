// Where does the MinHash slot loop lose its issue slots?  The kernel's own step / filter code (included from the source file),
// stripped to variants: A = step only, B = + fixed-depth filter and the (rarely taken) trigger branch, C = + per-slot depth
// (v_readlane + scalar enable words), D = + queue append of the candidates.  Prints wave-steps/s against the VALU ceiling.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bs_loop_probe.hip -o tools/bs_loop_probe && tools/bs_loop_probe
#include "../mhap_amd/csrc/sketch_kernels.hip"
#include <cstdio>
using namespace mhap;

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void probe(uint32_t* out, int rows, int H, const int32_t* thr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t* q = (uint32_t*)(smem + wv * 6400 + 4104);
  int32_t* besthi = (int32_t*)(smem + wv * 6400);
  for (int s = lane; s < H; s += 64) { besthi[2 * s] = 0; besthi[2 * s + 1] = thr[s]; }
  __builtin_amdgcn_wave_barrier();
  uint32_t P[64];
#pragma unroll
  for (int b = 0; b < 64; b++) P[b] = (threadIdx.x + 1) * 2654435761u * (b + 3) + blockIdx.x * 40503u + b;
  uint32_t ACT = 0xFFFFFFFFu;
  uint32_t acc = 0;
  int qn = 0;
  for (int r = 0; r < rows; r++) {
    int vz = 0;
    for (int s = 0; s < H; s++) {
      int zs = 12;
      if (MODE >= 2) {
        if ((s & 63) == 0) vz = bs_depth(besthi[2 * (s + lane < H ? s + lane : H - 1) + 1]);
        zs = __builtin_amdgcn_readlane(vz, s & 63);
      }
      const BsEnable en = bs_enable(zs);
      bs_step(P);
      if (MODE >= 1) {
        uint32_t nacc = bs_filter(P, ACT, en, zs);
        if (__any(nacc != 0xFFFFFFFFu)) {
          if (MODE >= 3) {
            uint32_t cand = ~nacc;
            const uint32_t head = ((uint32_t)s << 17) | ((uint32_t)lane << 5);
            unsigned long long m = __ballot(cand != 0u);
            do {
              const int n = __popcll(m);
              if (qn + n > 512) qn = 0;
              if (cand) {
                const int idx = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                q[idx] = head | (uint32_t)__builtin_ctz(cand);
                cand &= cand - 1u;
              }
              qn += n;
              m = __ballot(cand != 0u);
            } while (m);
          } else acc += nacc;
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 64; b++) acc ^= P[b];
  out[blockIdx.x * 256 + threadIdx.x] = acc + (uint32_t)qn;
}


// ---- candidate loop structure for the kernel: all-bitop3 filter (full rate), mask bits precomputed per lane, shallow slots
// routed through the trigger, nested 64-slot blocks, trigger path marked unlikely
__device__ __forceinline__ uint32_t vm_of(int32_t bhs) {
  const int z = bs_depth(bhs);
  return z < 0 ? 0u : ((2u << z) - 1u);
}
__device__ __forceinline__ uint32_t sbit(uint32_t m, int b) {   // bit b of a scalar as a mask word
  uint32_t r = (uint32_t)((int)(m << (31 - b)) >> 31);
  asm volatile("" : "+s"(r));
  return r;
}
__device__ __forceinline__ uint32_t hot_filter(const uint32_t (&P)[64], uint32_t nACT, uint32_t sm) {
  const uint32_t keep = sbit(sm, 8), e9 = sbit(sm, 9), e10 = sbit(sm, 10), e11 = sbit(sm, 11), e12 = sbit(sm, 12);
  uint32_t n0 = __builtin_amdgcn_bitop3_b32(P[62], P[61], P[60], 0xFE);
  uint32_t t1 = __builtin_amdgcn_bitop3_b32(P[59], P[58], P[57], 0xFE);
  n0 = __builtin_amdgcn_bitop3_b32(n0, P[56], P[55], 0xFE);
  uint32_t n1 = __builtin_amdgcn_bitop3_b32(t1, P[63], nACT, 0xFB);
  n0 = __builtin_amdgcn_bitop3_b32(P[54], e9, n0, 0xEA);
  n1 = __builtin_amdgcn_bitop3_b32(P[53], e10, n1, 0xEA);
  n0 = __builtin_amdgcn_bitop3_b32(P[52], e11, n0, 0xEA);
  n1 = __builtin_amdgcn_bitop3_b32(P[51], e12, n1, 0xEA);
  return __builtin_amdgcn_bitop3_b32(n0, n1, keep, 0xA8);
}
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void probe2(uint32_t* out, int rows, int H, const int32_t* thr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t* q = (uint32_t*)(smem + wv * 6400 + 4104);
  int32_t* besthi = (int32_t*)(smem + wv * 6400);
  for (int s = lane; s < H; s += 64) { besthi[2 * s] = 0; besthi[2 * s + 1] = thr[s]; }
  __builtin_amdgcn_wave_barrier();
  uint32_t P[64];
#pragma unroll
  for (int b = 0; b < 64; b++) P[b] = (threadIdx.x + 1) * 2654435761u * (b + 3) + blockIdx.x * 40503u + b;
  uint32_t ACT = 0xFFFFFFFFu;
  asm volatile("" : "+v"(ACT));
  const uint32_t nACT = ~ACT;
  uint32_t acc = 0;
  int qn = 0;
  for (int r = 0; r < rows; r++) {
    for (int s0 = 0; s0 < H; s0 += 64) {
      const uint32_t vm = vm_of(besthi[2 * (s0 + lane < H ? s0 + lane : H - 1) + 1]);
      const int tn = H - s0 < 64 ? H - s0 : 64;
      for (int t = 0; t < tn; t++) {
        const uint32_t sm = (uint32_t)__builtin_amdgcn_readlane((int)vm, t);
        bs_step(P);
        uint32_t nacc = hot_filter(P, nACT, sm);
        if (__builtin_expect(__any(nacc != 0xFFFFFFFFu), 0)) {
          const int s = s0 + t;
          if (MODE >= 1) {
            uint32_t cand = ~nacc;
            const uint32_t head = ((uint32_t)s << 17) | ((uint32_t)lane << 5);
            unsigned long long m = __ballot(cand != 0u);
            do {
              const int n = __popcll(m);
              if (qn + n > 512) qn = 0;
              if (cand) {
                const int idx = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                q[idx] = head | (uint32_t)__builtin_ctz(cand);
                cand &= cand - 1u;
              }
              qn += n;
              m = __ballot(cand != 0u);
            } while (m);
          } else acc += nacc + (uint32_t)s;
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 64; b++) acc ^= P[b];
  out[blockIdx.x * 256 + threadIdx.x] = acc + (uint32_t)qn;
}
template <int MODE> void run2(const char* name, const int32_t* dthr, int wgs_per_cu) {
  const int rows = 40, H = 512;
  const int blocks = 256 * wgs_per_cu;
  uint32_t* d; (void)hipMalloc(&d, blocks * 256 * 4);
  const size_t lds = wgs_per_cu >= 6 ? 25600 : wgs_per_cu == 5 ? 31 * 1024 : wgs_per_cu == 4 ? 39 * 1024 : 52 * 1024;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(probe2<MODE>, dim3(blocks), dim3(256), lds, 0, d, 2, H, dthr); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); hipLaunchKernelGGL(probe2<MODE>, dim3(blocks), dim3(256), lds, 0, d, rows, H, dthr); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double wave_steps = (double)blocks * 4 * rows * H;
  printf("%-44s %d WG/CU %8.3f ms  %.3e wave-steps/s  = %.1f ns per step per SIMD\n", name, wgs_per_cu, ms, wave_steps / (ms * 1e-3), ms * 1e6 * 1024 / wave_steps);
  (void)hipFree(d);
}

template <int MODE> void run(const char* name, const int32_t* dthr, int wgs_per_cu) {
  const int rows = 40, H = 512;
  const int blocks = 256 * wgs_per_cu;
  uint32_t* d; hipMalloc(&d, blocks * 256 * 4);
  const size_t lds = wgs_per_cu >= 6 ? 25600 : wgs_per_cu == 5 ? 31 * 1024 : wgs_per_cu == 4 ? 39 * 1024 : 52 * 1024;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), lds, 0, d, 2, H, dthr); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), lds, 0, d, rows, H, dthr); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double wave_steps = (double)blocks * 4 * rows * H;
  // ceiling: 107 step ops (+ 10 filter) at 2 cycles per wave-instruction on 1024 SIMDs
  printf("%-44s %d WG/CU %8.3f ms  %.3e wave-steps/s  = %.1f ns per step per SIMD  (%.0f cycles at 2.25 GHz)\n", name, wgs_per_cu, ms, wave_steps / (ms * 1e-3),
         ms * 1e6 * 1024 / wave_steps, ms * 1e-3 * 1024 / wave_steps * 2.25e9);
  hipFree(d);
}

int main() {
  int32_t h[512];
  for (int i = 0; i < 512; i++) h[i] = (int32_t)(0x80000000u | (0x00040000u >> (i % 3)) | 0x1234u);   // negative minima, 12..14 leading zero magnitude bits
  int32_t* dthr; hipMalloc(&dthr, sizeof h); hipMemcpy(dthr, h, sizeof h, hipMemcpyHostToDevice);
  for (int wg = 4; wg <= 6; wg++) {
    run<0>("A step only (107 ops)", dthr, wg);
    run<1>("B + filter depth 12 + trigger branch", dthr, wg);
    run<2>("C + per-slot depth (readlane, enable words)", dthr, wg);
    run<3>("D + queue append", dthr, wg);
    run2<0>("E new loop, trigger -> trivial", dthr, wg);
    run2<1>("F new loop + queue append", dthr, wg);
  }
  return 0;
}
