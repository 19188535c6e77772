// mhap_internal.hpp — what the translation units of libmhaphip.so share besides the kernel launch interface (kernels.hpp):
// the device-buffer helper and the narrow view of a handle that the multi-GPU module (mhap_dist.hip) works through.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>

#include "../../include/mhap_hip.h"

namespace mhap {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes, bool keep = false, hipStream_t st = nullptr) {
    if (bytes <= cap) return hipSuccess;
    size_t ncap = std::max(bytes, keep ? cap * 2 : cap);
    void* np = nullptr;
    hipError_t e = hipMalloc(&np, ncap);
    if (e != hipSuccess) return e;
    if (keep && p && cap) {
      e = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) { (void)hipFree(np); return e; }
    }
    if (p) (void)hipFree(p);
    p = np; cap = ncap;
    return hipSuccess;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return (T*)p; }
};


// The parts of a handle the multi-GPU exchange reads: the rank's own tables (entry 2j / 2j+1 = forward / reverse strand of its
// j-th read) and the host mirrors of ids and strands.  `dist` is an opaque slot owned by mhap_dist.hip (freed by mhap_destroy
// through mhap_dist_release).
struct HandleView {
  int device; hipStream_t stream;
  int Hrow, S;
  int64_t n_entries;
  const int32_t *d_minhash, *d_ordered, *d_meta;
  const int64_t* h_ids; const uint8_t* h_fwd;
  std::string* err;
  void** dist;
};
HandleView handle_view(mhap_handle* h);
void mhap_dist_release(void* dist_state);
int internal_sketch_queries(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, int64_t n, void* d_mh, void* d_od, void* d_mt);

}  // namespace mhap
