// mhap_internal.hpp — what the translation units of libmhaphip.so share besides the kernel launch interface (kernels.hpp):
// the device-buffer helper and the narrow view of a handle that the multi-GPU module (mhap_dist.hip) works through.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "../../include/mhap_hip.h"

namespace mhap {

// Host threads worth starting: the hardware threads, capped by the container's CPU quota when there is one (cgroup v2 cpu.max) —
// more runnable threads than quota only buy CFS throttling (stalls of up to a period, 100 ms) — and by MHAP_HOST_THREADS.
inline int usable_host_threads(int hard_cap = 64) {
  if (const char* e = getenv("MHAP_HOST_THREADS")) { if (atoi(e) > 0) return atoi(e); }
  unsigned hc = std::thread::hardware_concurrency();
  if (hc == 0) hc = 1;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64] = {0}; long long per = 0;
    if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
      const long long c = (atoll(q) + per - 1) / per;
      if (c > 0 && (unsigned long long)c < hc) hc = (unsigned)c;
    }
    fclose(f);
  }
  return (int)std::max(1u, std::min(hc, (unsigned)hard_cap));
}

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
  int Hrow, S, k, min_olap_length;
  int64_t n_entries;
  uint64_t index_gen;      // bumped by every change of the entry set
  const int32_t *d_minhash, *d_ordered, *d_meta;
  const int64_t* h_ids; const uint8_t* h_fwd;
  std::string* err;
  void** dist;
};
HandleView handle_view(mhap_handle* h);
void mhap_dist_release(void* dist_state);
// Eager exchange (mhap_dist.hip; mhap_dist_set_eager): an add on a rank of a multi-GPU job gathers its forward rows while the add is
// still computing — the ordered rows (6/7 of the bytes) as soon as the ordered-sketch kernel has written them, under the MinHash
// kernel; the MinHash rows, meta and ids right behind the MinHash kernel, under the index build — so that the collective search
// that follows finds every rank's rows in place.  dist_eager_begin is a COLLECTIVE rendezvous (row counts and whether every rank
// can take part): 1 = this add is eager on every rank, 0 = it is not, < 0 an error code.
int dist_eager_begin(mhap_handle* h, int64_t rows, const int64_t* ids, bool eligible);
int dist_eager_ordered(mhap_handle* h, hipStream_t producer, const int32_t* d_ordered_rows);   // rows of this add: [2 rows][S][2], forward = even
int dist_eager_minhash(mhap_handle* h, hipStream_t producer, const int32_t* d_minhash_rows, const int32_t* d_meta_rows);
bool dist_eager_wanted(mhap_handle* h);
// An ingest of several adds (mhap_index_add_scan): COLLECTIVE when the eager exchange is on — the ranks exchange their group counts; 1 = every
// rank adds exactly one group and the add's own rendezvous follow, 0 = the eager exchange is suspended on every rank until
// dist_ingest_scope_end (no add makes a rendezvous), < 0 an error code.  Not a rank / eager off: 0 without any communication.
int dist_ingest_scope(mhap_handle* h, int64_t ngroups);
void dist_ingest_scope_end(mhap_handle* h);
int dist_eager_reserve_wgs(mhap_handle* h);   // workgroup slots the persistent MinHash grid leaves free for the collective's own kernels (RCCL; 0: copy engines)
void dist_eager_commit(mhap_handle* h);   // the add is complete (host mirrors included): what was gathered describes the index as it is now
// install a group of reads that the caller packed itself (2 bits per base / raw bytes, laid out like stage_reads does) as the handle's
// staged reads: descs[i] = {base_off, length, flags (MHAP_RD_SKIP / MHAP_RD_RAW)}, packed = `bytes` bytes of host memory (pinned: the
// upload then runs at PCIe speed)
struct ReadDesc;
int internal_stage_packed(mhap_handle* h, const ReadDesc* descs, const int64_t* ids, int64_t n, const void* packed, size_t bytes);
// records start, start + stride, ... of a scanned FASTA file into the handle's index (mhap_ingest.hip)
struct FastaScanImpl;
int ingest_add_subset(mhap_handle* h, const FastaScanImpl* scan, int64_t start, int64_t stride);
const FastaScanImpl* scan_impl(const mhap_fasta_scan* s);
int internal_find_matches_device(mhap_handle* h, const void* d_q_minhash, const void* d_q_ordered, const void* d_q_meta, const int64_t* ids,
                                 const int64_t* d_ids_dev, int64_t m, int to_self, mhap_record_sink sink, void* user);
int internal_sketch_queries(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, int64_t n, void* d_mh, void* d_od, void* d_mt);

}  // namespace mhap
