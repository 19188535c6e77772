// mhap_dist.hip — several GPUs, one sharded index: the exchange step of SURVEY.md §8(e) inside libmhaphip.so.
//
// The reference is one JVM with one index and a thread pool (J/impl/AbstractMatchSearch.java:67-117 addData, :121-199
// findMatches()); its only data-parallel axis is the read.  Here rank r of N holds the sketches and the inverted index of the
// reads it was dealt, and a search is: pack the forward-strand rows of the rank's tables -> all-gather them (MinHash rows, meta,
// ids first; the ordered rows, 6x the bytes, behind the candidate stage) -> every rank searches all forward rows against its own
// shard with the toSelf id rules (J/impl/MinHashSearch.java:200-225), so every unordered pair is reported once, by the rank
// that stores its lower-id read.  Two transports carry the gather:
//   * RCCL (one process per GPU, or MHAP_GROUP_TRANSPORT=rccl): ncclAllGather on a communicator over the ranks' devices.  The
//     library is bound at run time (dlopen): a host that already loaded an RCCL (PyTorch-ROCm ships one) shares it.
//   * peer copies (one process, N devices: mhap_group_*): every rank pulls the other ranks' rows with hipMemcpyPeerAsync —
//     on a fully connected xGMI node the direct all-gather uses all 7 links of a GPU at once; ranks may also share a device.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A ROCm install without the RCCL development headers: the library is bound with dlopen at run time anyway, so the handful of
// types the calls below need are declared here (rccl.h of RCCL 2.x: an opaque communicator, a 128-byte id, plain enums).
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7, ncclNumResults = 8 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
}
#endif

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hpp"
#include "mhap_internal.hpp"

using namespace mhap;

namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------------------
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;   // optional (watchdog)
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                          // optional (watchdog)
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;              // optional (mhap_dist_info)
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;           // optional
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;           // optional
  ncclResult_t (*GetVersion)(int*) = nullptr;                               // optional
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, []() {
    // an RCCL the process already loaded comes first (same HIP runtime as the rest of the process), then the ROCm one
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    if (const char* e = getenv("MHAP_RCCL_LIB")) api.lib = dlopen(e, RTLD_NOW | RTLD_LOCAL);
    for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) { api.why = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?"); return; }
    auto sym = [&](const char* n) { void* p = dlsym(api.lib, n); if (!p && api.why.empty()) api.why = std::string("librccl lacks ") + n; return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.CommGetAsyncError = (decltype(api.CommGetAsyncError))dlsym(api.lib, "ncclCommGetAsyncError");
    api.CommAbort = (decltype(api.CommAbort))dlsym(api.lib, "ncclCommAbort");
    api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
    api.CommUserRank = (decltype(api.CommUserRank))dlsym(api.lib, "ncclCommUserRank");
    api.CommCuDevice = (decltype(api.CommCuDevice))dlsym(api.lib, "ncclCommCuDevice");
    api.GetVersion = (decltype(api.GetVersion))dlsym(api.lib, "ncclGetVersion");
    if (!api.why.empty()) { api.lib = nullptr; }
  });
  return api;
}

// ---- transports ------------------------------------------------------------------------------------------------------------
struct Transport {
  int rank = 0, nranks = 1;
  virtual ~Transport() {}
  // device all-gather of `bytes` per rank, enqueued on st (of this rank's device); recv holds nranks * bytes
  virtual int allgather(const void* send, void* recv, size_t bytes, hipStream_t st, std::string& err) = 0;
  // small host-side all-gather (counts)
  virtual int allgather_host(const void* in, void* out, size_t bytes, std::string& err) = 0;
  // A rendezvous of the collective entry points: every rank contributes up to three words; the payload has ONE size whatever the call
  // (four words) and starts with a tag = the kind of rendezvous + how many this transport has made.  Ranks that are out of step — one
  // rank's add split into more ingest groups than another's, so an add's rendezvous meets a search's — then fail HERE with
  // the two kinds named, instead of reading each other's differently sized payloads as counts.  all: nranks x 3 words.
  enum { RV_SEARCH = 1, RV_SEARCH_ALLOC = 2, RV_ADD = 3, RV_ADD_ALLOC = 4, RV_SELFTEST = 5, RV_INGEST = 6 };
  uint64_t rv_seq = 0;
  int rendezvous(int kind, const int64_t (&mine)[3], int64_t* all, std::string& err) {
    const int64_t tag = (int64_t)(((uint64_t)kind << 56) | (rv_seq++ & 0x00FFFFFFFFFFFFFFULL));
    const int64_t out[4] = {tag, mine[0], mine[1], mine[2]};
    std::vector<int64_t> in((size_t)nranks * 4, 0);
    const int rc = allgather_host(out, in.data(), sizeof out, err);
    if (rc != MHAP_OK) return rc;
    for (int r = 0; r < nranks; r++) {
      // (the KIND must agree; the counts may differ after a rank left an earlier call with an error of its own — they are for the message)
      if ((((uint64_t)in[(size_t)r * 4]) >> 56) != (uint64_t)kind) {
        static const char* const names[] = {"?", "search", "search (buffers)", "add", "add (buffers)", "self-test", "ingest plan"};
        const int kr = (int)(((uint64_t)in[(size_t)r * 4]) >> 56);
        err = std::string("the ranks' collective calls are out of step: this rank is in its ") + names[kind] + " rendezvous no. " + std::to_string(rv_seq - 1) +
              ", rank " + std::to_string(r) + " in a " + (kr >= 1 && kr <= 6 ? names[kr] : "?") + " rendezvous no. " +
              std::to_string((unsigned long long)((uint64_t)in[(size_t)r * 4] & 0x00FFFFFFFFFFFFFFULL)) + " (every rank must make the same sequence of add / search calls)";
        return MHAP_E_STATE;
      }
      for (int i = 0; i < 3; i++) all[(size_t)r * 3 + i] = in[(size_t)r * 4 + 1 + i];
    }
    return MHAP_OK;
  }
  // every rank has drained its streams: send buffers may be rewritten
  virtual void quiesce() {}
  // this rank gives up: ranks of the same process waiting for it must not wait for ever
  virtual void abort() {}
  // this rank leaves a collective call with a LOCAL error (its sink said stop, it ran out of memory in the search) after it has
  // enqueued everything the other ranks need from it.  In-process peers still wait for it at the hub's barriers, so the default is
  // abort(); an RCCL communicator stays usable — tearing it down for a recoverable local error would kill the rank for good, and the
  // peers, who are owed nothing more by this call, would only learn of it in their next collective's time-out
  virtual void local_failure() { abort(); }
  // false (+ why): a peer of this exchange has failed or left — nothing it was to send will arrive
  virtual bool healthy(std::string&) { return true; }
  virtual const char* name() const = 0;

  // Wait for an event that depends on the other ranks without trusting them to be alive: poll it, ask the transport now and then
  // whether a peer has failed (RCCL: ncclCommGetAsyncError; peers of one process: the hub's flag), and give up after
  // MHAP_DIST_TIMEOUT_S (default 1800).  Giving up aborts the transport — an RCCL communicator is torn down with ncclCommAbort,
  // which also ends the collective kernels this rank has in flight — so neither this rank nor the surviving ones hang in a gather.
  int wait_event(hipEvent_t ev, std::string& err) {
    static const double limit_ms = []() { const char* e = getenv("MHAP_DIST_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return (v > 0 ? v : 1800.0) * 1e3; }();
    const double t0 = now_ms();
    for (unsigned spin = 1;; spin++) {
      const hipError_t e = hipEventQuery(ev);
      if (e == hipSuccess) return MHAP_OK;
      if (e != hipErrorNotReady) { err = std::string("exchange event: ") + hipGetErrorString(e); abort(); return MHAP_E_HIP; }
      if ((spin & 255u) == 0) {
        std::string why;
        if (!healthy(why)) { err = "exchange aborted: " + why; abort(); return MHAP_E_STATE; }
        if (now_ms() - t0 > limit_ms) { err = "exchange timed out after " + std::to_string((int)(limit_ms / 1e3)) + " s waiting for the other ranks (MHAP_DIST_TIMEOUT_S)"; abort(); return MHAP_E_STATE; }
      }
      if (spin > 4096) std::this_thread::sleep_for(std::chrono::microseconds(50)); else std::this_thread::yield();
    }
  }
};

struct RcclTransport : Transport {
  ncclComm_t comm = nullptr;
  bool owns = true;
  DevBuf scratch;
  int device = 0;
  hipEvent_t ev_host = nullptr;
  ~RcclTransport() override {
    scratch.release();
    if (ev_host) (void)hipEventDestroy(ev_host);
    if (comm && owns && rccl().CommDestroy) (void)rccl().CommDestroy(comm);
  }
  bool healthy(std::string& why) override {
    if (!comm) { why = "the RCCL communicator was aborted"; return false; }
    if (!rccl().CommGetAsyncError) return true;
    ncclResult_t st = ncclSuccess;
    if (rccl().CommGetAsyncError(comm, &st) != ncclSuccess) { why = "ncclCommGetAsyncError failed"; return false; }
    if (st == ncclSuccess || st == ncclInProgress) return true;
    why = std::string("RCCL reports an asynchronous error: ") + rccl().GetErrorString(st);
    return false;
  }
  void abort() override {
    if (comm && rccl().CommAbort) { (void)rccl().CommAbort(comm); comm = nullptr; }   // (aborted communicators are not destroyed again)
  }
  void local_failure() override {}   // (the communicator is kept: see Transport::local_failure)
  int allgather(const void* send, void* recv, size_t bytes, hipStream_t st, std::string& err) override {
    if (!comm) { err = "the RCCL communicator was aborted"; return MHAP_E_STATE; }
    const ncclResult_t r = rccl().AllGather(send, recv, bytes, ncclChar, comm, st);
    if (r != ncclSuccess) { err = std::string("ncclAllGather: ") + rccl().GetErrorString(r); return MHAP_E_HIP; }
    return MHAP_OK;
  }
  int allgather_host(const void* in, void* out, size_t bytes, std::string& err) override {
    if (scratch.ensure(bytes * (size_t)(nranks + 1)) != hipSuccess) { err = "out of device memory (exchange scratch)"; return MHAP_E_NOMEM; }
    char* d = scratch.as<char>();
    hipStream_t st = nullptr;   // the null stream: this is a rendezvous, not a hot path
    if (hipMemcpy(d, in, bytes, hipMemcpyHostToDevice) != hipSuccess) { err = "hipMemcpy (exchange scratch)"; return MHAP_E_HIP; }
    int rc = allgather(d, d + bytes, bytes, st, err);
    if (rc != MHAP_OK) return rc;
    if (!ev_host && hipEventCreateWithFlags(&ev_host, hipEventDisableTiming) != hipSuccess) { err = "cannot create the exchange event"; return MHAP_E_HIP; }
    if (hipEventRecord(ev_host, st) != hipSuccess) { err = "hipEventRecord (exchange)"; return MHAP_E_HIP; }
    rc = wait_event(ev_host, err);   // (not hipStreamSynchronize: a rank that never arrives must not hang this one for ever)
    if (rc != MHAP_OK) return rc;
    if (hipMemcpy(out, d + bytes, bytes * (size_t)nranks, hipMemcpyDeviceToHost) != hipSuccess) {
      err = "exchange of the row counts failed"; return MHAP_E_HIP;
    }
    return MHAP_OK;
  }
  const char* name() const override { return "rccl"; }
};

// One process, N ranks on N host threads: a rank publishes its send buffer + an event recorded behind the writes, all ranks
// meet at a host barrier, and every rank enqueues its own N copies (pull).  Ranks on one device degrade to plain device copies.
struct PeerHub {
  int n;
  std::mutex mu; std::condition_variable cv; int arrived = 0; uint64_t gen = 0; bool failed = false;
  std::vector<const void*> ptr; std::vector<hipEvent_t> ev; std::vector<int> dev;
  std::vector<char> hostbuf; size_t host_bytes = 0;
  explicit PeerHub(int n_) : n(n_), ptr((size_t)n_, nullptr), ev((size_t)n_, nullptr), dev((size_t)n_, 0) {}
  // false: a rank failed and left (every later barrier returns false at once)
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (failed) return false;
    const uint64_t g = gen;
    if (++arrived == n) { arrived = 0; gen++; cv.notify_all(); }
    else cv.wait(lk, [&]() { return gen != g || failed; });
    return !failed;
  }
  void abort() { std::lock_guard<std::mutex> lk(mu); failed = true; cv.notify_all(); }
  bool has_failed() { std::lock_guard<std::mutex> lk(mu); return failed; }
  // every rank thread has returned (on_ranks joined them): the next collective call starts from a clean hub, so one failed search
  // (a transient out-of-memory, say) does not kill the group for good
  void reset() { std::lock_guard<std::mutex> lk(mu); failed = false; arrived = 0; gen++; }
};
struct PeerTransport : Transport {
  PeerHub* hub = nullptr;
  int allgather(const void* send, void* recv, size_t bytes, hipStream_t st, std::string& err) override {
    hub->ptr[(size_t)rank] = send;
    bool ok = hipEventRecord(hub->ev[(size_t)rank], st) == hipSuccess;
    if (!hub->barrier()) { err = "another rank failed"; return MHAP_E_STATE; }
    for (int r = 0; r < nranks && ok; r++) {
      char* dst = (char*)recv + (size_t)r * bytes;
      if (r != rank) ok = ok && hipStreamWaitEvent(st, hub->ev[(size_t)r], 0) == hipSuccess;
      // (MHAP_GROUP_FORCE_PEER=1: the peer-copy call also between ranks that share a device — a copy dev0 -> dev0 is legal — so the
      //  cross-device branch can be exercised on a one-GPU box)
      const char* fpe = getenv("MHAP_GROUP_FORCE_PEER");
      const bool force_peer = fpe && fpe[0] == '1';
      if (hub->dev[(size_t)r] == hub->dev[(size_t)rank] && !force_peer) ok = ok && hipMemcpyAsync(dst, hub->ptr[(size_t)r], bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
      else ok = ok && hipMemcpyPeerAsync(dst, hub->dev[(size_t)rank], hub->ptr[(size_t)r], hub->dev[(size_t)r], bytes, st) == hipSuccess;
    }
    // the events of this round are consumed (waits enqueued) before any rank records the next one
    if (!hub->barrier()) { err = "another rank failed"; return MHAP_E_STATE; }
    if (!ok) { err = "peer-to-peer gather failed"; return MHAP_E_HIP; }
    return MHAP_OK;
  }
  int allgather_host(const void* in, void* out, size_t bytes, std::string& err) override {
    {
      std::lock_guard<std::mutex> lk(hub->mu);
      if (hub->hostbuf.size() < bytes * (size_t)nranks) hub->hostbuf.resize(bytes * (size_t)nranks);
      memcpy(hub->hostbuf.data() + (size_t)rank * bytes, in, bytes);
    }
    if (!hub->barrier()) { err = "another rank failed"; return MHAP_E_STATE; }
    memcpy(out, hub->hostbuf.data(), bytes * (size_t)nranks);
    if (!hub->barrier()) { err = "another rank failed"; return MHAP_E_STATE; }
    return MHAP_OK;
  }
  void abort() override { hub->abort(); }
  bool healthy(std::string& why) override { if (hub->has_failed()) { why = "another rank failed"; return false; } return true; }
  void quiesce() override { (void)hub->barrier(); }
  const char* name() const override { return "peer"; }
};

// ---- per-rank exchange state -------------------------------------------------------------------------------------------------
struct DistState {
  Transport* tr = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_pack = nullptr, ev_small = nullptr, ev_big = nullptr;
  hipEvent_t ev_xt[4] = {nullptr, nullptr, nullptr, nullptr};   // timing events on the exchange stream: ordered rows' gather begin / end, MinHash + meta + id rows' begin / end
  bool xt_ordered = false, xt_small = false;                   // ... recorded since the last read-out
  double xt_bytes[2] = {0, 0};                                 // bytes this rank received in them
  DevBuf s_mh, s_od, s_mt, s_ids;      // this rank's forward rows, packed
  DevBuf g_mh, g_od, g_mt, g_ids;      // gathered rows of all ranks, rank after rank
  DevBuf q_mh, q_od, q_mt;             // -q mode: this rank's query sketches (both strand slots, forward filled)
  std::vector<int64_t> ids_all, ids_local;
  double t_small = 0, t_wait_big = 0, t_total = 0;
  bool gate_ran = false;
  std::string gate_err;
  // eager exchange: eager = the host asked for it; eager_go = the add in progress gathers eagerly (every rank agreed); eager_valid = g_*
  // hold the forward rows of the index as it is now (eager_rows local rows, eager_npad per rank)
  bool eager = false, eager_go = false, eager_done = false, eager_valid = false, eager_suspended = false;
  uint64_t eager_gen = 0;
  int64_t eager_searches = 0;    // searches that found their rows gathered by the add
  int64_t eager_rows = 0, eager_npad = 0;
  hipEvent_t ev_prod = nullptr;
  ~DistState() {
    DevBuf* bufs[] = {&s_mh, &s_od, &s_mt, &s_ids, &g_mh, &g_od, &g_mt, &g_ids, &q_mh, &q_od, &q_mt};
    for (DevBuf* b : bufs) b->release();
    if (ev_pack) (void)hipEventDestroy(ev_pack);
    if (ev_small) (void)hipEventDestroy(ev_small);
    if (ev_big) (void)hipEventDestroy(ev_big);
    if (ev_prod) (void)hipEventDestroy(ev_prod);
    for (hipEvent_t e : ev_xt) if (e) (void)hipEventDestroy(e);
    if (comm_stream) (void)hipStreamDestroy(comm_stream);
    delete tr;
  }
};

int dfail(const HandleView& v, int code, const std::string& msg) { *v.err = msg; return code; }

#define DCHK(v, expr)                                                                                 \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      return dfail((v), _e == hipErrorOutOfMemory ? MHAP_E_NOMEM : MHAP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// attach() OWNS tr from here on, on every path: it ends up in the handle's DistState or is deleted with it
int attach(mhap_handle* h, Transport* tr) {
  HandleView v = handle_view(h);
  (void)hipSetDevice(v.device);
  if (*v.dist) { mhap_dist_release(*v.dist); *v.dist = nullptr; }
  DistState* d = new DistState();
  d->tr = tr;
  if (hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&d->ev_pack, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&d->ev_small, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&d->ev_big, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&d->ev_prod, hipEventDisableTiming) != hipSuccess) {
    delete d;
    return dfail(v, MHAP_E_HIP, "cannot create the exchange stream");
  }
  *v.dist = d;
  return MHAP_OK;
}

int gate_cb(void* user) {
  DistState* d = (DistState*)user;
  const double t0 = now_ms();
  const int rc = d->tr->wait_event(d->ev_big, d->gate_err);   // the ordered rows of every rank have arrived
  d->t_wait_big += now_ms() - t0;
  d->gate_ran = true;
  return rc == MHAP_OK ? 0 : 1;
}

// Gather `rows` forward rows of this rank (row j at src + j * src_pitch_rows rows) from every rank and search them against this
// rank's index.  d_* = tables holding the query rows of this rank at stride `stride` rows (2: forward entries of an index;
// the -q tables use the same pairing), ids = one id per local row.
int exchange_and_search(mhap_handle* h, DistState* d, const int32_t* d_mh, const int32_t* d_od, const int32_t* d_mt, int stride, const int64_t* ids,
                        int64_t rows, int to_self, mhap_record_sink sink, void* user) {
  HandleView v = handle_view(h);
  Transport* tr = d->tr;
  const int N = tr->nranks;
  const double t0 = now_ms();
  int rc = MHAP_OK;
  // How an error return treats the transport.  ABORT: the other ranks are owed something this rank will never send (a failed collective,
  // a HIP error between the rendezvous and the last gather) — in-process peers are released from the hub, an RCCL communicator is torn
  // down so that nobody hangs in a gather.  LOCAL: everything the others need is enqueued and the error is this rank's own (sink, out
  // of memory in the search): Transport::local_failure — the RCCL communicator survives.  NONE: every rank returns together (success,
  // or an error all of them computed from the same gathered numbers).
  enum { LEAVE_ABORT, LEAVE_LOCAL, LEAVE_NONE } leave = LEAVE_ABORT;
  struct Guard { Transport* t; decltype(leave)* how; ~Guard() { if (*how == LEAVE_ABORT) t->abort(); else if (*how == LEAVE_LOCAL) t->local_failure(); } } guard{tr, &leave};
  d->t_small = d->t_wait_big = 0; d->gate_ran = false;
  // The rows may have been gathered while the add was still computing (eager exchange).  Whether THIS rank's gathered rows describe its
  // index as it is now is rank-local knowledge (a failed commit, an add of precomputed sketches, a clear on one rank only), so the ranks
  // agree on the branch here: one small rendezvous carries every rank's row count and its "my rows are in place" flag, and the eager
  // branch is taken only if all of them say so — otherwise some ranks would enter the gathers below and others would not.
  const bool eager_mine = d->eager_valid && d->eager_gen == v.index_gen && to_self && stride == 2 && d_mh == v.d_minhash && rows == d->eager_rows &&
                          2 * rows == v.n_entries;
  int64_t n_pad = 0, total = 0;
  bool use_eager = eager_mine;
  {
    const int64_t mine[3] = {rows, eager_mine ? 1 : 0, 0};
    std::vector<int64_t> all((size_t)N * 3, 0);
    rc = tr->rendezvous(Transport::RV_SEARCH, mine, all.data(), *v.err);
    if (rc != MHAP_OK) return rc;                         // (a failed rendezvous: ABORT)
    for (int r = 0; r < N; r++) { n_pad = std::max(n_pad, all[(size_t)r * 3]); total += all[(size_t)r * 3]; use_eager = use_eager && all[(size_t)r * 3 + 1] != 0; }
  }
  if (use_eager && n_pad != d->eager_npad) use_eager = false;   // (cannot happen: every rank's count is the one its add announced)
  if (!use_eager) d->eager_valid = false;       // (the gather below overwrites the buffers)
  else d->eager_searches++;
  if (total == 0) { leave = LEAVE_NONE; tr->quiesce(); return MHAP_OK; }
  if ((int64_t)N * n_pad > (int64_t)INT32_MAX / 2) { leave = LEAVE_NONE; tr->quiesce(); return dfail(v, MHAP_E_INVALID, "too many query rows for one gather"); }
  const size_t mh_row = (size_t)v.Hrow * 4, od_row = (size_t)v.S * 8, mt_row = (size_t)META_W * 4;
  const size_t np = (size_t)n_pad;
  hipStream_t st = v.stream, cs = d->comm_stream;
  if (!use_eager) {
  // the exchange buffers; a rank that cannot allocate them says so in a second rendezvous, and all ranks return together with the
  // communicator intact (without it the others would sit in the gathers until the watchdog's time-out)
  {
    const bool got = d->s_mh.ensure(np * mh_row) == hipSuccess && d->s_od.ensure(np * od_row) == hipSuccess && d->s_mt.ensure(np * mt_row) == hipSuccess &&
                     d->s_ids.ensure(np * 8) == hipSuccess && d->g_mh.ensure((size_t)N * np * mh_row) == hipSuccess &&
                     d->g_od.ensure((size_t)N * np * od_row) == hipSuccess && d->g_mt.ensure((size_t)N * np * mt_row) == hipSuccess &&
                     d->g_ids.ensure((size_t)N * np * 8) == hipSuccess;
    if (!got) (void)hipGetLastError();
    const int64_t mine[3] = {got ? 0 : 1, 0, 0};
    std::vector<int64_t> all((size_t)N * 3, 0);
    rc = tr->rendezvous(Transport::RV_SEARCH_ALLOC, mine, all.data(), *v.err);
    if (rc != MHAP_OK) return rc;
    int bad = -1;
    for (int r = 0; r < N; r++) if (all[(size_t)r * 3] != 0 && bad < 0) bad = r;
    if (bad >= 0) {
      leave = LEAVE_NONE; tr->quiesce();
      return dfail(v, MHAP_E_NOMEM, got ? "rank " + std::to_string(bad) + " is out of device memory for the exchange buffers" : std::string("out of device memory (exchange buffers)"));
    }
  }
  // pack: every `stride`-th row of the tables; padding rows get status -1 (skipped as queries)
  if (rows > 0) {
    DCHK(v, hipMemcpy2DAsync(d->s_mh.p, mh_row, d_mh, mh_row * (size_t)stride, mh_row, (size_t)rows, hipMemcpyDeviceToDevice, st));
    DCHK(v, hipMemcpy2DAsync(d->s_mt.p, mt_row, d_mt, mt_row * (size_t)stride, mt_row, (size_t)rows, hipMemcpyDeviceToDevice, st));
    DCHK(v, hipMemcpy2DAsync(d->s_od.p, od_row, d_od, od_row * (size_t)stride, od_row, (size_t)rows, hipMemcpyDeviceToDevice, st));
    DCHK(v, hipMemcpyAsync(d->s_ids.p, ids, (size_t)rows * 8, hipMemcpyHostToDevice, st));
  }
  if (n_pad > rows) {
    DCHK(v, hipMemsetAsync(d->s_mt.as<char>() + (size_t)rows * mt_row, 0xFF, (size_t)(n_pad - rows) * mt_row, st));
    DCHK(v, hipMemsetAsync(d->s_ids.as<char>() + (size_t)rows * 8, 0, (size_t)(n_pad - rows) * 8, st));
    DCHK(v, hipMemsetAsync(d->s_mh.as<char>() + (size_t)rows * mh_row, 0, (size_t)(n_pad - rows) * mh_row, st));
  }
  DCHK(v, hipEventRecord(d->ev_pack, st));
  DCHK(v, hipStreamWaitEvent(cs, d->ev_pack, 0));
  // the small tables first (all the candidate stage needs), then the ordered rows: they travel while the candidates are counted
  rc = tr->allgather(d->s_mh.p, d->g_mh.p, np * mh_row, cs, *v.err); if (rc != MHAP_OK) return rc;
  rc = tr->allgather(d->s_mt.p, d->g_mt.p, np * mt_row, cs, *v.err); if (rc != MHAP_OK) return rc;
  rc = tr->allgather(d->s_ids.p, d->g_ids.p, np * 8, cs, *v.err); if (rc != MHAP_OK) return rc;
  DCHK(v, hipEventRecord(d->ev_small, cs));
  rc = tr->allgather(d->s_od.p, d->g_od.p, np * od_row, cs, *v.err); if (rc != MHAP_OK) return rc;
  DCHK(v, hipEventRecord(d->ev_big, cs));
  }
  // From here on the other ranks are owed nothing more by this call: an error below is this rank's own.  It still waits for the
  // gathers it takes part in (they write this rank's buffers) before it returns.
  leave = LEAVE_LOCAL;
  auto leave_local = [&](int code) {
    if (!d->gate_ran) { std::string why; if (tr->wait_event(d->ev_big, why) != MHAP_OK) leave = LEAVE_NONE; }   // (wait_event aborts by itself when the gather is dead)
    d->t_total = now_ms() - t0;
    return code;
  };
  // meanwhile: this rank's inverted index (a no-op when the add built it eagerly)
  rc = mhap_index_prepare(h); if (rc != MHAP_OK) return leave_local(rc);
  rc = tr->wait_event(d->ev_small, *v.err); if (rc != MHAP_OK) { leave = LEAVE_NONE; return rc; }   // (a dead gather: wait_event has aborted the transport)
  d->ids_all.resize((size_t)N * np);
  { const hipError_t e = hipMemcpy(d->ids_all.data(), d->g_ids.p, (size_t)N * np * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return leave_local(dfail(v, MHAP_E_HIP, std::string("read-back of the gathered ids: ") + hipGetErrorString(e))); }
  d->t_small = now_ms() - t0;
  rc = mhap_set_second_stage_gate(h, gate_cb, d); if (rc != MHAP_OK) return leave_local(rc);
  rc = internal_find_matches_device(h, d->g_mh.p, d->g_od.p, d->g_mt.p, d->ids_all.data(), d->g_ids.as<int64_t>(), (int64_t)N * n_pad, to_self, sink, user);
  (void)mhap_set_second_stage_gate(h, nullptr, nullptr);
  const bool gate_failed = !d->gate_err.empty();
  if (rc != MHAP_OK && gate_failed) *v.err = d->gate_err;   // (the gate's reason, not "the gate aborted the search")
  d->gate_err.clear();
  if (rc != MHAP_OK) { if (gate_failed) { leave = LEAVE_NONE; d->t_total = now_ms() - t0; return rc; } return leave_local(rc); }
  if (!d->gate_ran) { rc = tr->wait_event(d->ev_big, *v.err); if (rc != MHAP_OK) { leave = LEAVE_NONE; return rc; } }   // no candidates here: the gather still has to finish before the buffers are reused
  leave = LEAVE_NONE;
  tr->quiesce();
  d->t_total = now_ms() - t0;
  return rc;
}

}  // namespace

// ---- eager exchange hooks (declared in mhap_internal.hpp; called by the add path in mhap_capi.hip) ---------------------------------
namespace mhap {
bool dist_eager_wanted(mhap_handle* h) {
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  return d && d->eager && !d->eager_suspended;
}
int dist_ingest_scope(mhap_handle* h, int64_t ngroups) {
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d || !d->eager) return 0;
  Transport* tr = d->tr;
  const int64_t mine[3] = {ngroups, 0, 0};
  std::vector<int64_t> all((size_t)tr->nranks * 3, 0);
  const int rc = tr->rendezvous(Transport::RV_INGEST, mine, all.data(), *v.err);
  if (rc != MHAP_OK) { tr->abort(); return rc; }
  bool one_each = true;
  for (int r = 0; r < tr->nranks; r++) one_each = one_each && all[(size_t)r * 3] == 1;
  d->eager_suspended = !one_each;
  return one_each ? 1 : 0;
}
void dist_ingest_scope_end(mhap_handle* h) {
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (d) d->eager_suspended = false;
}
// The ordered rows' all-gather and the persistent MinHash grid become runnable at the same moment (both wait for the ordered kernel).
// A grid that takes every workgroup slot of every CU leaves an RCCL kernel nothing to run on until it ends — the gather would simply
// follow the MinHash kernel.  So the grid is launched a few workgroups short (MHAP_EAGER_RESERVE_WGS, default 48 of 1024: those CUs
// keep a free slot of four wave64s and 128 VGPRs each).  Peer copies run on the copy engines and need none.
int dist_eager_reserve_wgs(mhap_handle* h) {
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d || !d->eager_go || strcmp(d->tr->name(), "rccl") != 0 || d->tr->nranks < 2) return 0;
  if (const char* e = getenv("MHAP_EAGER_RESERVE_WGS")) { const int x = atoi(e); if (x >= 0 && x <= 512) return x; }
  return 48;
}
int dist_eager_begin(mhap_handle* h, int64_t rows, const int64_t* ids, bool eligible) {
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d || !d->eager) return 0;
  Transport* tr = d->tr;
  const int N = tr->nranks;
  d->eager_go = false; d->eager_done = false; d->eager_valid = false;
  // rendezvous: every rank's row count, negative = "I cannot" (not the first add of an empty index, or more than one launch group)
  const int64_t mine[3] = {eligible ? rows : -1, 0, 0};
  std::vector<int64_t> counts((size_t)N * 3, 0);
  int rc = tr->rendezvous(Transport::RV_ADD, mine, counts.data(), *v.err);
  if (rc != MHAP_OK) { tr->abort(); return rc; }
  int64_t n_pad = 0, total = 0;
  for (int r = 0; r < N; r++) { const int64_t c = counts[(size_t)r * 3]; if (c < 0) return 0; n_pad = std::max(n_pad, c); total += c; }
  if (total == 0 || (int64_t)N * n_pad > (int64_t)INT32_MAX / 2) return 0;
  const size_t mh_row = (size_t)v.Hrow * 4, od_row = (size_t)v.S * 8, mt_row = (size_t)META_W * 4, np = (size_t)n_pad;
  auto chk = [&](hipError_t e) { return e == hipSuccess; };
  // A rank that cannot allocate its exchange buffers says so in a second rendezvous and ALL ranks fall back to the exchange at search
  // time together (which reports the shortage properly); tearing the transport down here left the others in their gathers.
  const bool got = chk(d->s_mh.ensure(np * mh_row)) && chk(d->s_od.ensure(np * od_row)) && chk(d->s_mt.ensure(np * mt_row)) && chk(d->s_ids.ensure(np * 8)) &&
                   chk(d->g_mh.ensure((size_t)N * np * mh_row)) && chk(d->g_od.ensure((size_t)N * np * od_row)) && chk(d->g_mt.ensure((size_t)N * np * mt_row)) &&
                   chk(d->g_ids.ensure((size_t)N * np * 8));
  if (!got) (void)hipGetLastError();
  {
    const int64_t flag[3] = {got ? 0 : 1, 0, 0};
    std::vector<int64_t> all((size_t)N * 3, 0);
    rc = tr->rendezvous(Transport::RV_ADD_ALLOC, flag, all.data(), *v.err);
    if (rc != MHAP_OK) { tr->abort(); return rc; }
    for (int r = 0; r < N; r++) if (all[(size_t)r * 3] != 0) return 0;
  }
  d->ids_local.assign(ids, ids + rows);
  d->eager_rows = rows; d->eager_npad = n_pad; d->eager_go = true;
  return 1;
}
// the ordered rows of this add exist once `producer` has run what it holds now: pack the forward ones and start their all-gather
int dist_eager_ordered(mhap_handle* h, hipStream_t producer, const int32_t* d_od) {
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d || !d->eager_go) return MHAP_OK;
  const size_t od_row = (size_t)v.S * 8, np = (size_t)d->eager_npad;
  const int64_t rows = d->eager_rows;
  hipStream_t cs = d->comm_stream;
  DCHK(v, hipEventRecord(d->ev_prod, producer));
  DCHK(v, hipStreamWaitEvent(cs, d->ev_prod, 0));
  if (rows > 0) DCHK(v, hipMemcpy2DAsync(d->s_od.p, od_row, d_od, od_row * 2, od_row, (size_t)rows, hipMemcpyDeviceToDevice, cs));
  for (auto& e : d->ev_xt) if (!e) DCHK(v, hipEventCreate(&e));
  DCHK(v, hipEventRecord(d->ev_xt[0], cs));
  const int rc = d->tr->allgather(d->s_od.p, d->g_od.p, np * od_row, cs, *v.err);
  if (rc != MHAP_OK) { d->eager_go = false; d->tr->abort(); return rc; }
  DCHK(v, hipEventRecord(d->ev_xt[1], cs));
  d->xt_ordered = true; d->xt_bytes[0] = (double)(d->tr->nranks - 1) * (double)np * (double)od_row;
  DCHK(v, hipEventRecord(d->ev_big, cs));
  return MHAP_OK;
}
// ... and the MinHash rows, the meta rows (sizes by the ordered kernel, statuses by the MinHash kernel: both done) and the ids
int dist_eager_minhash(mhap_handle* h, hipStream_t producer, const int32_t* d_mh, const int32_t* d_mt) {
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d || !d->eager_go) return MHAP_OK;
  const size_t mh_row = (size_t)v.Hrow * 4, mt_row = (size_t)META_W * 4, np = (size_t)d->eager_npad;
  const int64_t rows = d->eager_rows, n_pad = d->eager_npad;
  hipStream_t cs = d->comm_stream;
  DCHK(v, hipEventRecord(d->ev_prod, producer));
  DCHK(v, hipStreamWaitEvent(cs, d->ev_prod, 0));
  if (rows > 0) {
    DCHK(v, hipMemcpy2DAsync(d->s_mh.p, mh_row, d_mh, mh_row * 2, mh_row, (size_t)rows, hipMemcpyDeviceToDevice, cs));
    DCHK(v, hipMemcpy2DAsync(d->s_mt.p, mt_row, d_mt, mt_row * 2, mt_row, (size_t)rows, hipMemcpyDeviceToDevice, cs));
    DCHK(v, hipMemcpyAsync(d->s_ids.p, d->ids_local.data(), (size_t)rows * 8, hipMemcpyHostToDevice, cs));
  }
  if (n_pad > rows) {
    DCHK(v, hipMemsetAsync(d->s_mt.as<char>() + (size_t)rows * mt_row, 0xFF, (size_t)(n_pad - rows) * mt_row, cs));
    DCHK(v, hipMemsetAsync(d->s_ids.as<char>() + (size_t)rows * 8, 0, (size_t)(n_pad - rows) * 8, cs));
    DCHK(v, hipMemsetAsync(d->s_mh.as<char>() + (size_t)rows * mh_row, 0, (size_t)(n_pad - rows) * mh_row, cs));
  }
  Transport* tr = d->tr;
  for (auto& e : d->ev_xt) if (!e) DCHK(v, hipEventCreate(&e));
  DCHK(v, hipEventRecord(d->ev_xt[2], cs));
  int rc = tr->allgather(d->s_mh.p, d->g_mh.p, np * mh_row, cs, *v.err);
  if (rc == MHAP_OK) rc = tr->allgather(d->s_mt.p, d->g_mt.p, np * mt_row, cs, *v.err);
  if (rc == MHAP_OK) rc = tr->allgather(d->s_ids.p, d->g_ids.p, np * 8, cs, *v.err);
  if (rc != MHAP_OK) { d->eager_go = false; tr->abort(); return rc; }
  DCHK(v, hipEventRecord(d->ev_xt[3], cs));
  d->xt_small = true; d->xt_bytes[1] = (double)(tr->nranks - 1) * (double)np * (double)(mh_row + mt_row + 8);
  DCHK(v, hipEventRecord(d->ev_small, cs));
  d->eager_go = false; d->eager_done = true;
  return MHAP_OK;
}
void dist_eager_commit(mhap_handle* h) {
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d || !d->eager_done) return;
  d->eager_done = false; d->eager_valid = true; d->eager_gen = v.index_gen;
}
}  // namespace mhap

namespace mhap {
void mhap_dist_release(void* dist_state) { delete (DistState*)dist_state; }
int internal_sketch_queries(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, int64_t n, void* d_mh, void* d_od, void* d_mt);
}  // namespace mhap

// ================================================================================================================================
extern "C" {

int mhap_dist_unique_id(void* id, size_t cap) {
  if (!id || cap < MHAP_DIST_ID_BYTES) return MHAP_E_INVALID;
  RcclApi& api = rccl();
  if (!api.lib) { fprintf(stderr, "mhap_dist_unique_id: %s\n", api.why.c_str()); return MHAP_E_HIP; }
  ncclUniqueId u;
  if (api.GetUniqueId(&u) != ncclSuccess) return MHAP_E_HIP;
  static_assert(sizeof(u) == MHAP_DIST_ID_BYTES, "ncclUniqueId size");
  memcpy(id, &u, sizeof u);
  return MHAP_OK;
}

int mhap_dist_init(mhap_handle* h, int32_t rank, int32_t nranks, const void* id) {
  if (!h) return MHAP_E_INVALID;
  HandleView v = handle_view(h);
  if (nranks < 1 || rank < 0 || rank >= nranks || !id) return dfail(v, MHAP_E_INVALID, "bad rank / world size / id");
  RcclApi& api = rccl();
  if (!api.lib) return dfail(v, MHAP_E_HIP, api.why);
  (void)hipSetDevice(v.device);
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  RcclTransport* tr = new RcclTransport();
  tr->rank = rank; tr->nranks = nranks; tr->device = v.device;
  const ncclResult_t r = api.CommInitRank(&tr->comm, nranks, u, rank);
  if (r != ncclSuccess) { tr->comm = nullptr; delete tr; return dfail(v, MHAP_E_HIP, std::string("ncclCommInitRank: ") + api.GetErrorString(r)); }
  return attach(h, tr);   // (owns tr, also when it fails)
}

int mhap_dist_finalize(mhap_handle* h) {
  if (!h) return MHAP_E_INVALID;
  HandleView v = handle_view(h);
  (void)hipSetDevice(v.device);
  if (*v.dist) { mhap_dist_release(*v.dist); *v.dist = nullptr; }
  return MHAP_OK;
}

int mhap_dist_find_matches_self(mhap_handle* h, mhap_record_sink sink, void* user) {
  if (!h) return MHAP_E_INVALID;
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d) return dfail(v, MHAP_E_STATE, "not a rank of a multi-GPU job (mhap_dist_init / mhap_group_create first)");
  (void)hipSetDevice(v.device);
  if (v.n_entries & 1) return dfail(v, MHAP_E_STATE, "the sharded search needs an index of sketched reads (forward + reverse entries in pairs)");
  const int64_t rows = v.n_entries / 2;
  for (int64_t j = 0; j < rows; j++)
    if (!v.h_fwd[(size_t)(2 * j)] || v.h_fwd[(size_t)(2 * j + 1)] || v.h_ids[(size_t)(2 * j)] != v.h_ids[(size_t)(2 * j + 1)])
      return dfail(v, MHAP_E_STATE, "the sharded search needs an index of sketched reads (forward + reverse entries in pairs)");
  d->ids_local.resize((size_t)rows);
  for (int64_t j = 0; j < rows; j++) d->ids_local[(size_t)j] = v.h_ids[(size_t)(2 * j)];
  return exchange_and_search(h, d, v.d_minhash, v.d_ordered, v.d_meta, 2, d->ids_local.data(), rows, 1, sink, user);
}

int mhap_dist_find_matches_reads(mhap_handle* h, const char* bases, const int64_t* offsets, const int32_t* lengths, const int64_t* ids, int64_t n,
                                 mhap_record_sink sink, void* user) {
  if (!h) return MHAP_E_INVALID;
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d) return dfail(v, MHAP_E_STATE, "not a rank of a multi-GPU job (mhap_dist_init / mhap_group_create first)");
  if (n < 0 || (n > 0 && (!bases || !offsets || !lengths || !ids))) return dfail(v, MHAP_E_INVALID, "null argument");
  (void)hipSetDevice(v.device);
  const size_t m = (size_t)std::max<int64_t>(2 * n, 2);
  DCHK(v, d->q_mh.ensure(m * (size_t)v.Hrow * 4)); DCHK(v, d->q_od.ensure(m * (size_t)v.S * 8)); DCHK(v, d->q_mt.ensure(m * META_W * 4));
  if (n > 0) {
    const int rc = internal_sketch_queries(h, bases, offsets, lengths, n, d->q_mh.p, d->q_od.p, d->q_mt.p);
    if (rc != MHAP_OK) return rc;
  }
  return exchange_and_search(h, d, d->q_mh.as<int32_t>(), d->q_od.as<int32_t>(), d->q_mt.as<int32_t>(), 2, ids, n, 0, sink, user);
}

int mhap_dist_set_eager(mhap_handle* h, int32_t on) {
  if (!h) return MHAP_E_INVALID;
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d) return dfail(v, MHAP_E_STATE, "not a rank of a multi-GPU job (mhap_dist_init / mhap_group_create first)");
  d->eager = on != 0; d->eager_go = false; d->eager_done = false; d->eager_valid = false;
  return MHAP_OK;
}

int64_t mhap_dist_eager_searches(mhap_handle* h) {
  if (!h) return 0;
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  return d ? d->eager_searches : 0;
}

// What the transport itself says about this rank: out[0] = ranks in the communicator (ncclCommCount), out[1] = this rank's number in
// it (ncclCommUserRank), out[2] = the device the communicator is bound to (ncclCommCuDevice), out[3] = the handle's device, out[4] =
// RCCL's version code (0 for the in-process peer transport, whose numbers come from the hub); pci = that device's PCI bus id.
// A driver that launched N processes can check that N ranks on N different devices were really seen.
int mhap_dist_info(mhap_handle* h, int32_t* out5, char* pci, size_t pci_cap) {
  if (!h || !out5) return MHAP_E_INVALID;
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d) return dfail(v, MHAP_E_STATE, "not a rank of a multi-GPU job (mhap_dist_init / mhap_group_create first)");
  out5[0] = d->tr->nranks; out5[1] = d->tr->rank; out5[2] = v.device; out5[3] = v.device; out5[4] = 0;
  if (strcmp(d->tr->name(), "rccl") == 0) {
    RcclTransport* rt = (RcclTransport*)d->tr;
    RcclApi& api = rccl();
    if (!rt->comm) return dfail(v, MHAP_E_STATE, "the RCCL communicator was aborted");
    int x = 0;
    if (api.CommCount && api.CommCount(rt->comm, &x) == ncclSuccess) out5[0] = x;
    if (api.CommUserRank && api.CommUserRank(rt->comm, &x) == ncclSuccess) out5[1] = x;
    if (api.CommCuDevice && api.CommCuDevice(rt->comm, &x) == ncclSuccess) out5[2] = x;
    if (api.GetVersion && api.GetVersion(&x) == ncclSuccess) out5[4] = x;
  }
  if (pci && pci_cap) { pci[0] = 0; if (hipDeviceGetPCIBusId(pci, (int)pci_cap, v.device) != hipSuccess) { (void)hipGetLastError(); pci[0] = 0; } }
  return MHAP_OK;
}

// The collective by itself, no sketching and no search: every rank fills `bytes` bytes with (rank + 1), all-gathers them on the
// exchange stream through the handle's transport (the watchdog applies), reads the result back and checks every rank's block.
// ms_out = wall time of the gather.  A fabric or rendezvous problem shows here, separately from any compute kernel.
int mhap_dist_selftest(mhap_handle* h, size_t bytes, double* ms_out) {
  if (!h || bytes == 0 || bytes > ((size_t)1 << 30)) return MHAP_E_INVALID;
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d) return dfail(v, MHAP_E_STATE, "not a rank of a multi-GPU job (mhap_dist_init / mhap_group_create first)");
  (void)hipSetDevice(v.device);
  Transport* tr = d->tr;
  const int N = tr->nranks;
  DevBuf sb, rb;
  hipStream_t cs = d->comm_stream;
  // (the guard first: a failed second allocation must not leak the first; and it drains the exchange stream before the buffers go —
  //  a gather may still be queued on it when a later step fails.  ADVICE r05)
  struct Rel { DevBuf& a; DevBuf& b; hipStream_t s; ~Rel() { (void)hipStreamSynchronize(s); a.release(); b.release(); } } rel{sb, rb, cs};
  // this is a collective: a rank that cannot get its buffers (or fill them) says so in a rendezvous and ALL ranks return together —
  // leaving alone would park the others in the gather until the watchdog's time-out
  bool got = sb.ensure(bytes) == hipSuccess && rb.ensure(bytes * (size_t)N) == hipSuccess;
  got = got && hipMemsetAsync(sb.p, (tr->rank + 1) & 0xFF, bytes, cs) == hipSuccess && hipMemsetAsync(rb.p, 0, bytes * (size_t)N, cs) == hipSuccess;
  if (!got) (void)hipGetLastError();
  {
    const int64_t mine[3] = {got ? 0 : 1, (int64_t)bytes, 0};
    std::vector<int64_t> all((size_t)N * 3, 0);
    const int rc0 = tr->rendezvous(Transport::RV_SELFTEST, mine, all.data(), *v.err);
    if (rc0 != MHAP_OK) { tr->abort(); return rc0; }
    for (int r = 0; r < N; r++) {
      if (all[(size_t)r * 3] != 0) return dfail(v, MHAP_E_NOMEM, "all-gather self-test: rank " + std::to_string(r) + " could not set its buffers up");
      if (all[(size_t)r * 3 + 1] != (int64_t)bytes) return dfail(v, MHAP_E_INVALID, "all-gather self-test: rank " + std::to_string(r) + " asked for another size");
    }
  }
  const double t0 = now_ms();
  int rc = tr->allgather(sb.p, rb.p, bytes, cs, *v.err);
  if (rc != MHAP_OK) { tr->abort(); return rc; }
  if (hipEventRecord(d->ev_small, cs) != hipSuccess) { (void)hipGetLastError(); tr->abort(); return dfail(v, MHAP_E_HIP, "all-gather self-test: hipEventRecord failed"); }
  rc = tr->wait_event(d->ev_small, *v.err);
  if (rc != MHAP_OK) return rc;
  if (ms_out) *ms_out = now_ms() - t0;
  std::vector<unsigned char> host(bytes * (size_t)N);
  DCHK(v, hipMemcpy(host.data(), rb.p, host.size(), hipMemcpyDeviceToHost));
  for (int r = 0; r < N; r++) {
    const unsigned char want = (unsigned char)((r + 1) & 0xFF);
    for (size_t i = 0; i < bytes; i += (bytes > 4096 ? 509 : 1))
      if (host[(size_t)r * bytes + i] != want) return dfail(v, MHAP_E_STATE, "all-gather self-test: rank " + std::to_string(r) + "'s block arrived damaged (byte " + std::to_string(i) + ")");
    if (host[(size_t)r * bytes + bytes - 1] != want) return dfail(v, MHAP_E_STATE, "all-gather self-test: rank " + std::to_string(r) + "'s block arrived short");
  }
  tr->quiesce();
  return MHAP_OK;
}

// The eager exchange of the last add, as the exchange stream saw it (events around the gathers, which run UNDER the add's kernels):
// out4 = {ms of the ordered rows' gather, bytes this rank received in it, ms of the MinHash + meta + id rows' gathers, bytes received}.
// -1 ms: that gather has not run since the last call (no eager add).  Waits for the gathers to finish.  (bench.py --exchange-only)
int mhap_dist_exchange_timing(mhap_handle* h, double* out4) {
  if (!h || !out4) return MHAP_E_INVALID;
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d) return dfail(v, MHAP_E_STATE, "not a rank of a multi-GPU job");
  (void)hipSetDevice(v.device);
  out4[0] = out4[2] = -1.0; out4[1] = d->xt_bytes[0]; out4[3] = d->xt_bytes[1];
  float ms = 0.f;
  if (d->xt_ordered && hipEventSynchronize(d->ev_xt[1]) == hipSuccess && hipEventElapsedTime(&ms, d->ev_xt[0], d->ev_xt[1]) == hipSuccess) out4[0] = ms;
  if (d->xt_small && hipEventSynchronize(d->ev_xt[3]) == hipSuccess && hipEventElapsedTime(&ms, d->ev_xt[2], d->ev_xt[3]) == hipSuccess) out4[2] = ms;
  (void)hipGetLastError();
  d->xt_ordered = d->xt_small = false;
  return MHAP_OK;
}

int mhap_dist_last_timing(mhap_handle* h, double* out3) {
  if (!h || !out3) return MHAP_E_INVALID;
  HandleView v = handle_view(h);
  DistState* d = (DistState*)*v.dist;
  if (!d) return dfail(v, MHAP_E_STATE, "not a rank of a multi-GPU job");
  out3[0] = d->t_small; out3[1] = d->t_wait_big; out3[2] = d->t_total;
  return MHAP_OK;
}

}  // extern "C"

// ================================================================================================================================
// One process, N devices.
struct mhap_group {
  int n = 0;
  std::vector<mhap_handle*> h;
  std::vector<int> dev;
  PeerHub* hub = nullptr;
  int64_t reads_added = 0;
  std::string err;
  std::mutex sink_mu;
};

namespace {

struct SinkFan { mhap_group* g; mhap_record_sink sink; void* user; };
int fan_sink(const mhap_record* r, int64_t n, void* user) {
  SinkFan* f = (SinkFan*)user;
  if (!f->sink) return 0;
  std::lock_guard<std::mutex> lk(f->g->sink_mu);   // one call at a time, as the reference's outputResults lock (AbstractMatchSearch.java:323)
  return f->sink(r, n, f->user);
}

// run fn(rank) on one thread per rank; first error wins
int on_ranks(mhap_group* g, const std::function<int(int)>& fn) {
  std::vector<int> rcs((size_t)g->n, MHAP_OK);
  std::vector<std::thread> th;
  // (a rank that fails before it reaches the exchange must not leave the others at the hub's barrier)
  auto run = [&](int r) { rcs[(size_t)r] = fn(r); if (rcs[(size_t)r] != MHAP_OK && g->hub) g->hub->abort(); };
  for (int r = 1; r < g->n; r++) th.emplace_back([&, r]() { run(r); });
  run(0);
  for (auto& t : th) t.join();
  if (g->hub) g->hub->reset();
  // the error to report is the root cause: a rank that says "another rank failed" is a victim
  int pick = -1;
  for (int r = 0; r < g->n; r++) {
    if (rcs[(size_t)r] == MHAP_OK) continue;
    const bool victim = strstr(mhap_last_error(g->h[(size_t)r]), "another rank failed") != nullptr;
    if (pick < 0 || (!victim && strstr(mhap_last_error(g->h[(size_t)pick]), "another rank failed") != nullptr)) pick = r;
  }
  if (pick >= 0) { g->err = "rank " + std::to_string(pick) + ": " + mhap_last_error(g->h[(size_t)pick]); return rcs[(size_t)pick]; }
  return MHAP_OK;
}

}  // namespace

extern "C" {

int mhap_group_create(const mhap_params* params, const int32_t* devices, int32_t n, mhap_group** out, char* err, size_t errcap) {
  auto seterr = [&](const std::string& m) { if (err && errcap) snprintf(err, errcap, "%s", m.c_str()); };
  if (!params || !out || n < 1 || n > 1024) { seterr("bad group arguments"); return MHAP_E_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { seterr("no HIP device available"); return MHAP_E_HIP; }
  mhap_group* g = new mhap_group();
  g->n = n;
  for (int r = 0; r < n; r++) g->dev.push_back(devices ? devices[r] : r % ndev);
  for (int r = 0; r < n; r++) if (g->dev[(size_t)r] < 0 || g->dev[(size_t)r] >= ndev) { seterr("device ordinal out of range"); delete g; return MHAP_E_INVALID; }
  g->h.assign((size_t)n, nullptr);
  // the handles come up concurrently (most of mhap_create is tables built on the host)
  std::vector<int> rcs((size_t)n, MHAP_OK);
  std::vector<std::string> errs((size_t)n);
  {
    std::vector<std::thread> th;
    for (int r = 0; r < n; r++) th.emplace_back([&, r]() {
      mhap_params p = *params; p.device = g->dev[(size_t)r];
      char e[512] = {0};
      rcs[(size_t)r] = mhap_create(&p, &g->h[(size_t)r], e, sizeof e);
      errs[(size_t)r] = e;
    });
    for (auto& t : th) t.join();
  }
  for (int r = 0; r < n; r++) if (rcs[(size_t)r] != MHAP_OK) { seterr(errs[(size_t)r]); const int rc = rcs[(size_t)r]; mhap_group_destroy(g); return rc; }
  const char* tenv = getenv("MHAP_GROUP_TRANSPORT");
  bool distinct = true;
  for (int a = 0; a < n; a++) for (int b = a + 1; b < n; b++) if (g->dev[(size_t)a] == g->dev[(size_t)b]) distinct = false;
  const bool want_rccl = tenv && strcmp(tenv, "rccl") == 0;
  if (want_rccl && !distinct) { seterr("MHAP_GROUP_TRANSPORT=rccl needs one device per rank"); mhap_group_destroy(g); return MHAP_E_INVALID; }
  if (want_rccl) {
    RcclApi& api = rccl();
    if (!api.lib) { seterr(api.why); mhap_group_destroy(g); return MHAP_E_HIP; }
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    const ncclResult_t r = api.CommInitAll(comms.data(), n, g->dev.data());
    if (r != ncclSuccess) { seterr(std::string("ncclCommInitAll: ") + api.GetErrorString(r)); mhap_group_destroy(g); return MHAP_E_HIP; }
    for (int k = 0; k < n; k++) {
      RcclTransport* tr = new RcclTransport();
      tr->rank = k; tr->nranks = n; tr->comm = comms[(size_t)k]; tr->device = g->dev[(size_t)k];
      if (attach(g->h[(size_t)k], tr) != MHAP_OK) { seterr(mhap_last_error(g->h[(size_t)k])); mhap_group_destroy(g); return MHAP_E_HIP; }
    }
  } else {
    g->hub = new PeerHub(n);
    for (int a = 0; a < n; a++) {
      g->hub->dev[(size_t)a] = g->dev[(size_t)a];
      (void)hipSetDevice(g->dev[(size_t)a]);
      if (hipEventCreateWithFlags(&g->hub->ev[(size_t)a], hipEventDisableTiming) != hipSuccess) { seterr("cannot create the exchange events"); mhap_group_destroy(g); return MHAP_E_HIP; }
      for (int b = 0; b < n; b++) {
        if (g->dev[(size_t)b] == g->dev[(size_t)a]) continue;
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, g->dev[(size_t)a], g->dev[(size_t)b]);
        if (can) { const hipError_t e = hipDeviceEnablePeerAccess(g->dev[(size_t)b], 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError(); }
      }
    }
    (void)hipGetLastError();
    for (int k = 0; k < n; k++) {
      PeerTransport* tr = new PeerTransport();
      tr->rank = k; tr->nranks = n; tr->hub = g->hub;
      if (attach(g->h[(size_t)k], tr) != MHAP_OK) { seterr(mhap_last_error(g->h[(size_t)k])); mhap_group_destroy(g); return MHAP_E_HIP; }
    }
  }
  *out = g;
  return MHAP_OK;
}

void mhap_group_destroy(mhap_group* g) {
  if (!g) return;
  for (mhap_handle* h : g->h) if (h) mhap_destroy(h);
  if (g->hub) {
    for (size_t a = 0; a < g->hub->ev.size(); a++) if (g->hub->ev[a]) { (void)hipSetDevice(g->hub->dev[a]); (void)hipEventDestroy(g->hub->ev[a]); }
    delete g->hub;
  }
  delete g;
}

int32_t mhap_group_size(const mhap_group* g) { return g ? g->n : 0; }
mhap_handle* mhap_group_rank(mhap_group* g, int32_t rank) { return (g && rank >= 0 && rank < g->n) ? g->h[(size_t)rank] : nullptr; }
const char* mhap_group_last_error(const mhap_group* g) { return g ? g->err.c_str() : "null group"; }

int mhap_group_add_reads(mhap_group* g, const char* bases, const int64_t* offsets, const int32_t* lengths, const int64_t* ids, int64_t n) {
  if (!g) return MHAP_E_INVALID;
  if (n <= 0) return MHAP_OK;
  if (!bases || !offsets || !lengths || !ids) { g->err = "null argument"; return MHAP_E_INVALID; }
  const int N = g->n;
  const int64_t first = g->reads_added;
  const int rc = on_ranks(g, [&](int r) {
    // rank r takes the reads whose ordinal in the data set is congruent to r: its share of this call, as index arrays over the caller's bases
    std::vector<int64_t> off, id; std::vector<int32_t> len;
    const int64_t i0 = ((r - first) % N + N) % N;
    for (int64_t i = i0; i < n; i += N) { off.push_back(offsets[i]); len.push_back(lengths[i]); id.push_back(ids[i]); }
    if (off.empty()) {   // (no read for this rank — with the eager exchange on it still joins the add's rendezvous, to say "not this time")
      if (dist_eager_wanted(g->h[(size_t)r])) { const int rx = dist_eager_begin(g->h[(size_t)r], 0, nullptr, false); return rx < 0 ? rx : (int)MHAP_OK; }
      return (int)MHAP_OK;
    }
    return mhap_index_add_reads(g->h[(size_t)r], bases, off.data(), len.data(), id.data(), (int64_t)off.size());
  });
  if (rc == MHAP_OK) g->reads_added += n;
  return rc;
}

int mhap_group_add_scan(mhap_group* g, const mhap_fasta_scan* s) {
  if (!g || !s) return MHAP_E_INVALID;
  const int64_t n = mhap_fasta_scan_reads(s);
  if (n <= 0) return MHAP_OK;
  const int N = g->n;
  const int64_t first = g->reads_added;
  const int rc = on_ranks(g, [&](int r) {
    const int64_t i0 = ((r - first) % N + N) % N;          // rank r takes the records whose ordinal in the data set is congruent to r
    if (i0 >= n) {
      if (dist_eager_wanted(g->h[(size_t)r])) { const int rx = dist_eager_begin(g->h[(size_t)r], 0, nullptr, false); return rx < 0 ? rx : (int)MHAP_OK; }
      return (int)MHAP_OK;
    }
    int64_t entries = 0;
    (void)mhap_index_size(g->h[(size_t)r], &entries);
    if (entries == 0) { const int rr = mhap_index_reserve(g->h[(size_t)r], (n - i0 + N - 1) / N); if (rr != MHAP_OK) return rr; }
    return ingest_add_subset(g->h[(size_t)r], scan_impl(s), i0, N);
  });
  if (rc == MHAP_OK) g->reads_added += n;
  return rc;
}

int mhap_group_clear(mhap_group* g) {
  if (!g) return MHAP_E_INVALID;
  for (mhap_handle* h : g->h) { const int rc = mhap_index_clear(h); if (rc != MHAP_OK) return rc; }
  g->reads_added = 0;
  return MHAP_OK;
}

int mhap_group_find_matches_self(mhap_group* g, mhap_record_sink sink, void* user) {
  if (!g) return MHAP_E_INVALID;
  SinkFan f{g, sink, user};
  return on_ranks(g, [&](int r) { return mhap_dist_find_matches_self(g->h[(size_t)r], fan_sink, &f); });
}

int mhap_group_find_matches_reads(mhap_group* g, const char* bases, const int64_t* offsets, const int32_t* lengths, const int64_t* ids, int64_t n,
                                  mhap_record_sink sink, void* user) {
  if (!g) return MHAP_E_INVALID;
  if (n > 0 && (!bases || !offsets || !lengths || !ids)) { g->err = "null argument"; return MHAP_E_INVALID; }
  SinkFan f{g, sink, user};
  const int N = g->n;
  return on_ranks(g, [&](int r) {
    std::vector<int64_t> off, id; std::vector<int32_t> len;
    for (int64_t i = r; i < n; i += N) { off.push_back(offsets[i]); len.push_back(lengths[i]); id.push_back(ids[i]); }
    return mhap_dist_find_matches_reads(g->h[(size_t)r], bases, off.data(), len.data(), id.data(), (int64_t)off.size(), fan_sink, &f);
  });
}

int mhap_group_get_stats(mhap_group* g, mhap_stats* sum) {
  if (!g || !sum) return MHAP_E_INVALID;
  memset(sum, 0, sizeof *sum);
  for (mhap_handle* h : g->h) {
    mhap_stats s;
    const int rc = mhap_get_stats(h, &s);
    if (rc != MHAP_OK) return rc;
    sum->strands_indexed += s.strands_indexed; sum->queries_searched += s.queries_searched; sum->candidates_compared += s.candidates_compared;
    sum->matches_found += s.matches_found; sum->slot_compares += s.slot_compares; sum->table_elements += s.table_elements;
    sum->slow_pairs += s.slow_pairs; sum->index_splits += s.index_splits;
  }
  // every rank probes every query: the reference's "sequences searched" counts a query once
  sum->queries_searched /= g->n;
  return MHAP_OK;
}

}  // extern "C"
