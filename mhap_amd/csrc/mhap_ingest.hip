// mhap_ingest.hip — streamed FASTA ingest: file text -> 2-bit packed reads in pinned staging -> HBM, overlapped with the kernels.
//
// The reference reads and sketches through a queue (FastaData.dequeue / enqueueNextSequenceInFile, J/impl/FastaData.java:101-204;
// SequenceSketchStreamer.enqueueFullFile with T threads, J/impl/SequenceSketchStreamer.java:179-222).  Round 2's driver did the
// same steps one after the other — parse the whole file into one byte per base, pack it, upload it, sketch it — and 83 % of
// its wall time was host time.  Here the file is mapped and scanned once on all host threads (record boundaries, lengths, is-it-ACGT;
// nothing is copied), and the index is fed in groups: while the GPU sketches and indexes group g, host threads pack group g+1 from
// the mapped text straight into the other of two pinned staging buffers.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <dlfcn.h>

#include <algorithm>
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

namespace mhap {
struct FastaScanImpl {
  const char* D = nullptr; size_t N = 0;      // the text
  void* map_base = nullptr; size_t map_len = 0;
  std::string owned;                          // inflated gz / bz2 input, pipes
  // per non-empty record
  std::vector<uint64_t> hdr, body, end;       // offsets of '>', of the first byte after the header line, of the record's end
  std::vector<int32_t> len;                   // bases
  std::vector<uint8_t> raw;                   // 1: has a char other than ACGT/acgt (kept as upper-cased bytes on the device)
  std::vector<uint8_t> oneline;               // 1: the sequence is one line (packed straight from the text)
  std::vector<int64_t> ids;
  int64_t total_bases = 0;
  std::string names; bool names_done = false;
  ~FastaScanImpl() { if (map_base) munmap(map_base, map_len); }
};
}  // namespace mhap
struct mhap_fasta_scan { FastaScanImpl impl; };

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool prof() { static int p = -1; if (p < 0) p = getenv("MHAP_HOST_PROF") ? 1 : 0; return p == 1; }

int host_threads() { return usable_host_threads(64); }

void parallel_chunks(int64_t n, int nthreads, const std::function<void(int64_t, int64_t, int)>& fn) {
  if (n <= 0) return;
  nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, n));
  if (nthreads == 1) { fn(0, n, 0); return; }
  std::vector<std::thread> th;
  const int64_t chunk = (n + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; t++) {
    const int64_t lo = t * chunk, hi = std::min(n, lo + chunk);
    if (lo >= hi) break;
    th.emplace_back([=, &fn]() { fn(lo, hi, t); });
  }
  for (auto& t : th) t.join();
}

// 2-bit code of a base in either case: (c >> 1) & 3 maps A,C,G,T (a,c,g,t) to 0,1,3,2; x ^ (x >> 1) turns that into 0,1,2,3
inline uint32_t base_code(uint8_t c) { const uint32_t x = (c >> 1) & 3u; return x ^ (x >> 1); }
inline bool is_base(uint8_t c) { return ((0x54474341u >> (8 * base_code(c))) & 0xFFu) == (uint32_t)(c & 0xDFu); }
inline uint8_t upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 'a' + 'A') : c; }   // toUpperCase(Locale.ENGLISH), FastaData.java:194
inline int64_t align4(int64_t v) { return (v + 3) & ~(int64_t)3; }

// pack L bases at s (ACGT in either case) into dst, 4 per byte, zero-padded to a multiple of 4 bytes
void pack_bases(const uint8_t* s, int L, uint8_t* dst) {
  const int full = L >> 2;
  for (int q = 0; q < full; q++)
    dst[q] = (uint8_t)(base_code(s[4 * q]) | (base_code(s[4 * q + 1]) << 2) | (base_code(s[4 * q + 2]) << 4) | (base_code(s[4 * q + 3]) << 6));
  const int nb = (int)align4((L + 3) / 4);
  for (int q = full; q < nb; q++) {
    uint32_t v = 0;
    for (int j = 4 * q; j < L && j < 4 * q + 4; j++) v |= base_code(s[j]) << (2 * (j & 3));
    dst[q] = (uint8_t)v;
  }
}

bool inflate_file(const std::string& name, std::string& data, std::string& err) {
  char buf[1 << 16];
  auto ends = [&](const char* suf) { const size_t n = strlen(suf); return name.size() >= n && name.compare(name.size() - n, n, suf) == 0; };
  if (ends("bz2")) {
    void* lib = dlopen("libbz2.so.1.0", RTLD_NOW);
    if (!lib) lib = dlopen("libbz2.so.1", RTLD_NOW);
    typedef void* (*open_t)(const char*, const char*); typedef int (*read_t)(void*, void*, int); typedef void (*close_t)(void*);
    open_t bzopen = lib ? (open_t)dlsym(lib, "BZ2_bzopen") : nullptr;
    read_t bzread = lib ? (read_t)dlsym(lib, "BZ2_bzread") : nullptr;
    close_t bzclose = lib ? (close_t)dlsym(lib, "BZ2_bzclose") : nullptr;
    if (!bzopen || !bzread || !bzclose) { err = "bzip2 input needs libbz2.so.1.0"; return false; }
    void* bf = bzopen(name.c_str(), "rb");
    if (!bf) { err = "cannot open " + name; return false; }
    int got;
    while ((got = bzread(bf, buf, (int)sizeof buf)) > 0) data.append(buf, (size_t)got);
    bzclose(bf);
    return true;
  }
  gzFile gf = gzopen(name.c_str(), "rb");
  if (!gf) { err = "cannot open " + name; return false; }
  int got;
  while ((got = gzread(gf, buf, (unsigned)sizeof buf)) > 0) data.append(buf, (size_t)got);
  gzclose(gf);
  return true;
}

// One group of reads on its way to the GPU.
struct Slot {
  uint8_t* pin = nullptr; size_t cap = 0;     // pinned staging
  std::vector<ReadDesc> descs; std::vector<int64_t> ids;
  size_t bytes = 0;
  bool ready = false, last = false;
};

// the records idx[lo, hi) of the scan packed into slot (descs, ids, bytes); false on allocation failure
bool pack_group(const FastaScanImpl& sc, const std::vector<int64_t>& idx, int64_t lo, int64_t hi, int min_olap, bool fwd_only, Slot& slot, int nthreads) {
  const int64_t n = hi - lo;
  slot.descs.assign((size_t)n, ReadDesc{});
  slot.ids.resize((size_t)n);
  int64_t store = 0;
  for (int64_t i = 0; i < n; i++) {
    const int64_t r = idx[(size_t)(lo + i)];
    ReadDesc& d = slot.descs[(size_t)i];
    d.length = sc.len[(size_t)r];
    d.flags = 0;
    if (d.length < min_olap) d.flags |= MHAP_RD_SKIP;                 // SequenceSketchStreamer.java:129-133
    if (sc.raw[(size_t)r]) d.flags |= MHAP_RD_RAW;
    if (fwd_only) d.flags |= MHAP_RD_FWDONLY;
    d.base_off = store;
    if (!(d.flags & MHAP_RD_SKIP)) store += (d.flags & MHAP_RD_RAW) ? align4(d.length) : align4((d.length + 3) / 4);
    slot.ids[(size_t)i] = sc.ids[(size_t)r];
  }
  const size_t need = (size_t)std::max<int64_t>(store, 4);
  if (slot.cap < need) {
    if (slot.pin) (void)hipHostFree(slot.pin);
    slot.pin = nullptr; slot.cap = 0;
    if (hipHostMalloc((void**)&slot.pin, need + need / 8, hipHostMallocPortable) != hipSuccess) return false;
    slot.cap = need + need / 8;
  }
  slot.bytes = need;
  uint8_t* hs = slot.pin;
  parallel_chunks(n, nthreads, [&](int64_t a, int64_t b, int) {
    std::vector<uint8_t> tmp;
    for (int64_t i = a; i < b; i++) {
      const ReadDesc& d = slot.descs[(size_t)i];
      if (d.flags & MHAP_RD_SKIP) continue;
      const int64_t r = idx[(size_t)(lo + i)];
      const uint8_t* s = (const uint8_t*)sc.D + sc.body[(size_t)r];
      const int L = d.length;
      if (!sc.oneline[(size_t)r]) {   // several sequence lines: joined in a thread-local buffer (stays in cache), then packed like a single line
        tmp.resize((size_t)L);
        const uint8_t* e = (const uint8_t*)sc.D + sc.end[(size_t)r];
        int w = 0;
        for (const uint8_t* p = s; p < e; p++) { const uint8_t c = *p; if (c != '\n' && c != '\r') tmp[(size_t)w++] = c; }
        s = tmp.data();
      } else {
        while (*s == '\n' || *s == '\r') s++;   // (the line break that ends the header line)
      }
      uint8_t* dst = hs + d.base_off;
      if (d.flags & MHAP_RD_RAW) {
        for (int j = 0; j < L; j++) dst[j] = upper(s[j]);
        for (int j = L; j < (int)align4(L); j++) dst[j] = 0;
      } else pack_bases(s, L, dst);
    }
  });
  return true;
}

}  // namespace

namespace mhap {

// groups of <= group_bases bases (and <= 2^21 reads) over the subset start, start + stride, ...; for each group `consume(slot)` runs on
// the calling thread while a producer thread packs the next group into the other slot
// on_plan(groups): called once before anything is consumed, also for an empty subset (a collective caller agrees with its peers there)
int ingest_pipeline(const FastaScanImpl& sc, int64_t start, int64_t stride, int min_olap, bool fwd_only,
                    const std::function<int(Slot&)>& consume, std::string& err, const std::function<int(size_t)>& on_plan = nullptr) {
  std::vector<int64_t> idx;
  const int64_t nrec = (int64_t)sc.len.size();
  for (int64_t r = start; r < nrec; r += stride) idx.push_back(r);
  if (idx.empty()) return on_plan ? on_plan(0) : MHAP_OK;
  int64_t group_bases = 256LL << 20;
  if (const char* e = getenv("MHAP_INGEST_GROUP_BASES")) { const long long v = atoll(e); if (v > 0) group_bases = v; }
  // a quarter-size first group lets the GPU start early; after it the packing (15 GB/s on 16 threads) stays ahead of the kernels
  // (9 Gbase/s) and every launch is large enough to fill the GPU.  Measured at C2 (1 Gbase): ramped groups of 16..128 Mbase 159 ms
  // for 108 ms of kernels — small launches pay their fixed costs —, this scheme see profiles/r03_e2e_probe.txt
  std::vector<std::pair<int64_t, int64_t>> groups;
  for (int64_t lo = 0; lo < (int64_t)idx.size();) {
    const int64_t cap = groups.empty() ? std::max<int64_t>(group_bases >> 2, 1 << 20) : group_bases;
    int64_t hi = lo, tot = 0;
    while (hi < (int64_t)idx.size() && (hi == lo || tot + sc.len[(size_t)idx[(size_t)hi]] <= cap) && hi - lo < (1 << 21)) { tot += sc.len[(size_t)idx[(size_t)hi]]; hi++; }
    groups.emplace_back(lo, hi);
    lo = hi;
  }
  if (on_plan) { const int rp = on_plan(groups.size()); if (rp != MHAP_OK) return rp; }
  Slot slots[2];
  std::mutex mu; std::condition_variable cv;
  bool failed = false;
  const int nthreads = host_threads();
  std::thread producer([&]() {
    for (size_t g = 0; g < groups.size(); g++) {
      Slot& s = slots[g & 1];
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return !s.ready || failed; }); if (failed) return; }
      const double t0 = now_s();
      const bool ok = pack_group(sc, idx, groups[g].first, groups[g].second, min_olap, fwd_only, s, nthreads);
      if (prof()) fprintf(stderr, "[ingest] group %zu packed: %lld reads, %.1f MB, %.3f s (at %.3f)\n", g, (long long)(groups[g].second - groups[g].first), s.bytes / 1e6, now_s() - t0, now_s());
      { std::lock_guard<std::mutex> lk(mu); if (!ok) failed = true; s.ready = true; s.last = g + 1 == groups.size(); }
      cv.notify_all();
      if (!ok) return;
    }
  });
  int rc = MHAP_OK;
  for (size_t g = 0; g < groups.size(); g++) {
    Slot& s = slots[g & 1];
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return s.ready || failed; }); }
    if (failed) { rc = MHAP_E_NOMEM; err = "cannot allocate pinned staging memory"; break; }
    const double t0 = now_s();
    rc = consume(s);
    if (prof()) fprintf(stderr, "[ingest] group %zu consumed in %.3f s (at %.3f)\n", g, now_s() - t0, now_s());
    { std::lock_guard<std::mutex> lk(mu); s.ready = false; if (rc != MHAP_OK) failed = true; }
    cv.notify_all();
    if (rc != MHAP_OK) break;
  }
  { std::lock_guard<std::mutex> lk(mu); if (rc != MHAP_OK) failed = true; }
  cv.notify_all();
  producer.join();
  for (Slot& s : slots) if (s.pin) (void)hipHostFree(s.pin);
  return rc;
}

const FastaScanImpl* scan_impl(const mhap_fasta_scan* s) { return &s->impl; }

int ingest_add_subset(mhap_handle* h, const FastaScanImpl* scan, int64_t start, int64_t stride) {
  HandleView v = handle_view(h);
  (void)hipSetDevice(v.device);
  std::string err;
  // Under the eager exchange every add is a rendezvous of the ranks, and a rank's share may split into a different number of ingest groups
  // than another's: the ranks agree on the plan FIRST — one rendezvous per call of this function whatever the group counts —
  // and keep the eager exchange only when every rank adds exactly one group; otherwise it is suspended on all of them for this ingest
  // (the exchange then happens at search time) and no add makes a rendezvous.
  const int rc = ingest_pipeline(*scan, start, stride, v.min_olap_length, false, [&](Slot& s) {
    int r = internal_stage_packed(h, s.descs.data(), s.ids.data(), (int64_t)s.descs.size(), s.pin, s.bytes);
    if (r == MHAP_OK) r = mhap_index_add_staged(h);
    return r;
  }, err, [&](size_t ngroups) { const int r = dist_ingest_scope(h, (int64_t)ngroups); return r < 0 ? r : MHAP_OK; });
  dist_ingest_scope_end(h);
  if (rc != MHAP_OK && !err.empty()) *v.err = err;
  return rc;
}

}  // namespace mhap

extern "C" {

int mhap_fasta_scan_open(const char* path, int64_t id_offset, mhap_fasta_scan** out, char* errbuf, size_t errcap) {
  auto seterr = [&](const std::string& m) { if (errbuf && errcap) snprintf(errbuf, errcap, "%s", m.c_str()); };
  if (!path || !out) { seterr("null argument"); return MHAP_E_INVALID; }
  const double t0 = now_s();
  mhap_fasta_scan* S = new mhap_fasta_scan();
  FastaScanImpl& sc = S->impl;
  // Utils.getFile (J/utils/Utils.java:228-266): *bz2 -> bzip2, *gz -> gzip, otherwise the name must carry a FASTA suffix
  const std::string name(path);
  auto ends = [&](const char* suf) { const size_t n = strlen(suf); return name.size() >= n && name.compare(name.size() - n, n, suf) == 0; };
  if (ends("bz2") || ends("gz")) {
    std::string e;
    if (!inflate_file(name, sc.owned, e)) { seterr(e); delete S; return MHAP_E_INVALID; }
    sc.D = sc.owned.data(); sc.N = sc.owned.size();
  } else {
    static const char* suffixes[] = {"fna", "contigs", "contig", "final", "fasta", "fa"};   // FastaData.java:50
    bool ok = false;
    for (const char* suf : suffixes) ok = ok || ends(suf);
    if (!ok) { seterr(std::string("Unknown file format of file ") + path + "."); delete S; return MHAP_E_INVALID; }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { seterr(std::string("cannot open ") + path); delete S; return MHAP_E_INVALID; }
    struct stat sb;
    if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
      void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);   // (no MAP_POPULATE: the scan threads fault their own parts in)
      if (m != MAP_FAILED) { sc.map_base = m; sc.map_len = (size_t)sb.st_size; (void)madvise(m, sc.map_len, MADV_WILLNEED); }
    }
    if (!sc.map_base) {   // pipes, empty files, mmap refused: read it
      char buf[1 << 16]; ssize_t got;
      while ((got = read(fd, buf, sizeof buf)) > 0) sc.owned.append(buf, (size_t)got);
      sc.D = sc.owned.data(); sc.N = sc.owned.size();
    } else { sc.D = (const char*)sc.map_base; sc.N = sc.map_len; }
    close(fd);
  }
  const char* D = sc.D; const size_t N = sc.N;
  if (N > 0 && D[0] != '>') { seterr("Next sequence does not start with >. Invalid format."); delete S; return MHAP_E_INVALID; }   // FastaData.java:150-151
  // pass 1 (parallel over the text): the '>' that open records (BufferedReader.readLine: a line ends at \n, \r or \r\n)
  const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), (int64_t)(N >> 22) + 1));   // >= 4 MB of text per thread
  std::vector<std::vector<uint64_t>> found((size_t)nthreads);
  parallel_chunks((int64_t)N, nthreads, [&](int64_t lo, int64_t hi, int t) {
    std::vector<uint64_t>& v = found[(size_t)t];
    for (const char* q = D + lo; q < D + hi;) {
      const char* g = (const char*)memchr(q, '>', (size_t)(D + hi - q));
      if (!g) break;
      if (g == D || g[-1] == '\n' || g[-1] == '\r') v.push_back((uint64_t)(g - D));
      q = g + 1;
    }
  });
  std::vector<uint64_t> hdr;
  for (auto& v : found) hdr.insert(hdr.end(), v.begin(), v.end());
  const int64_t nrec = (int64_t)hdr.size();
  // pass 2 (parallel over records): header line, bases, is-it-ACGT, is-it-one-line
  std::vector<uint64_t> body((size_t)nrec), rend((size_t)nrec);
  std::vector<int64_t> rlen((size_t)nrec);
  std::vector<uint8_t> rraw((size_t)nrec), rone((size_t)nrec);
  parallel_chunks(nrec, host_threads(), [&](int64_t lo, int64_t hi, int) {
    for (int64_t r = lo; r < hi; r++) {
      const size_t e = (r + 1 < nrec) ? (size_t)hdr[(size_t)r + 1] : N;
      size_t p = (size_t)hdr[(size_t)r];
      const char* nl = (const char*)memchr(D + p, '\n', e - p);
      const char* cr = (const char*)memchr(D + p, '\r', nl ? (size_t)(nl - (D + p)) : e - p);
      p = cr ? (size_t)(cr - D) : (nl ? (size_t)(nl - D) : e);
      body[(size_t)r] = p; rend[(size_t)r] = e;
      int64_t breaks = 0; uint32_t bad = 0;
      const uint8_t* s = (const uint8_t*)D;
      for (size_t i = p; i < e; i++) {
        const uint8_t c = s[i];
        const uint32_t isnl = (c == '\n' || c == '\r') ? 1u : 0u;
        breaks += isnl;
        bad |= (isnl | (is_base(c) ? 1u : 0u)) ^ 1u;
      }
      rlen[(size_t)r] = (int64_t)(e - p) - breaks;
      rraw[(size_t)r] = (uint8_t)bad;
      // one line: every line break sits at the two ends of the body
      size_t a = p, b = e;
      while (a < b && (s[a] == '\n' || s[a] == '\r')) a++;
      while (b > a && (s[b - 1] == '\n' || s[b - 1] == '\r')) b--;
      rone[(size_t)r] = (int64_t)(b - a) == rlen[(size_t)r] ? 1 : 0;
    }
  });
  int64_t count = 0;
  for (int64_t r = 0; r < nrec; r++) {
    if (rlen[(size_t)r] <= 0) continue;          // ids count non-empty records only, 1-based (FastaData.java:180-181)
    if (rlen[(size_t)r] > INT32_MAX) { seterr("sequence longer than 2^31-1"); delete S; return MHAP_E_INVALID; }
    count++;
    sc.hdr.push_back(hdr[(size_t)r]); sc.body.push_back(body[(size_t)r]); sc.end.push_back(rend[(size_t)r]);
    sc.len.push_back((int32_t)rlen[(size_t)r]); sc.raw.push_back(rraw[(size_t)r]); sc.oneline.push_back(rone[(size_t)r]);
    sc.ids.push_back(count + id_offset);
    sc.total_bases += rlen[(size_t)r];
  }
  if (prof()) fprintf(stderr, "[ingest] scan of %s: %lld reads, %.1f Mbase, %.3f s on %d threads\n", path, (long long)count, sc.total_bases / 1e6, now_s() - t0, host_threads());
  *out = S;
  return MHAP_OK;
}

void mhap_fasta_scan_free(mhap_fasta_scan* s) { delete s; }
int64_t mhap_fasta_scan_reads(const mhap_fasta_scan* s) { return s ? (int64_t)s->impl.len.size() : 0; }
int64_t mhap_fasta_scan_bases(const mhap_fasta_scan* s) { return s ? s->impl.total_bases : 0; }

int mhap_fasta_scan_info(mhap_fasta_scan* s, int64_t* ids, int32_t* lengths, const char** headers, int64_t* headers_bytes) {
  if (!s) return MHAP_E_INVALID;
  FastaScanImpl& sc = s->impl;
  const size_t n = sc.len.size();
  if (ids && n) memcpy(ids, sc.ids.data(), n * 8);
  if (lengths && n) memcpy(lengths, sc.len.data(), n * 4);
  if (headers || headers_bytes) {
    if (!sc.names_done) {   // substring(1).split("[\\s,]+", 2)[0] (FastaData.java:155-156)
      for (size_t r = 0; r < n; r++) {
        size_t a = (size_t)sc.hdr[r] + 1, e = a;
        while (e < (size_t)sc.body[r] && !(isspace((unsigned char)sc.D[e]) || sc.D[e] == ',')) e++;
        sc.names.append(sc.D + a, e - a);
        sc.names.push_back('\0');
      }
      sc.names_done = true;
    }
    if (headers) *headers = sc.names.data();
    if (headers_bytes) *headers_bytes = (int64_t)sc.names.size();
  }
  return MHAP_OK;
}

int mhap_index_add_scan(mhap_handle* h, const mhap_fasta_scan* s) {
  if (!h || !s) return MHAP_E_INVALID;
  int64_t entries = 0;
  (void)mhap_index_size(h, &entries);
  if (entries == 0) { const int rc = mhap_index_reserve(h, (int64_t)s->impl.len.size()); if (rc != MHAP_OK) return rc; }
  return ingest_add_subset(h, &s->impl, 0, 1);
}

int mhap_find_matches_scan(mhap_handle* h, const mhap_fasta_scan* s, mhap_record_sink sink, void* user) {
  if (!h || !s) return MHAP_E_INVALID;
  // -q mode: groups of query reads; the library's entry point takes one byte per base, so a group's text is joined here (queries are
  // a small share of a run)
  const FastaScanImpl& sc = s->impl;
  const int64_t n = (int64_t)sc.len.size();
  const int64_t group_bases = 256LL << 20;
  for (int64_t lo = 0; lo < n;) {
    int64_t hi = lo, tot = 0;
    while (hi < n && (hi == lo || tot + sc.len[(size_t)hi] <= group_bases)) { tot += sc.len[(size_t)hi]; hi++; }
    std::vector<char> bases((size_t)std::max<int64_t>(tot, 1));
    std::vector<int64_t> offs((size_t)(hi - lo));
    int64_t at = 0;
    for (int64_t r = lo; r < hi; r++) { offs[(size_t)(r - lo)] = at; at += sc.len[(size_t)r]; }
    parallel_chunks(hi - lo, host_threads(), [&](int64_t a, int64_t b, int) {
      for (int64_t i = a; i < b; i++) {
        const int64_t r = lo + i;
        char* w = bases.data() + offs[(size_t)i];
        for (size_t p = (size_t)sc.body[(size_t)r]; p < (size_t)sc.end[(size_t)r]; p++) { const char c = sc.D[p]; if (c != '\n' && c != '\r') *w++ = (char)upper((uint8_t)c); }
      }
    });
    const int rc = mhap_find_matches_reads(h, bases.data(), offs.data(), sc.len.data() + lo, sc.ids.data() + lo, hi - lo, sink, user);
    if (rc != MHAP_OK) return rc;
    lo = hi;
  }
  return MHAP_OK;
}

}  // extern "C"
