// host_util.cpp — host-side helpers of libmhaphip.so that need no GPU: the reference's IO conventions
// (FASTA ingest, overlap-record text format) and the deterministic synthetic read generator.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mhap_hip.h"
#include "mhap_internal.hpp"

namespace {

// ---- Java Formatter "%.6f": shortest round-trip decimal digits, then HALF_UP at 6 places -------
// (JDK 8 java.util.Formatter -> sun.misc.FormattedFloatingDecimal; J/impl/MatchResult.java:100).
int fmt6(double v, char* out, size_t cap) {
  if (std::isnan(v)) return snprintf(out, cap, "NaN");
  if (std::isinf(v)) return snprintf(out, cap, v > 0 ? "Infinity" : "-Infinity");
  const bool neg = std::signbit(v);
  const double a = std::fabs(v);
  // Fast path (every record pays for two of these): unless the value sits within 10^-9 of a HALF_UP tie at the sixth decimal, the
  // shortest round-trip digits (which differ from the exact value by < 1 ulp) round like the exact value, so a multiply and a floor
  // decide.  Ties and huge values take the digit-string path below.
  if (a < 4e6) {   // (beyond 2^22 the product's rounding error approaches the 1e-3 tie margin: those take the digit-string path)
    const double x = a * 1e6, fl = std::floor(x), frac = x - fl;
    if (std::fabs(frac - 0.5) > 1e-3) {
      unsigned long long q = (unsigned long long)fl + (frac > 0.5 ? 1ULL : 0ULL);
      char tmp[40];
      int n = 0;
      for (int i = 0; i < 6; i++) { tmp[n++] = (char)('0' + q % 10); q /= 10; }
      tmp[n++] = '.';
      do { tmp[n++] = (char)('0' + q % 10); q /= 10; } while (q);
      if (neg) tmp[n++] = '-';
      if ((size_t)n + 1 > cap) return -1;
      for (int i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
      out[n] = 0;
      return n;
    }
  }
  // shortest digit string d1.d2...dn x 10^e that parses back to a
  char sci[40];
  int nd = 1;
  for (; nd <= 17; nd++) {
    snprintf(sci, sizeof sci, "%.*e", nd - 1, a);
    if (strtod(sci, nullptr) == a) break;
  }
  char digs[24];
  int n = 0;
  const char* epos = strchr(sci, 'e');
  for (const char* p = sci; p < epos; p++) if (*p != '.') digs[n++] = *p;
  const int e10 = atoi(epos + 1);
  // decimal expansion: integer part has (e10+1) digits when e10 >= 0
  std::string ip, fp;
  for (int i = 0; i <= e10; i++) ip.push_back(i < n ? digs[i] : '0');
  if (ip.empty()) ip = "0";
  for (int i = -1; i > e10; i--) fp.push_back('0');          // leading zeros of a pure fraction
  for (int i = std::max(0, e10 + 1); i < n; i++) fp.push_back(digs[i]);
  bool carry = false;
  if (fp.size() > 6) { carry = fp[6] >= '5'; fp.resize(6); }   // HALF_UP on the decimal digits
  while (fp.size() < 6) fp.push_back('0');
  if (carry) {
    std::string all = ip + fp;
    int i = (int)all.size() - 1;
    for (; i >= 0; i--) { if (all[i] == '9') all[i] = '0'; else { all[i]++; break; } }
    if (i < 0) all.insert(all.begin(), '1');
    ip = all.substr(0, all.size() - 6);
    fp = all.substr(all.size() - 6);
  }
  return snprintf(out, cap, "%s%s.%s", neg ? "-" : "", ip.c_str(), fp.c_str());
}

// ---- deterministic PRNG -------------------------------------------------------------------------
struct SplitMix64 { uint64_t s; uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); } };
struct Xoshiro256ss {
  uint64_t s[4];
  explicit Xoshiro256ss(uint64_t seed) { SplitMix64 sm{seed}; for (auto& x : s) x = sm.next(); }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  uint64_t below(uint64_t n) { return (uint64_t)(unit() * (double)n); }
};

}  // namespace

extern "C" {

int mhap_abi_version(void) { return MHAP_ABI_VERSION; }
int mhap_abi_sizes(int32_t* out4) {
  if (!out4) return MHAP_E_INVALID;
  out4[0] = (int32_t)sizeof(mhap_params); out4[1] = (int32_t)sizeof(mhap_record); out4[2] = (int32_t)sizeof(mhap_stats); out4[3] = (int32_t)sizeof(mhap_kernel_times);
  return MHAP_OK;
}

int mhap_format_record(const mhap_record* r, char* out, size_t cap) {
  if (!r || !out || cap == 0) return -1;
  const double score = r->score > 1.0 ? 1.0 : r->score;            // MatchResult.java:61-64
  // "%d %d %f %f %d %d %d %d %d %d %d %d" (MatchResult.java:98-113), digits written by hand: a snprintf of twelve fields costs more
  // than the GPU spends on the record
  char buf[320];
  char* p = buf;
  auto put = [&p](long long v) {
    char t[24];
    int n = 0;
    unsigned long long u = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
    do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *p++ = '-';
    while (n) *p++ = t[--n];
  };
  put((long long)r->from_id); *p++ = ' ';
  put((long long)r->to_id); *p++ = ' ';
  int k = fmt6(1.0 - score, p, 64); if (k < 0) return -1; p += k; *p++ = ' ';
  k = fmt6(r->raw, p, 64); if (k < 0) return -1; p += k; *p++ = ' ';
  *p++ = '0'; *p++ = ' ';
  put(r->a1); *p++ = ' '; put(r->a2); *p++ = ' '; put(r->alen); *p++ = ' ';
  *p++ = r->to_rc ? '1' : '0'; *p++ = ' ';
  put(r->b1); *p++ = ' '; put(r->b2); *p++ = ' '; put(r->blen);
  const size_t len = (size_t)(p - buf);
  if (len + 1 > cap) { if (cap) { memcpy(out, buf, cap - 1); out[cap - 1] = 0; } return (int)len; }   // snprintf semantics: truncated, full length returned
  memcpy(out, buf, len); out[len] = 0;
  return (int)len;
}

// FastaData.enqueueNextSequenceInFile (J/impl/FastaData.java:125-204).  Deviation (documented in DESIGN.md):
// an empty record is skipped instead of stopping a worker thread (reference behaviour is thread-count dependent).
int mhap_fasta_read(const char* path, int64_t id_offset, mhap_fasta* out, char* err, size_t errcap) {
  auto seterr = [&](const std::string& m) { if (err && errcap) snprintf(err, errcap, "%s", m.c_str()); };
  if (!path || !out) { seterr("null argument"); return MHAP_E_INVALID; }
  memset(out, 0, sizeof *out);
  // Utils.getFile (J/utils/Utils.java:228-266): *bz2 -> bzip2, *gz -> gzip, otherwise the name must carry a FASTA suffix
  const std::string name(path);
  auto ends = [&](const char* suf) { const size_t n = strlen(suf); return name.size() >= n && name.compare(name.size() - n, n, suf) == 0; };
  std::string data;
  void* map_base = nullptr;
  size_t map_len = 0;
  char buf[1 << 16];
  if (ends("bz2")) {
    // libbz2 ships without headers in this image: bind the three stdio-style entry points at run time
    void* lib = dlopen("libbz2.so.1.0", RTLD_NOW);
    if (!lib) lib = dlopen("libbz2.so.1", RTLD_NOW);
    typedef void* (*open_t)(const char*, const char*); typedef int (*read_t)(void*, void*, int); typedef void (*close_t)(void*);
    open_t bzopen = lib ? (open_t)dlsym(lib, "BZ2_bzopen") : nullptr;
    read_t bzread = lib ? (read_t)dlsym(lib, "BZ2_bzread") : nullptr;
    close_t bzclose = lib ? (close_t)dlsym(lib, "BZ2_bzclose") : nullptr;
    if (!bzopen || !bzread || !bzclose) { seterr("bzip2 input needs libbz2.so.1.0"); return MHAP_E_INVALID; }
    void* bf = bzopen(path, "rb");
    if (!bf) { seterr(std::string("cannot open ") + path); return MHAP_E_INVALID; }
    int got;
    while ((got = bzread(bf, buf, (int)sizeof buf)) > 0) data.append(buf, (size_t)got);
    bzclose(bf);
  } else if (ends("gz")) {
    gzFile gf = gzopen(path, "rb");
    if (!gf) { seterr(std::string("cannot open ") + path); return MHAP_E_INVALID; }
    int got;
    while ((got = gzread(gf, buf, (unsigned)sizeof buf)) > 0) data.append(buf, (size_t)got);
    gzclose(gf);
  } else {
    static const char* suffixes[] = {"fna", "contigs", "contig", "final", "fasta", "fa"};   // FastaData.java:50
    bool ok = false;
    for (const char* suf : suffixes) ok = ok || ends(suf);
    if (!ok) { seterr(std::string("Unknown file format of file ") + path + "."); return MHAP_E_INVALID; }
    // plain text is parsed in place from a read-only mapping (no copy of the file)
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { seterr(std::string("cannot open ") + path); return MHAP_E_INVALID; }
    struct stat sb;
    if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
      void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
      if (m != MAP_FAILED) { map_base = m; map_len = (size_t)sb.st_size; }
    }
    if (!map_base) {   // pipes, empty files, mmap refused: read it
      ssize_t got;
      while ((got = read(fd, buf, sizeof buf)) > 0) data.append(buf, (size_t)got);
    }
    close(fd);
  }
  struct Unmap { void* p; size_t n; ~Unmap() { if (p) munmap(p, n); } } unmap{map_base, map_len};
  // BufferedReader.readLine: a line ends at \n, \r or \r\n; a line that starts with '>' opens a record, every other line is
  // sequence (concatenated, upper-cased :194).  Three passes over the text: record boundaries (memchr, serial), sequence
  // lengths and the copy (both parallel over records).
  const size_t N = map_base ? map_len : data.size();
  const char* D = map_base ? (const char*)map_base : data.data();
  if (N > 0 && D[0] != '>') { seterr("Next sequence does not start with >. Invalid format."); return MHAP_E_INVALID; }   // :150-151
  std::vector<size_t> hdr;   // offsets of the '>' that open records
  for (const char* q = D; q && q < D + N;) {
    const char* g = (const char*)memchr(q, '>', (size_t)(D + N - q));
    if (!g) break;
    if (g == D || g[-1] == '\n' || g[-1] == '\r') hdr.push_back((size_t)(g - D));
    q = g + 1;
  }
  const int64_t nrec = (int64_t)hdr.size();
  std::vector<size_t> body((size_t)nrec), bend((size_t)nrec);
  std::vector<int64_t> rlen((size_t)nrec);
  const int hw = mhap::usable_host_threads(64);
  const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)hw, (int64_t)(N >> 22) + 1));   // >= 4 MB of text per thread
  auto par = [&](const std::function<void(int64_t, int64_t)>& fn) {
    if (nthreads == 1 || nrec < 2) { fn(0, nrec); return; }
    std::vector<std::thread> th;
    const int64_t chunk = (nrec + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
      const int64_t lo = t * chunk, hi = std::min<int64_t>(nrec, lo + chunk);
      if (lo >= hi) break;
      th.emplace_back([=, &fn]() { fn(lo, hi); });
    }
    for (auto& t : th) t.join();
  };
  par([&](int64_t lo, int64_t hi) {
    for (int64_t r = lo; r < hi; r++) {
      const size_t end = (r + 1 < nrec) ? hdr[(size_t)r + 1] : N;
      size_t p = hdr[(size_t)r];
      while (p < end && D[p] != '\n' && D[p] != '\r') p++;   // header line
      body[(size_t)r] = p; bend[(size_t)r] = end;
      int64_t cnt = 0;
      for (size_t i = p; i < end; i++) cnt += (D[i] != '\n' && D[i] != '\r') ? 1 : 0;
      rlen[(size_t)r] = cnt;
    }
  });
  std::vector<int64_t> offs, ids, dst((size_t)nrec);
  std::vector<int32_t> lens;
  int64_t total = 0, count = 0;
  for (int64_t r = 0; r < nrec; r++) {
    dst[(size_t)r] = total;
    if (rlen[(size_t)r] > 0) {   // ids count non-empty records only, 1-based (:180-181)
      if (rlen[(size_t)r] > INT32_MAX) { seterr("sequence longer than 2^31-1"); return MHAP_E_INVALID; }
      count++; offs.push_back(total); lens.push_back((int32_t)rlen[(size_t)r]); ids.push_back(count + id_offset);
      total += rlen[(size_t)r];
    }
  }
  std::string names;
  for (int64_t r = 0; r < nrec; r++) {
    if (rlen[(size_t)r] <= 0) continue;
    size_t a = hdr[(size_t)r] + 1, e = a;
    while (e < body[(size_t)r] && !(isspace((unsigned char)D[e]) || D[e] == ',')) e++;   // substring(1).split("[\\s,]+", 2)[0]
    names.append(D + a, e - a);
    names.push_back('\0');
  }
  out->n = (int64_t)offs.size();
  out->total_bases = total;
  out->headers_bytes = (int64_t)names.size();
  out->headers = (char*)malloc(std::max<size_t>(names.size(), 1));
  if (out->headers) memcpy(out->headers, names.data(), names.size());
  out->bases = (char*)malloc(std::max<size_t>((size_t)total, 1));
  out->offsets = (int64_t*)malloc(std::max<size_t>(offs.size(), 1) * 8);
  out->lengths = (int32_t*)malloc(std::max<size_t>(offs.size(), 1) * 4);
  out->ids = (int64_t*)malloc(std::max<size_t>(offs.size(), 1) * 8);
  if (!out->bases || !out->offsets || !out->lengths || !out->ids || !out->headers) { mhap_fasta_free(out); seterr("out of memory"); return MHAP_E_NOMEM; }
  char* B = out->bases;
  par([&](int64_t lo, int64_t hi) {
    for (int64_t r = lo; r < hi; r++) {
      char* w = B + dst[(size_t)r];
      for (size_t i = body[(size_t)r]; i < bend[(size_t)r]; i++) {
        const char c = D[i];
        if (c == '\n' || c == '\r') continue;
        *w++ = (c >= 'a' && c <= 'z') ? (char)(c - 'a' + 'A') : c;       // toUpperCase(Locale.ENGLISH) :194
      }
    }
  });
  if (!offs.empty()) { memcpy(out->offsets, offs.data(), offs.size() * 8); memcpy(out->lengths, lens.data(), lens.size() * 4); memcpy(out->ids, ids.data(), ids.size() * 8); }
  return MHAP_OK;
}

void mhap_fasta_free(mhap_fasta* f) {
  if (!f) return;
  free(f->bases); free(f->offsets); free(f->lengths); free(f->ids); free(f->headers);
  memset(f, 0, sizeof *f);
}

// Synthetic PacBio-style reads.  Circular random genome of n*len/coverage bp; every read
// has its own generator seeded from (seed, index) so generation is order- and thread-independent.
int mhap_synth_reads(uint64_t seed, int64_t n, int32_t len, double coverage, double error_rate, char* bases) {
  return mhap_synth_reads_shard(seed, n, len, coverage, error_rate, 0, 1, bases);
}

// Reads r = shard, shard+nshards, ... of the n-read data set (identical bytes to the full generation).
int mhap_synth_reads_shard(uint64_t seed, int64_t n, int32_t len, double coverage, double error_rate, int64_t shard, int64_t nshards,
                           char* bases) {
  return mhap_synth_reads_repeats(seed, n, len, coverage, error_rate, shard, nshards, 0, 0, 0.0, bases);
}
int mhap_synth_reads_repeats(uint64_t seed, int64_t n, int32_t len, double coverage, double error_rate, int64_t shard, int64_t nshards,
                             int32_t rep_len, int32_t rep_spacing, double rep_div, char* bases) {
  if (rep_len < 0 || (rep_len > 0 && (rep_spacing <= rep_len || rep_div < 0.0 || rep_div >= 1.0))) return MHAP_E_INVALID;
  if (n < 0 || len <= 0 || !bases || coverage <= 0.0 || error_rate < 0.0 || error_rate >= 1.0) return MHAP_E_INVALID;
  if (nshards < 1 || shard < 0 || shard >= nshards) return MHAP_E_INVALID;
  if (n == 0) return MHAP_OK;
  const int64_t G = std::max<int64_t>((int64_t)((double)n * (double)len / coverage), (int64_t)len + 1);
  std::vector<uint8_t> genome((size_t)G);
  {
    Xoshiro256ss g(SplitMix64{seed}.next());
    for (int64_t i = 0; i < G; i += 32) {
      uint64_t r = g.next();
      for (int j = 0; j < 32 && i + j < G; j++) genome[(size_t)(i + j)] = (uint8_t)((r >> (2 * j)) & 3);
    }
    // planted repeat family (BASELINE configs[4]: the workload the -f k-mer filter exists for): one random element of
    // rep_len bases, a copy with rep_div substitutions in every rep_spacing-base stretch of the genome (jittered position)
    if (rep_len > 0) {
      Xoshiro256ss e(SplitMix64{seed ^ 0x5245504541545321ULL}.next());
      std::vector<uint8_t> elem((size_t)rep_len);
      for (auto& b : elem) b = (uint8_t)(e.next() >> 62);
      for (int64_t c0 = 0; c0 + rep_spacing <= G; c0 += rep_spacing) {
        const int64_t at = c0 + (int64_t)e.below((uint64_t)(rep_spacing - rep_len));
        for (int i = 0; i < rep_len; i++) {
          uint8_t b = elem[(size_t)i];
          if (e.unit() < rep_div) b = (uint8_t)((b + 1 + (e.next() >> 62) % 3) & 3);
          genome[(size_t)(at + i)] = b;
        }
      }
    }
  }
  // ins:del:sub = 0.1188:0.0183:0.0129 (J/utils/RandomSequenceGenerator.java:93-96), scaled to error_rate
  const double p_ins = error_rate * (0.1188 / 0.15), p_del = error_rate * (0.0183 / 0.15), p_sub = error_rate * (0.0129 / 0.15);
  static const char ALPHA[4] = {'A', 'C', 'G', 'T'};
  const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(mhap::usable_host_threads(32), n));
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) {
    th.emplace_back([&, t]() {
      std::vector<uint8_t> tmp((size_t)len);
      for (int64_t q = t; shard + q * nshards < n; q += nthreads) {
        const int64_t r = shard + q * nshards;
        Xoshiro256ss g(SplitMix64{seed ^ (0x9e3779b97f4a7c15ULL * (uint64_t)(r + 1))}.next());
        int64_t gp = (int64_t)g.below((uint64_t)G);
        const bool rev = (g.next() >> 63) != 0;
        int w = 0;
        while (w < len) {
          const double u = g.unit();
          if (u < p_ins) { tmp[(size_t)w++] = (uint8_t)(g.next() >> 62); continue; }
          const uint8_t b = genome[(size_t)gp];
          gp++; if (gp == G) gp = 0;
          if (u < p_ins + p_del) continue;
          if (u < p_ins + p_del + p_sub) { tmp[(size_t)w++] = (uint8_t)((b + 1 + (g.next() >> 62) % 3) & 3); continue; }
          tmp[(size_t)w++] = b;
        }
        char* dst = bases + q * (int64_t)len;
        if (!rev) for (int i = 0; i < len; i++) dst[i] = ALPHA[tmp[(size_t)i]];
        else for (int i = 0; i < len; i++) dst[i] = ALPHA[3 - tmp[(size_t)(len - 1 - i)]];
      }
    });
  }
  for (auto& x : th) x.join();
  return MHAP_OK;
}

// Reads of given lengths drawn from a SUPPLIED circular genome (codes 0..3, one per byte), same error model and per-read generators as
// above.  The caller builds the genome — workloads.ecoli_like_genome plants rRNA-operon and IS-element copies in 4.6 Mbp — and the
// length mix; read r goes to bases[offsets[r] .. offsets[r] + lengths[r]).
int mhap_synth_reads_genome(uint64_t seed, const uint8_t* genome, int64_t G, int64_t n, const int32_t* lengths, const int64_t* offsets,
                            double error_rate, char* bases) {
  if (!genome || G < 1 || n < 0 || !lengths || !offsets || !bases || error_rate < 0.0 || error_rate >= 1.0) return MHAP_E_INVALID;
  const double p_ins = error_rate * (0.1188 / 0.15), p_del = error_rate * (0.0183 / 0.15), p_sub = error_rate * (0.0129 / 0.15);
  static const char ALPHA[4] = {'A', 'C', 'G', 'T'};
  const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(mhap::usable_host_threads(32), n));
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) {
    th.emplace_back([&, t]() {
      std::vector<uint8_t> tmp;
      for (int64_t r = t; r < n; r += nthreads) {
        const int len = lengths[r];
        if (len <= 0) continue;
        tmp.resize((size_t)len);
        Xoshiro256ss g(SplitMix64{seed ^ (0x9e3779b97f4a7c15ULL * (uint64_t)(r + 1))}.next());
        int64_t gp = (int64_t)g.below((uint64_t)G);
        const bool rev = (g.next() >> 63) != 0;
        int w = 0;
        while (w < len) {
          const double u = g.unit();
          if (u < p_ins) { tmp[(size_t)w++] = (uint8_t)(g.next() >> 62); continue; }
          const uint8_t b = genome[(size_t)gp] & 3;
          gp++; if (gp == G) gp = 0;
          if (u < p_ins + p_del) continue;
          if (u < p_ins + p_del + p_sub) { tmp[(size_t)w++] = (uint8_t)((b + 1 + (g.next() >> 62) % 3) & 3); continue; }
          tmp[(size_t)w++] = b;
        }
        char* dst = bases + offsets[r];
        if (!rev) for (int i = 0; i < len; i++) dst[i] = ALPHA[tmp[(size_t)i]];
        else for (int i = 0; i < len; i++) dst[i] = ALPHA[3 - tmp[(size_t)(len - 1 - i)]];
      }
    });
  }
  for (auto& x : th) x.join();
  return MHAP_OK;
}

// new FrequencyCounts(reader, filterCutoff, offset, removeUnique, noTf, numThreads, range, doRC) (J/sketch/FrequencyCounts.java:63-229)
int mhap_set_filter_file(mhap_handle* h, const char* path, double filter_cutoff, double offset, int32_t remove_unique, int32_t no_tf,
                         double range, int32_t do_rc, char* kmer_sizes, size_t kmer_sizes_cap) {
  if (!h || !path) return MHAP_E_INVALID;
  if (kmer_sizes && kmer_sizes_cap) kmer_sizes[0] = 0;
  FILE* f = fopen(path, "r");
  if (!f) return MHAP_E_IO;
  std::vector<int64_t> hs; std::vector<double> fr;
  std::vector<int64_t> all;                       // every line's k-mer goes into the whitelist (:192), with or without a fraction
  char* line = nullptr; size_t cap = 0; ssize_t len; bool first = true;
  long long size_bloom = 1;
  std::vector<int> sizes;
  int rc = MHAP_OK;
  while ((len = getline(&line, &cap, f)) > 0) {
    if (first) {   // "sizeBloom sizeRepeat" (:102-121): the first number sizes the Bloom filter
      first = false;
      long long a = 0, b = 0;
      if (sscanf(line, "%lld %lld", &a, &b) < 2 || a < 0 || b < 0) { rc = MHAP_E_INVALID; break; }   // :139-142
      size_bloom = a == 0 ? 1 : a;
      continue;
    }
    // String.split("\\s+", 3): k-mer, optional fraction, rest ignored (:158-173)
    char kmer[4096], ftok[256];
    const int got = sscanf(line, "%4095s %255s", kmer, ftok);
    if (got < 1) continue;
    int64_t hv; const int kl = (int)strlen(kmer);
    if (mhap_hash_kmer(kmer, kl, do_rc, &hv) != MHAP_OK) continue;
    if (std::find(sizes.begin(), sizes.end(), kl) == sizes.end()) sizes.push_back(kl);   // kmerSizes.add comes before the parse (:163-166)
    double frac = 0.0;
    if (got >= 2) {
      // Double.parseDouble throws on a malformed number and the exception handler drops the WHOLE line — it never reaches the
      // whitelist either (:173,190-198)
      char* end = nullptr;
      frac = strtod(ftok, &end);
      if (end == ftok || *end != 0) continue;
    }
    if (remove_unique > 0) all.push_back(hv);
    if (got >= 2) { hs.push_back(hv); fr.push_back(frac); }
  }
  free(line); fclose(f);
  if (rc != MHAP_OK) return rc;
  // a file without a single (k-mer, fraction) line still installs the filter, with an empty table (see mhap_set_filter)
  static const int64_t none_h = 0; static const double none_f = 0.0;
  rc = mhap_set_filter(h, hs.empty() ? &none_h : hs.data(), hs.empty() ? &none_f : fr.data(), (int64_t)hs.size(), filter_cutoff, offset, range, no_tf);
  if (rc == MHAP_OK && remove_unique > 0) rc = mhap_set_filter_whitelist(h, all.data(), (int64_t)all.size(), size_bloom, remove_unique);
  if (kmer_sizes && kmer_sizes_cap) {
    std::sort(sizes.begin(), sizes.end());
    std::string ks; for (int v : sizes) ks += (ks.empty() ? "" : ", ") + std::to_string(v);
    snprintf(kmer_sizes, kmer_sizes_cap, "%s", ks.c_str());
  }
  return rc;
}

}  // extern "C"
