// mhap_cli.cpp — `mhap-hip`: native host driver over libmhaphip.so that keeps MHAP's command line, stdout
// record format, stderr timing lines and `.dat` sketch files (J/main/MhapMain.java:93-590,
// J/utils/ParseOptions.java, J/impl/SequenceSketchStreamer.java:278-395, J/impl/SequenceSketch.java:61-148).
// The reference host is Java; this image has no JVM, so the host side above the C ABI is C++ (see INTEGRATION.md
// for the JNI stub that lets the stock Java host call the same library).
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mhap_hip.h"

namespace {

struct Opt { std::string help; bool is_flag; std::string value; bool set = false; };
struct Options {
  std::vector<std::string> order;
  std::map<std::string, Opt> m;
  void add(const std::string& name, const std::string& help, const std::string& def, bool flag = false) { order.push_back(name); m[name] = Opt{help, flag, def, false}; }
  std::string s(const std::string& n) const { return m.at(n).value; }
  int i(const std::string& n) const { return atoi(m.at(n).value.c_str()); }
  double d(const std::string& n) const { return atof(m.at(n).value.c_str()); }
  bool b(const std::string& n) const { return m.at(n).value == "true"; }
  bool isset(const std::string& n) const { return m.at(n).set; }
  void setdef(const std::string& n, const std::string& v) { if (!m[n].set) m[n].value = v; }
  std::string helpText() const {
    std::string t = "MHAP: MinHash Alignment Protocol. A tool for finding overlaps of long-read sequences (such as PacBio or Nanopore) in bioinformatics.\n"
                    "\tmhap-hip: MI355X (gfx950) implementation of the MHAP overlap path.\n"
                    "\tUsage 1 (direct execution): mhap-hip -s<fasta/dat from/self file> [-q<fasta/dat to file>] [-f<kmer filter list, must be sorted>]\n"
                    "\tUsage 2 (generate precomputed binaries): mhap-hip -p<directory of fasta files> -q <output directory> [-f<kmer filter list, must be sorted>]\n";
    for (auto& n : order) { const Opt& o = m.at(n); t += "\t" + n + ", default = " + (o.is_flag ? "false" : o.value) + "\n\t\t" + o.help + "\n"; }
    return t;
  }
  std::string dump() const { std::string t; for (auto& n : order) t += n + " = " + m.at(n).value + "\n"; return t; }
  // ParseOptions.process (J/utils/ParseOptions.java:209-236,327-368): flags are presence flags, others consume the next arg
  bool parse(int argc, char** argv) {
    for (int a = 1; a < argc; a++) {
      std::string k = argv[a];
      if (k == "-h" || k == "--help") { printf("%s", helpText().c_str()); return false; }
      auto it = m.find(k);
      if (it == m.end()) { printf("Unknown parameter %s, please see help menu below.\n%s", k.c_str(), helpText().c_str()); return false; }
      it->second.set = true;
      if (it->second.is_flag) it->second.value = "true";
      else { if (a + 1 >= argc) { printf("Parameter %s requires a value.\n", k.c_str()); return false; } it->second.value = argv[++a]; }
    }
    return true;
  }
};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
bool is_dir(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
bool ends_with(const std::string& s, const std::string& e) { return s.size() >= e.size() && s.compare(s.size() - e.size(), e.size(), e) == 0; }

std::vector<std::string> list_files(const std::string& path) {   // non-hidden entries, sorted (MhapMain.java:403-420,489-512)
  std::vector<std::string> out;
  if (!is_dir(path)) { out.push_back(path); return out; }
  DIR* d = opendir(path.c_str());
  if (!d) return out;
  while (dirent* e = readdir(d)) { std::string n = e->d_name; if (n.empty() || n[0] == '.') continue; out.push_back(path + "/" + n); }
  closedir(d);
  std::sort(out.begin(), out.end());
  return out;
}

[[noreturn]] void die(const std::string& m) { fprintf(stderr, "Exception in mhap-hip: %s\n", m.c_str()); exit(1); }
void chk(mhap_handle* h, int rc) { if (rc != MHAP_OK) die(std::string(mhap_last_error(h)) + " (code " + std::to_string(rc) + ")"); }

// One GPU (a handle) or several (a group: one rank per device, reads dealt round-robin, the exchange inside the library).
struct Engine {
  mhap_handle* h = nullptr;
  mhap_group* g = nullptr;
  int n = 1;
  mhap_handle* rank(int r) const { return g ? mhap_group_rank(g, r) : h; }
  void check(int rc) const {
    if (rc == MHAP_OK) return;
    die(std::string(g ? mhap_group_last_error(g) : mhap_last_error(h)) + " (code " + std::to_string(rc) + ")");
  }
  void add_reads(const mhap_fasta& fa) const {
    if (fa.n <= 0) return;
    check(g ? mhap_group_add_reads(g, fa.bases, fa.offsets, fa.lengths, fa.ids, fa.n) : mhap_index_add_reads(h, fa.bases, fa.offsets, fa.lengths, fa.ids, fa.n));
  }
  void add_scan(const mhap_fasta_scan* sc) const { check(g ? mhap_group_add_scan(g, sc) : mhap_index_add_scan(h, sc)); }
  void find_self(mhap_record_sink sink, void* user) const { check(g ? mhap_group_find_matches_self(g, sink, user) : mhap_find_matches_self(h, 0, -1, sink, user)); }
  void find_reads(const mhap_fasta& fa, mhap_record_sink sink, void* user) const {
    if (fa.n <= 0) return;
    check(g ? mhap_group_find_matches_reads(g, fa.bases, fa.offsets, fa.lengths, fa.ids, fa.n, sink, user)
            : mhap_find_matches_reads(h, fa.bases, fa.offsets, fa.lengths, fa.ids, fa.n, sink, user));
  }
  mhap_stats stats() const { mhap_stats st; check(g ? mhap_group_get_stats(g, &st) : mhap_get_stats(h, &st)); return st; }
  void clear() const { check(g ? mhap_group_clear(g) : mhap_index_clear(h)); }
  void destroy() { if (g) mhap_group_destroy(g); else if (h) mhap_destroy(h); g = nullptr; h = nullptr; }
};

// ---- big-endian `.dat` primitives (DataOutputStream / ByteBuffer) ----
void put8(std::string& b, uint8_t v) { b.push_back((char)v); }
void put32(std::string& b, int32_t v) { for (int s = 24; s >= 0; s -= 8) b.push_back((char)((uint32_t)v >> s)); }
void put64(std::string& b, int64_t v) { for (int s = 56; s >= 0; s -= 8) b.push_back((char)((uint64_t)v >> s)); }
void putUTF(std::string& b, const std::string& s) { b.push_back((char)(s.size() >> 8)); b.push_back((char)s.size()); b += s; }   // ASCII headers only
struct Rd {
  const uint8_t* p; size_t n, o = 0; bool ok = true;
  uint8_t u8() { if (o + 1 > n) { ok = false; return 0; } return p[o++]; }
  int32_t i32() { if (o + 4 > n) { ok = false; return 0; } uint32_t v = 0; for (int i = 0; i < 4; i++) v = (v << 8) | p[o++]; return (int32_t)v; }
  int64_t i64() { if (o + 8 > n) { ok = false; return 0; } uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[o++]; return (int64_t)v; }
  std::string utf() { if (o + 2 > n) { ok = false; return ""; } size_t l = ((size_t)p[o] << 8) | p[o + 1]; o += 2; if (o + l > n) { ok = false; return ""; } std::string s((const char*)p + o, l); o += l; return s; }
};

struct Headers { std::map<int64_t, std::string> byid; bool full = false; };
Headers g_headers;

std::string header_of(int64_t id) {   // SequenceId.getHeader (J/impl/SequenceId.java:102-108)
  if (g_headers.full) { auto it = g_headers.byid.find(id); if (it != g_headers.byid.end()) return it->second; }
  return std::to_string(id);
}

// --store-full-id: header = first token after '>' split on [\s,]+ (FastaData.java:155-156)
// --store-full-id: names of the records just read (mhap_fasta.headers), keyed by the ids they were given
void collect_headers(const mhap_fasta& fa) {
  const char* p = fa.headers;
  for (int64_t i = 0; i < fa.n && p && p < fa.headers + fa.headers_bytes; i++) {
    g_headers.byid[fa.ids[i]] = std::string(p);
    p += strlen(p) + 1;
  }
}

// --store-full-id: the names of a scanned file, keyed by the ids its records were given
void collect_headers(mhap_fasta_scan* sc) {
  const int64_t n = mhap_fasta_scan_reads(sc);
  std::vector<int64_t> ids((size_t)std::max<int64_t>(n, 1));
  const char* p = nullptr; int64_t bytes = 0;
  if (mhap_fasta_scan_info(sc, ids.data(), nullptr, &p, &bytes) != MHAP_OK) return;
  const char* e = p + bytes;
  for (int64_t i = 0; i < n && p && p < e; i++) { g_headers.byid[ids[(size_t)i]] = std::string(p); p += strlen(p) + 1; }
}

// the file the index is built from is mapped and scanned while the HIP runtime comes up
struct Preload { std::string path; mhap_fasta_scan* scan = nullptr; } g_preload;
struct Sink { FILE* out; std::string buf; int64_t n = 0; };
int sink_cb(const mhap_record* r, int64_t n, void* user) {
  Sink* s = (Sink*)user;
  char line[512];
  for (int64_t i = 0; i < n; i++) {
    if (!g_headers.full) { int len = mhap_format_record(&r[i], line, sizeof line); s->buf.append(line, (size_t)len); }
    else {
      // same formatter, headers swapped in for the two id columns
      int len = mhap_format_record(&r[i], line, sizeof line);
      const char* p = line; int sp = 0; while (*p && sp < 2) { if (*p == ' ') sp++; p++; }
      s->buf += header_of(r[i].from_id) + " " + header_of(r[i].to_id) + " "; s->buf.append(p, (size_t)(line + len - p));
    }
    s->buf.push_back('\n');
    if (s->buf.size() > (8u << 20)) { fwrite(s->buf.data(), 1, s->buf.size(), s->out); s->buf.clear(); }   // 8 MB buffer (Utils.BUFFER_BYTE_SIZE)
  }
  s->n += n;
  return 0;
}
void sink_flush(Sink& s) { if (!s.buf.empty()) fwrite(s.buf.data(), 1, s.buf.size(), s.out); s.buf.clear(); fflush(s.out); }

// FrequencyCounts file -> (hash, fraction) arrays (J/sketch/FrequencyCounts.java:63-229)
void load_filter(const Engine& E, const Options& o) {
  const double rw = o.d("--repeat-weight");
  const double offset = (rw >= 0.0 && rw < 1.0) ? rw : 0.0;   // MhapMain.java:346-350
  char sizes[256];
  for (int r = 0; r < E.n; r++) {   // every rank applies the filter to the reads it sketches
    mhap_handle* h = E.rank(r);
    const int rc = mhap_set_filter_file(h, o.s("-f").c_str(), o.d("--filter-threshold"), offset, o.i("--supress-noise"), o.b("--no-tf") ? 1 : 0,
                                        o.d("--repeat-idf-scale"), o.b("--no-rc") ? 0 : 1, sizes, sizeof sizes);
    if (rc == MHAP_E_IO) die("Could not parse k-mer filter file.");
    if (rc == MHAP_E_INVALID && !*mhap_last_error(h)) die("K-mer filter file first line must contain estimated number of k-mers in the file (long).");
    chk(h, rc);
  }
  fprintf(stderr, "Read in k-mer filter for sizes: [%s]\n", sizes);
}

// `.dat` reader (SequenceSketchStreamer.readFromBinary :278-320 + SequenceSketch.fromByteStream :61-96)
struct DatEntries { std::vector<int64_t> ids; std::vector<uint8_t> fwd; std::vector<int32_t> seqlen, mh, ord, osz, olen; std::vector<std::string> hdr; };
void read_dat(const std::string& path, int64_t offset, int H, int S, bool fwd_only, DatEntries& d) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) die("cannot open " + path);
  std::vector<uint8_t> data; uint8_t buf[1 << 16]; size_t got;
  while ((got = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + got);
  fclose(f);
  Rd r{data.data(), data.size()};
  while (r.o < r.n) {
    uint8_t isFwd = r.u8(); int32_t size = r.i32();
    if (!r.ok || size < 0 || r.o + (size_t)size > r.n) break;
    Rd p{data.data() + r.o, (size_t)size}; r.o += (size_t)size;
    if (fwd_only && isFwd != 1) continue;
    uint8_t pf = p.u8(); int64_t id = p.i64() + offset; std::string hdr = p.utf(); int32_t sl = p.i32();
    int32_t hn = p.i32();
    if (!p.ok || hn != H) die("Number of MinHashes of the sequence does not match current settings.");
    d.ids.push_back(id); d.fwd.push_back(pf ? 1 : 0); d.seqlen.push_back(sl); d.hdr.push_back(hdr);
    for (int i = 0; i < H; i++) d.mh.push_back(p.i32());
    int32_t n = p.i32(); (void)p.i32(); int32_t sz = p.i32();
    if (!p.ok || sz < 0 || sz > S) die("ordered sketch in .dat larger than --ordered-sketch-size");
    d.olen.push_back(n); d.osz.push_back(sz);
    size_t base = d.ord.size(); d.ord.resize(base + (size_t)S * 2, 0);
    for (int i = 0; i < sz; i++) { d.ord[base + 2 * i] = p.i32(); d.ord[base + 2 * i + 1] = p.i32(); }
    if (!p.ok) die("Unexpected data read error.");
  }
}

// `.dat` writer (SequenceSketchStreamer.writeToBinary :322-395; SequenceSketch.getAsByteArray :123-148)
void write_dat(mhap_handle* h, const std::string& path, int H, int S, int k2) {
  int64_t n = 0; chk(h, mhap_index_size(h, &n));
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) die("cannot write " + path);
  const int64_t CH = 4096;
  std::vector<int64_t> ids(CH); std::vector<uint8_t> fwd(CH), st(CH); std::vector<int32_t> sl(CH), mh(CH * H), od(CH * (size_t)S * 2), osz(CH), olen(CH);
  for (int64_t e0 = 0; e0 < n; e0 += CH) {
    const int64_t c = std::min(CH, n - e0);
    chk(h, mhap_index_export(h, e0, c, ids.data(), fwd.data(), sl.data(), mh.data(), od.data(), osz.data(), olen.data(), st.data()));
    std::string out;
    for (int64_t e = 0; e < c; e++) {
      if (st[e] != 0) continue;
      std::string pay;
      put8(pay, fwd[e]); put64(pay, ids[e]); putUTF(pay, header_of(ids[e])); put32(pay, sl[e]);
      put32(pay, H); for (int i = 0; i < H; i++) put32(pay, mh[(size_t)e * H + i]);
      put32(pay, olen[e]); put32(pay, k2); put32(pay, osz[e]);
      for (int i = 0; i < osz[e]; i++) { put32(pay, od[((size_t)e * S + i) * 2]); put32(pay, od[((size_t)e * S + i) * 2 + 1]); }
      put8(out, fwd[e]); put32(out, (int32_t)pay.size()); out += pay;
    }
    fwrite(out.data(), 1, out.size(), f);
  }
  fclose(f);
}

int64_t add_file_to_index(const Engine& E, const std::string& path, int64_t id_offset, const Options& o, int64_t* strands) {
  const int H = o.i("--num-hashes"), S = o.i("--ordered-sketch-size");
  if (ends_with(path, ".dat")) {   // MhapMain.java:563-564
    if (E.g) die("--gpus N takes FASTA input (precomputed .dat sketches are searched on one GPU)");
    mhap_handle* h = E.h;
    DatEntries d; read_dat(path, id_offset, H, S, false, d);
    if (!d.ids.empty()) chk(h, mhap_index_add_sketches(h, d.ids.data(), d.fwd.data(), d.seqlen.data(), d.mh.data(), d.ord.data(), d.osz.data(), d.olen.data(), (int64_t)d.ids.size()));
    if (g_headers.full) for (size_t i = 0; i < d.ids.size(); i++) g_headers.byid[d.ids[i]] = d.hdr[i];
    *strands = (int64_t)d.ids.size();
    return (int64_t)d.ids.size() / 2;
  }
  // FASTA: the file is mapped and scanned (record boundaries, lengths, on all host threads), then fed to the index in groups — host
  // threads pack the next group from the text while the GPU sketches the previous one (mhap_index_add_scan)
  mhap_fasta_scan* sc = nullptr; char err[512];
  const double t_read = now();
  if (g_preload.scan && g_preload.path == path && id_offset == 0) { sc = g_preload.scan; g_preload.scan = nullptr; }   // scanned while the runtime came up
  else if (mhap_fasta_scan_open(path.c_str(), id_offset, &sc, err, sizeof err) != MHAP_OK) die(err);
  if (g_headers.full) collect_headers(sc);
  const double t_add = now();
  E.add_scan(sc);
  if (getenv("MHAP_HOST_PROF")) fprintf(stderr, "[cli] fasta scan %.3f s, add (pack + upload + sketch + index, pipelined) %.3f s\n", t_add - t_read, now() - t_add);
  // (the mapping is released at exit: unmapping gigabytes holds the address-space lock the search's allocations need)
  const mhap_stats st = E.stats();
  *strands = st.strands_indexed;
  // seqNumberProcessed += seqStreamer.getNumberProcessed()/2 (MhapMain.java:462): the streamer counts the sketches it
  // produced (SequenceSketchStreamer.java:145,155,268-271), so reads below --min-olap-length and reads without a valid
  // n-gram do not advance the id offset of the -q files
  return st.strands_indexed / 2;
}

}  // namespace

int main(int argc, char** argv) {
  Options o;
  // --num-threads defaults to the CPUs the process may use: the hardware threads, capped by the container's CPU quota (cgroup v2)
  unsigned hc = std::max(1u, std::thread::hardware_concurrency());
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64] = {0}; long long per = 0;
    if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) { const long long c = (atoll(q) + per - 1) / per; if (c > 0 && (unsigned long long)c < hc) hc = (unsigned)c; }
    fclose(f);
  }
  o.add("-s", "Usage 1 only. The FASTA or binary dat file (see Usage 2) of reads that will be stored in a box, and that all subsequent reads will be compared to.", "");
  o.add("-q", "Usage 1: The FASTA file of reads, or a directory of files, that will be compared to the set of reads in the box (see -s). Usage 2: The output directory for the binary formatted dat files.", "");
  o.add("-p", "Usage 2 only. The directory containing FASTA files that should be converted to binary format for storage.", "");
  o.add("-f", "k-mer filter file used for filtering out highly repetative k-mers. Must be sorted in descending order of frequency (second column).", "");
  o.add("-k", "[int], k-mer size used for MinHashing. The k-mer size for second stage filter is seperate, and can also be modified.", "16");
  o.add("--num-hashes", "[int], Number of min-mers to be used in MinHashing.", "512");
  o.add("--threshold", "[double], The threshold cutoff for the second stage sort-merge filter.", "0.78");
  o.add("--filter-threshold", "[double], The cutoff at which the k-mer in the k-mer filter file is considered repetitive.", "1.0E-5");
  o.add("--max-shift", "[double], Region size to the left and right of the estimated overlap where k-mer matches are still considered valid. Second stage filter only.", "0.2");
  o.add("--num-min-matches", "[int], Minimum # min-mer that must be shared before computing second stage filter.", "3");
  o.add("--num-threads", "[int], Number of host threads (FASTA packing / record conversion); the compute runs on the GPU.", std::to_string(hc));
  o.add("--repeat-weight", "[double] Repeat suppression strength for tf-idf weighing. <0.0 do unweighted MinHash (version 1.0), >=1.0 do only the tf weighing.", "0.9");
  o.add("--repeat-idf-scale", "[double] The upper range of the idf (from tf-idf) scale. The full scale will be [1,X].", "3.0");
  o.add("--ordered-kmer-size", "[int] The size of k-mers used in the ordered second stage filter.", "12");
  o.add("--ordered-sketch-size", "[int] The sketch size for second stage filter.", "1536");
  o.add("--min-store-length", "[int], The minimum length of the read that is stored in the box.", "0");
  o.add("--min-olap-length", "[int], The minimum length of the read that used for overlapping.", "116");
  o.add("--no-self", "Do not compute the overlaps between sequences inside a box.", "false", true);
  o.add("--store-full-id", "Store full IDs as seen in FASTA files, rather than storing just the sequence position in the file.", "false", true);
  o.add("--supress-noise", "[int] 0) Does nothing, 1) completely removes any k-mers not specified in the filter file, 2) supresses k-mers not specified in the filter file, similar to repeats.", "0");
  o.add("--no-tf", "Do not perform the tf weighing, in the tf-idf weighing.", "false", true);
  o.add("--no-rc", "Do not store or do comparison of the reverse compliment strings (in this MHAP version it only changes how -f k-mers are hashed).", "false", true);
  o.add("--settings", "Set all unset parameters for the default settings. 0) None, 1) Default, 2) Fast, 3) Sensitive.", "0");
  o.add("--device", "[int] HIP device ordinal (the first one with --gpus N).", "0");
  o.add("--gpus", "[int] Number of GPUs: the reads are dealt round-robin over devices --device .. --device+N-1, every GPU sketches and indexes its share, and a search gathers the forward query sketches of all GPUs (over xGMI) against every share.", "1");
  o.add("--devices", "Comma-separated HIP device ordinals, one per rank (overrides --device/--gpus; an ordinal may repeat).", "");
  if (!o.parse(argc, argv)) return 0;

  auto bad = [&](const char* m) { printf("%s\n", m); exit(1); };
  const int settings = o.i("--settings");
  if (settings < 0 || settings > 3) { printf("Please enter valid --settings flag. See options below:\n%s", o.helpText().c_str()); return 1; }
  if (settings == 1) { o.setdef("-k", "16"); o.setdef("--num-min-matches", "3"); o.setdef("--num-hashes", "512"); o.setdef("--threshold", "0.78"); o.setdef("--ordered-sketch-size", "1536"); o.setdef("--ordered-kmer-size", "12"); }
  if (settings == 2) { o.setdef("-k", "16"); o.setdef("--num-min-matches", "3"); o.setdef("--num-hashes", "256"); o.setdef("--threshold", "0.80"); o.setdef("--ordered-sketch-size", "1000"); o.setdef("--ordered-kmer-size", "14"); }
  if (settings == 3) { o.setdef("-k", "16"); o.setdef("--num-min-matches", "2"); o.setdef("--num-hashes", "768"); o.setdef("--threshold", "0.73"); o.setdef("--ordered-sketch-size", "1536"); o.setdef("--ordered-kmer-size", "12"); }
  if (o.s("-s").empty() && o.s("-p").empty()) { printf("Please set the -s or the -p options. See options below:\n%s", o.helpText().c_str()); return 1; }
  if (!o.s("-p").empty() && o.s("-q").empty()) { printf("Please set the -q option. See options below:\n%s", o.helpText().c_str()); return 1; }
  for (const char* k : {"-p", "-s", "-q", "-f"}) if (!o.s(k).empty() && !exists(o.s(k))) { printf("Could not find requested file/folder: %s\n", o.s(k).c_str()); return 1; }
  if (o.i("--num-threads") <= 0) bad("Number of threads must be positive.");
  if (o.i("-k") <= 0) bad("k-mer size must be positive.");
  if (o.i("--num-min-matches") <= 0) bad("Minimum number of matches must be positive.");
  if (o.i("--min-store-length") < 0) bad("The minimum read length stored must be >=0.");
  if (o.d("--repeat-idf-scale") < 1.0) bad("The minimum repeat idf scale must be >=1.0.");
  if (o.d("--max-shift") < -1.0) bad("The minimum shift must be greater than -1.");
  if (o.d("--threshold") < 0.0 || o.d("--threshold") > 1.0) bad("The second stage filter threshold must be 0<=threshold<=1.0.");
  if (o.i("--supress-noise") < 0 || o.i("--supress-noise") > 2) bad("The --supress-noise parameter must be in [0,2].");
  g_headers.full = o.b("--store-full-id");
  setenv("MHAP_HOST_THREADS", o.s("--num-threads").c_str(), 0);
  fprintf(stderr, "Running with these settings:\n%s", o.dump().c_str());

  mhap_params P; mhap_default_params(&P);
  P.kmer_size = o.i("-k"); P.num_hashes = o.i("--num-hashes"); P.ordered_kmer_size = o.i("--ordered-kmer-size");
  P.ordered_sketch_size = o.i("--ordered-sketch-size"); P.num_min_matches = o.i("--num-min-matches");
  P.min_store_length = o.i("--min-store-length"); P.min_olap_length = o.i("--min-olap-length"); P.device = o.i("--device");
  P.threshold = o.d("--threshold"); P.max_shift = o.d("--max-shift"); P.repeat_weight = o.d("--repeat-weight");
  // ranks: --devices a,b,c  |  --gpus N from --device on  |  one handle
  std::vector<int32_t> devs;
  if (!o.s("--devices").empty()) {
    const std::string dl = o.s("--devices");
    for (size_t i = 0; i < dl.size();) { size_t j = dl.find(',', i); if (j == std::string::npos) j = dl.size(); if (j > i) devs.push_back(atoi(dl.substr(i, j - i).c_str())); i = j + 1; }
  } else {
    if (o.i("--gpus") < 1) bad("The number of GPUs must be positive.");
    for (int r = 0; r < o.i("--gpus"); r++) devs.push_back(o.i("--device") + r);
  }
  if (devs.empty()) bad("No device given.");
  const bool precompute = !o.s("-p").empty();
  if (precompute && devs.size() > 1) { fprintf(stderr, "Usage 2 (-p) writes its .dat files from one GPU: using device %d only.\n", devs[0]); devs.resize(1); }
  Engine E; E.n = (int)devs.size();
  char err[512] = {0};
  const double t_create = now();
  // mhap_create is mostly the HIP runtime coming up (~0.25 s): it runs on its own thread while this one reads and parses the
  // FASTA file the index is built from
  int rc_create = MHAP_OK;
  std::thread creator([&]() {
    if (devs.size() == 1) { P.device = devs[0]; rc_create = mhap_create(&P, &E.h, err, sizeof err); }
    else rc_create = mhap_group_create(&P, devs.data(), (int32_t)devs.size(), &E.g, err, sizeof err);
  });
  if (!precompute && !is_dir(o.s("-s")) && !ends_with(o.s("-s"), ".dat")) {
    char perr[512] = {0};
    if (mhap_fasta_scan_open(o.s("-s").c_str(), 0, &g_preload.scan, perr, sizeof perr) == MHAP_OK) g_preload.path = o.s("-s");
    else g_preload.scan = nullptr;
    // (a failure is reported by the regular scan below)
  }
  creator.join();
  if (rc_create != MHAP_OK) die(err);
  if (getenv("MHAP_HOST_PROF")) fprintf(stderr, "[cli] mhap_create + first FASTA scan %.3f s\n", now() - t_create);
  if (E.g) fprintf(stderr, "Using %d GPU ranks (reads dealt round-robin; every rank indexes its share).\n", E.n);

  const double t_total = now();
  if (!o.s("-f").empty()) {
    const double t = now();
    fprintf(stderr, "Reading in filter file %s.\n", o.s("-f").c_str());
    load_filter(E, o);
    fprintf(stderr, "Time (s) to read filter file: %g\n", now() - t);
  }

  if (precompute) {   // Usage 2: precompute `.dat` (MhapMain.java:384-451)
    mhap_handle* h = E.h;
    fprintf(stderr, "Processing FASTA files for binary compression...\n");
    if (!is_dir(o.s("-q"))) die("Target directory doesn't exit.");
    for (const std::string& pf : list_files(o.s("-p"))) {
      const double t = now();
      chk(h, mhap_index_clear(h));
      int64_t strands = 0;
      add_file_to_index(E, pf, 0, o, &strands);
      std::string name = pf.substr(pf.find_last_of('/') + 1);
      size_t dot = name.find_last_of('.'); if (dot != std::string::npos && dot > 0) name = name.substr(0, dot);
      const std::string out = o.s("-q") + "/" + name + ".dat";
      write_dat(h, out, P.num_hashes, P.ordered_sketch_size, P.ordered_kmer_size);
      fprintf(stderr, "Processed %lld sequences (fwd and rev).\n", (long long)strands);
      fprintf(stderr, "Read, hashed, and stored file %s to %s.\n", pf.c_str(), out.c_str());
      fprintf(stderr, "Time (s): %g\n", now() - t);
    }
    fprintf(stderr, "Total time (s): %g\n", now() - t_total);
    E.destroy();
    return 0;
  }

  fprintf(stderr, "Processing files for storage in reverse index...\n");
  const double t_proc = now();
  int64_t strands = 0;
  int64_t seq_processed = add_file_to_index(E, o.s("-s"), 0, o, &strands);
  fprintf(stderr, "Stored %lld sequences in the index.\n", (long long)strands);
  fprintf(stderr, "Processed %lld unique sequences (fwd and rev).\n", (long long)strands);
  fprintf(stderr, "Time (s) to read and hash from file: %g\n", now() - t_proc);

  Sink sink{stdout};
  const double t_score = now();
  if (o.s("-q").empty()) {
    const double t = now();
    E.find_self(sink_cb, &sink);
    sink_flush(sink);
    fprintf(stderr, "Time (s) to score and output to self: %g\n", now() - t);
  } else {
    double t = now();
    if (!o.b("--no-self")) {
      E.find_self(sink_cb, &sink);
      sink_flush(sink);
      fprintf(stderr, "Time (s) to score and output to self: %g\n", now() - t);
    }
    for (const std::string& cf : list_files(o.s("-q"))) {
      t = now();
      fprintf(stderr, "Opened fasta file %s.\n", cf.c_str());
      int64_t nq = 0;
      if (ends_with(cf, ".dat")) {   // precomputed query sketches: forward entries only (SequenceSketchStreamer.java:291-303)
        if (E.g) die("--gpus N takes FASTA input (precomputed .dat sketches are searched on one GPU)");
        mhap_handle* h = E.h;
        DatEntries d;
        read_dat(cf, seq_processed, P.num_hashes, P.ordered_sketch_size, true, d);
        if (g_headers.full) for (size_t i = 0; i < d.ids.size(); i++) g_headers.byid[d.ids[i]] = d.hdr[i];
        if (!d.ids.empty())
          chk(h, mhap_find_matches_sketches(h, d.ids.data(), d.seqlen.data(), d.mh.data(), d.ord.data(), d.osz.data(), d.olen.data(),
                                            (int64_t)d.ids.size(), sink_cb, &sink));
        sink_flush(sink);
        seq_processed += (int64_t)d.ids.size();
        fprintf(stderr, "Processed %lld to sequences.\n", (long long)d.ids.size());
        fprintf(stderr, "Time (s) to score, hash to-file, and output: %g\n", now() - t);
        continue;
      }
      mhap_fasta fa;
      if (mhap_fasta_read(cf.c_str(), seq_processed, &fa, err, sizeof err) != MHAP_OK) die(err);   // id offset = reads so far (MhapMain.java:527)
      if (g_headers.full) collect_headers(fa);
      const mhap_stats s0 = E.stats();
      E.find_reads(fa, sink_cb, &sink);
      const mhap_stats s1 = E.stats();
      nq = s1.queries_searched - s0.queries_searched;   // forward sketches produced = getNumberProcessed() (MhapMain.java:537)
      mhap_fasta_free(&fa);
      sink_flush(sink);
      seq_processed += nq;
      fprintf(stderr, "Processed %lld to sequences.\n", (long long)nq);
      fprintf(stderr, "Time (s) to score, hash to-file, and output: %g\n", now() - t);
    }
  }
  sink_flush(sink);
  fprintf(stderr, "Total scoring time (s): %g\n", now() - t_score);
  fprintf(stderr, "Total time (s): %g\n", now() - t_total);
  // outputFinalStat (MhapMain.java:572-590); the inverted-index counters have no brute-force analogue
  const mhap_stats st = E.stats();
  mhap_kernel_times kt; memset(&kt, 0, sizeof kt);
  for (int r = 0; r < E.n; r++) {   // kernel time summed over the ranks (they run concurrently)
    mhap_kernel_times k1; chk(E.rank(r), mhap_get_kernel_times(E.rank(r), &k1));
    for (int i = 0; i < MHAP_K_COUNT; i++) { kt.ms[i] += k1.ms[i]; kt.launches[i] += k1.launches[i]; }
  }
  fprintf(stderr, "MinHash search time (s): %g\n", (kt.ms[MHAP_K_CANDIDATE] + kt.ms[MHAP_K_INDEX_QUERY]) * 1e-3);
  fprintf(stderr, "Total matches found: %lld\n", (long long)st.matches_found);
  fprintf(stderr, "Average number of matches per lookup: %g\n", (double)st.matches_found / (double)std::max<int64_t>(1, st.queries_searched));
  fprintf(stderr, "Average %% of hashed sequences fully compared that are matches: %g\n", (double)st.matches_found / (double)std::max<int64_t>(1, st.candidates_compared) * 100.0);
  fprintf(stderr, "GPU kernel time (ms): hash %.3f, weights %.3f, minhash %.3f, ordered %.3f, index build %.3f, index query %.3f, candidates %.3f, overlap %.3f\n",
          kt.ms[MHAP_K_HASH], kt.ms[MHAP_K_WEIGHT], kt.ms[MHAP_K_MINHASH], kt.ms[MHAP_K_ORDERED], kt.ms[MHAP_K_INDEX_BUILD], kt.ms[MHAP_K_INDEX_QUERY],
          kt.ms[MHAP_K_CANDIDATE], kt.ms[MHAP_K_OVERLAP]);
  // everything is written: leave without tearing down gigabytes of device and host mappings one by one (the kernel does it faster)
  fflush(nullptr);
  _exit(0);
}
