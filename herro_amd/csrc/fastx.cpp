// fastx.cpp — host-side file formats on either side of the hot path, behind the C ABI.
//
//   herro_fastx_read        what get_reads (haec_io.rs:37-75) hands to the path.  The reference reads through needletail's
//                           parse_fastx_file: FASTA or FASTQ by the first byte, gzip by magic, sequences (and FASTQ qualities)
//                           allowed to span lines, '\r' dropped.  get_reads then: records shorter than min_length dropped; the
//                           header split at the first blank or tab into id / description; qualities mandatory ("Qualities should
//                           be present." — a FASTA record is an error); the core / neighbour filter (kept if in either set).
//                           A stream (gzip, a pipe, a small file) is read in chunks of 32 MiB (HERRO_FASTX_CHUNK overrides, for
//                           the tests); a record that reaches the end of the buffered text is parsed again once more text is
//                           behind it: 1.4 GB/s of plain FASTQ on one thread.  A plain regular file of some size is MAPPED and
//                           read by byte ranges on several threads (HERRO_FASTX_THREADS, default min(hardware threads, 16)):
//                           boundaries guessed, then proven against the true record ends from the front, two passes (counts,
//                           then every record straight to its final place) — 9.5 GB/s on 8 cores (tools/fastxrate.py); whatever
//                           cannot be proven (multi-line records cut badly, an error) goes to the sequential reader, so the
//                           result — messages included — is always the sequential one.
//   herro_write_window_features   the `herro features` sink (features.rs:724-764): <dir>/<wid>.features.npy = u8 [2, L', 31]
//                           (ASCII bases, then qualities), <wid>.supported.npy = records {pos: <u2, ins: u1}, <wid>.ids.txt.
//                           The reference writes NPY through the npyz crate (format 1.0, C order, default dtype strings); the
//                           header written here is the one numpy itself writes for these arrays: magic, version 1.0, the dict
//                           {'descr': ..., 'fortran_order': False, 'shape': (...), } padded with spaces to a multiple of 64
//                           bytes, '\n' last.
//
// zlib is used through the system's libz.so.1 (stable C ABI), loaded on first use: the library keeps no link-time dependency
// beyond the HIP runtime and libstdc++.
#include <dlfcn.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <memory>
#include <new>
#include <stdexcept>
#include <algorithm>
#include <chrono>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "host_cpus.h"
#include "../../include/herro_amd.h"

struct herro_reads {
  std::vector<std::string> ids, descs;
  std::vector<uint8_t> has_desc;
  std::vector<uint8_t> seq, qual;          // the streaming reader grows these ...
  uint8_t *seq_raw = nullptr, *qual_raw = nullptr;   // ... the parallel one allocates the exact size once (uninitialised: no 100-GB memset)
  std::vector<uint64_t> off;
  std::vector<const char*> id_ptr, desc_ptr;
  ~herro_reads() { free(seq_raw); free(qual_raw); }
};

namespace {

struct Zlib {
  void* h = nullptr;
  void* (*gzopen)(const char*, const char*) = nullptr;
  int (*gzread)(void*, void*, unsigned) = nullptr;
  int (*gzclose)(void*) = nullptr;
  const char* (*gzerror)(void*, int*) = nullptr;
  bool ok = false;
  Zlib() {
    for (const char* n : {"libz.so.1", "libz.so"}) {
      h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) return;
    gzopen = (void* (*)(const char*, const char*))dlsym(h, "gzopen");
    gzread = (int (*)(void*, void*, unsigned))dlsym(h, "gzread");
    gzclose = (int (*)(void*))dlsym(h, "gzclose");
    gzerror = (const char* (*)(void*, int*))dlsym(h, "gzerror");
    ok = gzopen && gzread && gzclose && gzerror;
  }
};

// The file as a stream of chunks (plain or gzip by magic): the reader never holds more than one chunk plus the record that
// straddles its end — a read set is hundreds of gigabytes of text, and needletail streams it too.
struct Source {
  FILE* f = nullptr;
  void* g = nullptr;
  bool eof = false;
  uint64_t plain_size = 0;   // size of a plain regular file (0: unknown), for the output reservation
  std::string why;
  bool open(const char* path) {
    f = fopen(path, "rb");
    if (!f) { why = "Cannot open file containing reads."; return false; }
    unsigned char magic[2] = {0, 0};
    const size_t got = fread(magic, 1, 2, f);
    const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (!gz) {
      struct stat st;
      if (fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) plain_size = (uint64_t)st.st_size;
      rewind(f);
      return true;
    }
    fclose(f);
    f = nullptr;
    static Zlib z;
    if (!z.ok) { why = "gzip input, but libz.so.1 is not available"; return false; }
    g = z.gzopen(path, "rb");
    if (!g) { why = "Cannot open file containing reads."; return false; }
    zl = &z;
    return true;
  }
  // appends up to `want` bytes to `buf`; false on a read error
  bool fill(std::string& buf, size_t want) {
    const size_t at = buf.size();
    buf.resize(at + want);
    size_t have = 0;
    while (have < want && !eof) {
      if (f) {
        const size_t n = fread(&buf[at + have], 1, want - have, f);
        if (n == 0) { eof = true; if (ferror(f)) { why = "Error parsing fastx file. (read error)"; buf.resize(at + have); return false; } }
        have += n;
      } else {
        const int n = zl->gzread(g, &buf[at + have], (unsigned)std::min<size_t>(want - have, 1u << 30));
        if (n < 0) { why = "Error parsing fastx file. (gzip stream)"; buf.resize(at + have); return false; }
        if (n == 0) {   // end of the stream — or a stream cut short (at a record boundary it would otherwise pass for a clean end: gzread
          int zerr = 0;  // returns 0 there too, and only gzerror tells: Z_BUF_ERROR = -5, "unexpected end of file")
          (void)zl->gzerror(g, &zerr);
          if (zerr != 0 && zerr != 1 /* Z_STREAM_END */) { why = "Error parsing fastx file. (truncated or damaged gzip stream)"; buf.resize(at + have); return false; }
          eof = true;
        }
        have += (size_t)n;
      }
    }
    buf.resize(at + have);
    return true;
  }
  ~Source() {
    if (f) fclose(f);
    if (g) zl->gzclose(g);
  }
  Zlib* zl = nullptr;
};

void set_err(char* err, uint64_t cap, const std::string& m) {
  if (err && cap) { strncpy(err, m.c_str(), cap - 1); err[cap - 1] = 0; }
}

// one line [b, e) without its terminator ('\n', and a '\r' before it); returns the start of the next line
template <class TextT>
size_t next_line(const TextT& t, size_t pos, size_t& b, size_t& e) {
  b = pos;
  size_t nl = t.find('\n', pos);
  if (nl == std::string::npos) nl = t.size();
  e = nl;
  if (e > b && t[e - 1] == '\r') e--;
  return nl < t.size() ? nl + 1 : t.size();
}

std::string npy_header(const std::string& descr, const std::string& shape) {
  std::string dict = "{'descr': " + descr + ", 'fortran_order': False, 'shape': " + shape + ", }";
  // magic (6) + version (2) + header length (2) + dict + padding + '\n', total a multiple of 64 (numpy's format 1.0 writer)
  size_t total = 10 + dict.size() + 1;
  const size_t pad = (64 - total % 64) % 64;
  dict.append(pad, ' ');
  dict.push_back('\n');
  std::string h("\x93NUMPY\x01\x00", 8);
  const uint16_t hl = (uint16_t)dict.size();
  h.push_back((char)(hl & 0xff));
  h.push_back((char)(hl >> 8));
  return h + dict;
}

bool mkdirs(const std::string& dir) {
  std::string cur;
  for (size_t i = 0; i <= dir.size(); i++) {
    if (i == dir.size() || dir[i] == '/') {
      if (!cur.empty() && mkdir(cur.c_str(), 0777) != 0) {
        struct stat st;
        if (stat(cur.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
      }
    }
    if (i < dir.size()) cur.push_back(dir[i]);
  }
  return true;
}

bool write_file(const std::string& path, const std::string& head, const void* body, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  bool ok = fwrite(head.data(), 1, head.size(), f) == head.size();
  if (ok && n) ok = fwrite(body, 1, n, f) == n;
  return fclose(f) == 0 && ok;
}

}  // namespace

extern "C" {

static herro_reads* fastx_read_impl(const char* path, uint32_t min_length, const char* const* keep_ids, uint64_t n_keep, char* err,
                                    uint64_t err_cap, std::unique_ptr<herro_reads>& guard);

herro_reads* herro_fastx_read(const char* path, uint32_t min_length, const char* const* keep_ids, uint64_t n_keep, char* err,
                              uint64_t err_cap) {
  std::unique_ptr<herro_reads> guard;   // no exception crosses the C ABI: an allocation failure is an error return
  try {
    return fastx_read_impl(path, min_length, keep_ids, n_keep, err, err_cap, guard);
  } catch (const std::bad_alloc&) {
    set_err(err, err_cap, "out of memory while reading the reads file");
  } catch (const std::exception& e) {
    set_err(err, err_cap, std::string("reads reader: ") + e.what());
  }
  return nullptr;
}

}  // extern "C"

namespace {

// The records of a text stream, one after the other: on(head, seq, qual, len) for every record that passes the length filter.
// `base` = file offset of the first byte `src` delivers; the walk stops in front of the first record that starts at or behind
// `stop_at` (*end_off = its offset, or the end of the file).  A record is final only if its parse stopped short of the end of the
// buffered text or the stream is exhausted: otherwise what looks like its end (or like an error) may just be the end of the chunk,
// and it is parsed again with more text behind it.
// A whole plain file mapped read-only, seen through the few members of std::string the record walk uses (the parallel reader: no
// chunk buffer, no copy of the text before the one into the result arrays).
struct MapText {
  const char* d = nullptr;
  size_t n = 0;
  size_t size() const { return n; }
  const char* data() const { return d; }
  char operator[](size_t i) const { return d[i]; }
  size_t find(char c, size_t pos) const {
    if (pos >= n) return std::string::npos;
    const void* p = memchr(d + pos, c, n - pos);
    return p ? (size_t)((const char*)p - d) : std::string::npos;
  }
  void erase(size_t, size_t) {}   // (never reached: the source of a mapped text is at its end from the start)
  void clear() {}
};
struct NoSource {
  bool eof = true;
  std::string why;
  bool fill(MapText&, size_t) { return true; }
};

template <class TextT, class SourceT, class OnRec>
bool parse_records(TextT& text, size_t pos, SourceT& src, uint64_t base, uint64_t stop_at, uint32_t min_length, size_t chunk, OnRec&& on, uint64_t* end_off,
                   std::string& why) {
  std::string seq, qual, head;
  size_t b = 0, e = 0;
  enum { KEEP, SKIP, ERROR };
  const char* emsg = nullptr;
  const char *seq_p = nullptr, *qual_p = nullptr;
  size_t seq_len = 0;
  auto parse_record = [&]() -> int {
    const char kind = text[b];   // [b, e): the header line, already fetched
    if (kind != '@' && kind != '>') { emsg = "Error parsing fastx file. (record does not start with '@' or '>')"; return ERROR; }
    head.assign(text.data() + b + 1, e - b - 1);
    seq.clear(); qual.clear();
    if (kind == '>') {                          // FASTA: sequence lines up to the next header
      size_t n = 0;
      while (pos < text.size() && text[pos] != '>') { pos = next_line(text, pos, b, e); n += e - b; }
      if (n < min_length) return SKIP;          // (the length filter comes first in get_reads)
      emsg = "Qualities should be present.";
      return ERROR;
    }
    bool plus = false;                          // FASTQ: sequence lines up to the '+' line, then as many quality bytes
    size_t sb = 0, se = 0, qb = 0, qe = 0, n_seq_lines = 0, n_qual_lines = 0;   // the record's only sequence / quality line, if single-line
    while (pos < text.size()) {
      pos = next_line(text, pos, b, e);
      if (b < e && text[b] == '+') { plus = true; break; }
      if (n_seq_lines++ == 0) { sb = b; se = e; }
      else { if (n_seq_lines == 2) seq.assign(text.data() + sb, se - sb); seq.append(text.data() + b, e - b); }
    }
    if (!plus) { emsg = "Error parsing fastx file. (no '+' line)"; return ERROR; }
    seq_len = n_seq_lines <= 1 ? se - sb : seq.size();
    size_t qual_len = 0;
    while (qual_len < seq_len && pos < text.size()) {
      pos = next_line(text, pos, b, e);
      if (n_qual_lines++ == 0) { qb = b; qe = e; }
      else { if (n_qual_lines == 2) qual.assign(text.data() + qb, qe - qb); qual.append(text.data() + b, e - b); }
      qual_len = n_qual_lines <= 1 ? qe - qb : qual.size();
    }
    if (qual_len != seq_len) { emsg = "Error parsing fastx file. (sequence and quality lengths differ)"; return ERROR; }
    if (seq_len < min_length) return SKIP;      // haec_io.rs:48-50
    seq_p = n_seq_lines <= 1 ? text.data() + sb : seq.data();      // four-line records are copied once, from the text
    qual_p = n_qual_lines <= 1 ? text.data() + qb : qual.data();
    return KEEP;
  };
  for (;;) {
    if (pos >= text.size()) {                   // everything buffered is consumed: next chunk
      if (src.eof) break;
      base += text.size();
      text.clear();
      pos = 0;
      if (!src.fill(text, chunk)) { why = src.why; return false; }
      continue;
    }
    const size_t rec0 = pos;
    pos = next_line(text, pos, b, e);
    int what = SKIP;                            // a blank line between records is skipped
    if (b != e) {
      if (base + rec0 >= stop_at) { if (end_off) *end_off = base + rec0; return true; }   // the next worker's first record
      what = parse_record();
    }
    if (pos >= text.size() && !src.eof) {       // the parse ran into the end of the chunk: again, with more text behind it
      text.erase(0, rec0);
      base += rec0;
      pos = 0;
      if (!src.fill(text, chunk)) { why = src.why; return false; }
      continue;
    }
    if (what == ERROR) { why = emsg; return false; }
    if (what == SKIP) continue;
    on(head, seq_p, qual_p, seq_len);
  }
  if (end_off) *end_off = base + text.size();
  return true;
}

// id / description of a header (splitn(2, ' ' | '\t'), haec_io.rs:52-61)
inline size_t id_cut(const std::string& head) { return head.find_first_of(" \t"); }

// Offset of the first line at or behind `from` that looks like the head of a plain four-line record — '@...', one non-empty
// line, '+...' — and starts in front of `limit`; UINT64_MAX if there is none.  A guess: the caller checks it against the true
// end of the record in front of it (a quality line may start with '@', and in a multi-line file the third line proves nothing).
uint64_t find_record_start(const MapText& t, uint64_t from, uint64_t limit) {
  size_t p = from;
  if (from && t[from - 1] != '\n') {            // `from` is inside a line: its successor is the first candidate
    const size_t nl = t.find('\n', from);
    if (nl == std::string::npos) return UINT64_MAX;
    p = nl + 1;
  }
  while (p < t.size() && p < limit) {
    const size_t l1 = t.find('\n', p);
    if (l1 == std::string::npos) return UINT64_MAX;
    if (t[p] == '@') {
      const size_t l2 = t.find('\n', l1 + 1);
      if (l2 == std::string::npos) return UINT64_MAX;
      if (l2 > l1 + 1 && l2 + 1 < t.size() && t[l2 + 1] == '+') return p;
    }
    p = l1 + 1;
  }
  return UINT64_MAX;
}

uint32_t fastx_threads() {
  if (const char* e = getenv("HERRO_FASTX_THREADS")) return (uint32_t)std::max(1, atoi(e));
  return herro::host_threads(16);   // HERRO_HOST_THREADS, else the CPUs the process may use (affinity / cgroup quota, not the hardware threads: ADVICE r4)
}

// The parallel reader of a plain regular file: T byte ranges, one worker each.  A worker guesses the first record of its range
// (find_record_start), walks the records that START in its range — the last one may end far behind it — and reports where it
// stopped; the guesses are then checked against those true ends from the front: start[k] == end[k - 1] for every k proves by
// induction that every worker parsed from a record boundary, i.e. exactly what one sequential pass sees.  Any mismatch, parse
// error or range without a boundary sends the whole file to the sequential reader (which also owns the error messages).  Two
// passes over the MAPPED file (no chunk buffers: the only copy of a base is the one into the result): the first counts records and
// bases per range, the second writes every record at its final offset — no merge, and the arrays are allocated once at their
// exact size.  Returns false for "use the sequential reader".
bool fastx_read_parallel(const char* path, uint64_t file_size, uint32_t T, uint32_t min_length, bool filter,
                         const std::unordered_set<std::string>& keep, size_t chunk, herro_reads* r) {
  struct Part {
    uint64_t lo = 0, hi = 0, start = UINT64_MAX, end = 0, n_rec = 0, n_bases = 0;
    bool ok = true;
    std::vector<std::string> ids, descs;
    std::vector<uint8_t> has_desc;
  };
  std::vector<Part> part(T);
  struct Mapping {
    void* p = MAP_FAILED; size_t n = 0; int fd = -1;
    ~Mapping() { if (p != MAP_FAILED) munmap(p, n); if (fd >= 0) close(fd); }
  } map;
  map.fd = ::open(path, O_RDONLY);
  if (map.fd < 0) return false;
  map.n = file_size;
  // The file is read through a private mapping: a file that is TRUNCATED by another process while it is being read raises SIGBUS (as for
  // any mmap reader: needletail's buffered reader in the reference would report a short read instead).  Growth and rewrites that keep the
  // size are caught by the size checks around the two passes; set HERRO_FASTX_THREADS=1 for the sequential (read()-based) reader if the
  // input may shrink under the process.
  map.p = mmap(nullptr, file_size, PROT_READ, MAP_PRIVATE, map.fd, 0);
  if (map.p == MAP_FAILED) return false;
  (void)madvise(map.p, file_size, MADV_SEQUENTIAL);
  MapText text{(const char*)map.p, (size_t)file_size};
  const uint64_t S = (file_size + T - 1) / T;
  for (uint32_t k = 0; k < T; k++) { part[k].lo = std::min<uint64_t>(file_size, (uint64_t)k * S); part[k].hi = std::min<uint64_t>(file_size, (uint64_t)(k + 1) * S); }
  auto run = [&](auto&& fn) {
    std::vector<std::thread> th;
    for (uint32_t k = 1; k < T; k++) th.emplace_back([&, k] { try { fn(k); } catch (...) { part[k].ok = false; } });
    try { fn(0); } catch (...) { part[0].ok = false; }
    for (auto& t : th) t.join();
  };
  auto wanted = [&](const std::string& head) { return !filter || keep.count(head.substr(0, id_cut(head))) != 0; };
  const bool trace = getenv("HERRO_FASTX_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  // ---- pass 1: boundaries and counts
  run([&](uint32_t k) {
    Part& P = part[k];
    P.start = k == 0 ? 0 : find_record_start(text, P.lo, P.hi);
    if (P.start == UINT64_MAX) return;           // no record starts in this range (checked against the neighbours below)
    NoSource src;
    std::string why;
    MapText t = text;
    P.ok = parse_records(t, (size_t)P.start, src, 0, k + 1 < T ? P.hi : UINT64_MAX, min_length, chunk,
                         [&](const std::string& head, const char*, const char*, size_t len) { if (wanted(head)) { P.n_rec++; P.n_bases += len; } }, &P.end, why);
  });
  const double t_pass1 = ms();
  uint64_t cur = 0, n_rec = 0, n_bases = 0;
  for (uint32_t k = 0; k < T; k++) {
    Part& P = part[k];
    if (!P.ok) return false;
    if (P.start == UINT64_MAX) {                 // fine only if the record in front of it covers the whole range
      if (cur < P.hi) return false;
      continue;
    }
    if (P.start != cur) return false;
    cur = P.end;
    n_rec += P.n_rec; n_bases += P.n_bases;
  }
  if (cur != file_size || n_rec > 0xffffffffull) return false;
  // ---- pass 2: every record to its place
  // (huge pages for these were tried — madvise(MADV_HUGEPAGE) on 2 MiB-aligned blocks: 11 GB/s once warm, but the FIRST call of a process, the one
  // that matters, waits 0.5-1.5 s per GB for the kernel to assemble them; plain pages: 6-8 GB/s from the first call on)
  r->seq_raw = (uint8_t*)malloc(std::max<uint64_t>(n_bases, 1));
  r->qual_raw = (uint8_t*)malloc(std::max<uint64_t>(n_bases, 1));
  if (!r->seq_raw || !r->qual_raw) throw std::bad_alloc();
  r->off.assign(n_rec + 1, 0);
  std::vector<uint64_t> rec0(T + 1, 0), base0(T + 1, 0);
  for (uint32_t k = 0; k < T; k++) { rec0[k + 1] = rec0[k] + part[k].n_rec; base0[k + 1] = base0[k] + part[k].n_bases; }
  run([&](uint32_t k) {
    Part& P = part[k];
    if (P.start == UINT64_MAX || P.n_rec == 0) return;
    NoSource src;
    MapText t = text;
    P.ids.reserve(P.n_rec); P.descs.reserve(P.n_rec); P.has_desc.reserve(P.n_rec);
    uint64_t i = rec0[k], o = base0[k], end = 0;
    std::string why;
    const bool ok = parse_records(t, (size_t)P.start, src, 0, k + 1 < T ? P.hi : UINT64_MAX, min_length, chunk,
                                  [&](const std::string& head, const char* sp, const char* qp, size_t len) {
                                    const size_t cut = id_cut(head);
                                    std::string id = head.substr(0, cut);
                                    if (filter && !keep.count(id)) return;
                                    if (i >= rec0[k + 1] || o + len > base0[k + 1]) { P.ok = false; return; }   // (the file changed under us)
                                    P.ids.push_back(std::move(id));
                                    P.has_desc.push_back(cut != std::string::npos);
                                    P.descs.push_back(cut != std::string::npos ? head.substr(cut + 1) : std::string());
                                    memcpy(r->seq_raw + o, sp, len);
                                    memcpy(r->qual_raw + o, qp, len);
                                    r->off[i] = o;
                                    o += len; i++;
                                  }, &end, why);
    if (!ok || end != P.end || i != rec0[k + 1] || o != base0[k + 1]) P.ok = false;
  });
  r->off[n_rec] = n_bases;
  if (trace) fprintf(stderr, "herro_fastx_read: boundaries + counts %.1f ms, records to their places %.1f ms (%u threads, %.2f GB)\n", t_pass1, ms() - t_pass1, T, file_size / 1e9);
  for (uint32_t k = 0; k < T; k++) if (!part[k].ok) { free(r->seq_raw); free(r->qual_raw); r->seq_raw = r->qual_raw = nullptr; r->off.assign(1, 0); return false; }
  r->ids.reserve(n_rec); r->descs.reserve(n_rec); r->has_desc.reserve(n_rec);
  for (uint32_t k = 0; k < T; k++) {
    for (auto& x : part[k].ids) r->ids.push_back(std::move(x));
    for (auto& x : part[k].descs) r->descs.push_back(std::move(x));
    r->has_desc.insert(r->has_desc.end(), part[k].has_desc.begin(), part[k].has_desc.end());
  }
  return true;
}

}  // namespace

extern "C" {

static herro_reads* fastx_read_impl(const char* path, uint32_t min_length, const char* const* keep_ids, uint64_t n_keep, char* err,
                                    uint64_t err_cap, std::unique_ptr<herro_reads>& guard) {
  if (!path) { set_err(err, err_cap, "null path"); return nullptr; }
  Source src;
  if (!src.open(path)) { set_err(err, err_cap, src.why); return nullptr; }
  std::unordered_set<std::string> keep;
  const bool filter = keep_ids != nullptr;
  for (uint64_t i = 0; filter && i < n_keep; i++) if (keep_ids[i]) keep.insert(keep_ids[i]);
  guard.reset(new herro_reads());
  herro_reads* r = guard.get();
  r->off.push_back(0);
  auto fail = [&](const std::string& m) -> herro_reads* { set_err(err, err_cap, m); guard.reset(); return nullptr; };
  size_t chunk = 32u << 20;
  if (const char* e = getenv("HERRO_FASTX_CHUNK")) chunk = std::max<size_t>(1, (size_t)strtoull(e, nullptr, 10));   // (tests: tiny chunks)
  auto finish = [&]() -> herro_reads* {
    for (size_t i = 0; i < r->ids.size(); i++) {
      r->id_ptr.push_back(r->ids[i].c_str());
      r->desc_ptr.push_back(r->has_desc[i] ? r->descs[i].c_str() : nullptr);
    }
    return guard.release();
  };
  // a plain regular file of some size: byte ranges on several threads (HERRO_FASTX_THREADS, default min(hardware threads, 16);
  // HERRO_FASTX_RANGE_MIN = smallest range worth a thread, 16 MiB); anything it is not sure about comes back here
  uint64_t range_min = 16u << 20;
  if (const char* e = getenv("HERRO_FASTX_RANGE_MIN")) range_min = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
  const uint32_t T = src.plain_size ? (uint32_t)std::min<uint64_t>(fastx_threads(), src.plain_size / range_min) : 0;
  if (T >= 2) {
    const bool par = fastx_read_parallel(path, src.plain_size, T, min_length, filter, keep, chunk, r);
    if (getenv("HERRO_FASTX_TRACE")) fprintf(stderr, "herro_fastx_read: %u byte ranges %s\n", T, par ? "read in parallel" : "not provably on record boundaries (or an error): sequential pass");
    if (par) return finish();
  }
  if (src.plain_size && !filter && min_length <= 1) {   // a FASTQ file is half bases, half qualities: one allocation instead of a doubling series
    r->seq.reserve(src.plain_size / 2);                  // (not when a filter may drop most of the file: the vectors then grow with what is kept)
    r->qual.reserve(src.plain_size / 2);
  }
  std::string why;
  std::string text;
  const bool ok = parse_records(text, 0, src, 0, UINT64_MAX, min_length, chunk,
                                [&](const std::string& head, const char* seq_p, const char* qual_p, size_t seq_len) {
                                  const size_t cut = id_cut(head);                 // splitn(2, ' ' | '\t')
                                  std::string id = head.substr(0, cut);
                                  if (filter && !keep.count(id)) return;           // haec_io.rs:63-69 (the caller passes core u neighbour when both are given)
                                  r->ids.push_back(std::move(id));
                                  r->has_desc.push_back(cut != std::string::npos);
                                  r->descs.push_back(cut != std::string::npos ? head.substr(cut + 1) : std::string());
                                  r->seq.insert(r->seq.end(), seq_p, seq_p + seq_len);
                                  r->qual.insert(r->qual.end(), qual_p, qual_p + seq_len);
                                  r->off.push_back(r->seq.size());
                                }, nullptr, why);
  if (!ok) return fail(why);
  return finish();
}

uint32_t herro_reads_count(const herro_reads* r) { return r ? (uint32_t)r->ids.size() : 0; }
const uint8_t* herro_reads_seq(const herro_reads* r) { return r ? (r->seq_raw ? r->seq_raw : r->seq.data()) : nullptr; }
const uint8_t* herro_reads_qual(const herro_reads* r) { return r ? (r->qual_raw ? r->qual_raw : r->qual.data()) : nullptr; }
const uint64_t* herro_reads_off(const herro_reads* r) { return r ? r->off.data() : nullptr; }
const char* const* herro_reads_ids(const herro_reads* r) { return r ? r->id_ptr.data() : nullptr; }
const char* const* herro_reads_descs(const herro_reads* r) { return r ? r->desc_ptr.data() : nullptr; }
void herro_reads_free(herro_reads* r) { delete r; }

int herro_write_window_features(const char* dir, uint32_t wid, const char* const* ids, uint32_t n_ids, const uint8_t* bases,
                                const uint8_t* quals, uint32_t length, const uint16_t* sup_pos, const uint8_t* sup_ins,
                                uint32_t n_sup) {
  if (!dir || (n_ids && !ids) || (length && (!bases || !quals)) || (n_sup && (!sup_pos || !sup_ins))) return HERRO_E_INVALID;
  const std::string d(dir);
  if (!mkdirs(d)) return HERRO_E_INVALID;
  const std::string stem = d + "/" + std::to_string(wid);
  std::string idtxt;
  for (uint32_t i = 0; i < n_ids; i++) { idtxt += ids[i]; idtxt += "\n"; }
  if (!write_file(stem + ".ids.txt", idtxt, nullptr, 0)) return HERRO_E_INVALID;
  const size_t cells = (size_t)length * 31;
  std::vector<uint8_t> feats(2 * cells);          // stack![Axis(0), bases, quals] (features.rs:742-743)
  if (cells) { memcpy(feats.data(), bases, cells); memcpy(feats.data() + cells, quals, cells); }
  if (!write_file(stem + ".features.npy", npy_header("'|u1'", "(2, " + std::to_string(length) + ", 31)"), feats.data(), feats.size()))
    return HERRO_E_INVALID;
  std::vector<uint8_t> sup((size_t)n_sup * 3);    // packed records {pos: <u2, ins: u1}
  for (uint32_t k = 0; k < n_sup; k++) {
    sup[3 * (size_t)k] = (uint8_t)(sup_pos[k] & 0xff);
    sup[3 * (size_t)k + 1] = (uint8_t)(sup_pos[k] >> 8);
    sup[3 * (size_t)k + 2] = sup_ins[k];
  }
  if (!write_file(stem + ".supported.npy", npy_header("[('pos', '<u2'), ('ins', '|u1')]", "(" + std::to_string(n_sup) + ",)"), sup.data(), sup.size()))
    return HERRO_E_INVALID;
  return HERRO_OK;
}

}  // extern "C"
