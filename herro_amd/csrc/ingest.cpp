// ingest.cpp — PAF / .oec.zst ingest on the host.
//
// Replaces, behind the C ABI of include/herro_amd.h:
//   parse_paf        overlaps.rs:117-202  (one PAF line per overlap, `cg:Z:` CIGAR in the last column)
//   read_batches     overlaps.rs:292-323  (zstd stream: line "n_targets", n id lines, then PAF lines)
// and hands out exactly what herro_job_create takes: target ids, per-target alignment ranges and an array of
// herro_alignment whose cigar pointers refer to the text owned by the returned object.
//
// Per-line rules restated from the reference (in this order, overlaps.rs:130-196):
//   * the line is read up to '\n' and its LAST BYTE IS DROPPED, newline or not (`buffer[..len - 1]`: a final line
//     without '\n' loses its last character);
//   * unknown query name -> skip; qlen/qstart/qend must be all digits (else panic; overflow wraps in u32);
//   * strand must be '+' or '-' (else panic); target not in `core` -> skip; unknown target -> skip;
//   * tlen/tstart/tend digits; CIGAR = last tab field minus its first 5 bytes ("cg:Z:"; shorter -> panic,
//     a line with no 10th field -> panic);
//   * self overlap -> skip; a (query, target) pair already seen -> skip (first one wins);
//   * kept lines are grouped by target.  The reference groups in a HashMap (iteration order unspecified);
//     here targets come out in order of first appearance, alignments in file order.
// A name that occurs more than once maps to its LAST index (HashMap::collect, lib.rs).
//
// Lines are independent up to the duplicate test, so they are parsed by a pool of threads over newline-aligned
// chunks; the duplicate test and the grouping run once, in file order, over the parsed records.
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <memory>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "host_cpus.h"
#include "../../include/herro_amd.h"

namespace {

struct Rec {
  herro_alignment a;
  uint64_t line;      // index of the line in the text (for error messages)
  int32_t status;     // 0 keep, 1 skip, <0 panic
  const char* msg;
};

struct SvHash {
  size_t operator()(std::string_view s) const noexcept {  // FNV-1a
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return (size_t)h;
  }
};
using NameMap = std::unordered_map<std::string_view, uint32_t, SvHash>;

bool parse_u32(std::string_view f, uint32_t& v) {
  uint32_t acc = 0;  // wraps like the release-mode fold of bytes_to_u32 (haec_io.rs:175-183)
  for (unsigned char c : f) {
    if (c < '0' || c > '9') return false;
    acc = acc * 10u + (uint32_t)(c - '0');
  }
  v = acc;
  return true;
}

// one line (without its dropped last byte) -> record
void parse_line(std::string_view ln, const NameMap& names, const uint8_t* core, Rec& r) {
  r.status = 1;
  r.msg = nullptr;
  size_t pos = 0;
  bool exhausted = false;
  auto next = [&](std::string_view& f) -> bool {  // split on '\t' like slice::split (an empty line yields one empty field)
    if (exhausted) return false;
    const size_t t = ln.find('\t', pos);
    if (t == std::string_view::npos) { f = ln.substr(pos); exhausted = true; }
    else { f = ln.substr(pos, t - pos); pos = t + 1; }
    return true;
  };
  auto panic = [&](const char* m) { r.status = HERRO_E_REFERENCE_PANIC; r.msg = m; };
  std::string_view f;
  if (!next(f)) return panic("called `Option::unwrap()` on a `None` value");
  auto q = names.find(f);
  if (q == names.end()) return;  // unknown query: skip
  r.a.qid = q->second;
  uint32_t* nums[3] = {&r.a.qlen, &r.a.qstart, &r.a.qend};
  for (uint32_t* p : nums) {
    if (!next(f)) return panic("called `Option::unwrap()` on a `None` value");
    if (!parse_u32(f, *p)) return panic("Character is not a valid digit");
  }
  if (!next(f)) return panic("called `Option::unwrap()` on a `None` value");
  if (f.empty()) return panic("index out of bounds: the len is 0 but the index is 0");
  if (f[0] == '+') r.a.strand = 0;
  else if (f[0] == '-') r.a.strand = 1;
  else return panic("Invalid strand character.");
  if (!next(f)) return panic("called `Option::unwrap()` on a `None` value");
  auto t = names.find(f);
  // `core` is consulted by NAME before the id lookup; with the per-read flag form used here an unknown name
  // is skipped either way
  if (t == names.end()) return;
  if (core && !core[t->second]) return;
  r.a.tid = t->second;
  uint32_t* tn[3] = {&r.a.tlen, &r.a.tstart, &r.a.tend};
  for (uint32_t* p : tn) {
    if (!next(f)) return panic("called `Option::unwrap()` on a `None` value");
    if (!parse_u32(f, *p)) return panic("Character is not a valid digit");
  }
  if (exhausted) return panic("called `Option::unwrap()` on a `None` value");  // data.last() on an empty iterator
  // (memrchr, not string_view::rfind: the last field is the CIGAR, ~1.6 KB per line, and rfind walks it a byte at a time —
  // that walk was two thirds of the whole parse)
  const void* ltp = memrchr(ln.data(), '\t', ln.size());
  const size_t lt = ltp ? (size_t)((const char*)ltp - ln.data()) : std::string_view::npos;
  const std::string_view last = ln.substr(lt + 1);
  if (last.size() < 5) return panic("range start index 5 out of range for slice");
  r.a.cigar = reinterpret_cast<const uint8_t*>(last.data() + 5);
  r.a.cigar_len = (uint32_t)(last.size() - 5);
  if (r.a.tid == r.a.qid) return;  // self overlap
  r.status = 0;
}

}  // namespace

struct herro_paf {
  // the bytes the cigar pointers refer to: either a buffer filled by the parser's threads (herro_paf_parse; not a std::string —
  // resize() would zero 50 MB on one thread first) or the string the zstd decoder produced (herro_oec_read)
  std::unique_ptr<char[]> buf;
  std::string text;
  std::vector<uint32_t> tids;
  std::vector<uint64_t> aln_off;
  std::vector<herro_alignment> alns;
};

// read name -> read id, built once per read set (the reference's `name_to_id`, lib.rs:136-140) and shared by every parse
struct herro_name_index {
  std::string blob;      // the names, owned: the map's keys point into it
  NameMap map;
  uint32_t n_reads = 0;
};

namespace {

void set_err(char* err, uint64_t cap, const std::string& m) {
  if (err && cap) {
    const size_t n = std::min<size_t>(m.size(), cap - 1);
    memcpy(err, m.data(), n);
    err[n] = 0;
  }
}

// `src` != nullptr: the caller's bytes, copied into a buffer the result owns (by the same threads that look for the line
// ends); else `owned` holds them already.
void fill_map(NameMap& map, uint32_t n_reads, const char* names, const uint64_t* name_off) {
  map.reserve((size_t)n_reads * 2);
  for (uint32_t i = 0; i < n_reads; i++) map[std::string_view(names + name_off[i], (size_t)(name_off[i + 1] - name_off[i]))] = i;
}

herro_paf* parse_owned(const char* src, size_t src_len, std::string&& owned, size_t body, uint32_t n_reads, const NameMap& map,
                       const uint8_t* core, int n_threads, char* err, uint64_t err_cap, bool borrow = false) {
  auto out = new herro_paf();
  size_t len;
  if (src && borrow) len = src_len;   // the caller keeps the bytes alive (herro_paf_parse_view): nothing is copied
  else if (src) { len = src_len; out->buf.reset(new char[std::max<size_t>(len, 1)]); }
  else { out->text = std::move(owned); len = out->text.size(); }
  const char* base = src ? (borrow ? src : out->buf.get()) : out->text.data();
  char* wbase = (src && !borrow) ? out->buf.get() : nullptr;   // where the caller's bytes are copied to, if they are

  const uint32_t want = n_threads > 0 ? (uint32_t)n_threads : herro::host_threads(32);   // the CPUs the process may use, not the hardware threads
  auto run = [&](uint32_t nthr, const std::function<void(uint32_t)>& f) {
    if (nthr <= 1) { f(0); return; }
    std::vector<std::thread> th;
    for (uint32_t i = 1; i < nthr; i++) th.emplace_back(f, i);
    f(0);
    for (auto& x : th) x.join();
  };
  // ---- pass 1, byte ranges: copy (if the bytes are the caller's) and note every '\n'
  const size_t span = len > body ? len - body : 0;
  const uint32_t nt1 = (uint32_t)std::max<size_t>(1, std::min<size_t>(want, (span + (1u << 20) - 1) >> 20));
  std::vector<std::vector<size_t>> nls(nt1);
  if (wbase && body) memcpy(wbase, src, std::min(body, len));
  run(nt1, [&](uint32_t k) {
    const size_t b = body + span * k / nt1, e = body + span * (k + 1) / nt1;
    if (wbase && e > b) memcpy(wbase + b, src + b, e - b);
    std::vector<size_t>& v = nls[k];
    v.reserve((e - b) / 512 + 16);
    for (size_t p = b; p < e;) {
      const void* nl = memchr(base + p, '\n', e - p);
      if (!nl) break;
      p = (size_t)((const char*)nl - base) + 1;
      v.push_back(p);                      // start of the next line
    }
  });
  // line starts: `body`, then the byte after every newline (a newline at the very end starts no line)
  std::vector<size_t> ls;
  {
    size_t total = 1;
    for (auto& v : nls) total += v.size();
    ls.reserve(total + 1);
    if (body < len) ls.push_back(body);
    for (auto& v : nls) for (size_t p : v) if (p < len) ls.push_back(p);
  }
  const size_t nl = ls.size();
  ls.push_back(len);
  std::vector<Rec> recs(nl);
  const uint32_t nthr = (uint32_t)std::max<size_t>(1, std::min<size_t>(want, (nl + 1023) / 1024));
  std::atomic<size_t> next{0};
  run(nthr, [&](uint32_t) {
    for (;;) {
      const size_t b = next.fetch_add(1024);
      if (b >= nl) break;
      const size_t e = std::min(nl, b + 1024);
      for (size_t i = b; i < e; i++) {
        const size_t l0 = ls[i], l1 = ls[i + 1];  // read_until: bytes l0..l1 incl. the delimiter if present
        recs[i].line = i;
        memset(&recs[i].a, 0, sizeof(herro_alignment));
        parse_line(std::string_view(base + l0, l1 - l0 - 1), map, core, recs[i]);  // `buffer[..len - 1]`
      }
    }
  });
  // duplicates + grouping, in file order.  (query, target) pairs go through an open-addressing table (a node-based set cost
  // 70 ns per overlap, as much as parsing its line on eight threads); a target's group is found by direct index.
  size_t n_keep = 0;
  for (size_t i = 0; i < nl; i++) {
    const Rec& r = recs[i];
    if (r.status < 0) {
      set_err(err, err_cap, std::string(r.msg) + " (PAF line " + std::to_string(i + 1) + ")");
      delete out;
      return nullptr;
    }
    n_keep += r.status == 0;
  }
  size_t cap = 16;
  while (cap < 2 * n_keep) cap <<= 1;
  std::vector<uint64_t> seen(cap, 0);                       // key + 1 (0 = empty); a pair of read ids is never 2^64 - 1
  std::vector<uint32_t> slot(n_reads, 0);                   // group + 1 of a target (tid < n_reads: it came out of the name index)
  std::vector<std::vector<uint32_t>> groups;
  for (size_t i = 0; i < nl; i++) {
    const Rec& r = recs[i];
    if (r.status) continue;
    const uint64_t key = (((uint64_t)r.a.qid << 32) | r.a.tid) + 1;
    size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 17) & (cap - 1);
    bool dup = false;
    while (seen[h]) {
      if (seen[h] == key) { dup = true; break; }
      h = (h + 1) & (cap - 1);
    }
    if (dup) continue;                                        // the first overlap of a pair is kept
    seen[h] = key;
    uint32_t& g = slot[r.a.tid];
    if (!g) {
      groups.emplace_back();
      g = (uint32_t)groups.size();
      out->tids.push_back(r.a.tid);
    }
    groups[g - 1].push_back((uint32_t)i);
  }
  out->alns.reserve(n_keep);
  out->aln_off.assign(1, 0);
  for (auto& g : groups) {
    for (uint32_t i : g) out->alns.push_back(recs[i].a);
    out->aln_off.push_back(out->alns.size());
  }
  return out;
}

// ---- zstd through the system's libzstd.so.1 (stable C ABI; the image ships the library without headers) ------------
struct ZIn { const void* src; size_t size; size_t pos; };
struct ZOut { void* dst; size_t size; size_t pos; };
struct Zstd {
  void* h = nullptr;
  void* (*createDStream)() = nullptr;
  size_t (*freeDStream)(void*) = nullptr;
  size_t (*initDStream)(void*) = nullptr;
  size_t (*decompressStream)(void*, ZOut*, ZIn*) = nullptr;
  unsigned (*isError)(size_t) = nullptr;
  const char* (*getErrorName)(size_t) = nullptr;
  bool ok = false;
  Zstd() {
    for (const char* n : {"libzstd.so.1", "libzstd.so"}) {
      h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) return;
    createDStream = (void* (*)())dlsym(h, "ZSTD_createDStream");
    freeDStream = (size_t(*)(void*))dlsym(h, "ZSTD_freeDStream");
    initDStream = (size_t(*)(void*))dlsym(h, "ZSTD_initDStream");
    decompressStream = (size_t(*)(void*, ZOut*, ZIn*))dlsym(h, "ZSTD_decompressStream");
    isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
    getErrorName = (const char* (*)(size_t))dlsym(h, "ZSTD_getErrorName");
    ok = createDStream && freeDStream && initDStream && decompressStream && isError && getErrorName;
  }
};

bool zstd_decode_file(const char* path, std::string& out, std::string& why) {
  static Zstd z;
  if (!z.ok) { why = "libzstd.so.1 not available"; return false; }
  FILE* f = fopen(path, "rb");
  if (!f) { why = std::string("cannot open ") + path; return false; }
  void* ds = z.createDStream();
  z.initDStream(ds);
  std::vector<char> ib(1 << 17), ob(1 << 18);
  bool good = true;
  for (;;) {
    const size_t n = fread(ib.data(), 1, ib.size(), f);
    if (!n) break;
    ZIn in{ib.data(), n, 0};
    while (in.pos < in.size) {
      ZOut o{ob.data(), ob.size(), 0};
      const size_t rc = z.decompressStream(ds, &o, &in);
      if (z.isError(rc)) { why = std::string("zstd: ") + z.getErrorName(rc); good = false; break; }
      out.append(ob.data(), o.pos);
    }
    if (!good) break;
  }
  z.freeDStream(ds);
  fclose(f);
  return good;
}

}  // namespace

extern "C" {

// header of an .oec.zst stream (overlaps.rs:304-320): "<n_targets>\n" then n_targets id lines (read and ignored by the
// reference too); returns where the PAF lines start
static size_t oec_body(const std::string& text) {
  size_t p = 0;
  uint32_t n_targets = 0;
  {
    const size_t e = text.find('\n', p);
    const size_t end = e == std::string::npos ? text.size() : e + 1;  // read_until: including the delimiter
    for (size_t i = p; i + 1 < end; i++) n_targets = n_targets * 10u + (uint32_t)((unsigned char)text[i] - '0');  // buf[..len - 1], unchecked
    p = end;
  }
  for (uint32_t i = 0; i < n_targets && p < text.size(); i++) {
    const size_t e = text.find('\n', p);
    p = e == std::string::npos ? text.size() : e + 1;
  }
  return p;
}

herro_name_index* herro_name_index_create(uint32_t n_reads, const char* names, const uint64_t* name_off) {
  if (n_reads && (!names || !name_off)) return nullptr;
  auto ix = new herro_name_index();
  ix->n_reads = n_reads;
  if (n_reads) {
    ix->blob.assign(names + name_off[0], (size_t)(name_off[n_reads] - name_off[0]));
    std::vector<uint64_t> off(n_reads + 1);
    for (uint32_t i = 0; i <= n_reads; i++) off[i] = name_off[i] - name_off[0];
    fill_map(ix->map, n_reads, ix->blob.data(), off.data());
  }
  return ix;
}
void herro_name_index_free(herro_name_index* ix) { delete ix; }

herro_paf* herro_paf_parse_indexed(const char* text, uint64_t len, const herro_name_index* ix, const uint8_t* core, int n_threads,
                                   char* err, uint64_t err_cap) {
  if ((!text && len) || !ix) { set_err(err, err_cap, "invalid argument"); return nullptr; }
  return parse_owned(text ? text : "", (size_t)len, std::string(), 0, ix->n_reads, ix->map, core, n_threads, err, err_cap);
}

herro_paf* herro_paf_parse_view(const char* text, uint64_t len, const herro_name_index* ix, const uint8_t* core, int n_threads,
                                char* err, uint64_t err_cap) {
  if ((!text && len) || !ix) { set_err(err, err_cap, "invalid argument"); return nullptr; }
  return parse_owned(text ? text : "", (size_t)len, std::string(), 0, ix->n_reads, ix->map, core, n_threads, err, err_cap, true);
}

herro_paf* herro_oec_read_indexed(const char* path, const herro_name_index* ix, const uint8_t* core, int n_threads, char* err,
                                  uint64_t err_cap) {
  if (!path || !ix) { set_err(err, err_cap, "invalid argument"); return nullptr; }
  std::string text, why;
  if (!zstd_decode_file(path, text, why)) { set_err(err, err_cap, why); return nullptr; }
  const size_t p = oec_body(text);
  return parse_owned(nullptr, 0, std::move(text), p, ix->n_reads, ix->map, core, n_threads, err, err_cap);
}

herro_paf* herro_paf_parse(const char* text, uint64_t len, uint32_t n_reads, const char* names, const uint64_t* name_off,
                           const uint8_t* core, int n_threads, char* err, uint64_t err_cap) {
  if ((!text && len) || (n_reads && (!names || !name_off))) { set_err(err, err_cap, "invalid argument"); return nullptr; }
  NameMap map;                                   // keys point into the caller's names: valid for the call
  fill_map(map, n_reads, names, name_off);
  return parse_owned(text ? text : "", (size_t)len, std::string(), 0, n_reads, map, core, n_threads, err, err_cap);
}

herro_paf* herro_oec_read(const char* path, uint32_t n_reads, const char* names, const uint64_t* name_off, const uint8_t* core,
                          int n_threads, char* err, uint64_t err_cap) {
  if (!path || (n_reads && (!names || !name_off))) { set_err(err, err_cap, "invalid argument"); return nullptr; }
  std::string text, why;
  if (!zstd_decode_file(path, text, why)) { set_err(err, err_cap, why); return nullptr; }
  NameMap map;
  fill_map(map, n_reads, names, name_off);
  const size_t p = oec_body(text);
  return parse_owned(nullptr, 0, std::move(text), p, n_reads, map, core, n_threads, err, err_cap);
}

uint32_t herro_paf_n_targets(const herro_paf* p) { return p ? (uint32_t)p->tids.size() : 0; }
const uint32_t* herro_paf_target_ids(const herro_paf* p) { return p ? p->tids.data() : nullptr; }
const uint64_t* herro_paf_aln_off(const herro_paf* p) { return p ? p->aln_off.data() : nullptr; }
const herro_alignment* herro_paf_alignments(const herro_paf* p) { return p ? p->alns.data() : nullptr; }
void herro_paf_free(herro_paf* p) { delete p; }

}  // extern "C"
