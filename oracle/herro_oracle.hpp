// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (C++17, single-threaded, bit-exact for all integer/byte work) of the
// lbcb-sci/herro v0.1.1 hot path: 2-bit read codec, CIGAR iterator, windowing, per-window
// pileup feature generation, batch collation and the consensus decoder.  Every function
// cites the reference file:line it follows (paths are relative to the reference's src/).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything
// under oracle/.  The product (herro_amd/) never includes, links or calls this code.
//
// PARITY STATUS: the reference cannot be built here (no Rust toolchain, crates not
// vendored).  Pinned: the 2-bit codec against the reference's 11 codec tests
// (haec_io.rs:185-300, tests/golden/codec_vectors.json); extract_windows against the
// expected values of the reference's seven windowing tests (windowing.rs:309-606; they no
// longer compile, their sequences and per-window numbers are data —
// tests/test_oracle_ref_windowing.py).  For pileup / consensus the reference has no tests
// or fixtures, so those parts are "parity unpinned" beyond a hand-traced known-answer
// example re-derived from the code.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace oracle {

// A reference `panic!`/`assert!`/`unwrap()` failure.  The reference aborts the process
// (Cargo.toml:18 panic = "abort"); the oracle surfaces it as an exception -> error text.
struct RefPanic : std::runtime_error {
  using std::runtime_error::runtime_error;
};
[[noreturn]] inline void ref_panic(const std::string& m) { throw RefPanic(m); }

// ---------------------------------------------------------------------------------------
// haec_io.rs — 2-bit codec
// ---------------------------------------------------------------------------------------

// haec_io.rs:7-15 BASE_ENCODING: A/a 0, C/c 1, G/g 2, T/t 3, everything else <128 is 255.
inline uint64_t base_encoding(uint8_t b) {
  switch (b) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default:
      if (b >= 128) ref_panic("BASE_ENCODING index out of bounds");  // [u64;128] indexing
      return 255;
  }
}
static const uint8_t BASE_DECODING[4] = {'A', 'C', 'G', 'T'};  // haec_io.rs:17

struct HAECSeq {  // haec_io.rs:78-81
  std::vector<uint64_t> data;
  size_t length = 0;
  size_t len() const { return length; }
};

// haec_io.rs:121-136 encode.  Note the unmasked OR of 255 for non-ACGT bytes (quirk: an
// 'N' turns bases i..i+3 of the same 32-base word into 'T').
inline HAECSeq encode(const uint8_t* seq, size_t n) {
  HAECSeq out;
  out.data.reserve((n + 31) / 32);
  uint64_t block = 0;
  for (size_t i = 0; i < n; i++) {
    uint64_t c = base_encoding(seq[i]);
    block |= c << ((i << 1) & 63);
    if (((i + 1) & 31) == 0 || i == n - 1) {
      out.data.push_back(block);
      block = 0;
    }
  }
  out.length = n;
  return out;
}

// haec_io.rs:138-173 decode(range start..end, is_reversed) into buffer[0..end-start).
inline void decode(const HAECSeq& s, size_t start, size_t end, bool is_reversed, uint8_t* buffer) {
  if (end > s.length) ref_panic("Out of bounds for 2-bit sequence decoding.");
  if (start >= end) return;
  const uint64_t rc_mask = is_reversed ? 3 : 0;
  for (size_t i0 = start; i0 < end; i0++) {
    size_t idx = i0 - start;
    size_t i = is_reversed ? end - idx - 1 : i0;
    uint64_t code = ((s.data[i >> 5] >> ((i << 1) & 63)) & 3) ^ rc_mask;
    buffer[idx] = BASE_DECODING[code];
  }
}

struct HAECRecord {  // haec_io.rs:19-24
  std::string id;
  std::optional<std::string> description;
  HAECSeq seq;
  std::vector<uint8_t> qual;
};

// ---------------------------------------------------------------------------------------
// overlaps.rs — Overlap / Alignment
// ---------------------------------------------------------------------------------------
enum class Strand : uint8_t { Forward = 0, Reverse = 1 };  // overlaps.rs:28-31

struct Overlap {  // overlaps.rs:45-55
  uint32_t qid, qlen, qstart, qend;
  Strand strand;
  uint32_t tid, tlen, tstart, tend;
  uint32_t return_other_id(uint32_t id) const { return qid == id ? tid : qid; }  // :82-88
};

struct Alignment {  // overlaps.rs:92-95
  Overlap overlap;
  std::string cigar;  // ASCII "\d+[MID]" ...
};

// ---------------------------------------------------------------------------------------
// aligners.rs — CigarOp / CigarIter
// ---------------------------------------------------------------------------------------
enum class Op : uint8_t { Match, Mismatch, Insertion, Deletion };  // aligners.rs:7-12
struct CigarOp {
  Op op;
  uint32_t len;
};
struct CigarItem {
  CigarOp op;
  size_t start, end;  // byte range within the slice handed to the iterator
};

// aligners.rs:252-293 — yields (op, byte range) lazily; panics on len==0 / unknown op.
struct CigarIter {
  const uint8_t* data;
  size_t n, pos = 0;
  CigarIter(const uint8_t* d, size_t len) : data(d), n(len) {}
  bool next(CigarItem& out) {
    if (pos >= n) return false;
    size_t start = pos;
    uint32_t len = 0;
    while (true) {
      if (pos >= n) ref_panic("CigarIter: index out of bounds");
      uint8_t c = data[pos];
      if (c < '0' || c > '9') break;
      len = len * 10 + (uint32_t)(c - '0');
      pos++;
    }
    if (!(len > 0)) ref_panic("Length has to be longer than 0");
    Op op;
    switch (data[pos]) {
      case 'M': op = Op::Match; break;
      case 'I': op = Op::Insertion; break;
      case 'D': op = Op::Deletion; break;
      default: ref_panic(std::string("Unexpected cigar operation ") + (char)data[pos]);
    }
    pos++;
    out = {{op, len}, start, pos};
    return true;
  }
};

// ---------------------------------------------------------------------------------------
// windowing.rs
// ---------------------------------------------------------------------------------------
struct OverlapWindow {  // windowing.rs:7-16
  const Overlap* overlap;
  uint32_t tstart, qstart, qend;
  size_t cigar_start_idx;
  uint32_t cigar_start_offset;
  size_t cigar_end_idx;
  uint32_t cigar_end_offset;
};
using Windows = std::vector<std::vector<OverlapWindow>>;

// windowing.rs:275-293
inline CigarOp get_last_cigar_op(const std::string& cigar) {
  if (cigar.empty()) ref_panic("get_last_cigar_op: empty cigar");
  uint8_t op = cigar[cigar.size() - 1];
  uint32_t len = 0, p10 = 1;
  for (size_t k = cigar.size() - 1; k-- > 0;) {
    uint8_t c = cigar[k];
    if (c < '0' || c > '9') break;
    len += (uint32_t)(c - '0') * p10;
    p10 *= 10;
  }
  switch (op) {
    case 'M': return {Op::Match, len};
    case 'I': return {Op::Insertion, len};
    case 'D': return {Op::Deletion, len};
    default: ref_panic("Invalid cigar op");
  }
}

// windowing.rs:44-273.  u32 arithmetic mirrors the reference (release build wraps; the
// oracle treats an underflow that the reference would only survive by wrapping as a panic
// only where the reference indexes with it).
inline void extract_windows(Windows& windows, const Overlap* overlap, const std::string& cigar,
                            uint32_t tshift, uint32_t qshift, bool is_target,
                            uint32_t window_size) {
  if ((is_target && (overlap->tend - overlap->tstart) < window_size) ||
      ((overlap->qend - overlap->qstart) < window_size))
    return;  // :53-57

  uint32_t first_window, last_window, tstart, tpos, qpos = 0;
  const uint32_t zeroth_window_thresh = (uint32_t)(0.1f * (float)window_size);  // :65
  const uint32_t nth_window_thresh =
      is_target ? overlap->tlen - zeroth_window_thresh : overlap->qlen - zeroth_window_thresh;

  if (is_target) {  // :74-88
    first_window = overlap->tstart < zeroth_window_thresh
                       ? 0
                       : (overlap->tstart + window_size - 1) / window_size;
    last_window = overlap->tend > nth_window_thresh ? (overlap->tend - 1) / window_size + 1
                                                    : overlap->tend / window_size;
    tstart = overlap->tstart;
    tpos = overlap->tstart;
  } else {  // :89-104 (latent: parse_paf groups by target id only)
    first_window = overlap->qstart < zeroth_window_thresh
                       ? 0
                       : (overlap->qstart + window_size - 1) / window_size;
    last_window = overlap->qend > nth_window_thresh ? (overlap->qend - 1) / window_size + 1
                                                    : overlap->qend / window_size;
    tstart = overlap->qstart;
    tpos = overlap->qstart;
  }

  // :106 `last_window - first_window < 1` in u32: a wrapped negative is a huge number (not <1).
  if ((uint32_t)(last_window - first_window) < 1) return;

  std::optional<uint32_t> t_window_start, q_window_start, cigar_start_offset;
  std::optional<size_t> cigar_start_idx;

  tpos += tshift;
  qpos += qshift;

  if (tpos % window_size == 0 || tstart < zeroth_window_thresh) {  // :120-125
    t_window_start = tpos;
    q_window_start = qpos;
    cigar_start_idx = 0;
    cigar_start_offset = 0;
  }

  auto unwrap_u32 = [](const std::optional<uint32_t>& o) -> uint32_t {
    if (!o) ref_panic("called `Option::unwrap()` on a `None` value");
    return *o;
  };
  auto unwrap_sz = [](const std::optional<size_t>& o) -> size_t {
    if (!o) ref_panic("called `Option::unwrap()` on a `None` value");
    return *o;
  };
  auto push = [&](size_t widx, OverlapWindow ow) {
    if (widx >= windows.size()) ref_panic("windows index out of bounds");
    windows[widx].push_back(ow);
  };

  CigarIter it((const uint8_t*)cigar.data(), cigar.size());
  CigarItem cur;
  bool have_next;
  CigarItem nxt;
  have_next = it.next(nxt);  // peekable
  while (have_next) {
    cur = nxt;
    have_next = it.next(nxt);
    const CigarOp op = cur.op;
    const bool is_m = (op.op == Op::Match || op.op == Op::Mismatch);
    uint32_t tnew, qnew;
    if (is_m) {
      tnew = tpos + op.len;
      qnew = qpos + op.len;
    } else if (op.op == Op::Deletion) {
      tnew = tpos + op.len;
      qnew = qpos;
    } else {  // Insertion :132-135
      qpos += op.len;
      continue;
    }

    const uint32_t current_w = tpos / window_size;
    const uint32_t new_w = tnew / window_size;
    const uint32_t diff_w = new_w - current_w;
    if (diff_w == 0) {  // :142-147
      tpos = tnew;
      qpos = qnew;
      continue;
    }

    for (uint32_t i = 1; i < diff_w; i++) {  // :150-195
      const uint32_t offset = (current_w + i) * window_size - tpos;
      const uint32_t q_start_new = is_m ? qpos + offset : qpos;
      if (cigar_start_idx.has_value()) {
        push((size_t)(current_w + i) - 1,
             OverlapWindow{overlap, unwrap_u32(t_window_start), unwrap_u32(q_window_start),
                           q_start_new, unwrap_sz(cigar_start_idx),
                           unwrap_u32(cigar_start_offset), cur.end, offset});
        t_window_start = tpos + offset;
        q_window_start = is_m ? qpos + offset : qpos;
        cigar_start_idx = cur.start;
        cigar_start_offset = offset;
      } else {
        t_window_start = tpos + offset;
        q_window_start = is_m ? qpos + offset : qpos;
        cigar_start_idx = cur.start;
        cigar_start_offset = offset;
      }
    }

    const uint32_t offset = new_w * window_size - tpos;  // :198
    uint32_t qend = is_m ? qpos + offset : qpos;
    size_t cigar_end_idx, next_cigar_start_idx;
    uint32_t cigar_end_offset, next_cigar_start_offset;
    if (tnew == new_w * window_size) {  // :210-223
      if (have_next && nxt.op.op == Op::Insertion) {
        qend += nxt.op.len;
        cigar_end_idx = nxt.end;
        cigar_end_offset = nxt.op.len;
      } else {
        cigar_end_idx = cur.end;
        cigar_end_offset = op.len;
      }
      next_cigar_start_idx = cigar_end_idx;
      next_cigar_start_offset = 0;
    } else {  // :224-230
      cigar_end_idx = cur.end;
      cigar_end_offset = offset;
      next_cigar_start_idx = cur.start;
      next_cigar_start_offset = cigar_end_offset;
    }

    if (cigar_start_idx.has_value()) {  // :232-248
      push((size_t)new_w - 1,
           OverlapWindow{overlap, unwrap_u32(t_window_start), unwrap_u32(q_window_start), qend,
                         unwrap_sz(cigar_start_idx), unwrap_u32(cigar_start_offset),
                         cigar_end_idx, cigar_end_offset});
      t_window_start = tpos + offset;
      q_window_start = qend;
      cigar_start_idx = next_cigar_start_idx;
      cigar_start_offset = next_cigar_start_offset;
    } else {
      t_window_start = tpos + offset;
      q_window_start = qend;
      cigar_start_idx = next_cigar_start_idx;
      cigar_start_offset = next_cigar_start_offset;
    }

    tpos = tnew;
    qpos = qnew;
  }

  if (tpos > nth_window_thresh && tpos % window_size != 0) {  // :261-272
    push((size_t)last_window - 1,
         OverlapWindow{overlap, unwrap_u32(t_window_start), unwrap_u32(q_window_start), qpos,
                       unwrap_sz(cigar_start_idx), unwrap_u32(cigar_start_offset), cigar.size(),
                       get_last_cigar_op(cigar).len});
  }
}

// ---------------------------------------------------------------------------------------
// features.rs
// ---------------------------------------------------------------------------------------
static const size_t TOP_K_SORT = 30;  // features.rs:22

inline uint8_t base_lower(uint8_t b) {  // features.rs:24-32 BASE_LOWER
  switch (b) {
    case 'A': return 'a';
    case 'C': return 'c';
    case 'G': return 'g';
    case 'T': return 't';
    default: return 255;
  }
}
inline uint8_t base_forward(uint8_t b) {  // features.rs:34-42 BASE_FORWARD
  switch (b) {
    case '#': case '*': return '*';
    case 'A': case 'a': return 'A';
    case 'C': case 'c': return 'C';
    case 'G': case 'g': return 'G';
    case 'T': case 't': return 'T';
    default: return 255;
  }
}

struct SupportedPos {  // features.rs:896-900
  uint16_t pos;
  uint8_t ins;
  bool operator==(const SupportedPos& o) const { return pos == o.pos && ins == o.ins; }
};

// Row-major [rows, cols] u8 matrix (ndarray Array2<u8> standard layout).
struct Mat {
  size_t rows = 0, cols = 0;
  std::vector<uint8_t> v;
  Mat() = default;
  Mat(size_t r, size_t c, uint8_t fill) : rows(r), cols(c), v(r * c, fill) {}
  uint8_t& at(size_t r, size_t c) { return v[r * cols + c]; }
  uint8_t at(size_t r, size_t c) const { return v[r * cols + c]; }
};

// Effective op length rule (features.rs:82-90 / :181-188 / :591-614).
inline uint32_t effective_len(const OverlapWindow& ow, const CigarItem& it) {
  const size_t S = ow.cigar_end_idx - ow.cigar_start_idx;
  if (it.start == 0 && it.end == S) return ow.cigar_end_offset - ow.cigar_start_offset;
  if (it.start == 0) return it.op.len - ow.cigar_start_offset;
  if (it.end == S) return ow.cigar_end_offset;
  return it.op.len;
}

using CigarMap = std::unordered_map<uint32_t, const std::string*>;

inline const std::string& cigar_of(const CigarMap& m, uint32_t qid) {
  auto it = m.find(qid);
  if (it == m.end()) ref_panic("called `Option::unwrap()` on a `None` value (cigar map)");
  return *it->second;
}
inline void check_slice(const std::string& c, const OverlapWindow& ow) {
  if (ow.cigar_start_idx > ow.cigar_end_idx || ow.cigar_end_idx > c.size())
    ref_panic("cigar slice out of range");
}

// features.rs:44-95
inline std::vector<uint16_t> get_max_ins_for_window(const std::vector<OverlapWindow>& overlaps,
                                                    const CigarMap& cmap, uint32_t tid,
                                                    size_t tstart, size_t window_length) {
  std::vector<uint16_t> max_ins(window_length, 0);
  for (const auto& ow : overlaps) {
    size_t tpos = (size_t)ow.tstart - tstart;
    const uint32_t qid = ow.overlap->return_other_id(tid);
    const std::string& cigar = cigar_of(cmap, qid);
    check_slice(cigar, ow);
    CigarIter it((const uint8_t*)cigar.data() + ow.cigar_start_idx,
                 ow.cigar_end_idx - ow.cigar_start_idx);
    CigarItem ci;
    while (it.next(ci)) {
      if (ci.op.op == Op::Insertion) {
        if (!(tpos <= max_ins.size())) ref_panic("max_ins: tpos bigger than tseq");
        if (tpos == 0 || tpos - 1 >= max_ins.size()) ref_panic("max_ins[tpos - 1] out of bounds");
        max_ins[tpos - 1] = std::max<uint16_t>(max_ins[tpos - 1], (uint16_t)ci.op.len);
        continue;
      }
      const size_t l = ci.op.len;
      const size_t S = ow.cigar_end_idx - ow.cigar_start_idx;
      if (ci.start == 0 && ci.end == S)
        tpos += (size_t)(ow.cigar_end_offset - ow.cigar_start_offset);
      else if (ci.start == 0)
        tpos += l - (size_t)ow.cigar_start_offset;
      else if (ci.end == S)
        tpos += (size_t)ow.cigar_end_offset;
      else
        tpos += l;
    }
  }
  return max_ins;
}

// features.rs:97-108
inline std::pair<uint32_t, uint32_t> get_query_region(const OverlapWindow& w, uint32_t tid) {
  uint32_t qstart, qend;
  if (w.overlap->tid == tid) {
    qstart = w.overlap->qstart;
    qend = w.overlap->qend;
  } else {
    qstart = w.overlap->tstart;
    qend = w.overlap->tend;
  }
  if (w.overlap->strand == Strand::Forward) return {qstart + w.qstart, qstart + w.qend};
  return {qend - w.qend, qend - w.qstart};
}

// features.rs:110-237 — writes column `col` of bases/quals.
inline void get_features_for_ol_window(Mat& bases, Mat& quals, size_t col,
                                       const OverlapWindow& window, const std::string& cigar_full,
                                       const HAECRecord& query, size_t offset, uint32_t tid,
                                       const std::vector<uint16_t>& max_ins,
                                       std::vector<uint8_t>& qbuffer) {
  auto [rs, re] = get_query_region(window, tid);
  const size_t qlen = (size_t)(window.qend - window.qstart);
  if (re > query.qual.size() || rs > re) ref_panic("query qual range out of bounds");
  if (qbuffer.size() < qlen) ref_panic("qbuffer too small");
  const bool fwd = window.overlap->strand == Strand::Forward;
  decode(query.seq, rs, re, !fwd, qbuffer.data());
  // iterator state: k-th (base, qual)
  size_t qk = 0;
  auto next_bq = [&](uint8_t& b, uint8_t& q) {
    if (qk >= qlen) ref_panic("Base and its quality should be present.");
    if (fwd) {
      b = qbuffer[qk];
      q = query.qual[rs + qk];
    } else {
      b = base_lower(qbuffer[qk]);
      q = query.qual[re - 1 - qk];
    }
    qk++;
  };

  check_slice(cigar_full, window);
  CigarIter it((const uint8_t*)cigar_full.data() + window.cigar_start_idx,
               window.cigar_end_idx - window.cigar_start_idx);
  const uint8_t gap = fwd ? '*' : '#';
  const size_t L = bases.rows;
  for (size_t r = 0; r < L; r++) bases.at(r, col) = gap;

  size_t tpos = offset;
  if (offset > max_ins.size()) ref_panic("max_ins[..offset] out of range");
  size_t idx = offset;
  for (size_t k = 0; k < offset; k++) idx += max_ins[k];
  if (idx > 0) {
    if (idx > L) ref_panic("slice_mut(..idx) out of bounds");
    for (size_t r = 0; r < idx; r++) bases.at(r, col) = '.';
  }

  CigarItem ci;
  while (it.next(ci)) {
    const size_t l = effective_len(window, ci);
    switch (ci.op.op) {
      case Op::Match:
      case Op::Mismatch:
        for (size_t i = 0; i < l; i++) {
          uint8_t b, q;
          next_bq(b, q);
          if (idx >= L) ref_panic("bases[idx] out of bounds");
          bases.at(idx, col) = b;
          quals.at(idx, col) = q;
          if (tpos + i >= max_ins.size()) ref_panic("max_ins[tpos + i] out of bounds");
          idx += 1 + max_ins[tpos + i];
        }
        tpos += l;
        break;
      case Op::Deletion:
        for (size_t i = 0; i < l; i++) {
          if (tpos + i >= max_ins.size()) ref_panic("max_ins[tpos + i] out of bounds");
          idx += 1 + max_ins[tpos + i];
        }
        tpos += l;
        break;
      case Op::Insertion: {
        if (tpos == 0 || tpos - 1 >= max_ins.size()) ref_panic("max_ins[tpos - 1] out of bounds");
        const size_t mi = max_ins[tpos - 1];
        if (idx < mi) ref_panic("attempt to subtract with overflow");
        idx -= mi;
        for (size_t i = 0; i < l; i++) {
          uint8_t b, q;
          next_bq(b, q);
          if (idx + i >= L) ref_panic("bases[idx + i] out of bounds");
          bases.at(idx + i, col) = b;
          quals.at(idx + i, col) = q;
        }
        idx += mi;
        break;
      }
    }
  }
  if (idx < L)
    for (size_t r = idx; r < L; r++) bases.at(r, col) = '.';
}

// features.rs:239-266
inline void write_target_for_window(size_t tstart, const HAECRecord& target,
                                    const std::vector<uint16_t>& max_ins, Mat& bases, Mat& quals,
                                    size_t window_length, const std::vector<uint8_t>& tbuffer) {
  for (size_t r = 0; r < bases.rows; r++) bases.at(r, 0) = '*';
  size_t tpos = 0;
  for (size_t i = 0; i < window_length; i++) {
    bases.at(tpos, 0) = tbuffer[tstart + i];
    quals.at(tpos, 0) = target.qual[tstart + i];
    tpos += 1 + max_ins[i];
  }
}

// features.rs:315-324
inline bool overlap_window_filter(const uint8_t* cigar, size_t n) {
  CigarIter it(cigar, n);
  CigarItem ci;
  bool long_indel = false;
  while (it.next(ci)) {
    if ((ci.op.op == Op::Insertion || ci.op.op == Op::Deletion) && ci.op.len > 50) {
      long_indel = true;
      break;  // Iterator::any short-circuits
    }
  }
  return !long_indel;
}

// features.rs:585-679
inline float calculate_accuracy(const OverlapWindow& window, const std::string& cigar,
                                const uint8_t* tseq, size_t tlen, const uint8_t* qseq,
                                size_t qlen) {
  size_t tpos = 0, qpos = 0;
  int32_t m = 0, s = 0, i = 0, d = 0;  // Rust infers i32 for m,s; usize-added i,d -> all usize
  size_t mi = 0, md = 0;
  check_slice(cigar, window);
  CigarIter it((const uint8_t*)cigar.data() + window.cigar_start_idx,
               window.cigar_end_idx - window.cigar_start_idx);
  CigarItem ci;
  const size_t S = window.cigar_end_idx - window.cigar_start_idx;
  while (it.next(ci)) {
    size_t len;
    if (ci.start == 0 && ci.end == S) {
      if (!(window.cigar_end_offset > window.cigar_start_offset)) ref_panic("accuracy: end<=start offset");
      len = window.cigar_end_offset - window.cigar_start_offset;
    } else if (ci.start == 0) {
      if (!(ci.op.len > window.cigar_start_offset)) ref_panic("accuracy: op len <= start offset");
      len = ci.op.len - window.cigar_start_offset;
    } else if (ci.end == S) {
      len = window.cigar_end_offset;
    } else {
      len = ci.op.len;
    }
    if (!(len > 0)) ref_panic("Operation length cannot be 0");
    if (ci.op.op != Op::Insertion && !(tpos + len <= tlen)) ref_panic("accuracy: tseq overrun");
    if (ci.op.op != Op::Deletion && !(qpos + len <= qlen)) ref_panic("accuracy: qseq overrun");
    switch (ci.op.op) {
      case Op::Match:
        for (size_t j = 0; j < len; j++) {
          if (tseq[tpos + j] == qseq[qpos + j]) m++; else s++;
        }
        tpos += len;
        qpos += len;
        break;
      case Op::Mismatch: ref_panic("unreachable");
      case Op::Insertion: mi += len; qpos += len; break;
      case Op::Deletion: md += len; tpos += len; break;
    }
  }
  (void)i; (void)d;
  const size_t denom = (size_t)m + (size_t)s + mi + md;
  return (float)(size_t)m / (float)denom;  // (m as f32) / ((m+s+i+d) as f32)
}

// features.rs:681-722
inline std::vector<SupportedPos> get_supported(const Mat& bases) {
  std::vector<SupportedPos> out;
  int16_t tpos = -1;
  uint8_t ins = 0;
  const size_t thresh = (size_t)((double)bases.cols * 0.1);  // :712
  for (size_t r = 0; r < bases.rows; r++) {
    if (bases.at(r, 0) == '*') {
      ins = (uint8_t)(ins + 1);
    } else {
      tpos = (int16_t)(tpos + 1);
      ins = 0;
    }
    size_t cnt[5] = {0, 0, 0, 0, 0};  // A C G T *
    for (size_t c = 0; c < bases.cols; c++) {
      uint8_t b = bases.at(r, c);
      if (b == '.') continue;
      if (b >= 128) ref_panic("BASE_FORWARD index out of bounds");
      switch (base_forward(b)) {
        case 'A': cnt[0]++; break;
        case 'C': cnt[1]++; break;
        case 'G': cnt[2]++; break;
        case 'T': cnt[3]++; break;
        case '*': cnt[4]++; break;
        default: ref_panic("called `Option::unwrap()` on a `None` value (counter)");
      }
    }
    uint8_t n_supported = 0;
    for (int k = 0; k < 5; k++)
      if (cnt[k] >= thresh) n_supported++;
    if (n_supported >= 2) out.push_back({(uint16_t)tpos, ins});
  }
  return out;
}

inline uint8_t ascii_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

// One window as handed to FeaturesOutput::update (features.rs:571-579).
struct WindowFeatures {
  uint32_t rid;
  uint16_t wid;
  Mat bases, quals;  // [L', 31] ASCII
  std::vector<SupportedPos> supported;
  std::vector<uint32_t> qids;  // ranked overlap read ids ("ids", all of them, :569)
  uint16_t n_wids;
  // --- intermediate state (for stage-by-stage debugging of the HIP path; not reference API)
  std::vector<uint32_t> p1_qids;            // after filter + accuracy sort (:376-409)
  std::vector<float> p1_acc;                // accuracy per p1_qids entry
  std::vector<uint16_t> max_ins;            // :411
  size_t p1_L = 0;                          // rows of the pass-1 matrix
  std::vector<SupportedPos> p1_supported;   // :438
  std::vector<double> scores;               // per p1_qids entry (:505-510)
  uint8_t n_alns() const { return (uint8_t)std::min(qids.size(), TOP_K_SORT); }  // :877
};

// features.rs:326-583.  Returns the windows in the order `update` is called.
inline std::vector<WindowFeatures> extract_features(uint32_t rid,
                                                    const std::vector<HAECRecord>& reads,
                                                    const std::vector<Alignment>& overlaps,
                                                    uint32_t window_size) {
  if (rid >= reads.size()) ref_panic("reads[rid] out of bounds");
  const HAECRecord& read = reads[rid];
  size_t max_len = 0;
  for (const auto& r : reads) max_len = std::max(max_len, r.seq.len());  // lib.rs:134,168-169
  std::vector<uint8_t> tbuf(max_len), qbuf(max_len);
  decode(read.seq, 0, read.seq.len(), false, tbuf.data());  // :335

  const size_t W = window_size;
  const size_t n_windows = (read.seq.len() + W - 1) / W;  // :338
  Windows windows(n_windows);
  CigarMap cmap;
  for (const auto& aln : overlaps) {  // :343-361
    const uint32_t qid = aln.overlap.return_other_id(rid);
    const bool is_target = aln.overlap.tid == rid;
    extract_windows(windows, &aln.overlap, aln.cigar, 0, 0, is_target, window_size);
    cmap[qid] = &aln.cigar;  // HashMap::insert overwrites
  }

  struct P1 {
    Mat bases, quals;
    std::vector<SupportedPos> supported;
    std::vector<uint32_t> qids;
    std::vector<float> acc;
    std::vector<uint16_t> max_ins;
  };
  std::vector<P1> all(n_windows);

  for (size_t i = 0; i < n_windows; i++) {  // :364-459
    const size_t win_len = (i == n_windows - 1) ? read.seq.len() - i * W : W;
    auto& wv = windows[i];
    // retain(overlap_window_filter) :376-383
    {
      std::vector<OverlapWindow> kept;
      for (const auto& ow : wv) {
        const std::string& c = cigar_of(cmap, ow.overlap->return_other_id(rid));
        check_slice(c, ow);
        if (overlap_window_filter((const uint8_t*)c.data() + ow.cigar_start_idx,
                                  ow.cigar_end_idx - ow.cigar_start_idx))
          kept.push_back(ow);
      }
      wv.swap(kept);
    }
    // sort_by_key(-accuracy), stable :386-409
    std::vector<float> acc(wv.size());
    for (size_t k = 0; k < wv.size(); k++) {
      const auto& ow = wv[k];
      const uint32_t qid = ow.overlap->return_other_id(rid);
      const std::string& c = cigar_of(cmap, qid);
      const size_t tstart = ow.tstart;
      const size_t tend = i * W + win_len;
      if (tstart > tend) ref_panic("tbuf[tstart..tend] slice index order");
      auto [qs, qe] = get_query_region(ow, rid);
      if (qid >= reads.size()) ref_panic("reads[qid] out of bounds");
      if (qe < qs) ref_panic("attempt to subtract with overflow");
      const size_t qlen = (size_t)(qe - qs);
      if (qlen > qbuf.size()) ref_panic("qbuf too small");
      decode(reads[qid].seq, qs, qe, ow.overlap->strand == Strand::Reverse, qbuf.data());
      acc[k] = calculate_accuracy(ow, c, tbuf.data() + tstart, tend - tstart, qbuf.data(), qlen);
    }
    std::vector<size_t> order(wv.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
      // key = OrderedFloat(-acc): ascending; NaN sorts last; -0.0 == 0.0
      const float ka = -acc[a], kb = -acc[b];
      if (std::isnan(ka)) return false;
      if (std::isnan(kb)) return true;
      return ka < kb;
    });
    {
      std::vector<OverlapWindow> sorted;
      std::vector<float> sacc;
      for (size_t k : order) {
        sorted.push_back(wv[k]);
        sacc.push_back(acc[k]);
      }
      wv.swap(sorted);
      acc.swap(sacc);
    }

    auto max_ins = get_max_ins_for_window(wv, cmap, rid, i * W, win_len);  // :411

    // get_features_for_window :268-313
    size_t length = max_ins.size();
    for (auto v : max_ins) length += v;
    const size_t ncols = 1 + std::max(wv.size(), TOP_K_SORT);
    P1& p = all[i];
    p.bases = Mat(length, ncols, '.');
    p.quals = Mat(length, ncols, '!');
    write_target_for_window(i * W, read, max_ins, p.bases, p.quals, win_len, tbuf);
    for (size_t k = 0; k < wv.size(); k++) {
      const auto& ow = wv[k];
      const uint32_t qid = ow.overlap->return_other_id(rid);
      if ((size_t)ow.tstart < i * W) ref_panic("attempt to subtract with overflow");
      get_features_for_ol_window(p.bases, p.quals, k + 1, ow, cigar_of(cmap, qid), reads[qid],
                                 (size_t)ow.tstart - i * W, rid, max_ins, qbuf);
      p.qids.push_back(qid);
    }
    p.acc = acc;
    p.max_ins = max_ins;
    p.supported = get_supported(p.bases);  // :438
  }

  // Ratios :461-500 — keyed by the query read *name* (string), across all windows.
  std::unordered_map<std::string, std::pair<double, double>> ratios;
  for (size_t i = 0; i < n_windows; i++) {
    const P1& p = all[i];
    std::vector<size_t> pos_to_idx;
    for (size_t r = 0; r < p.bases.rows; r++)
      if (p.bases.at(r, 0) != '*') pos_to_idx.push_back(r);
    std::unordered_set<size_t> indices;
    for (const auto& s : p.supported) {
      if (s.pos >= pos_to_idx.size()) ref_panic("pos_to_idx out of bounds");
      indices.insert(pos_to_idx[s.pos] + s.ins);
    }
    for (size_t k = 0; k < p.qids.size(); k++) {
      const std::string& qname = reads[p.qids[k]].id;
      const size_t col = k + 1;
      for (size_t pos = 0; pos < p.bases.rows; pos++) {
        if (!indices.count(pos)) continue;
        const uint8_t t = ascii_upper(p.bases.at(pos, 0));
        const uint8_t q = ascii_upper(p.bases.at(pos, col));
        if (t == '*') continue;
        auto& e = ratios[qname];  // or_insert((0.,0.))
        if (q == t) e.first += 1.; else e.second += 1.;
      }
    }
  }

  std::vector<WindowFeatures> out;
  out.reserve(n_windows);
  for (size_t i = 0; i < n_windows; i++) {  // :502-580
    P1& p = all[i];
    std::vector<double> iden;
    iden.push_back(std::numeric_limits<double>::max());
    std::vector<double> scores;
    for (uint32_t q : p.qids) {
      auto it = ratios.find(reads[q].id);
      double s = 0.;
      if (it != ratios.end()) {
        const double n = it->second.first, d = it->second.second;
        s = n / (n + d) * std::log(n + d + 1.);
      }
      iden.push_back(s);
      scores.push_back(s);
    }
    std::vector<size_t> sr(iden.size());
    for (size_t k = 0; k < sr.size(); k++) sr[k] = k;
    std::stable_sort(sr.begin(), sr.end(), [&](size_t a, size_t b) {
      // sort_by_key(Reverse(OrderedFloat)): descending, NaN (greatest) first
      const double ka = iden[a], kb = iden[b];
      if (std::isnan(kb)) return false;
      if (std::isnan(ka)) return true;
      return ka > kb;
    });

    std::vector<size_t> sel;
    for (size_t k = 0; k < sr.size() && k < TOP_K_SORT + 1; k++) sel.push_back(sr[k]);
    for (size_t k = sr.size(); k < TOP_K_SORT + 1; k++) sel.push_back(k);  // :522-525
    if (sel.size() != TOP_K_SORT + 1) ref_panic("assert_eq new_bases.len()");

    std::vector<size_t> retain;
    for (size_t r = 0; r < p.bases.rows; r++) {  // :531-545
      bool all_gap = true;
      for (size_t c : sel) {
        const uint8_t b = p.bases.at(r, c);
        if (b == '.') continue;
        if (!(b == '*' || b == '#')) { all_gap = false; break; }
      }
      if (!all_gap) retain.push_back(r);
    }
    WindowFeatures wf;
    wf.rid = rid;
    wf.wid = (uint16_t)i;
    wf.n_wids = (uint16_t)n_windows;
    wf.bases = Mat(retain.size(), TOP_K_SORT + 1, 0);
    wf.quals = Mat(retain.size(), TOP_K_SORT + 1, 0);
    for (size_t rr = 0; rr < retain.size(); rr++)
      for (size_t c = 0; c < sel.size(); c++) {
        wf.bases.at(rr, c) = p.bases.at(retain[rr], sel[c]);
        wf.quals.at(rr, c) = p.quals.at(retain[rr], sel[c]);
      }
    wf.supported = get_supported(wf.bases);  // :558
    for (size_t k = 1; k < sr.size(); k++) wf.qids.push_back(p.qids[sr[k] - 1]);  // :569
    wf.p1_qids = p.qids;
    wf.p1_acc = p.acc;
    wf.max_ins = p.max_ins;
    wf.p1_L = p.bases.rows;
    wf.p1_supported = p.supported;
    wf.scores = scores;
    out.push_back(std::move(wf));
  }
  return out;
}

// ---------------------------------------------------------------------------------------
// inference.rs — token encoding, target indices, batching, collate
// ---------------------------------------------------------------------------------------
static const uint8_t BASE_PADDING = 11;     // inference.rs:15
static const uint8_t QUAL_MAX_VAL_U8 = 126;  // inference.rs:17 (pad value for quals)

inline uint8_t bases_map(uint8_t b) {  // inference.rs:23-31 BASES_MAP
  switch (b) {
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
    case '*': return 4;
    case 'a': return 5; case 'c': return 6; case 'g': return 7; case 't': return 8;
    case '#': return 9; case '.': return 10;
    default: return 255;
  }
}

struct ConsensusWindow {  // consensus.rs:22-33
  uint32_t rid;
  uint16_t wid;
  uint8_t n_alns;
  uint16_t n_total_wins;
  Mat bases, quals;  // bases are *encoded* tokens
  std::vector<size_t> indices;
  std::vector<SupportedPos> supported;
  bool has_logits = false;
  std::vector<float> info_logits;
  std::vector<std::array<float, 5>> bases_logits;
};

// inference.rs:255-268
inline std::vector<size_t> get_target_indices(const Mat& bases) {
  std::vector<size_t> out;
  for (size_t r = 0; r < bases.rows; r++)
    if (bases.at(r, 0) != bases_map('*')) out.push_back(r);
  return out;
}

struct InferenceBatch {  // inference.rs:33-39 (host tensors)
  std::vector<uint32_t> wids;  // index into the flush's window list
  size_t B = 0, L = 0, R = 0;
  std::vector<uint8_t> bases, quals;       // [B, L, R] padded with 11 / 126
  std::vector<int32_t> lens;               // [B]
  std::vector<std::vector<int32_t>> indices;
};

// inference.rs:73-145
inline InferenceBatch collate(const std::vector<std::pair<uint32_t, const ConsensusWindow*>>& batch) {
  InferenceBatch ib;
  size_t length = 0;
  for (auto& [w, f] : batch) length = std::max(length, f->bases.rows);
  ib.B = batch.size();
  ib.L = length;
  ib.R = batch[0].second->bases.cols;
  ib.bases.assign(ib.B * ib.L * ib.R, BASE_PADDING);
  ib.quals.assign(ib.B * ib.L * ib.R, QUAL_MAX_VAL_U8);
  for (size_t idx = 0; idx < batch.size(); idx++) {
    const auto* f = batch[idx].second;
    ib.wids.push_back(batch[idx].first);
    const size_t l = f->bases.rows;
    std::memcpy(&ib.bases[idx * ib.L * ib.R], f->bases.v.data(), l * ib.R);
    std::memcpy(&ib.quals[idx * ib.L * ib.R], f->quals.v.data(), l * ib.R);
    ib.lens.push_back((int32_t)f->supported.size());
    std::vector<int32_t> tidx;
    for (const auto& sp : f->supported) {
      if (sp.pos >= f->indices.size()) ref_panic("f.indices[sp.pos] out of bounds");
      tidx.push_back((int32_t)(f->indices[sp.pos] + sp.ins));
    }
    ib.indices.push_back(std::move(tidx));
  }
  return ib;
}

struct InferenceData {  // inference.rs:59-62
  std::vector<ConsensusWindow> consensus_data;
  std::vector<InferenceBatch> batches;
};

// inference.rs:214-253
inline InferenceData prepare_examples(const std::vector<WindowFeatures>& features, size_t batch_size) {
  InferenceData d;
  for (const auto& ex : features) {
    ConsensusWindow cw;
    cw.rid = ex.rid;
    cw.wid = ex.wid;
    cw.n_alns = ex.n_alns();
    cw.n_total_wins = ex.n_wids;
    cw.bases = ex.bases;
    for (auto& b : cw.bases.v) {
      if (b >= 128) ref_panic("BASES_MAP index out of bounds");
      b = bases_map(b);
    }
    cw.quals = ex.quals;
    cw.indices = get_target_indices(cw.bases);
    cw.supported = ex.supported;
    d.consensus_data.push_back(std::move(cw));
  }
  std::vector<std::pair<uint32_t, const ConsensusWindow*>> cur;
  for (uint32_t k = 0; k < d.consensus_data.size(); k++) {
    if (d.consensus_data[k].supported.empty()) continue;
    cur.push_back({k, &d.consensus_data[k]});
    if (cur.size() == batch_size) {
      d.batches.push_back(collate(cur));
      cur.clear();
    }
  }
  if (!cur.empty()) d.batches.push_back(collate(cur));
  return d;
}

// inference.rs:16-21,153 — qual normalisation applied on device in the reference.
inline float normalise_qual(uint8_t q) {
  const double QUAL_RANGE_DIFF = (double)(126.f - 33.f);
  const double QUAL_SCALE = 2. / QUAL_RANGE_DIFF;
  const double QUAL_OFFSET = 2. * (double)33.f / QUAL_RANGE_DIFF + 1.;
  // tch: `f64 * Tensor(f32) - f64` -> scalar ops on an f32 tensor are computed in f32.
  return (float)QUAL_SCALE * (float)q - (float)QUAL_OFFSET;
}

// ---------------------------------------------------------------------------------------
// consensus.rs
// ---------------------------------------------------------------------------------------
static const uint8_t BASES_UPPER[10] = {'A', 'C', 'G', 'T', '*', 'A', 'C', 'G', 'T', '*'};  // :18
static const size_t BASES_UPPER_COUNTER[10] = {0, 1, 2, 3, 4, 0, 1, 2, 3, 4};               // :19

// consensus.rs:86-227.  `data` must be one read's windows sorted by wid (:251).
// Returns false for `None`.
inline bool consensus(const std::vector<ConsensusWindow>& data,
                      std::vector<std::vector<uint8_t>>& corrected_seqs) {
  corrected_seqs.clear();
  std::vector<uint8_t> corrected;
  bool any = false;
  size_t wid_st = 0, wid_en = 0;
  for (size_t idx = 0; idx < data.size(); idx++)
    if (data[idx].n_alns > 1) {
      if (!any) wid_st = idx;
      wid_en = idx + 1;
      any = true;
    }
  if (!any) return false;

  for (size_t w = wid_st; w < wid_en; w++) {
    const ConsensusWindow& window = data[w];
    if (window.n_alns < 2) {
      if (!corrected.empty()) {
        corrected_seqs.push_back(corrected);
        corrected.clear();
      }
      continue;
    }
    const size_t n_rows = (size_t)window.n_alns + 1;
    if (n_rows > window.bases.cols) ref_panic("slice ..n_rows out of bounds");
    std::map<std::pair<uint16_t, uint8_t>, size_t> maybe_info;  // SupportedPos -> logits row
    if (!window.supported.empty()) {
      if (!window.has_logits) ref_panic("called `Option::unwrap()` on a `None` value (logits)");
      const size_t n = std::min({window.supported.size(), window.info_logits.size(),
                                 window.bases_logits.size()});  // zip stops at the shortest
      for (size_t k = 0; k < n; k++)
        maybe_info[{window.supported[k].pos, window.supported[k].ins}] = k;  // later dup wins
    }
    int32_t pos = -1;
    uint8_t ins = 0;
    for (size_t r = 0; r < window.bases.rows; r++) {
      if (window.bases.at(r, 0) == bases_map('*')) {
        ins = (uint8_t)(ins + 1);
      } else {
        pos += 1;
        ins = 0;
      }
      auto it = maybe_info.find({(uint16_t)pos, ins});
      if (it != maybe_info.end()) {
        const auto& b = window.bases_logits[it->second];
        // max_by_key(OrderedFloat) -> last maximum wins; NaN is greatest (:136-141)
        size_t argmax = 0;
        for (size_t k = 1; k < 5; k++) {
          const float v = b[k], cur = b[argmax];
          bool ge;
          if (std::isnan(v)) ge = true;
          else if (std::isnan(cur)) ge = false;
          else ge = v >= cur;
          if (ge) argmax = k;
        }
        static const uint8_t DEC[5] = {'A', 'C', 'G', 'T', '*'};
        const uint8_t base = DEC[argmax];
        if (base != '*') corrected.push_back(base);
      } else {
        uint8_t counts[5] = {0, 0, 0, 0, 0};
        for (size_t c = 0; c < n_rows; c++) {
          const uint8_t b = window.bases.at(r, c);
          if (b != bases_map('.')) {
            if (b >= 10) ref_panic("BASES_UPPER_COUNTER index out of bounds");
            counts[BASES_UPPER_COUNTER[b]] = (uint8_t)(counts[BASES_UPPER_COUNTER[b]] + 1);
          }
        }
        // sorted_by_key(Reverse(count)) is stable: ties keep A,C,G,T,* order (:186-193)
        size_t ord[5] = {0, 1, 2, 3, 4};
        std::stable_sort(ord, ord + 5, [&](size_t a, size_t b) { return counts[a] > counts[b]; });
        const uint8_t mc0c = counts[ord[0]], mc1c = counts[ord[1]];
        const uint8_t mc0b = BASES_UPPER[ord[0]], mc1b = BASES_UPPER[ord[1]];
        const uint8_t t0 = window.bases.at(r, 0);
        if (t0 >= 10) ref_panic("BASES_UPPER index out of bounds");
        const uint8_t tbase = BASES_UPPER[t0];
        const uint8_t base =
            (mc0c < 2 || (mc0c == mc1c && (mc0b == tbase || mc1b == tbase))) ? tbase : mc0b;
        if (base != '*') corrected.push_back(base);
      }
    }
  }
  if (!corrected.empty()) corrected_seqs.push_back(corrected);
  return true;
}

// lib.rs:282-317 — FASTA record(s) for one read.
inline std::string write_read(const HAECRecord& read, const std::vector<std::vector<uint8_t>>& seqs) {
  std::string out;
  auto write_sequence = [&](const std::vector<uint8_t>& seq, int idx) {
    out += ">";
    out += read.id;
    if (idx >= 0) out += ":" + std::to_string(idx) + " ";
    else out += " ";
    if (read.description) out += *read.description;
    out += "\n";
    out.append((const char*)seq.data(), seq.size());
    out += "\n";
  };
  if (seqs.size() == 1) write_sequence(seqs[0], -1);
  else
    for (size_t i = 0; i < seqs.size(); i++) write_sequence(seqs[i], (int)i);
  return out;
}

}  // namespace oracle
