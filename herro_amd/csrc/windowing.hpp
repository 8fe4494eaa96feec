// windowing.hpp — host half of the boundary: PAF alignments -> per-window overlap descriptors.
//
// Product code (not the oracle).  Does what the front of the reference's extract_features does
// on the CPU before any pileup work (features.rs:337-361): cut every alignment into the target
// windows it spans (`extract_windows`, windowing.rs:44-273).  Unlike the reference, the CIGAR is
// first converted to a binary op stream ((len<<2)|type, u32) so that slices are op-index ranges
// and the GPU never parses ASCII.
//
// Inputs on which the reference panics are reported as errors (HERRO_E_REFERENCE_PANIC) instead
// of aborting; a few inputs the reference tolerates but minimap2 never produces are rejected as
// HERRO_E_UNSUPPORTED (documented in DESIGN.md).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/herro_amd.h"
#include "pileup_core.h"

namespace herro {

struct HostOw {  // OverlapWindow (windowing.rs:7-16) with op-index slice
  uint32_t win;  // window index within the target read
  uint32_t tstart, qstart, qend;
  uint32_t op_lo, op_hi;  // slice [op_lo, op_hi) of the alignment's ops
  uint32_t start_off, end_off;
  uint32_t st = 0, sq = 0, si = 0;  // target / query / insertion bases of the UNTRIMMED slice (window_cuts only)
  uint32_t op_first = 0, op_last = 0;  // the ops at op_lo and op_hi - 1 (window_cuts only; 0 past the end)
};

struct BuildError {
  int code = HERRO_OK;
  std::string msg;
};

// ASCII "\d+[MID]" -> ops.  Mirrors CigarIter's panics (aligners.rs:252-293).
inline bool parse_cigar(const uint8_t* s, uint32_t n, std::vector<uint32_t>& ops, BuildError& e) {
  uint32_t i = 0;
  while (i < n) {
    uint64_t len = 0;
    const uint32_t st = i;
    while (i < n && s[i] >= '0' && s[i] <= '9') {
      len = len * 10 + (uint64_t)(s[i] - '0');
      if (len > 0x3fffffffull) { e = {HERRO_E_INVALID, "cigar op length overflows 30 bits"}; return false; }
      i++;
    }
    if (i >= n) { e = {HERRO_E_REFERENCE_PANIC, "cigar ends inside an op (CigarIter index out of bounds)"}; return false; }
    if (i == st || len == 0) { e = {HERRO_E_REFERENCE_PANIC, "Length has to be longer than 0"}; return false; }
    uint32_t ty;
    switch (s[i]) {
      case 'M': ty = OP_M; break;
      case 'I': ty = OP_I; break;
      case 'D': ty = OP_D; break;
      default:
        e = {HERRO_E_REFERENCE_PANIC, std::string("Unexpected cigar operation ") + (char)s[i]};
        return false;
    }
    ops.push_back(((uint32_t)len << 2) | ty);
    i++;
  }
  return true;
}

// ---- herro_job_create's hot loop (~5000 text ops per 4096-bp window of 32 overlaps) -----------------------------
// Everything that needs every op happens in ONE sweep: the binary ops are written, the running target / query /
// insertion totals are kept, and the few ops that reach a window boundary ("cuts", ~1 per window the alignment spans)
// are noted together with the totals in front of them.  The windowing proper (window_cuts below) then only visits the
// cuts.  The text is decoded in two branch-light stages so that consecutive ops do not wait for each other:
//   A  letter positions: bit 6 separates 'A'..'Z' from '0'..'9'; eight bytes per load, one ctz per letter;
//   B  per op, independent of its neighbours: the eight bytes ending in front of the letter, the digit bytes masked
//      out, SWAR decimal conversion (three multiplies), op type from a 256-entry table.
// Anything unusual — a length of 8+ digits, a byte that is neither digit nor M/I/D, text shorter than 8 bytes —
// sends the whole CIGAR through the byte-wise path, which also produces the reference's panic messages.
// A cut carries everything the windowing reads of the op array: the op itself and its two successors (0 past the end).
// The scan may therefore run somewhere else than the windowing — on the GPU, cigar_dev.hip — and hand over cuts only.
struct Cut { uint32_t k, t, q, ins, o0, o1, o2; };   // op index; absolute target position, query and insertion bases consumed before it; ops k, k + 1, k + 2
struct CigarScan {
  uint32_t n_ops = 0, t_end = 0, q_end = 0, ins_end = 0;   // totals after the last op (t_end absolute)
  uint32_t op0 = 0, opn = 0;                               // first and last op
  std::vector<Cut> cuts;
  std::vector<uint32_t> ins_pairs;   // every k with ops k and k + 1 both insertions (minimap2 never emits any)
  std::vector<uint32_t> pos;         // stage A scratch: letter positions
  bool ins_pair_in(uint32_t lo, uint32_t hi) const {  // a pair inside the slice [lo, hi)
    for (uint32_t k : ins_pairs) if (k >= lo && k + 1 < hi) return true;
    return false;
  }
};
namespace detail {
struct OpLut {
  uint8_t t[256];
  constexpr OpLut() : t() {
    for (int i = 0; i < 256; i++) t[i] = 0xff;
    t[(int)'M'] = OP_M; t[(int)'I'] = OP_I; t[(int)'D'] = OP_D;
  }
};
static constexpr OpLut kOpLut{};

// running totals + cut detection, shared by both decoders
struct ScanState {
  uint32_t k = 0, t, q = 0, ins = 0, prev_i = 0, W;
  uint64_t wend;
  uint32_t* ops;
  CigarScan& S;
  ScanState(uint32_t tstart, uint32_t W_, uint32_t* ops_, CigarScan& S_) : t(tstart), W(W_), wend(((uint64_t)(tstart / W_) + 1) * W_), ops(ops_), S(S_) {}
  inline void op(uint32_t len, uint32_t ty) {
    ops[k] = (len << 2) | ty;
    const uint32_t mi = 0u - (uint32_t)(ty == OP_I), md = 0u - (uint32_t)(ty == OP_D);   // all-ones masks: insertion / deletion
    const uint32_t tnew = t + (len & ~mi);
    if (__builtin_expect((uint64_t)tnew >= wend, 0)) {   // ~once per window (never true for an insertion: t < wend always holds)
      S.cuts.push_back(Cut{k, t, q, ins, ops[k], 0, 0});
      wend = ((uint64_t)(tnew / W) + 1) * W;
    }
    if (__builtin_expect(mi & prev_i, 0)) S.ins_pairs.push_back(k - 1);
    t = tnew;
    q += len & ~md;
    ins += len & mi;
    prev_i = mi;
    k++;
  }
  void finish() {
    S.n_ops = k; S.t_end = t; S.q_end = q; S.ins_end = ins;
    S.op0 = k ? ops[0] : 0u; S.opn = k ? ops[k - 1] : 0u;
    for (Cut& c : S.cuts) { c.o1 = c.k + 1 < k ? ops[c.k + 1] : 0u; c.o2 = c.k + 2 < k ? ops[c.k + 2] : 0u; }
  }
};

// CigarIter's behaviour byte by byte (aligners.rs:252-293)
inline bool scan_cigar_bytewise(const uint8_t* s, uint32_t n, ScanState& st, BuildError& e) {
  uint32_t i = 0;
  while (i < n) {
    uint64_t len = 0;
    const uint32_t i0 = i;
    uint32_t c;
    while (i < n && (c = (uint32_t)s[i] - (uint32_t)'0') < 10u) {
      len = len * 10 + c;
      if (len > 0x3fffffffull) { e = {HERRO_E_INVALID, "cigar op length overflows 30 bits"}; return false; }
      i++;
    }
    if (i >= n) { e = {HERRO_E_REFERENCE_PANIC, "cigar ends inside an op (CigarIter index out of bounds)"}; return false; }
    if (i == i0 || len == 0) { e = {HERRO_E_REFERENCE_PANIC, "Length has to be longer than 0"}; return false; }
    const uint32_t ty = kOpLut.t[s[i]];
    if (ty == 0xff) { e = {HERRO_E_REFERENCE_PANIC, std::string("Unexpected cigar operation ") + (char)s[i]}; return false; }
    st.op((uint32_t)len, ty);
    i++;
  }
  st.finish();
  return true;
}
}  // namespace detail

// `ops` must have room for cigar_len / 2 + 1 entries.  tstart / W: where the alignment starts on the target and the
// window length (cuts are ops whose target end reaches the next multiple of W).
inline bool scan_cigar(const uint8_t* s, uint32_t n, uint32_t tstart, uint32_t W, uint32_t* ops, CigarScan& S, BuildError& e) {
  S.cuts.clear(); S.ins_pairs.clear();
  bool fast = n >= 8;
  uint32_t n_let = 0;
  if (fast) {  // ---- stage A
    if (S.pos.size() < (size_t)n + 8) S.pos.resize((size_t)n + 8);
    uint32_t* pos = S.pos.data();
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {
      uint64_t x;
      std::memcpy(&x, s + i, 8);
      uint64_t m = x & 0x4040404040404040ull;
      while (m) { pos[n_let++] = i + ((uint32_t)__builtin_ctzll(m) >> 3); m &= m - 1; }
    }
    for (; i < n; i++) if (s[i] & 0x40) pos[n_let++] = i;
    fast = n_let <= n / 2 && n_let > 0 && pos[n_let - 1] == n - 1;   // more letters than room / text ending in digits: byte-wise reports it
  }
  if (fast) {  // ---- stage B
    detail::ScanState st(tstart, W, ops, S);
    const uint32_t* pos = S.pos.data();
    uint64_t head;
    std::memcpy(&head, s, 8);
    uint32_t start = 0;
    for (uint32_t j = 0; j < n_let; j++) {
      const uint32_t end = pos[j], nd = end - start;
      uint32_t len;
      if (__builtin_expect(nd - 1u <= 3u && end >= 4, 1)) {   // 1..4 digits: the four bytes in front of the letter
        uint32_t x;
        std::memcpy(&x, s + end - 4, 4);
        const uint32_t hm = ~0u << (32 - 8 * nd);             // its digit bytes (the top nd)
        const uint32_t dg = (x & hm) | (0x30303030u & ~hm);
        if ((dg & 0x80808080u) | (((dg + 0x46464646u) | (dg - 0x30303030u)) & 0x80808080u)) { fast = false; break; }   // not all digits
        uint32_t v = x & 0x0f0f0f0fu & hm;                    // d0 d1 d2 d3, d0 (lowest byte) most significant, zeros in front
        v = (v * 10 + (v >> 8)) & 0x00ff00ffu;                // 10 d0 + d1 | 10 d2 + d3
        len = (v & 0xff) * 100 + (v >> 16);
      } else {
        if (nd - 1u > 6u) { fast = false; break; }   // no digits, or 8+ of them
        uint64_t x;
        if (end >= 8) std::memcpy(&x, s + end - 8, 8); else x = head << (8 * (8 - end));   // the eight bytes in front of the letter
        const uint64_t hm = ~0ull << (64 - 8 * nd);
        const uint64_t dg = (x & hm) | (0x3030303030303030ull & ~hm);
        if ((dg & 0x8080808080808080ull) | (((dg + 0x4646464646464646ull) | (dg - 0x3030303030303030ull)) & 0x8080808080808080ull)) { fast = false; break; }
        uint64_t v = x & 0x0f0f0f0f0f0f0f0full & hm;
        v = (v * 2561) >> 8;
        v = ((v & 0x00ff00ff00ff00ffull) * 6553601) >> 16;
        v = ((v & 0x0000ffff0000ffffull) * 42949672960001ull) >> 32;
        len = (uint32_t)v;
      }
      const uint32_t ty = detail::kOpLut.t[s[end]];
      if (ty == 0xff || len == 0) { fast = false; break; }
      st.op(len, ty);
      start = end + 1;
    }
    if (fast) { st.finish(); return true; }
    S.cuts.clear(); S.ins_pairs.clear();
  }
  detail::ScanState st(tstart, W, ops, S);
  return detail::scan_cigar_bytewise(s, n, st, e);
}

// extract_windows (windowing.rs:44-273) for the is_target == true case, on binary ops.
// n_windows = windows of the target read.  Appends to `out` in emission order.
inline bool window_alignment(const std::vector<uint32_t>& ops, const herro_alignment& a, uint32_t W,
                             uint32_t n_windows, std::vector<HostOw>& out, BuildError& e) {
  if (a.tend < a.tstart || a.qend < a.qstart) { e = {HERRO_E_INVALID, "alignment with end < start"}; return false; }
  if ((a.tend - a.tstart) < W || (a.qend - a.qstart) < W) return true;  // :53-57
  const uint32_t zthr = (uint32_t)(0.1f * (float)W);                     // :65
  if (a.tlen < zthr) { e = {HERRO_E_INVALID, "tlen shorter than 0.1*window"}; return false; }
  const uint32_t nthr = a.tlen - zthr;
  const uint32_t first_window = a.tstart < zthr ? 0 : (a.tstart + W - 1) / W;  // :75-79
  const uint32_t last_window = a.tend > nthr ? (a.tend - 1) / W + 1 : a.tend / W;  // :81-85
  if (last_window <= first_window) return true;  // :106 (last >= first always holds here)

  bool started = false;
  uint32_t w_t = 0, w_q = 0, w_op = 0, w_off = 0;  // pending window start
  uint32_t tpos = a.tstart, qpos = 0;
  if (tpos % W == 0 || a.tstart < zthr) {  // :120-125
    started = true; w_t = tpos; w_q = 0; w_op = 0; w_off = 0;
  }
  auto emit = [&](uint32_t widx_plus1, uint32_t qend, uint32_t op_hi, uint32_t end_off) -> bool {
    if (widx_plus1 == 0 || widx_plus1 - 1 >= n_windows) {
      e = {HERRO_E_REFERENCE_PANIC, "alignment reaches past the target's windows (windows[] index out of bounds)"};
      return false;
    }
    out.push_back(HostOw{widx_plus1 - 1, w_t, w_q, qend, w_op, op_hi, w_off, end_off});
    return true;
  };

  const uint32_t n = (uint32_t)ops.size();
  uint64_t wend = ((uint64_t)(tpos / W) + 1) * W;  // first window boundary above tpos: ops that stay below it need no division
  for (uint32_t k = 0; k < n; k++) {
    const uint32_t ty = op_type(ops[k]), l = op_len(ops[k]);
    if (ty == OP_I) { qpos += l; continue; }  // :132-135
    const bool is_m = ty == OP_M;
    const uint32_t tnew = tpos + l, qnew = is_m ? qpos + l : qpos;
    if ((uint64_t)tnew < wend) { tpos = tnew; qpos = qnew; continue; }  // same window, :142-147
    const uint32_t cur_w = tpos / W, new_w = tnew / W;
    wend = ((uint64_t)new_w + 1) * W;
    for (uint32_t i = 1; i < new_w - cur_w; i++) {  // windows fully inside this op :150-195
      const uint32_t off = (cur_w + i) * W - tpos;
      const uint32_t qcut = is_m ? qpos + off : qpos;
      if (started && !emit(cur_w + i, qcut, k + 1, off)) return false;
      started = true; w_t = tpos + off; w_q = qcut; w_op = k; w_off = off;
    }
    const uint32_t off = new_w * W - tpos;  // :198
    uint32_t qend = is_m ? qpos + off : qpos;
    uint32_t op_hi, end_off, next_op, next_off;
    if (tnew == new_w * W) {  // op ends exactly on the boundary :210-223
      if (k + 1 < n && op_type(ops[k + 1]) == OP_I) {  // trailing insertion stays with this window
        qend += op_len(ops[k + 1]);
        op_hi = k + 2;
        end_off = op_len(ops[k + 1]);
      } else {
        op_hi = k + 1;
        end_off = l;
      }
      next_op = op_hi;
      next_off = 0;
    } else {  // :224-230
      op_hi = k + 1;
      end_off = off;
      next_op = k;
      next_off = off;
    }
    if (started && !emit(new_w, qend, op_hi, end_off)) return false;
    started = true; w_t = tpos + off; w_q = qend; w_op = next_op; w_off = next_off;
    tpos = tnew;
    qpos = qnew;
  }
  if (tpos > nthr && tpos % W != 0) {  // tail window :261-272
    if (!started) { e = {HERRO_E_REFERENCE_PANIC, "tail window without a start (Option::unwrap on None)"}; return false; }
    if (n == 0) { e = {HERRO_E_REFERENCE_PANIC, "empty cigar"}; return false; }
    if (!emit(last_window, qpos, n, op_len(ops[n - 1]))) return false;
  }
  return true;
}

// The same windowing from the output of scan_cigar: only ops that reach a window boundary are visited (every other op
// just advances the totals, which the scan has already done).  Produces exactly what window_alignment produces, plus
// the untrimmed target / query / insertion bases of every slice (differences of the running totals at op_lo / op_hi),
// so that nothing downstream walks a slice again.
inline bool window_cuts(const CigarScan& S, const herro_alignment& a, uint32_t W,
                        uint32_t n_windows, std::vector<HostOw>& out, BuildError& e) {
  if (a.tend < a.tstart || a.qend < a.qstart) { e = {HERRO_E_INVALID, "alignment with end < start"}; return false; }
  if ((a.tend - a.tstart) < W || (a.qend - a.qstart) < W) return true;  // :53-57
  const uint32_t zthr = (uint32_t)(0.1f * (float)W);                     // :65
  if (a.tlen < zthr) { e = {HERRO_E_INVALID, "tlen shorter than 0.1*window"}; return false; }
  const uint32_t nthr = a.tlen - zthr;
  const uint32_t first_window = a.tstart < zthr ? 0 : (a.tstart + W - 1) / W;  // :75-79
  const uint32_t last_window = a.tend > nthr ? (a.tend - 1) / W + 1 : a.tend / W;  // :81-85
  if (last_window <= first_window) return true;  // :106

  struct P { uint32_t t, q, i; };  // running totals in front of an op index
  bool started = false;
  uint32_t w_t = 0, w_q = 0, w_op = 0, w_off = 0;  // pending window start
  P w_p{a.tstart, 0, 0};
  uint32_t w_opv = S.op0;  // the op at w_op
  if (a.tstart % W == 0 || a.tstart < zthr) {  // :120-125
    started = true; w_t = a.tstart; w_q = 0; w_op = 0; w_off = 0;
  }
  auto emit = [&](uint32_t widx_plus1, uint32_t qend, uint32_t op_hi, uint32_t end_off, const P& hi, uint32_t last_opv) -> bool {
    if (widx_plus1 == 0 || widx_plus1 - 1 >= n_windows) {
      e = {HERRO_E_REFERENCE_PANIC, "alignment reaches past the target's windows (windows[] index out of bounds)"};
      return false;
    }
    HostOw h{widx_plus1 - 1, w_t, w_q, qend, w_op, op_hi, w_off, end_off};
    h.st = hi.t - w_p.t; h.sq = hi.q - w_p.q; h.si = hi.i - w_p.i;
    h.op_first = w_opv; h.op_last = last_opv;
    out.push_back(h);
    return true;
  };
  const uint32_t n = S.n_ops;
  for (const Cut& c : S.cuts) {
    const uint32_t k = c.k, tpos = c.t, qpos = c.q;
    const uint32_t ty = op_type(c.o0), l = op_len(c.o0);
    const bool is_m = ty == OP_M;
    const uint32_t tnew = tpos + l, qnew = is_m ? qpos + l : qpos;
    const P p_k{tpos, qpos, c.ins}, p_k1{tnew, qnew, c.ins};
    const uint32_t cur_w = tpos / W, new_w = tnew / W;
    for (uint32_t i = 1; i < new_w - cur_w; i++) {  // windows fully inside this op :150-195
      const uint32_t off = (cur_w + i) * W - tpos;
      const uint32_t qcut = is_m ? qpos + off : qpos;
      if (started && !emit(cur_w + i, qcut, k + 1, off, p_k1, c.o0)) return false;
      started = true; w_t = tpos + off; w_q = qcut; w_op = k; w_off = off; w_p = p_k; w_opv = c.o0;
    }
    const uint32_t off = new_w * W - tpos;  // :198
    uint32_t qend = is_m ? qpos + off : qpos;
    uint32_t op_hi, end_off, next_op, next_off;
    P p_hi = p_k1, p_next = p_k;
    uint32_t last_opv = c.o0, next_opv = c.o0;
    if (tnew == new_w * W) {  // op ends exactly on the boundary :210-223
      if (k + 1 < n && op_type(c.o1) == OP_I) {  // trailing insertion stays with this window
        const uint32_t li = op_len(c.o1);
        qend += li;
        op_hi = k + 2;
        end_off = li;
        p_hi = P{tnew, qnew + li, c.ins + li};
        last_opv = c.o1; next_opv = c.o2;
      } else {
        op_hi = k + 1;
        end_off = l;
        next_opv = c.o1;
      }
      next_op = op_hi;
      next_off = 0;
      p_next = p_hi;
    } else {  // :224-230
      op_hi = k + 1;
      end_off = off;
      next_op = k;
      next_off = off;
    }
    if (started && !emit(new_w, qend, op_hi, end_off, p_hi, last_opv)) return false;
    started = true; w_t = tpos + off; w_q = qend; w_op = next_op; w_off = next_off; w_p = p_next; w_opv = next_opv;
  }
  const uint32_t tpos = S.t_end, qpos = S.q_end;
  if (tpos > nthr && tpos % W != 0) {  // tail window :261-272
    if (!started) { e = {HERRO_E_REFERENCE_PANIC, "tail window without a start (Option::unwrap on None)"}; return false; }
    if (n == 0) { e = {HERRO_E_REFERENCE_PANIC, "empty cigar"}; return false; }
    if (!emit(last_window, qpos, n, op_len(S.opn), P{S.t_end, S.q_end, S.ins_end}, S.opn)) return false;
  }
  return true;
}


}  // namespace herro
