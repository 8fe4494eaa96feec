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

// The same parse for herro_job_create's hot loop (~4300 ops per 4096-bp window of 32 overlaps): besides the ops it
// leaves exclusive prefix sums of the target / query / insertion bases consumed before every op (n + 1 entries each),
// so that everything the job builder derives from a window's op slice is a difference of two entries instead of a
// walk over the slice, and it lists the places where two insertion ops follow each other.
struct ParsedCigar {
  std::vector<uint32_t> ops, pt, pq, pi;
  std::vector<uint32_t> ins_pairs;  // every k with ops[k] and ops[k + 1] both insertions (minimap2 never emits any)
  void clear() { ops.clear(); pt.clear(); pq.clear(); pi.clear(); ins_pairs.clear(); }
  bool ins_pair_in(uint32_t lo, uint32_t hi) const {  // a pair inside the slice [lo, hi)
    for (uint32_t k : ins_pairs) if (k >= lo && k + 1 < hi) return true;
    return false;
  }
};
inline bool parse_cigar_prefix(const uint8_t* s, uint32_t n, ParsedCigar& P, BuildError& e) {
  P.clear();
  const uint32_t guess = n / 2 + 2;
  P.ops.reserve(guess); P.pt.reserve(guess); P.pq.reserve(guess); P.pi.reserve(guess);
  uint32_t i = 0, t = 0, q = 0, ins = 0, prev = 3;
  while (i < n) {
    uint64_t len = 0;
    const uint32_t st = i;
    uint32_t c;
    while (i < n && (c = (uint32_t)s[i] - (uint32_t)'0') < 10u) {
      len = len * 10 + c;
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
    P.ops.push_back(((uint32_t)len << 2) | ty);
    P.pt.push_back(t); P.pq.push_back(q); P.pi.push_back(ins);
    const uint32_t l = (uint32_t)len;
    if (ty != OP_I) t += l;
    if (ty != OP_D) q += l;
    if (ty == OP_I) { ins += l; if (prev == OP_I) P.ins_pairs.push_back((uint32_t)P.ops.size() - 2); }
    prev = ty;
    i++;
  }
  P.pt.push_back(t); P.pq.push_back(q); P.pi.push_back(ins);
  return true;
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

}  // namespace herro
