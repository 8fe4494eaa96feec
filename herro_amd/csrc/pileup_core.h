// pileup_core.h — data layout + the small pieces of integer logic shared by the HIP kernels
// (pileup.hip, cigar_dev.hip) and the host-side descriptor builder (herro_api.hip, windowing.hpp).  Everything here is
// `__host__ __device__` so the exact device logic can also be unit-tested on the CPU.
//
// Reference semantics restated here (lbcb-sci/herro v0.1.1, src/):
//   * effective op length inside a window's CIGAR slice — features.rs:82-90 / :181-188 / :591-614
//   * one pileup cell of one overlap column                — features.rs:110-237
//   * token alphabet                                       — inference.rs:23-31 (BASES_MAP)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HERRO_HD __host__ __device__ __forceinline__
#else
#define HERRO_HD inline
#endif

namespace herro {

// ---- binary CIGAR op: (len << 2) | type -------------------------------------------------------
enum : uint32_t { OP_M = 0, OP_I = 1, OP_D = 2 };
HERRO_HD uint32_t op_type(uint32_t op) { return op & 3u; }
HERRO_HD uint32_t op_len(uint32_t op) { return op >> 2; }

// ---- tokens (inference.rs:23-31): A0 C1 G2 T3 *4 a5 c6 g7 t8 #9 .10, batch padding 11 ----------
enum : uint8_t { TOK_GAP_F = 4, TOK_GAP_R = 9, TOK_NONE = 10, TOK_PAD = 11 };
HERRO_HD bool tok_is_base(uint32_t t) { return t < 4u || (t >= 5u && t <= 8u); }
// BASE_FORWARD folding (features.rs:34-42): strand-insensitive symbol 0..4; TOK_NONE stays 10.
HERRO_HD uint32_t tok_fold(uint32_t t) { return (t >= 5u && t <= 9u) ? t - 5u : t; }
static const char TOK_ASCII[13] = "ACGT*acgt#.?";

// ---- one overlap restricted to one target window (windowing.rs:7-16 + overlap fields) ----------
struct OwDesc {
  uint32_t win;        // window index within the job
  uint32_t qid;        // query read
  uint32_t cls;        // ratio accumulator slot (same target + same query *name*; features.rs:494)
  uint32_t tstart;     // absolute target position where the overlap starts in this window
  uint32_t qbeg;       // stored-read coordinate of the query region start (features.rs:97-108)
  uint32_t qlen;       // window.qend - window.qstart
  uint32_t op_begin;   // first op in JobDev::ops
  uint32_t op_cnt;     // ops in the slice (>= 1)
  uint32_t start_off;  // cigar_start_offset
  uint32_t end_off;    // cigar_end_offset
  uint32_t scr_off;    // offset of this overlap's op_t/op_q scratch
  uint32_t strand;     // 0 forward, 1 reverse
  // copies of per-window / per-read facts, so that the per-overlap kernel starts from ONE record instead of a
  // chain of dependent loads (descriptor -> window -> read offsets -> data)
  uint32_t wtstart;    // first target position of the window (WinDesc::tstart)
  uint32_t wlen;       // target bases in the window (WinDesc::win_len)
  uint64_t t_woff;     // first 2-bit word of the target read
  uint64_t q_woff;     // first 2-bit word of the query read
  uint64_t q_qual_off; // first quality byte of the query read
};

struct WinDesc {
  uint32_t rid, wid, n_wids;
  uint32_t tstart;    // wid * window_size
  uint32_t win_len;   // target bases in the window
  uint32_t ow_begin;  // first OwDesc of this window (push order == alignment order)
  uint32_t ow_cnt;
  uint32_t lub;       // upper bound on rows L (multiple of 16)
  uint64_t col_off;   // first tile of this window in the job's tile list
  uint64_t fin_off;   // byte offset of this window's [31, lub] final planes
  uint64_t row_off;   // element offset of this window's lub-sized u32 row scratch
  uint64_t pos_off;   // element offset of this window's (window_size+1)-sized u32 position scratch
  uint64_t ev_off;    // first slot of this window's compact insertion-event arrays (JobDev::sev / tev); room: its overlaps' ops + 2 each
};

// Effective length of op k of a slice with cnt ops (the reference decides "first"/"last" by the
// op's byte range inside the slice; first op <=> range.start == 0, last <=> range.end == S).
HERRO_HD uint32_t eff_len(uint32_t op, uint32_t k, uint32_t cnt, uint32_t start_off, uint32_t end_off) {
  const uint32_t l = op_len(op);
  if (k == 0 && k + 1 == cnt) return end_off - start_off;
  if (k == 0) return l - start_off;
  if (k + 1 == cnt) return end_off;
  return l;
}

// Largest k in [0,cnt) with op_t[k] <= u (op_t is non-decreasing, op_t[0] == 0).
HERRO_HD uint32_t find_op(const uint32_t* op_t, uint32_t cnt, uint32_t u) {
  uint32_t lo = 0, hi = cnt;  // invariant: op_t[lo] <= u, (hi == cnt or op_t[hi] > u)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (op_t[mid] <= u) lo = mid; else hi = mid;
  }
  return lo;
}

// What one cell of an overlap column holds.
enum : uint32_t { CELL_NONE = 0, CELL_GAP = 1, CELL_BASE = 2 };
struct Cell {
  uint32_t kind;
  uint32_t q;  // CELL_BASE: index into the query region in alignment orientation
};

// Cell of the overlap column at target-relative position u (= p - off, may be negative) and
// insertion ordinal j (0 = the base row itself).  t_total = target bases the slice consumes.
// `hint` carries the covering op between calls for neighbouring cells (pure optimisation).
HERRO_HD Cell eval_cell(const uint32_t* ops, const uint32_t* op_t, const uint32_t* op_q, uint32_t cnt,
                        uint32_t start_off, uint32_t end_off, uint32_t t_total, int32_t u, uint32_t j) {
  Cell c;
  c.q = 0;
  if (u < 0 || (uint32_t)u >= t_total) {  // before the overlap starts / after it ends: '.'
    c.kind = CELL_NONE;
    return c;
  }
  const uint32_t uu = (uint32_t)u;
  const uint32_t k = find_op(op_t, cnt, uu);
  const uint32_t op = ops[k];
  if (j == 0) {
    if (op_type(op) == OP_M) {
      c.kind = CELL_BASE;
      c.q = op_q[k] + (uu - op_t[k]);
    } else {
      c.kind = CELL_GAP;  // deletion (features.rs:205-212)
    }
    return c;
  }
  // insertion slot j-1 behind position u: only ops directly following the covering op, and only
  // if u is that op's last target base (features.rs:213-229: idx - max_ins[tpos-1] + i).
  c.kind = CELL_GAP;
  const uint32_t tend = op_t[k] + eff_len(op, k, cnt, start_off, end_off);
  if (uu + 1 != tend) return c;
  for (uint32_t kk = k + 1; kk < cnt && op_type(ops[kk]) == OP_I; kk++) {
    if (eff_len(ops[kk], kk, cnt, start_off, end_off) > j - 1) {
      c.kind = CELL_BASE;
      c.q = op_q[kk] + (j - 1);  // a later insertion at the same place overwrites (as in the reference)
    }
  }
  return c;
}

// ---- base rows of the final matrix in position space (k_rows, pileup.hip; round 5) ----------------------------------
// One bit per target position, 32 positions per word.  A symbol's occurrences over the columns are counted in a saturating 2-bit
// bit-sliced counter (c1 c0 = 0, 1, 2, ">= 3"): enough for the informative-row rule (two symbols reach thresh = (31 * 0.1) as usize = 3,
// features.rs:558,712) AND for the decoder's vote (consensus.rs:178-200), which is only consulted on rows that are not informative —
// where at most one symbol reaches 3.  tests/test_vote_planes.py checks both against the reference's rules on every count vector.
HERRO_HD void sat2_add(uint32_t& c0, uint32_t& c1, uint32_t x) {   // one more column shows the symbol at the positions of x
  const uint32_t a = c0, b = c1;
  c0 = (a ^ x) | (a & b);
  c1 = b | (a & x);
}
HERRO_HD void sat2_merge(uint32_t& c0, uint32_t& c1, uint32_t b0, uint32_t b1) {   // the saturating sum of two such counters
  const uint32_t a0 = c0, a1 = c1;
  const uint32_t s0 = a0 ^ b0, k0 = a0 & b0, s1 = a1 ^ b1 ^ k0, k1 = (a1 & b1) | (k0 & (a1 ^ b1));   // k1: the sum is 4 or more
  c0 = s0 | k1;
  c1 = s1 | k1;
}
struct RowVotes { uint32_t sup, v0, v1, v2; };   // informative positions; the others' vote as three bit planes (code 0..4 = A C G T *)
// c0 / c1: the counters of A C G T * (the target's own base included); tsym: one-hot planes of the target's base; vm: positions inside the window.
// consensus.rs:186-200 on a row that is not informative: the two most common symbols by a stable descending sort (ties keep A C G T * order);
// base = (c0 < 2 || (c0 == c1 && (b0 == target || b1 == target))) ? target : b0.  With at most one symbol at >= 3:
//   a symbol at >= 3 -> it is b0 and c0 > c1: that symbol;  else no symbol at 2 -> c0 < 2: the target;
//   else b0 = the first symbol at 2; a second symbol at 2 makes c0 == c1 with b1 = that one: the target if it is one of the two, else b0.
// On informative rows the model decides (k_consensus_p patches its call in); they get the target's code so that every code is defined.
HERRO_HD RowVotes base_row_votes(const uint32_t (&c0)[5], const uint32_t (&c1)[5], const uint32_t (&tsym)[4], uint32_t vm) {
  uint32_t g3[5], e2[5];
#pragma unroll
  for (int q = 0; q < 5; q++) { g3[q] = c1[q] & c0[q]; e2[q] = c1[q] & ~c0[q]; }
  uint32_t one = 0, two = 0;
#pragma unroll
  for (int q = 0; q < 5; q++) { two |= one & g3[q]; one |= g3[q]; }
  const uint32_t sup = two & vm;
  uint32_t f[5], sc[5], seen1 = 0, seen2 = 0;   // first / second symbol with exactly two, in A C G T * order
#pragma unroll
  for (int q = 0; q < 5; q++) {
    f[q] = e2[q] & ~seen1;
    sc[q] = e2[q] & seen1 & ~seen2;
    seen2 |= seen1 & e2[q];
    seen1 |= e2[q];
  }
  uint32_t tb_in = 0;   // the target's base is one of the two (the target column never shows '*' on a base row)
#pragma unroll
  for (int q = 0; q < 4; q++) tb_in |= tsym[q] & (f[q] | sc[q]);
  const uint32_t any3 = one, tie_t = seen2 & tb_in;
  const uint32_t use_t = sup | (~any3 & (~seen1 | tie_t));
  const uint32_t use_g = any3 & ~sup, use_f = ~any3 & seen1 & ~tie_t;
  uint32_t v[5];
#pragma unroll
  for (int q = 0; q < 5; q++) v[q] = (use_g & g3[q]) | (use_f & f[q]) | (q < 4 ? use_t & tsym[q] : 0u);
  return RowVotes{sup, (v[1] | v[3]) & vm, (v[2] | v[3]) & vm, v[4] & vm};
}
// The decoder's vote on exact counts c5 = A C G T * (insertion rows: the target shows '*', tb = 4)
HERRO_HD uint32_t vote5(const uint32_t (&c5)[5], uint32_t tb) {
  uint32_t m0 = c5[0], i0 = 0;
#pragma unroll
  for (uint32_t q = 1; q < 5; q++) if (c5[q] > m0) { m0 = c5[q]; i0 = q; }
  uint32_t m1 = 0, i1 = 5;
  bool have = false;
#pragma unroll
  for (uint32_t q = 0; q < 5; q++)
    if (q != i0 && (!have || c5[q] > m1)) { m1 = c5[q]; i1 = q; have = true; }
  return (m0 < 2u || (m0 == m1 && (i0 == tb || i1 == tb))) ? tb : i0;
}

// 2-bit read store access (haec_io.rs:163-171).
HERRO_HD uint32_t read_code(const uint64_t* words, uint64_t word_off, uint32_t i) {
  return (uint32_t)((words[word_off + (i >> 5)] >> ((i & 31u) << 1)) & 3ull);
}

}  // namespace herro
