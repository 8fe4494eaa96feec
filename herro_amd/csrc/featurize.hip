// featurize.hip — pileup feature generation on gfx950 (reference src/features.rs:364-580).
//
// Integer / byte work, HBM- and LDS-bound; no MFMA on purpose.  Pipeline (one launch each, all
// windows of a job at once, >>256 workgroups):
//   k_ow_stats    per overlap-window : op prefix sums, long-indel filter (features.rs:315-324),
//                                      alignment accuracy (features.rs:585-679)
//   k_win_layout  per window         : stable rank by accuracy (features.rs:386-409), per-position
//                                      max insertion (features.rs:44-95), row map (prefix sum)
//   k_columns     per column         : pileup column of tokens+quals, written contiguously along
//                                      the row axis (features.rs:110-266) — the reference's
//                                      [L, C] matrix is stored transposed ([C][L]) so that every
//                                      store and every later load is coalesced
//   k_pass1       per window         : informative rows over all columns (features.rs:681-722),
//                                      match/mismatch tallies per query (features.rs:461-500)
//   k_select      per window         : haplotype score, stable re-rank, top-30, all-gap row
//                                      removal, final informative rows (features.rs:502-580)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "job_dev.h"
#include "pileup_core.h"

namespace herro {

static constexpr int NT = 256;  // threads per workgroup (4 waves)

// ---- block-wide exclusive scan of one u32 per thread; returns exclusive prefix, *total = sum ----
__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t* total, uint32_t* s_wave /*[NT/64]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();  // protect s_wave reuse
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) {
    const uint32_t x = s_wave[w];
    if (w < wave) base += x;
    tot += x;
  }
  *total = tot;
  return base + inc - v;
}

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* s_wave) {
  uint32_t tot;
  block_scan(v, &tot, s_wave);
  return tot;
}

// =====================================================================================================
// k_ow_stats — one workgroup per overlap-window
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_ow_stats(JobDev J) {
  __shared__ uint32_t s_wave[NT / 64];
  const uint32_t o = blockIdx.x;
  const OwDesc d = J.ow[o];
  const uint32_t* ops = J.ops + d.op_begin;
  uint32_t* op_t = J.op_t + d.scr_off;
  uint32_t* op_q = J.op_q + d.scr_off;
  const uint32_t cnt = d.op_cnt;

  uint32_t carry_t = 0, carry_q = 0, isum = 0, dsum = 0, longindel = 0;
  for (uint32_t base = 0; base < cnt; base += NT) {
    const uint32_t k = base + threadIdx.x;
    uint32_t tadv = 0, qadv = 0;
    if (k < cnt) {
      const uint32_t op = ops[k], ty = op_type(op);
      const uint32_t e = eff_len(op, k, cnt, d.start_off, d.end_off);
      if (ty != OP_M && op_len(op) > 50u) longindel = 1;  // untrimmed length (features.rs:317)
      if (ty != OP_I) tadv = e;
      if (ty != OP_D) qadv = e;
      if (ty == OP_I) isum += e;
      if (ty == OP_D) dsum += e;
    }
    uint32_t tot_t, tot_q;
    const uint32_t ex_t = block_scan(tadv, &tot_t, s_wave);
    const uint32_t ex_q = block_scan(qadv, &tot_q, s_wave);
    if (k < cnt) {
      op_t[k] = carry_t + ex_t;
      op_q[k] = carry_q + ex_q;
    }
    carry_t += tot_t;
    carry_q += tot_q;
  }
  const uint32_t t_total = carry_t;
  __syncthreads();  // op_t/op_q (global) written by this workgroup are read back below

  // accuracy: matches / mismatches over M ops (features.rs:650-665)
  const uint64_t t_woff = J.read_word_off[J.win[d.win].rid];
  const uint64_t q_woff = J.read_word_off[d.qid];
  uint32_t m = 0, s = 0;
  for (uint32_t u = threadIdx.x; u < t_total; u += NT) {
    const uint32_t k = find_op(op_t, cnt, u);
    if (op_type(ops[k]) == OP_M) {
      const uint32_t q = op_q[k] + (u - op_t[k]);
      const uint32_t tb = read_code(J.read_words, t_woff, d.tstart + u);
      uint32_t qb;
      if (d.strand == 0) qb = read_code(J.read_words, q_woff, d.qbeg + q);
      else qb = read_code(J.read_words, q_woff, d.qbeg + d.qlen - 1 - q) ^ 3u;
      if (tb == qb) m++; else s++;
    }
  }
  m = block_sum(m, s_wave);
  s = block_sum(s, s_wave);
  isum = block_sum(isum, s_wave);
  dsum = block_sum(dsum, s_wave);
  longindel = block_sum(longindel, s_wave);
  if (threadIdx.x == 0) {
    J.ow_keep[o] = longindel ? 0 : 1;
    // (m as f32) / ((m+s+i+d) as f32), correctly rounded (features.rs:678)
    J.ow_acc[o] = __fdiv_rn((float)m, (float)(m + s + isum + dsum));
    J.ow_ttotal[o] = t_total;
  }
}

// =====================================================================================================
// k_win_layout — one workgroup per window
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_win_layout(JobDev J) {
  __shared__ uint32_t s_wave[NT / 64];
  __shared__ uint32_t s_mi[HERRO_MAX_WINDOW];  // max insertion behind each target position
  const uint32_t w = blockIdx.x;
  const WinDesc wd = J.win[w];
  const uint32_t n = wd.ow_cnt;

  // ---- stable rank of kept overlaps by descending accuracy (sort_by_key(-acc), features.rs:386)
  uint32_t kept_local = 0;
  for (uint32_t i = threadIdx.x; i < n; i += NT) {
    const uint32_t oi = wd.ow_begin + i;
    uint32_t slot = 0;
    if (J.ow_keep[oi]) {
      kept_local++;
      const float ai = J.ow_acc[oi];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n; j++) {
        const uint32_t oj = wd.ow_begin + j;
        if (!J.ow_keep[oj]) continue;
        const float aj = J.ow_acc[oj];
        if (aj > ai || (aj == ai && j < i)) rank++;
      }
      slot = rank + 1;
      J.slot_ow[wd.ow_begin + rank] = oi;
    }
    J.ow_slot[oi] = slot;
  }
  const uint32_t n_kept = block_sum(kept_local, s_wave);

  // ---- max insertion per target position over ALL kept overlaps (features.rs:44-95)
  for (uint32_t p = threadIdx.x; p < wd.win_len; p += NT) s_mi[p] = 0;
  __syncthreads();
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t oi = wd.ow_begin + i;
    if (!J.ow_keep[oi]) continue;
    const OwDesc d = J.ow[oi];
    const uint32_t off = d.tstart - wd.tstart;
    for (uint32_t k = threadIdx.x; k < d.op_cnt; k += NT) {
      const uint32_t op = J.ops[d.op_begin + k];
      if (op_type(op) == OP_I) {
        const uint32_t tpos = off + J.op_t[d.scr_off + k];  // >= 1: a slice never starts with I
        if (tpos >= 1 && tpos - 1 < wd.win_len) atomicMax(&s_mi[tpos - 1], op_len(op));
      }
    }
  }
  __syncthreads();

  // ---- row_of_pos = exclusive prefix of (1 + max_ins); rowmap[row] = pos | ins_ordinal << 16
  uint32_t* row_of_pos = J.row_of_pos + wd.pos_off;
  uint32_t* rowmap = J.rowmap + wd.row_off;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < wd.win_len; base += NT) {
    const uint32_t p = base + threadIdx.x;
    const uint32_t v = p < wd.win_len ? 1u + s_mi[p] : 0u;
    uint32_t tot;
    const uint32_t ex = block_scan(v, &tot, s_wave);
    if (p < wd.win_len) {
      const uint32_t r0 = carry + ex;
      row_of_pos[p] = r0;
      for (uint32_t j = 0; j < v; j++)
        if (r0 + j < wd.lub) rowmap[r0 + j] = p | (j << 16);
    }
    carry += tot;
  }
  if (threadIdx.x == 0) {
    row_of_pos[wd.win_len] = carry;
    J.win_L[w] = carry;  // <= lub by construction of lub (host)
    J.win_nkept[w] = n_kept;
  }
}

// =====================================================================================================
// k_columns — one workgroup per pileup column (blocks [0,n_ow): overlaps, [n_ow, n_ow+n_win): targets)
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_columns(JobDev J) {
  const uint32_t b = blockIdx.x;
  if (b >= J.n_ow) {  // ---- target column (features.rs:239-266): base at j==0 rows, '*' elsewhere
    const uint32_t w = b - J.n_ow;
    const WinDesc wd = J.win[w];
    const uint32_t L = J.win_L[w];
    const uint32_t* rowmap = J.rowmap + wd.row_off;
    uint8_t* cb = J.cols_b + wd.col_off;
    uint8_t* cq = J.cols_q + wd.col_off;
    const uint64_t woff = J.read_word_off[wd.rid];
    const uint8_t* qual = J.read_qual + J.read_qual_off[wd.rid];
    for (uint32_t r4 = threadIdx.x * 4; r4 < L; r4 += NT * 4) {
      uint32_t pb = 0, pq = 0;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint32_t r = r4 + e;
        uint32_t tok = TOK_GAP_F, q = 33;
        if (r < L) {
          const uint32_t rm = rowmap[r];
          if ((rm >> 16) == 0) {
            const uint32_t p = wd.tstart + (rm & 0xffffu);
            tok = read_code(J.read_words, woff, p);
            q = qual[p];
          }
        }
        pb |= tok << (8 * e);
        pq |= q << (8 * e);
      }
      *reinterpret_cast<uint32_t*>(cb + r4) = pb;
      *reinterpret_cast<uint32_t*>(cq + r4) = pq;
    }
    return;
  }
  // ---- overlap column (features.rs:110-237)
  const uint32_t slot = J.ow_slot[b];
  if (slot == 0) return;  // filtered out
  const OwDesc d = J.ow[b];
  const WinDesc wd = J.win[d.win];
  const uint32_t L = J.win_L[d.win];
  const uint32_t* rowmap = J.rowmap + wd.row_off;
  uint8_t* cb = J.cols_b + wd.col_off + (uint64_t)slot * wd.lub;
  uint8_t* cq = J.cols_q + wd.col_off + (uint64_t)slot * wd.lub;
  const uint32_t* ops = J.ops + d.op_begin;
  const uint32_t* op_t = J.op_t + d.scr_off;
  const uint32_t* op_q = J.op_q + d.scr_off;
  const uint32_t t_total = J.ow_ttotal[b];
  const int32_t off = (int32_t)(d.tstart - wd.tstart);
  const uint64_t q_woff = J.read_word_off[d.qid];
  const uint8_t* qual = J.read_qual + J.read_qual_off[d.qid];
  const uint32_t gap = d.strand ? TOK_GAP_R : TOK_GAP_F;
  for (uint32_t r4 = threadIdx.x * 4; r4 < L; r4 += NT * 4) {
    uint32_t pb = 0, pq = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint32_t r = r4 + e;
      uint32_t tok = TOK_NONE, q = 33;
      if (r < L) {
        const uint32_t rm = rowmap[r];
        const Cell c = eval_cell(ops, op_t, op_q, d.op_cnt, d.start_off, d.end_off, t_total,
                                 (int32_t)(rm & 0xffffu) - off, rm >> 16);
        if (c.kind == CELL_GAP) tok = gap;
        else if (c.kind == CELL_BASE) {
          // forward: stored index qbeg+q; reverse: complement of stored index qbeg+qlen-1-q, lower
          // case (BASE_LOWER) and the quality of that same stored base (features.rs:128-153)
          const uint32_t si = d.strand ? d.qbeg + d.qlen - 1 - c.q : d.qbeg + c.q;
          const uint32_t code = read_code(J.read_words, q_woff, si);
          tok = d.strand ? 5u + (code ^ 3u) : code;
          q = qual[si];
        }
      }
      pb |= tok << (8 * e);
      pq |= q << (8 * e);
    }
    *reinterpret_cast<uint32_t*>(cb + r4) = pb;
    *reinterpret_cast<uint32_t*>(cq + r4) = pq;
  }
}

// informative row test (features.rs:681-722): >= 2 symbols with count >= thresh.
__device__ __forceinline__ bool supported_from_counts(const uint32_t* c, uint32_t thresh) {
  uint32_t ns = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) ns += (c[k] >= thresh) ? 1u : 0u;
  return ns >= 2;
}

// =====================================================================================================
// k_pass1 — one workgroup per window: informative rows over all columns + per-query tallies
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_pass1(JobDev J) {
  __shared__ uint32_t s_wave[NT / 64];
  const uint32_t w = blockIdx.x;
  const WinDesc wd = J.win[w];
  const uint32_t L = J.win_L[w], n_kept = J.win_nkept[w];
  const uint8_t* cb = J.cols_b + wd.col_off;
  const uint32_t ncols = 1u + (n_kept > 30u ? n_kept : 30u);           // features.rs:282
  const uint32_t thresh = (uint32_t)((double)ncols * 0.1);             // features.rs:712
  uint32_t nsup = 0;
  for (uint32_t r = threadIdx.x; r < L; r += NT) {
    uint32_t c[5] = {0, 0, 0, 0, 0};
    const uint32_t t = cb[r];
    for (uint32_t k = 0; k <= n_kept; k++) {
      const uint32_t f = tok_fold(cb[(uint64_t)k * wd.lub + r]);
      if (f < 5u) c[f]++;
    }
    if (!supported_from_counts(c, thresh)) continue;
    nsup++;
    if (t == TOK_GAP_F) continue;  // insertion row of the target: not tallied (features.rs:489-491)
    for (uint32_t k = 1; k <= n_kept; k++) {
      const uint32_t f = tok_fold(cb[(uint64_t)k * wd.lub + r]);
      const uint32_t cls = J.ow[J.slot_ow[wd.ow_begin + k - 1]].cls;
      atomicAdd(&J.nd[2 * (uint64_t)cls + (f == t ? 0 : 1)], 1u);
    }
  }
  nsup = block_sum(nsup, s_wave);
  if (threadIdx.x == 0) J.win_p1sup[w] = nsup;
}

// =====================================================================================================
// k_select — one workgroup per window
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_select(JobDev J) {
  __shared__ uint32_t s_wave[NT / 64];
  __shared__ uint32_t s_sel[HERRO_ROWS];  // column slot feeding final row c (0xffffffff: padding)
  const uint32_t w = blockIdx.x;
  const WinDesc wd = J.win[w];
  const uint32_t L = J.win_L[w], n_kept = J.win_nkept[w];
  const uint8_t* cb = J.cols_b + wd.col_off;
  const uint8_t* cq = J.cols_q + wd.col_off;

  // ---- score n/(n+d)*ln(n+d+1) (f64, features.rs:505-510); stable descending rank (:512-513)
  if (threadIdx.x < HERRO_ROWS) s_sel[threadIdx.x] = threadIdx.x == 0 ? 0u : 0xffffffffu;
  __syncthreads();
  double* score = J.score + wd.ow_begin;
  for (uint32_t k = threadIdx.x; k < n_kept; k += NT) {
    const uint32_t cls = J.ow[J.slot_ow[wd.ow_begin + k]].cls;
    const uint32_t nn = J.nd[2 * (uint64_t)cls], dd = J.nd[2 * (uint64_t)cls + 1];
    const uint32_t tot = nn + dd;
    double s = 0.0;
    if (tot) {
      const double lg = tot < J.ln_table_n ? J.ln_table[tot] : log((double)tot + 1.0);
      s = __dmul_rn(__ddiv_rn((double)nn, (double)tot), lg);
    }
    score[k] = s;
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < n_kept; k += NT) {
    const double sk = score[k];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n_kept; j++) {
      const double sj = score[j];
      if (sj > sk || (sj == sk && j < k)) rank++;
    }
    J.rank_qid[wd.ow_begin + rank] = J.ow[J.slot_ow[wd.ow_begin + k]].qid;
    if (rank < 30u) s_sel[rank + 1] = k + 1;
  }
  __syncthreads();

  // ---- drop rows where every selected column is a gap or empty (features.rs:531-545), compact
  uint32_t* newidx = J.newidx + wd.row_off;
  uint8_t* fb = J.fin_b + wd.fin_off;
  uint8_t* fq = J.fin_q + wd.fin_off;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < L; base += NT) {
    const uint32_t r = base + threadIdx.x;
    uint32_t keep = 0;
    uint8_t tb[HERRO_ROWS], tq[HERRO_ROWS];
    if (r < L) {
#pragma unroll
      for (int c = 0; c < HERRO_ROWS; c++) {
        const uint32_t sl = s_sel[c];
        tb[c] = sl == 0xffffffffu ? (uint8_t)TOK_NONE : cb[(uint64_t)sl * wd.lub + r];
        tq[c] = sl == 0xffffffffu ? (uint8_t)33 : cq[(uint64_t)sl * wd.lub + r];
        keep |= tok_is_base(tb[c]) ? 1u : 0u;
      }
    }
    uint32_t tot;
    const uint32_t ex = block_scan(keep, &tot, s_wave);
    if (r < L) {
      newidx[r] = carry + ex;  // for dropped rows: index of the next kept row
      if (keep) {
        const uint32_t nr = carry + ex;
#pragma unroll
        for (int c = 0; c < HERRO_ROWS; c++) {
          fb[(uint64_t)c * wd.lub + nr] = tb[c];
          fq[(uint64_t)c * wd.lub + nr] = tq[c];
        }
      }
    }
    carry += tot;
  }
  const uint32_t Lf = carry;
  __syncthreads();  // fin planes + newidx visible to the whole workgroup

  // ---- informative rows of the final [L',31] matrix (thresh = (31*0.1) as usize = 3)
  const uint32_t thresh2 = (uint32_t)((double)HERRO_ROWS * 0.1);
  const uint32_t* rowmap = J.rowmap + wd.row_off;
  const uint32_t* row_of_pos = J.row_of_pos + wd.pos_off;
  uint32_t* sup_row = J.sup_row + wd.row_off;
  uint32_t* sup_pi = J.sup_pi + wd.row_off;
  uint32_t scarry = 0;
  for (uint32_t base = 0; base < L; base += NT) {
    const uint32_t r = base + threadIdx.x;
    uint32_t sup = 0, nr = 0;
    if (r < L) {
      nr = newidx[r];
      const bool kept = (r + 1 < L) ? (newidx[r + 1] != nr) : (nr < Lf);
      if (kept) {
        uint32_t c[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < HERRO_ROWS; k++) {
          const uint32_t f = tok_fold(fb[(uint64_t)k * wd.lub + nr]);
          if (f < 5u) c[f]++;
        }
        sup = supported_from_counts(c, thresh2) ? 1u : 0u;
      }
    }
    uint32_t tot;
    const uint32_t ex = block_scan(sup, &tot, s_wave);
    if (sup) {
      const uint32_t rm = rowmap[r], p = rm & 0xffffu;
      const uint32_t ins = nr - newidx[row_of_pos[p]];  // ordinal among the *kept* insertion rows
      sup_row[scarry + ex] = nr;
      sup_pi[scarry + ex] = p | (ins << 16);
    }
    scarry += tot;
  }
  if (threadIdx.x == 0) {
    J.win_Lf[w] = Lf;
    J.win_nsup[w] = scarry;
  }
}

void launch_featurize(const JobDev& J, hipStream_t st, KernelTimer* tm) {
  if (J.n_ow) {
    KT_BEGIN(tm, "ow_stats", st);
    hipLaunchKernelGGL(k_ow_stats, dim3(J.n_ow), dim3(NT), 0, st, J);
    KT_END(tm, st);
  }
  KT_BEGIN(tm, "win_layout", st);
  hipLaunchKernelGGL(k_win_layout, dim3(J.n_win), dim3(NT), 0, st, J);
  KT_END(tm, st);
  KT_BEGIN(tm, "columns", st);
  hipLaunchKernelGGL(k_columns, dim3(J.n_ow + J.n_win), dim3(NT), 0, st, J);
  KT_END(tm, st);
  KT_BEGIN(tm, "pass1", st);
  hipLaunchKernelGGL(k_pass1, dim3(J.n_win), dim3(NT), 0, st, J);
  KT_END(tm, st);
  KT_BEGIN(tm, "select", st);
  hipLaunchKernelGGL(k_select, dim3(J.n_win), dim3(NT), 0, st, J);
  KT_END(tm, st);
}

}  // namespace herro
