// featurize.hip — pileup feature generation on gfx950 (reference src/features.rs:364-580).
//
// Integer / byte work, HBM- and LDS-bound; no MFMA on purpose.  Six launches per job, all windows
// at once; the heavy ones are tiled over pileup rows so a 128-window batch already gives >2000
// workgroups:
//   k_ow_stats       per overlap-window : op prefix sums, long-indel filter (features.rs:315-324),
//                                         accuracy by bit-parallel 2-bit compares (features.rs:585-679),
//                                         insertion events, op checkpoints
//   k_win_layout     per window         : stable rank by accuracy (features.rs:386-409), per-position
//                                         max insertion over all kept overlaps (features.rs:44-95),
//                                         row map (prefix sum)
//   k_pass1_tiles    per 256 rows       : every pileup cell of every kept overlap is evaluated from an
//                                         LDS-staged CIGAR segment (features.rs:110-266); informative
//                                         rows (features.rs:681-722) and match/mismatch tallies per
//                                         query (features.rs:461-500).  Nothing is written but tallies:
//                                         the reference's [L, 1+n] pass-1 matrix never exists in HBM.
//   k_select_layout  per window         : haplotype score, stable re-rank, top-30 (features.rs:502-525);
//                                         all-gap row removal (features.rs:531-556) == row map over the
//                                         max insertion of the *selected* overlaps only
//   k_final_tiles    per 256 rows       : final [31][L'] token + quality planes (row axis contiguous:
//                                         coalesced stores), informative-row flags (features.rs:558)
//   k_sup_compact    per window         : ordered list of informative positions
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "job_dev.h"
#include "pileup_core.h"

namespace herro {

static constexpr int NT = 256;       // threads per workgroup (4 waves)
static constexpr int OPCAP = 1024;   // ops of one overlap-window staged in LDS by k_ow_stats
static constexpr int EVCAP = 1024;   // overlaps whose insertion events are flattened through LDS

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

// 16 consecutive 2-bit bases starting at base index i, packed LSB-first into a u32.
__device__ __forceinline__ uint32_t get16(const uint64_t* __restrict__ words, uint64_t woff, uint32_t i) {
  const uint64_t* w = words + woff + (i >> 5);
  const uint32_t sh = (i & 31u) << 1;
  uint64_t v = w[0] >> sh;
  if (sh > 32) v |= w[1] << (64 - sh);  // the store carries one pad word at its end
  return (uint32_t)v;
}
// reverse the order of the sixteen 2-bit fields of x
__device__ __forceinline__ uint32_t rev_pairs(uint32_t x) {
  const uint32_t r = __brev(x);
  return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}

// =====================================================================================================
// k_ow_stats — one workgroup per overlap-window
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_ow_stats(JobDev J) {
  __shared__ uint32_t s_wave[NT / 64];
  __shared__ uint32_t s_op[OPCAP], s_t[OPCAP], s_q[OPCAP];
  const uint32_t o = blockIdx.x;
  const OwDesc d = J.ow[o];
  const WinDesc wd = J.win[d.win];
  const uint32_t* ops = J.ops + d.op_begin;
  uint32_t* op_t = J.op_t + d.scr_off;
  uint32_t* op_q = J.op_q + d.scr_off;
  uint32_t* ins_ev = J.ins_ev + d.scr_off;
  const uint32_t cnt = d.op_cnt;
  const uint32_t off = d.tstart - wd.tstart;
  const bool in_lds = cnt <= OPCAP;

  uint32_t carry_t = 0, carry_q = 0, carry_i = 0, isum = 0, dsum = 0, longindel = 0;
  for (uint32_t base = 0; base < cnt; base += NT) {
    const uint32_t k = base + threadIdx.x;
    uint32_t tadv = 0, qadv = 0, is_i = 0, op = 0;
    if (k < cnt) {
      op = ops[k];
      const uint32_t ty = op_type(op);
      const uint32_t e = eff_len(op, k, cnt, d.start_off, d.end_off);
      if (ty != OP_M && op_len(op) > 50u) longindel = 1;  // untrimmed length (features.rs:317)
      if (ty != OP_I) tadv = e;
      if (ty != OP_D) qadv = e;
      if (ty == OP_I) { isum += e; is_i = 1; }
      if (ty == OP_D) dsum += e;
    }
    uint32_t tot_t, tot_q, tot_i;
    const uint32_t ex_t = block_scan(tadv, &tot_t, s_wave);
    const uint32_t ex_q = block_scan(qadv, &tot_q, s_wave);
    const uint32_t ex_i = block_scan(is_i, &tot_i, s_wave);
    if (k < cnt) {
      const uint32_t t = carry_t + ex_t, q = carry_q + ex_q;
      op_t[k] = t;
      op_q[k] = q;
      if (in_lds) { s_op[k] = op; s_t[k] = t; s_q[k] = q; }
      // insertion behind window position off+t-1 (features.rs:77); t >= 1: a slice never starts with I
      if (is_i) ins_ev[carry_i + ex_i] = ((off + t - 1u) & 0xffffu) | (op_len(op) << 16);
    }
    carry_t += tot_t;
    carry_q += tot_q;
    carry_i += tot_i;
  }
  const uint32_t t_total = carry_t;
  __syncthreads();  // LDS tables / global op_t visible to the whole workgroup
  const uint32_t* po = in_lds ? s_op : ops;
  const uint32_t* pt = in_lds ? s_t : op_t;
  const uint32_t* pq = in_lds ? s_q : op_q;

  // op checkpoints every 128 target positions (entry points for the tile kernels)
  for (uint32_t c = threadIdx.x; c < J.n_ckpt; c += NT) {
    const uint32_t u = c << HERRO_CKPT_SHIFT;
    J.ckpt[(uint64_t)o * J.n_ckpt + c] = u < t_total ? find_op(pt, cnt, u) : 0u;
  }

  // accuracy: matches / mismatches over M ops (features.rs:650-665), 16 target bases per step
  const uint64_t t_woff = J.read_word_off[wd.rid];
  const uint64_t q_woff = J.read_word_off[d.qid];
  uint32_t m = 0, s = 0;
  for (uint32_t u0 = threadIdx.x * 16u; u0 < t_total; u0 += NT * 16u) {
    const uint32_t u1 = min(u0 + 16u, t_total);
    uint32_t k = find_op(pt, cnt, u0), u = u0;
    while (u < u1) {
      const uint32_t op = po[k], ty = op_type(op);
      if (ty == OP_I) { k++; continue; }
      const uint32_t t0 = pt[k], oend = t0 + eff_len(op, k, cnt, d.start_off, d.end_off);
      const uint32_t se = min(u1, oend);
      if (ty == OP_M) {
        const uint32_t n = se - u, q = pq[k] + (u - t0);
        const uint32_t tb = get16(J.read_words, t_woff, d.tstart + u);
        uint32_t qb;
        if (d.strand == 0) {
          qb = get16(J.read_words, q_woff, d.qbeg + q);
        } else {  // alignment-orientation bases q..q+n-1 = complement of stored bases s_lo+n-1 .. s_lo
          const uint32_t s_lo = d.qbeg + d.qlen - q - n;
          qb = ~(rev_pairs(get16(J.read_words, q_woff, s_lo)) >> (32u - 2u * n));
        }
        const uint32_t x = tb ^ qb;
        const uint32_t mask = n >= 16u ? 0x55555555u : (0x55555555u & ((1u << (2u * n)) - 1u));
        const uint32_t mm = __popc((x | (x >> 1)) & mask);
        s += mm;
        m += n - mm;
      }
      u = se;
      if (se == oend) k++;
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
    J.ins_cnt[o] = carry_i;
  }
}

// Scatter-max the insertion events of `n` overlaps (given by ow index list) into s_mi (LDS).
// Events of all listed overlaps are flattened over the threads through an LDS prefix of counts.
__device__ __forceinline__ void scatter_max_ins(const JobDev& J, const uint32_t* ow_list, uint32_t n,
                                                uint32_t win_len, uint32_t* s_mi, uint32_t* s_pref,
                                                uint32_t* s_wave) {
  for (uint32_t base = 0; base < n; base += EVCAP) {
    const uint32_t nn = min((uint32_t)EVCAP, n - base);
    uint32_t carry = 0;
    for (uint32_t b2 = 0; b2 < nn; b2 += NT) {
      const uint32_t i = b2 + threadIdx.x;
      const uint32_t c = i < nn ? J.ins_cnt[ow_list[base + i]] : 0u;
      uint32_t tot;
      const uint32_t ex = block_scan(c, &tot, s_wave);
      if (i < nn) s_pref[i] = carry + ex;
      carry += tot;
    }
    __syncthreads();
    const uint32_t E = carry;
    for (uint32_t e = threadIdx.x; e < E; e += NT) {
      uint32_t lo = 0, hi = nn;  // largest i with s_pref[i] <= e
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_pref[mid] <= e) lo = mid; else hi = mid;
      }
      const uint32_t ev = J.ins_ev[J.ow[ow_list[base + lo]].scr_off + (e - s_pref[lo])];
      const uint32_t p = ev & 0xffffu;
      if (p < win_len) atomicMax(&s_mi[p], ev >> 16);
    }
    __syncthreads();
  }
}

// row_of_pos = exclusive prefix of (1 + max_ins); rowmap[row] = pos | ins_ordinal << 16.  Returns L.
__device__ __forceinline__ uint32_t write_layout(const uint32_t* s_mi, uint32_t win_len, uint32_t lub,
                                                 uint32_t* row_of_pos, uint32_t* rowmap, uint32_t* s_wave) {
  const uint32_t ch = (win_len + NT - 1) / NT;  // consecutive positions per thread
  const uint32_t p0 = threadIdx.x * ch, p1 = min(p0 + ch, win_len);
  uint32_t local = 0;
  for (uint32_t p = p0; p < p1; p++) local += 1u + s_mi[p];
  uint32_t tot;
  uint32_t r = block_scan(local, &tot, s_wave);
  for (uint32_t p = p0; p < p1; p++) {
    row_of_pos[p] = r;
    const uint32_t v = 1u + s_mi[p];
    for (uint32_t j = 0; j < v; j++)
      if (r + j < lub) rowmap[r + j] = p | (j << 16);
    r += v;
  }
  if (threadIdx.x == 0) row_of_pos[win_len] = tot;
  return tot;
}

// =====================================================================================================
// k_win_layout — one workgroup per window
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_win_layout(JobDev J) {
  __shared__ uint32_t s_wave[NT / 64];
  __shared__ uint32_t s_mi[HERRO_MAX_WINDOW];
  __shared__ uint32_t s_pref[EVCAP];
  const uint32_t w = blockIdx.x;
  const WinDesc wd = J.win[w];
  const uint32_t n = wd.ow_cnt;

  // stable rank of kept overlaps by descending accuracy (sort_by_key(-acc), features.rs:386)
  uint32_t kept_local = 0;
  for (uint32_t i = threadIdx.x; i < n; i += NT) {
    const uint32_t oi = wd.ow_begin + i;
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
      J.slot_ow[wd.ow_begin + rank] = oi;
    }
  }
  for (uint32_t p = threadIdx.x; p < wd.win_len; p += NT) s_mi[p] = 0;
  const uint32_t n_kept = block_sum(kept_local, s_wave);  // also orders slot_ow / s_mi writes
  __syncthreads();

  // max insertion per target position over ALL kept overlaps (features.rs:44-95)
  scatter_max_ins(J, J.slot_ow + wd.ow_begin, n_kept, wd.win_len, s_mi, s_pref, s_wave);

  const uint32_t L = write_layout(s_mi, wd.win_len, wd.lub, J.row_of_pos + wd.pos_off,
                                  J.rowmap + wd.row_off, s_wave);
  if (threadIdx.x == 0) {
    J.win_L[w] = L;  // <= lub by construction of lub (host)
    J.win_nkept[w] = n_kept;
  }
}

// ---- pileup columns of one row tile, evaluated lane = row --------------------------------------------
// Consecutive lanes own consecutive pileup rows, so the loads of query bases / qualities are
// coalesced and control flow is uniform across the wave.  For each column the CIGAR ops that can
// touch the tile (a window of WOPS ops starting at the checkpointed op) are staged into LDS by all
// threads at once; a cell then needs a 6-step binary search in LDS.
static constexpr int NCOL = 32;   // columns staged per pass
static constexpr int WOPS = 64;   // ops staged per column

struct SHdr {
  int32_t off;        // window-relative position where the overlap starts
  uint32_t t_total;   // target bases the slice consumes
  uint32_t k0;        // first staged op (absolute index in the slice)
  uint32_t cnt;       // ops in the slice
  uint32_t fallback;  // staged window does not reach the end of the tile: use the global tables
  uint32_t ow, qbeg, qlen, strand, cls;
  uint64_t q_woff;
  const uint8_t* qual;
};

struct TileLds {
  SHdr hdr[NCOL];
  uint32_t op[NCOL * WOPS], t[NCOL * WOPS], q[NCOL * WOPS];
};

// Stage columns `ow_list[0..ng)` for the tile covering window positions [p_lo, p_hi].
__device__ __forceinline__ void stage_group(const JobDev& J, const WinDesc& wd, const uint32_t* ow_list,
                                            uint32_t ng, uint32_t p_lo, uint32_t p_hi, TileLds& S) {
  if (threadIdx.x < ng) {
    const uint32_t o = ow_list[threadIdx.x];
    SHdr h;
    h.ow = o;
    if (o != 0xffffffffu) {
      const OwDesc d = J.ow[o];
      h.off = (int32_t)(d.tstart - wd.tstart);
      h.t_total = J.ow_ttotal[o];
      h.cnt = d.op_cnt;
      h.qbeg = d.qbeg; h.qlen = d.qlen; h.strand = d.strand; h.cls = d.cls;
      h.q_woff = J.read_word_off[d.qid];
      h.qual = J.read_qual + J.read_qual_off[d.qid];
      const int32_t ulo = max((int32_t)p_lo - h.off, 0);
      const int32_t uhi = min((int32_t)p_hi - h.off, (int32_t)h.t_total - 1);
      h.k0 = 0;
      h.fallback = 0;
      if (ulo <= uhi) {
        h.k0 = J.ckpt[(uint64_t)o * J.n_ckpt + ((uint32_t)ulo >> HERRO_CKPT_SHIFT)];
        // the window must hold the op covering uhi plus the insertions right behind it
        const uint32_t last = h.k0 + WOPS - 1;
        if (last + 1 < d.op_cnt && J.op_t[d.scr_off + last] <= (uint32_t)uhi + 1u) h.fallback = 1;
      } else {
        h.t_total = 0;  // the tile lies outside the overlap: every cell is '.'
      }
    }
    S.hdr[threadIdx.x] = h;
  }
  __syncthreads();
  for (uint32_t idx = threadIdx.x; idx < ng * WOPS; idx += NT) {
    const uint32_t c = idx / WOPS, i = idx % WOPS;
    const SHdr& h = S.hdr[c];
    uint32_t op = 0, t = 0xffffffffu, q = 0;
    if (h.ow != 0xffffffffu && h.t_total && h.k0 + i < h.cnt) {
      const OwDesc& d = J.ow[h.ow];
      op = J.ops[d.op_begin + h.k0 + i];
      t = J.op_t[d.scr_off + h.k0 + i];
      q = J.op_q[d.scr_off + h.k0 + i];
    }
    S.op[idx] = op; S.t[idx] = t; S.q[idx] = q;
  }
  __syncthreads();
}

// Cell of staged column c at target-relative position u, insertion ordinal j (features.rs:173-231).
__device__ __forceinline__ Cell eval_lane(const JobDev& J, const TileLds& S, uint32_t c, int32_t u, uint32_t j) {
  const SHdr& h = S.hdr[c];
  Cell cell;
  cell.kind = CELL_NONE;
  cell.q = 0;
  if (u < 0 || (uint32_t)u >= h.t_total) return cell;
  if (h.fallback) {  // rare: more than WOPS ops between the checkpoint and the end of the tile
    const OwDesc d = J.ow[h.ow];
    return eval_cell(J.ops + d.op_begin, J.op_t + d.scr_off, J.op_q + d.scr_off, d.op_cnt, d.start_off,
                     d.end_off, h.t_total, u, j);
  }
  const uint32_t uu = (uint32_t)u;
  const uint32_t* T = S.t + c * WOPS;
  uint32_t pos = 0;  // largest i with T[i] <= uu (T sorted, padded with 0xffffffff; T[0] <= uu)
#pragma unroll
  for (uint32_t st = WOPS / 2; st; st >>= 1)
    if (T[pos + st] <= uu) pos += st;
  const uint32_t op = S.op[c * WOPS + pos];
  if (j == 0) {
    if (op_type(op) == OP_M) { cell.kind = CELL_BASE; cell.q = S.q[c * WOPS + pos] + (uu - T[pos]); }
    else cell.kind = CELL_GAP;  // deletion
    return cell;
  }
  // insertion slot j-1 behind u: only if u is the last target base of its op and insertions follow.
  // (an I op is never trimmed by the window offsets, so its effective length is its length)
  cell.kind = CELL_GAP;
  uint32_t x = pos + 1;
  const uint32_t t_next = (h.k0 + x < h.cnt) ? T[x] : h.t_total;
  if (uu + 1 != t_next) return cell;
  for (; h.k0 + x < h.cnt && x < WOPS; x++) {
    const uint32_t opx = S.op[c * WOPS + x];
    if (op_type(opx) != OP_I) break;
    if (op_len(opx) > j - 1) { cell.kind = CELL_BASE; cell.q = S.q[c * WOPS + x] + (j - 1); }
  }
  return cell;
}

// token (and quality) of a cell; forward: stored index qbeg+q, upper case; reverse: complement of
// stored index qbeg+qlen-1-q, lower case, quality of that same stored base (features.rs:128-153)
template <bool WITH_QUAL>
__device__ __forceinline__ uint32_t cell_token(const JobDev& J, const SHdr& h, const Cell& c, uint32_t* qual) {
  if (c.kind == CELL_NONE) return TOK_NONE;
  if (c.kind == CELL_GAP) return h.strand ? TOK_GAP_R : TOK_GAP_F;
  const uint32_t si = h.strand ? h.qbeg + h.qlen - 1 - c.q : h.qbeg + c.q;
  const uint32_t code = read_code(J.read_words, h.q_woff, si);
  if (WITH_QUAL) *qual = h.qual[si];
  return h.strand ? 5u + (code ^ 3u) : code;
}


// Tokens (and qualities) of NB consecutive staged columns at one row, with all NB base (and quality)
// loads issued back to back before any is consumed (memory-level parallelism instead of NB
// serialized HBM/L2 round trips per wave).
static constexpr int NB = 8;
template <bool WITH_QUAL>
__device__ __forceinline__ void eval_batch(const JobDev& J, const TileLds& S, uint32_t c0, uint32_t nc, int32_t p,
                                           uint32_t j, uint32_t (&tok)[NB], uint32_t (&ql)[NB]) {
  uint32_t si[NB];
  uint64_t wd64[NB];
  uint8_t qb[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) {
    tok[k] = TOK_NONE;
    ql[k] = 33;
    si[k] = 0xffffffffu;
    if ((uint32_t)k < nc && S.hdr[c0 + k].ow != 0xffffffffu) {
      const SHdr& h = S.hdr[c0 + k];
      const Cell cell = eval_lane(J, S, c0 + k, p - h.off, j);
      if (cell.kind == CELL_GAP) tok[k] = h.strand ? TOK_GAP_R : TOK_GAP_F;
      else if (cell.kind == CELL_BASE) si[k] = h.strand ? h.qbeg + h.qlen - 1 - cell.q : h.qbeg + cell.q;
    }
  }
#pragma unroll
  for (int k = 0; k < NB; k++) {
    wd64[k] = 0;
    qb[k] = 33;
    if (si[k] != 0xffffffffu) {
      const SHdr& h = S.hdr[c0 + k];
      wd64[k] = J.read_words[h.q_woff + (si[k] >> 5)];
      if (WITH_QUAL) qb[k] = h.qual[si[k]];
    }
  }
#pragma unroll
  for (int k = 0; k < NB; k++) {
    if (si[k] != 0xffffffffu) {
      const uint32_t code = (uint32_t)(wd64[k] >> ((si[k] & 31u) << 1)) & 3u;
      tok[k] = S.hdr[c0 + k].strand ? 5u + (code ^ 3u) : code;
      ql[k] = qb[k];
    }
  }
}

// Symbol counters A,C,G,T,* packed as five 12-bit fields (a runtime-indexed register array would
// live in scratch memory); the host caps overlaps per window at 4000.
__device__ __forceinline__ void count_sym(uint64_t& c, uint32_t folded) {
  if (folded < 5u) c += 1ull << (12u * folded);
}
// informative row test (features.rs:681-722): >= 2 symbols with count >= thresh.
__device__ __forceinline__ bool supported_from_counts(uint64_t c, uint32_t thresh) {
  uint32_t ns = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) ns += (((uint32_t)(c >> (12 * k)) & 0xfffu) >= thresh) ? 1u : 0u;
  return ns >= 2;
}

// =====================================================================================================
// k_pass1_tiles — one workgroup per 256 pass-1 rows
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_pass1_tiles(JobDev J) {
  __shared__ TileLds S;
  const uint32_t w = J.tile_win[blockIdx.x], r0 = J.tile_r0[blockIdx.x];
  const uint32_t L = J.win_L[w];
  if (r0 >= L) return;
  const WinDesc wd = J.win[w];
  const uint32_t n_kept = J.win_nkept[w];
  const uint32_t* rowmap = J.rowmap + wd.row_off;
  const uint32_t p_lo = rowmap[r0] & 0xffffu, p_hi = rowmap[min(r0 + NT, L) - 1] & 0xffffu;
  const uint32_t r = r0 + threadIdx.x;
  const bool valid = r < L;
  const uint32_t rm = valid ? rowmap[r] : 0u;
  const int32_t p = (int32_t)(rm & 0xffffu);
  const uint32_t j = rm >> 16;
  // target column (features.rs:239-266): base on j == 0 rows, '*' on insertion rows
  const uint32_t t = (valid && j == 0) ? read_code(J.read_words, J.read_word_off[wd.rid], wd.tstart + (uint32_t)p)
                                       : (uint32_t)TOK_GAP_F;
  const uint32_t ncols = 1u + (n_kept > 30u ? n_kept : 30u);  // features.rs:282
  const uint32_t thresh = (uint32_t)((double)ncols * 0.1);    // features.rs:712
  uint64_t cnt = 0;
  if (valid) count_sym(cnt, t);
  const bool single = n_kept <= NCOL;
  bool sup = false;
  for (int pass = 0; pass < 2; pass++) {
    for (uint32_t g0 = 0; g0 < n_kept; g0 += NCOL) {
      const uint32_t ng = min((uint32_t)NCOL, n_kept - g0);
      if (!(single && pass == 1)) {
        __syncthreads();
        stage_group(J, wd, J.slot_ow + wd.ow_begin + g0, ng, p_lo, p_hi, S);
      }
      if (pass == 0 ? valid : sup) {
        for (uint32_t c0 = 0; c0 < ng; c0 += NB) {
          uint32_t tok[NB], ql[NB];
          const uint32_t nc = min((uint32_t)NB, ng - c0);
          eval_batch<false>(J, S, c0, nc, p, j, tok, ql);
#pragma unroll
          for (int k = 0; k < NB; k++) {
            if ((uint32_t)k >= nc) break;
            const uint32_t f = tok_fold(tok[k]);
            if (pass == 0) count_sym(cnt, f);
            else  // tallies (features.rs:478-498): '.', '*', '#' count as mismatch
              atomicAdd(&J.nd[2 * (uint64_t)S.hdr[c0 + k].cls + (f == t ? 0 : 1)], 1u);
          }
        }
      }
    }
    if (pass == 0) {
      // '*' target rows (insertion rows) are never tallied (features.rs:489-491)
      sup = valid && t != TOK_GAP_F && supported_from_counts(cnt, thresh);
      if (!__syncthreads_or(sup)) break;
    }
  }
}

// =====================================================================================================
// k_select_layout — one workgroup per window
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_select_layout(JobDev J) {
  __shared__ uint32_t s_wave[NT / 64];
  __shared__ uint32_t s_mi[HERRO_MAX_WINDOW];
  __shared__ uint32_t s_pref[EVCAP];
  __shared__ double s_score[EVCAP];
  __shared__ uint32_t s_sel[32];
  const uint32_t w = blockIdx.x;
  const WinDesc wd = J.win[w];
  const uint32_t n_kept = J.win_nkept[w];
  uint32_t* sel = J.sel_ow + (uint64_t)w * 32;
  if (threadIdx.x < 32) s_sel[threadIdx.x] = 0xffffffffu;
  for (uint32_t p = threadIdx.x; p < wd.win_len; p += NT) s_mi[p] = 0;
  __syncthreads();

  // score n/(n+d)*ln(n+d+1) in f64 (features.rs:505-510); stable descending rank (features.rs:512-513)
  auto score_of = [&](uint32_t k) -> double {
    const uint32_t cls = J.ow[J.slot_ow[wd.ow_begin + k]].cls;
    const uint32_t nn = J.nd[2 * (uint64_t)cls], dd = J.nd[2 * (uint64_t)cls + 1];
    const uint32_t tot = nn + dd;
    if (!tot) return 0.0;
    const double lg = tot < J.ln_table_n ? J.ln_table[tot] : log((double)tot + 1.0);
    return __dmul_rn(__ddiv_rn((double)nn, (double)tot), lg);
  };
  const bool cached = n_kept <= EVCAP;
  if (cached) {
    for (uint32_t k = threadIdx.x; k < n_kept; k += NT) s_score[k] = score_of(k);
    __syncthreads();
  }
  for (uint32_t k = threadIdx.x; k < n_kept; k += NT) {
    const double sk = cached ? s_score[k] : score_of(k);
    uint32_t rank = 0;
    for (uint32_t i = 0; i < n_kept; i++) {
      const double si = cached ? s_score[i] : score_of(i);
      if (si > sk || (si == sk && i < k)) rank++;
    }
    const uint32_t o = J.slot_ow[wd.ow_begin + k];
    J.rank_qid[wd.ow_begin + rank] = J.ow[o].qid;
    if (rank < 30u) s_sel[rank + 1] = o;
  }
  __syncthreads();
  if (threadIdx.x < 32) sel[threadIdx.x] = s_sel[threadIdx.x];

  // rows where every selected column is a gap are dropped (features.rs:531-556).  Target rows always
  // carry a base; an insertion row (p, j) survives iff some *selected* overlap inserts >= j bases
  // behind p — i.e. the final layout is the row map over the selected overlaps' max insertion.
  const uint32_t n_sel = min(n_kept, 30u);
  scatter_max_ins(J, s_sel + 1, n_sel, wd.win_len, s_mi, s_pref, s_wave);
  const uint32_t Lf = write_layout(s_mi, wd.win_len, wd.lub, J.row_of_pos2 + wd.pos_off,
                                   J.rowmap2 + wd.row_off, s_wave);
  if (threadIdx.x == 0) J.win_Lf[w] = Lf;
}

// =====================================================================================================
// k_final_tiles — one workgroup per 256 final rows
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_final_tiles(JobDev J) {
  __shared__ TileLds S;
  const uint32_t w = J.tile_win[blockIdx.x], r0 = J.tile_r0[blockIdx.x];
  const uint32_t Lf = J.win_Lf[w];
  if (r0 >= Lf) return;
  const WinDesc wd = J.win[w];
  const uint32_t* rowmap = J.rowmap2 + wd.row_off;
  const uint32_t p_lo = rowmap[r0] & 0xffffu, p_hi = rowmap[min(r0 + NT, Lf) - 1] & 0xffffu;
  stage_group(J, wd, J.sel_ow + (uint64_t)w * 32 + 1, HERRO_ROWS - 1, p_lo, p_hi, S);
  const uint32_t r = r0 + threadIdx.x;
  if (r >= Lf) return;
  const uint32_t rm = rowmap[r];
  const int32_t p = (int32_t)(rm & 0xffffu);
  const uint32_t j = rm >> 16;
  uint8_t* fb = J.fin_b + wd.fin_off + r;
  uint8_t* fq = J.fin_q + wd.fin_off + r;
  uint64_t cnt = 0;
  {
    uint32_t t = TOK_GAP_F, q = 33;
    if (j == 0) {
      t = read_code(J.read_words, J.read_word_off[wd.rid], wd.tstart + (uint32_t)p);
      q = J.read_qual[J.read_qual_off[wd.rid] + wd.tstart + (uint32_t)p];
    }
    fb[0] = (uint8_t)t;
    fq[0] = (uint8_t)q;
    count_sym(cnt, t);
  }
  // fewer than 30 overlaps: untouched '.' / '!' columns (features.rs:522-525)
  for (uint32_t c0 = 0; c0 < HERRO_ROWS - 1; c0 += NB) {
    uint32_t tok[NB], ql[NB];
    const uint32_t nc = min((uint32_t)NB, (uint32_t)(HERRO_ROWS - 1) - c0);
    eval_batch<true>(J, S, c0, nc, p, j, tok, ql);
#pragma unroll
    for (int k = 0; k < NB; k++) {
      if ((uint32_t)k >= nc) break;
      fb[(uint64_t)(c0 + k + 1) * wd.lub] = (uint8_t)tok[k];
      fq[(uint64_t)(c0 + k + 1) * wd.lub] = (uint8_t)ql[k];
      count_sym(cnt, tok_fold(tok[k]));
    }
  }
  // informative rows of the final [L',31] matrix: thresh = (31 * 0.1) as usize = 3 (features.rs:558,712)
  J.sup_flag[wd.row_off + r] = supported_from_counts(cnt, (uint32_t)((double)HERRO_ROWS * 0.1)) ? 1 : 0;
}

// =====================================================================================================
// k_sup_compact — one workgroup per window: ordered informative-position list
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_sup_compact(JobDev J) {
  __shared__ uint32_t s_wave[NT / 64];
  const uint32_t w = blockIdx.x;
  const WinDesc wd = J.win[w];
  const uint32_t Lf = J.win_Lf[w];
  const uint8_t* flag = J.sup_flag + wd.row_off;
  const uint32_t ch = (Lf + NT - 1) / NT;
  const uint32_t a = min(threadIdx.x * ch, Lf), b = min(a + ch, Lf);
  uint32_t local = 0;
  for (uint32_t r = a; r < b; r++) local += flag[r];
  uint32_t tot;
  uint32_t k = block_scan(local, &tot, s_wave);
  if (local) {
    const uint32_t* rowmap = J.rowmap2 + wd.row_off;
    for (uint32_t r = a; r < b; r++)
      if (flag[r]) {
        J.sup_row[wd.row_off + k] = r;
        J.sup_pi[wd.row_off + k] = rowmap[r];  // pos | ins << 16 (SupportedPos, features.rs:896-900)
        k++;
      }
  }
  if (threadIdx.x == 0) J.win_nsup[w] = tot;
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
  KT_BEGIN(tm, "pass1_tiles", st);
  hipLaunchKernelGGL(k_pass1_tiles, dim3(J.n_tiles), dim3(NT), 0, st, J);
  KT_END(tm, st);
  KT_BEGIN(tm, "select_layout", st);
  hipLaunchKernelGGL(k_select_layout, dim3(J.n_win), dim3(NT), 0, st, J);
  KT_END(tm, st);
  KT_BEGIN(tm, "final_tiles", st);
  hipLaunchKernelGGL(k_final_tiles, dim3(J.n_tiles), dim3(NT), 0, st, J);
  KT_END(tm, st);
  KT_BEGIN(tm, "sup_compact", st);
  hipLaunchKernelGGL(k_sup_compact, dim3(J.n_win), dim3(NT), 0, st, J);
  KT_END(tm, st);
}

}  // namespace herro
