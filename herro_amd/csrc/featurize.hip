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
static constexpr int EVCAP = 128;    // overlaps whose insertion events are flattened through LDS

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

// the same for a u64 (several packed counters scanned at once: one pair of barriers instead of several)
__device__ __forceinline__ uint64_t block_scan64(uint64_t v, uint64_t* total, uint64_t* s_wave64 /*[NT/64]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();
  if (lane == 63) s_wave64[wave] = inc;
  __syncthreads();
  uint64_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) {
    const uint64_t x = s_wave64[w];
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
// k_ow_stats — one WAVE per overlap-window (4 per workgroup), no workgroup barriers
// =====================================================================================================
// The work of one overlap-window is small (~100 ops, <= 8192 positions) and is a chain of dependent
// global loads; one wave each keeps 4x more of those chains in flight per CU than one workgroup each.
static constexpr int OWCAP = 192;  // ops staged in LDS per wave (more: tables are read back from global)
static constexpr int TWCAP = HERRO_MAX_WINDOW / 32 + 2;  // staged target 2-bit words per wave
static constexpr int QWCAP = 192;                         // staged query 2-bit words per wave (6k bases; more: read from global)

__device__ __forceinline__ uint64_t wave_scan64(uint64_t v, uint64_t* total) {  // exclusive, within the wave
  const int lane = threadIdx.x & 63;
  uint64_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  *total = __shfl(inc, 63, 64);
  return inc - v;
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__host__ __device__ inline size_t ow_stats_lds_per_wave(uint32_t window_size) {
  return (((size_t)(window_size / 32 + 2) + QWCAP) * 8 + (size_t)3 * OWCAP * 4 + (size_t)(window_size / 32 + 1) * 4 + 15) & ~(size_t)15;
}

__global__ __launch_bounds__(NT) void k_ow_stats(JobDev J) {
  // per-wave LDS, sized for the job's window size (dynamic): [tw: ntw_cap u64][qw: QWCAP u64][op, t, q: OWCAP u32 each][bm: n_bw u32]
  extern __shared__ __attribute__((aligned(16))) unsigned char ow_smem[];
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave-uniform: the descriptor arrives by scalar loads
  const uint32_t o = blockIdx.x * 4 + wave;
  if (o >= J.n_ow) return;
  const uint32_t ntw_cap = J.window_size / 32 + 2;
  unsigned char* base_w = ow_smem + (size_t)wave * ow_stats_lds_per_wave(J.window_size);
  uint64_t* s_tw = reinterpret_cast<uint64_t*>(base_w);
  uint64_t* s_qw = s_tw + ntw_cap;
  uint32_t* s_op = reinterpret_cast<uint32_t*>(s_qw + QWCAP);
  uint32_t* s_t = s_op + OWCAP;
  uint32_t* s_q = s_t + OWCAP;
  uint32_t* s_bm = s_q + OWCAP;
  const OwDesc d = J.ow[o];  // carries the window's and the reads' offsets: no further lookups before the data
  const uint32_t* ops = J.ops + d.op_begin;
  uint32_t* op_t = J.op_t + d.scr_off;
  uint32_t* op_q = J.op_q + d.scr_off;
  uint32_t* ins_ev = J.ins_ev + d.scr_off;
  uint4* md = J.md + d.scr_off;
  const uint32_t cnt = d.op_cnt;
  const uint32_t off = d.tstart - d.wtstart;
  const bool in_lds = cnt <= OWCAP;

  for (uint32_t i = lane; i < J.n_bw; i += 64) s_bm[i] = 0;
  // 2-bit words of the target stretch and of the query stretch (for the accuracy pass below) and the ops of the slice:
  // EVERY global load of this wave is issued here, before anything waits.  (As loops with run-time trip counts —
  // three passes over the ops with two loads each, up to five + three passes over the words — they were ~13 memory
  // round trips in sequence, ~26 us in the life of a wave whose arithmetic takes two or three.)
  const uint64_t t_woff = d.t_woff;
  const uint64_t q_woff = d.q_woff;
  const uint32_t tw0 = d.tstart >> 5, ntw = ((d.wtstart + d.wlen) >> 5) - tw0 + 2;
  const uint32_t qw0 = d.qbeg >> 5, nqw = ((d.qbeg + d.qlen) >> 5) - qw0 + 2;
  const bool q_lds = nqw <= QWCAP;
  constexpr int TWI = (TWCAP + 63) / 64, QWI = (QWCAP + 63) / 64, OPI = (OWCAP + 63) / 64;
  uint64_t tw_r[TWI], qw_r[QWI];
  uint32_t op_r[OPI];
#pragma unroll
  for (int i = 0; i < TWI; i++) {
    const uint32_t idx = lane + 64u * i;
    tw_r[i] = (idx < ntw && t_woff + tw0 + idx <= J.read_n_words) ? J.read_words[t_woff + tw0 + idx] : 0ull;
  }
#pragma unroll
  for (int i = 0; i < QWI; i++) {
    const uint32_t idx = lane + 64u * i;
    qw_r[i] = (q_lds && idx < nqw && q_woff + qw0 + idx <= J.read_n_words) ? J.read_words[q_woff + qw0 + idx] : 0ull;
  }
#pragma unroll
  for (int i = 0; i < OPI; i++) {
    const uint32_t k = lane + 64u * i;
    op_r[i] = (in_lds && k < cnt) ? ops[k] : 0u;
  }
#pragma unroll
  for (int i = 0; i < TWI; i++) {
    const uint32_t idx = lane + 64u * i;
    if (idx < ntw) s_tw[idx] = tw_r[i];
  }
  if (q_lds) {
#pragma unroll
    for (int i = 0; i < QWI; i++) {
      const uint32_t idx = lane + 64u * i;
      if (idx < nqw) s_qw[idx] = qw_r[i];
    }
  }
  uint32_t carry_t = 0, carry_q = 0, carry_i = 0, carry_m = 0, isum = 0, dsum = 0, longindel = 0;
  auto op_step = [&](uint32_t base, uint32_t op_in, uint32_t nxt_in, bool have_regs) {
    const uint32_t k = base + lane;
    uint32_t tadv = 0, qadv = 0, is_i = 0, op = 0;
    if (k < cnt) {
      op = have_regs ? op_in : ops[k];
      const uint32_t ty = op_type(op);
      const uint32_t e = eff_len(op, k, cnt, d.start_off, d.end_off);
      if (ty != OP_M && op_len(op) > 50u) longindel = 1;  // untrimmed length (features.rs:317)
      if (ty != OP_I) tadv = e;
      if (ty != OP_D) qadv = e;
      if (ty == OP_I) { isum += e; is_i = 1; }
      if (ty == OP_D) dsum += e;
    }
    const uint32_t is_md = (k < cnt && !is_i) ? 1u : 0u;
    // four prefix sums in two 64-bit wave scans
    uint64_t tot_tq, tot_im;
    const uint64_t ex_tq = wave_scan64((uint64_t)tadv | ((uint64_t)qadv << 32), &tot_tq);
    const uint64_t ex_im = wave_scan64((uint64_t)is_i | ((uint64_t)is_md << 32), &tot_im);
    if (k < cnt) {
      const uint32_t t = carry_t + (uint32_t)ex_tq, q = carry_q + (uint32_t)(ex_tq >> 32);
      if (in_lds) { s_op[k] = op; s_t[k] = t; s_q[k] = q; }
      else { op_t[k] = t; op_q[k] = q; }  // only this kernel reads them back (accuracy pass of very long CIGARs)
      // insertion behind window position off+t-1 (features.rs:77); t >= 1: a slice never starts with I
      if (is_i) ins_ev[carry_i + (uint32_t)ex_im] = ((off + t - 1u) & 0xffffu) | (op_len(op) << 16);
      if (is_md) {
        // compact M/D op table + bitmap of op starts: the tile kernels find the op covering a target
        // position with one popcount (rank) instead of a search.  An M/D op is followed by at most one
        // insertion (the host rejects consecutive I ops); I ops are never trimmed by window offsets.
        const uint32_t nxt = (k + 1 < cnt) ? (have_regs ? nxt_in : ops[k + 1]) : 0u;
        const uint32_t ins_len = (k + 1 < cnt && op_type(nxt) == OP_I) ? op_len(nxt) : 0u;
        md[carry_m + (uint32_t)(ex_im >> 32)] = make_uint4(t, q, tadv | (op_type(op) == OP_M ? 0x80000000u : 0u), ins_len);
        atomicOr(&s_bm[t >> 5], 1u << (t & 31u));
      }
    }
    carry_t += (uint32_t)tot_tq;
    carry_q += (uint32_t)(tot_tq >> 32);
    carry_i += (uint32_t)tot_im;
    carry_m += (uint32_t)(tot_im >> 32);
  };
  if (in_lds) {
#pragma unroll
    for (int i = 0; i < OPI; i++) {
      if (64u * i < cnt) {   // wave-uniform
        // the op after lane l's: lane l+1's, or — for the last lane — the first op of the next register
        const uint32_t same = __shfl_down(op_r[i], 1, 64);
        const uint32_t next0 = i + 1 < OPI ? __shfl(op_r[i + 1 < OPI ? i + 1 : i], 0, 64) : 0u;
        op_step(64u * i, op_r[i], lane == 63 ? next0 : same, true);
      }
    }
  } else {
    for (uint32_t base = 0; base < cnt; base += 64) op_step(base, 0u, 0u, false);
  }
  const uint32_t t_total = carry_t;
  // this wave's LDS / global writes are read back below by the same wave.  LDS operations of one wave execute in order, so
  // for the staged tables a wavefront-scope fence (a compiler barrier, no wait) is enough; only the rare overlap whose ops
  // did not fit the LDS slots reads op_t / op_q back from GLOBAL memory and has to wait for its stores.  (A workgroup-scope
  // release here made every wave wait for the acknowledgement of its md / ins_ev stores — a full memory round trip in the
  // middle of a latency-bound kernel.)
  if (in_lds) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  const uint32_t* po = in_lds ? s_op : ops;
  const uint32_t* pt = in_lds ? s_t : op_t;
  const uint32_t* pq = in_lds ? s_q : op_q;

  // bitmap words + cumulative popcounts (rank directory), and the column header
  {
    uint2* bm = J.bm + (uint64_t)o * J.n_bw;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < J.n_bw; base += 64) {
      const uint32_t i = base + lane;
      const uint32_t bits = i < J.n_bw ? s_bm[i] : 0u;
      uint64_t tot;
      const uint32_t ex = (uint32_t)wave_scan64(__popc(bits), &tot);
      if (i < J.n_bw) bm[i] = make_uint2(bits, carry + ex);
      carry += (uint32_t)tot;
    }
    if (lane == 0) {
      ColHdr h;
      h.off = (int32_t)off;
      h.t_total = t_total;
      h.strand = d.strand;
      h.cls = d.cls;
      // stored index of alignment-orientation base q: sbase + sdir * q (features.rs:97-108,128-153)
      h.sbase = d.strand ? (int32_t)(d.qbeg + d.qlen - 1u) : (int32_t)d.qbeg;
      h.sdir = d.strand ? -1 : 1;
      h.md_off = d.scr_off;
      h.n_md = carry_m;
      h.q_woff = d.q_woff;
      h.qual_off = d.q_qual_off;
      J.chdr[o] = h;
    }
  }

  // accuracy: matches / mismatches over M ops (features.rs:650-665), 16 target bases per step
  const uint64_t* twp = s_tw;                              // staged: index relative to word tw0
  const uint64_t* qwp = q_lds ? s_qw : J.read_words + q_woff + qw0;
  uint32_t m = 0, s = 0;
  for (uint32_t u0 = lane * 16u; u0 < t_total; u0 += 64u * 16u) {
    const uint32_t u1 = min(u0 + 16u, t_total);
    uint32_t k = find_op(pt, cnt, u0), u = u0;
    while (u < u1) {
      const uint32_t op = po[k], ty = op_type(op);
      if (ty == OP_I) { k++; continue; }
      const uint32_t t0 = pt[k], oend = t0 + eff_len(op, k, cnt, d.start_off, d.end_off);
      const uint32_t se = min(u1, oend);
      if (ty == OP_M) {
        const uint32_t n = se - u, q = pq[k] + (u - t0);
        const uint32_t tb = get16(twp, 0, d.tstart + u - (tw0 << 5));
        uint32_t qb;
        if (d.strand == 0) {
          qb = get16(qwp, 0, d.qbeg + q - (qw0 << 5));
        } else {  // alignment-orientation bases q..q+n-1 = complement of stored bases s_lo+n-1 .. s_lo
          const uint32_t s_lo = d.qbeg + d.qlen - q - n;
          qb = ~(rev_pairs(get16(qwp, 0, s_lo - (qw0 << 5))) >> (32u - 2u * n));
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
  const uint64_t t1 = wave_sum64((uint64_t)m | ((uint64_t)s << 32));
  const uint64_t t2 = wave_sum64((uint64_t)isum | ((uint64_t)dsum << 32));
  longindel = __ballot(longindel != 0u) != 0ull ? 1u : 0u;
  if (lane == 0) {
    m = (uint32_t)t1; s = (uint32_t)(t1 >> 32); isum = (uint32_t)t2; dsum = (uint32_t)(t2 >> 32);
    J.ow_keep[o] = longindel ? 0 : 1;
    // (m as f32) / ((m+s+i+d) as f32), correctly rounded (features.rs:678)
    J.ow_acc[o] = __fdiv_rn((float)m, (float)(m + s + isum + dsum));
    J.ow_ttotal[o] = t_total;
    J.ins_cnt[o] = carry_i;
  }
}

// The per-position max-insertion array is walked 16 consecutive positions per thread (write_layout): unpadded, the lanes
// of a wave hit two LDS banks (stride 16 words) — PMC: 77 % of k_select_layout's LDS cycles were bank conflicts.  One pad
// word per 16 positions makes the stride 17.
__host__ __device__ __forceinline__ uint32_t mi_idx(uint32_t p) { return p + (p >> 4); }

// Scatter-max the insertion events of `n` overlaps (given by ow index list) into s_mi (LDS).
// Events of all listed overlaps are flattened over the threads through an LDS prefix of counts.
__device__ __forceinline__ void scatter_max_ins(const JobDev& J, const uint32_t* ow_list, uint32_t n,
                                                uint32_t win_len, uint32_t* s_mi, uint32_t* s_pref,
                                                uint32_t* s_wave) {
  __shared__ uint32_t s_scr[EVCAP];   // first insertion event of each listed overlap (OwDesc::scr_off), staged with the counts
  for (uint32_t base = 0; base < n; base += EVCAP) {
    const uint32_t nn = min((uint32_t)EVCAP, n - base);
    uint32_t carry = 0;
    for (uint32_t b2 = 0; b2 < nn; b2 += NT) {
      const uint32_t i = b2 + threadIdx.x;
      const uint32_t c = i < nn ? J.ins_cnt[ow_list[base + i]] : 0u;
      if (i < nn) s_scr[i] = J.ow[ow_list[base + i]].scr_off;     // same round trip as the count
      uint32_t tot;
      const uint32_t ex = block_scan(c, &tot, s_wave);
      if (i < nn) s_pref[i] = carry + ex;
      carry += tot;
    }
    __syncthreads();
    const uint32_t E = carry;
    // four events per thread and iteration: indices from LDS first, then the four loads together, then the atomics
    // (one event at a time was slot -> descriptor -> event: two dependent loads per event, ~3 events per thread in a row)
    for (uint32_t e0 = threadIdx.x; e0 < E; e0 += 4 * NT) {
      uint32_t ev[4];
      bool live[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t e = e0 + u * NT;
        live[u] = e < E;
        const uint32_t ec = live[u] ? e : e0;
        uint32_t lo = 0, hi = nn;  // largest i with s_pref[i] <= ec
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s_pref[mid] <= ec) lo = mid; else hi = mid;
        }
        ev[u] = J.ins_ev[s_scr[lo] + (ec - s_pref[lo])];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t p = ev[u] & 0xffffu;
        if (live[u] && p < win_len) atomicMax(&s_mi[mi_idx(p)], ev[u] >> 16);
      }
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
  for (uint32_t p = p0; p < p1; p++) local += 1u + s_mi[mi_idx(p)];
  uint32_t tot;
  uint32_t r = block_scan(local, &tot, s_wave);
  for (uint32_t p = p0; p < p1; p++) {
    row_of_pos[p] = r;
    const uint32_t v = 1u + s_mi[mi_idx(p)];
    for (uint32_t j = 0; j < v; j++)
      if (r + j < lub) rowmap[r + j] = p | (j << 16);
    r += v;
  }
  if (threadIdx.x == 0) row_of_pos[win_len] = tot;
  return tot;
}

// =====================================================================================================
// k_win_rank — one workgroup per window: stable rank of kept overlaps by descending accuracy
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_win_rank(JobDev J) {
  __shared__ uint32_t s_wave[NT / 64];
  const uint32_t w = blockIdx.x;
  const WinDesc wd = J.win[w];
  const uint32_t n = wd.ow_cnt;
  uint32_t kept_local = 0;
  for (uint32_t i = threadIdx.x; i < n; i += NT) {  // sort_by_key(-acc), stable (features.rs:386-409)
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
  const uint32_t n_kept = block_sum(kept_local, s_wave);
  if (threadIdx.x == 0) J.win_nkept[w] = n_kept;
  // match / mismatch tallies start from zero (pass 1, the next kernel, accumulates into them)
  for (uint32_t i = blockIdx.x * NT + threadIdx.x; i < 2 * J.n_cls; i += gridDim.x * NT) J.nd[i] = 0;
}

// =====================================================================================================
// k_pass1_pos — informative target positions + match/mismatch tallies, in POSITION space, bit-sliced
// =====================================================================================================
// What pass 1 of the reference leaves behind is only the per-query tallies (features.rs:461-500), and
// those are taken at informative rows that carry a target base — insertion rows are skipped
// (features.rs:489-491).  A target-base row is a target position, so pass 1 needs no row layout at all:
// a lane owns 16 consecutive positions as bit-vectors, each column contributes its symbols as bit
// planes (query bases come from the bit-plane copy of the read store, a whole M run per shift), and the
// per-position symbol counts over all columns live in bit-sliced counters (features.rs:681-722).
__device__ __forceinline__ uint32_t plane_bits(const uint32_t* __restrict__ pl, uint64_t woff, int32_t s) {
  // 32 consecutive plane bits starting at base index s (s may be negative: those bits read 0)
  if (s < 0) return s <= -32 ? 0u : (pl[woff] << (uint32_t)(-s));
  const uint32_t w = (uint32_t)s >> 5;
  return __funnelshift_r(pl[woff + w], pl[woff + w + 1], (uint32_t)s & 31u);
}

struct ColPlanes { uint32_t m, lo, hi, gap; };  // per position bit: M base present / code planes / deletion

__device__ __forceinline__ ColPlanes column_planes(const JobDev& J, const ColHdr& h, uint32_t o, uint32_t P) {
  ColPlanes out{0u, 0u, 0u, 0u};
  const int32_t u0 = (int32_t)P - h.off;
  const int32_t lo_u = max(u0, 0), hi_u = min(u0 + 16, (int32_t)h.t_total);
  if (lo_u >= hi_u) return out;
  const uint2 bw = J.bm[(uint64_t)o * J.n_bw + ((uint32_t)lo_u >> 5)];
  uint32_t r = bw.y + __popc(bw.x & (0xffffffffu >> (31u - ((uint32_t)lo_u & 31u)))) - 1u;
  int32_t u = lo_u;
  while (u < hi_u) {
    const uint4 e = J.md[h.md_off + r];
    const int32_t oend = (int32_t)(e.x + (e.z & 0x7fffffffu));
    const int32_t se = min(hi_u, oend);
    const uint32_t n = (uint32_t)(se - u), sh = (uint32_t)(u - u0);
    const uint32_t seg = ((1u << n) - 1u) << sh;
    if (e.z >> 31) {
      const int32_t q = (int32_t)e.y + (u - (int32_t)e.x);
      uint32_t b0, b1;
      if (h.sdir > 0) {
        b0 = plane_bits(J.read_p0, h.q_woff, h.sbase + q);
        b1 = plane_bits(J.read_p1, h.q_woff, h.sbase + q);
      } else {  // alignment-orientation base k = complement of stored base (sbase - q) - k
        const int32_t s_hi = h.sbase - q;
        b0 = ~__brev(plane_bits(J.read_p0, h.q_woff, s_hi - 31));
        b1 = ~__brev(plane_bits(J.read_p1, h.q_woff, s_hi - 31));
      }
      out.lo |= (b0 << sh) & seg;
      out.hi |= (b1 << sh) & seg;
      out.m |= seg;
    } else {
      out.gap |= seg;
    }
    u = se;
    r++;
  }
  return out;
}

template <int NB>
struct SlicedCounters {
  uint32_t c[5][NB];  // A C G T *
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int s = 0; s < 5; s++)
#pragma unroll
      for (int b = 0; b < NB; b++) c[s][b] = 0;
  }
  __device__ __forceinline__ void add1(int s, uint32_t x) {  // saturating at 2^NB - 1 (>= the threshold)
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const uint32_t carry = c[s][b] & x;
      c[s][b] ^= x;
      x = carry;
    }
#pragma unroll
    for (int b = 0; b < NB; b++) c[s][b] |= x;
  }
  __device__ __forceinline__ void add(const ColPlanes& p) {
    add1(0, p.m & ~p.lo & ~p.hi);
    add1(1, p.m & p.lo & ~p.hi);
    add1(2, p.m & ~p.lo & p.hi);
    add1(3, p.m & p.lo & p.hi);
    add1(4, p.gap);
  }
  __device__ __forceinline__ uint32_t ge(int s, uint32_t thresh) const {  // positions with count >= thresh
    uint32_t gt = 0, eq = 0xffffffffu;
#pragma unroll
    for (int b = NB - 1; b >= 0; b--) {
      const uint32_t tb = ((thresh >> b) & 1u) ? 0xffffffffu : 0u;
      gt |= eq & c[s][b] & ~tb;
      eq &= ~(c[s][b] ^ tb);
    }
    return gt | eq;
  }
};

struct CellOut { uint32_t tok, qual; };
template <bool WITH_QUAL>
__device__ __forceinline__ CellOut column_cell(const JobDev& J, const ColHdr& h, uint32_t o, int32_t p, uint32_t j);
__device__ __forceinline__ void count_sym(uint64_t& c, uint32_t folded);

// LDS staging for k_pass1_pos: PG columns at a time, each with its rank directory, op table and the
// bit planes of the query stretch the overlap-window covers.
static constexpr int PG = 4;   // columns staged per pass (8 measured 1.7x slower: 42 KB of LDS, 3 workgroups per CU)
static constexpr int MDCAP = 168;  // M/D ops per column   (typical: ~100)
static constexpr int PWCAP = 160;  // plane words per column (typical: ~150 for a 4096-bp window)
static constexpr int HCAP = 32;    // column headers resident in LDS at a time (a multiple of PG; typical windows keep <= 32 overlaps)

struct PCol {
  ColHdr h;
  uint32_t ow, pw0, fb, pad;  // pw0: first staged plane word (relative to the read's words)
};

__device__ __forceinline__ uint32_t lds_plane_bits(const uint32_t* pl, uint32_t npw, int32_t s_rel) {
  // 32 plane bits starting at staged bit index s_rel (may be negative / run past the staged words -> 0s)
  if (s_rel < 0) return s_rel <= -32 ? 0u : (pl[0] << (uint32_t)(-s_rel));
  const uint32_t w = (uint32_t)s_rel >> 5;
  const uint32_t a = w < npw ? pl[w] : 0u, b = w + 1 < npw ? pl[w + 1] : 0u;
  return __funnelshift_r(a, b, (uint32_t)s_rel & 31u);
}

template <int NB>
__global__ __launch_bounds__(NT) void k_pass1_pos(JobDev J) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  PCol* pc = reinterpret_cast<PCol*>(smem);
  PCol* s_hdr = pc + PG;                       // headers of the first HCAP kept columns, fetched once per window
  uint4* s_md = reinterpret_cast<uint4*>(s_hdr + HCAP);
  uint2* s_bm = reinterpret_cast<uint2*>(s_md + PG * MDCAP);
  uint32_t* s_p0 = reinterpret_cast<uint32_t*>(s_bm + PG * J.n_bw);
  uint32_t* s_p1 = s_p0 + PG * PWCAP;

  const uint32_t w = blockIdx.x;
  const WinDesc wd = J.win[w];
  const uint32_t n_kept = J.win_nkept[w];
  const uint32_t* slots = J.slot_ow + wd.ow_begin;
  const uint32_t ncols = 1u + (n_kept > 30u ? n_kept : 30u);  // features.rs:282
  const uint32_t thresh = (uint32_t)((double)ncols * 0.1);    // features.rs:712
  const uint64_t t_woff = J.read_word_off[wd.rid];

  // planes of staged column ci for the 16 positions starting at P (see column_planes)
  auto staged_planes = [&](uint32_t ci, uint32_t P) -> ColPlanes {
    const PCol& c = pc[ci];
    if (c.fb) return column_planes(J, c.h, c.ow, P);
    ColPlanes out{0u, 0u, 0u, 0u};
    const int32_t u0 = (int32_t)P - c.h.off;
    const int32_t lo_u = max(u0, 0), hi_u = min(u0 + 16, (int32_t)c.h.t_total);
    if (lo_u >= hi_u) return out;
    const uint2 bw = s_bm[ci * J.n_bw + ((uint32_t)lo_u >> 5)];
    uint32_t r = bw.y + __popc(bw.x & (0xffffffffu >> (31u - ((uint32_t)lo_u & 31u)))) - 1u;
    const uint32_t* p0 = s_p0 + ci * PWCAP;
    const uint32_t* p1 = s_p1 + ci * PWCAP;
    const int32_t rel = -(int32_t)(c.pw0 << 5);  // staged bit index = stored index + rel
    int32_t u = lo_u;
    while (u < hi_u) {
      const uint4 e = s_md[ci * MDCAP + r];
      const int32_t oend = (int32_t)(e.x + (e.z & 0x7fffffffu));
      const int32_t se = min(hi_u, oend);
      const uint32_t n = (uint32_t)(se - u), sh = (uint32_t)(u - u0);
      const uint32_t seg = ((1u << n) - 1u) << sh;
      if (e.z >> 31) {
        const int32_t q = (int32_t)e.y + (u - (int32_t)e.x);
        uint32_t b0, b1;
        if (c.h.sdir > 0) {
          b0 = lds_plane_bits(p0, PWCAP, c.h.sbase + q + rel);
          b1 = lds_plane_bits(p1, PWCAP, c.h.sbase + q + rel);
        } else {
          const int32_t s_hi = c.h.sbase - q;
          b0 = ~__brev(lds_plane_bits(p0, PWCAP, s_hi - 31 + rel));
          b1 = ~__brev(lds_plane_bits(p1, PWCAP, s_hi - 31 + rel));
        }
        out.lo |= (b0 << sh) & seg;
        out.hi |= (b1 << sh) & seg;
        out.m |= seg;
      } else {
        out.gap |= seg;
      }
      u = se;
      r++;
    }
    return out;
  };
  // ---- staging.  The first version fetched, for every group of PG columns, the column headers (slot -> header -> overlap
  // descriptor: three dependent loads), then the tables, with barriers in between: four exposed memory round trips per
  // group, eight groups per window, ~165 us in the life of a workgroup.  Now the headers of (up to HCAP) kept columns are
  // fetched ONCE per window, and the tables of group g+1 travel global -> registers while group g is being counted; a
  // group costs two barriers and no exposed round trip.
  auto load_hdr = [&](uint32_t slot) -> PCol {
    PCol c;
    c.ow = slots[slot];
    c.h = J.chdr[c.ow];
    const OwDesc& d = J.ow[c.ow];
    c.pw0 = d.qbeg >> 5;
    const uint32_t npw = ((d.qbeg + d.qlen) >> 5) - c.pw0 + 2;  // +1 for the funnel shift's second word
    c.fb = (c.h.n_md > MDCAP || npw > PWCAP) ? 1u : 0u;
    c.pad = 0;
    return c;
  };
  // s_hdr holds the headers of kept columns [hbase, hbase + HCAP); refilled (barriers by the caller) every HCAP columns
  auto fill_hdr = [&](uint32_t hbase) {
    const uint32_t n_h = min(n_kept - hbase, (uint32_t)HCAP);
    for (uint32_t i = threadIdx.x; i < n_h; i += NT) s_hdr[i] = load_hdr(hbase + i);
  };
  auto hdr_of = [&](uint32_t slot) -> const PCol& { return s_hdr[slot % HCAP]; };   // slot inside the resident chunk
  // per-thread share of a group's tables: thread t takes entry t of every table of every column of the group (the
  // tables are shorter than the workgroup: MDCAP, PWCAP <= NT; the rank directory has W / 32 + 1 <= 2 * NT words) — no
  // index arithmetic, the column header is uniform per load
  struct Regs { uint4 md[PG]; uint2 bm[PG][2]; uint32_t p0[PG], p1[PG]; };
  static_assert(MDCAP <= NT && PWCAP <= NT && HERRO_MAX_WINDOW / 32 + 1 <= 2 * NT, "one table entry per thread and column");
  const uint32_t t_ = threadIdx.x;
  auto prefetch = [&](uint32_t g0, uint32_t ng, Regs& r) {
#pragma unroll
    for (int ci = 0; ci < PG; ci++) {
      r.md[ci] = make_uint4(0, 0, 0, 0); r.bm[ci][0] = make_uint2(0, 0); r.bm[ci][1] = make_uint2(0, 0); r.p0[ci] = 0; r.p1[ci] = 0;
      if ((uint32_t)ci < ng) {
        const PCol& c = hdr_of(g0 + ci);
        if (!c.fb && t_ < c.h.n_md) r.md[ci] = J.md[c.h.md_off + t_];
        const uint2* bmp = J.bm + (uint64_t)c.ow * J.n_bw;
        if (t_ < J.n_bw) r.bm[ci][0] = bmp[t_];
        if (t_ + NT < J.n_bw) r.bm[ci][1] = bmp[t_ + NT];
        if (!c.fb && t_ < PWCAP && c.h.q_woff + c.pw0 + t_ < J.read_n_words + 2) {
          r.p0[ci] = J.read_p0[c.h.q_woff + c.pw0 + t_];
          r.p1[ci] = J.read_p1[c.h.q_woff + c.pw0 + t_];
        }
      }
    }
  };
  auto commit = [&](uint32_t g0, uint32_t ng, const Regs& r) {   // registers -> LDS; barriers by the caller
    if (t_ < ng) pc[t_] = hdr_of(g0 + t_);
#pragma unroll
    for (int ci = 0; ci < PG; ci++) {
      if ((uint32_t)ci < ng) {
        if (t_ < MDCAP) s_md[ci * MDCAP + t_] = r.md[ci];
        if (t_ < J.n_bw) s_bm[ci * J.n_bw + t_] = r.bm[ci][0];
        if (t_ + NT < J.n_bw) s_bm[ci * J.n_bw + t_ + NT] = r.bm[ci][1];
        if (t_ < PWCAP) { s_p0[ci * PWCAP + t_] = r.p0[ci]; s_p1[ci * PWCAP + t_] = r.p1[ci]; }
      }
    }
  };

  for (uint32_t base = 0; base < wd.win_len; base += NT * 16u) {  // wave-uniform trip count (shuffles below)
    const uint32_t P = base + threadIdx.x * 16u;
    const bool act = P < wd.win_len;
    const uint32_t npos = act ? min(16u, wd.win_len - P) : 0u, vmask = (1u << npos) - 1u;
    const uint32_t tlo = act ? plane_bits(J.read_p0, t_woff, (int32_t)(wd.tstart + P)) & vmask : 0u;
    const uint32_t thi = act ? plane_bits(J.read_p1, t_woff, (int32_t)(wd.tstart + P)) & vmask : 0u;
    SlicedCounters<NB> cnt;
    cnt.clear();
    cnt.add(ColPlanes{vmask, tlo, thi, 0u});  // the target column: always a base on these rows
    Regs rg;
    for (uint32_t g0 = 0; g0 < n_kept; g0 += PG) {
      const uint32_t ng = min((uint32_t)PG, n_kept - g0);
      if (g0 % HCAP == 0) {     // first group of a header chunk (for <= 32 kept overlaps: once per window)
        __syncthreads();        // nobody reads the previous chunk's headers any more
        fill_hdr(g0);
        __syncthreads();
        prefetch(g0, ng, rg);
      }
      __syncthreads();          // everybody is done with the previous group's tables (and with the tally scratch below)
      commit(g0, ng, rg);
      __syncthreads();
      if (g0 + PG < n_kept && (g0 + PG) % HCAP != 0) prefetch(g0 + PG, min((uint32_t)PG, n_kept - g0 - PG), rg);   // in flight under the counting
      for (uint32_t ci = 0; ci < ng; ci++) {
        ColPlanes cp = act ? staged_planes(ci, P) : ColPlanes{0u, 0u, 0u, 0u};
        cp.m &= vmask; cp.gap &= vmask;
        cnt.add(cp);
      }
    }
    // informative: at least two symbols reach the threshold
    uint32_t one = 0, two = 0;
#pragma unroll
    for (int s5 = 0; s5 < 5; s5++) {
      const uint32_t g = cnt.ge(s5, thresh) & vmask;
      two |= one & g;
      one |= g;
    }
    const uint32_t sup = thresh == 0 ? vmask : two;  // thresh 0 cannot happen (ncols >= 31) but stay exact
    // tallies (features.rs:478-498): every kept column is scored at every informative position;
    // anything but the target's base ('.', '*', '#', other bases) is a mismatch.  Informative
    // positions are rare (~0.4 %), so they are listed in LDS (the staging area is free now) and only
    // those (position, column) cells are evaluated, one per thread, straight from the rank directory.
    __syncthreads();
    uint32_t* s_n = reinterpret_cast<uint32_t*>(s_md);
    uint16_t* s_list = reinterpret_cast<uint16_t*>(s_n + 4);
    if (threadIdx.x == 0) *s_n = 0;
    __syncthreads();
    for (uint32_t m = sup; m; m &= m - 1u) s_list[atomicAdd(s_n, 1u)] = (uint16_t)(P + (uint32_t)__ffs(m) - 1u);
    __syncthreads();
    const uint32_t npair = *s_n * n_kept;
    // two (position, column) cells per thread and iteration, phase by phase: a cell is slot -> header -> rank word -> op
    // entry -> 2-bit word, five dependent loads from global memory, and one at a time they ran in sequence at the tail of
    // every workgroup (typically 15 positions x 32 columns = 2 cells per thread = ten round trips)
    for (uint32_t pr0 = threadIdx.x; pr0 < npair; pr0 += 2 * NT) {
      bool live[2], inr[2];
      uint32_t pos[2], o[2], uu[2], tw[2];
      ColHdr h[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const uint32_t pr = pr0 + u * NT;
        live[u] = pr < npair;
        const uint32_t prc = live[u] ? pr : pr0;
        pos[u] = s_list[prc / n_kept];
        o[u] = slots[prc % n_kept];
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        h[u] = J.chdr[o[u]];
        const uint32_t ti = wd.tstart + pos[u];
        tw[u] = (uint32_t)(J.read_words[t_woff + (ti >> 5)] >> ((ti & 31u) << 1)) & 3u;   // read_code
      }
      uint2 bw[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const uint32_t uu_raw = (uint32_t)((int32_t)pos[u] - h[u].off);
        inr[u] = uu_raw < h[u].t_total;
        uu[u] = inr[u] ? uu_raw : 0u;
        bw[u] = J.bm[(uint64_t)o[u] * J.n_bw + (uu[u] >> 5)];
      }
      uint4 e[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const uint32_t rank = bw[u].y + __popc(bw[u].x & (0xffffffffu >> (31u - (uu[u] & 31u))));
        e[u] = J.md[h[u].md_off + (inr[u] && h[u].t_total ? rank - 1u : 0u)];
      }
      uint64_t word[2];
      uint32_t si[2];
      bool isbase[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        isbase[u] = inr[u] && (e[u].z >> 31) != 0;                               // j == 0: M -> base, D -> gap
        const uint32_t q = e[u].y + (uu[u] - e[u].x);
        si[u] = isbase[u] ? (uint32_t)(h[u].sbase + h[u].sdir * (int32_t)q) : 0u;
        word[u] = J.read_words[h[u].q_woff + (si[u] >> 5)];
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const uint32_t code = (uint32_t)(word[u] >> ((si[u] & 31u) << 1)) & 3u;
        // folded symbol of the cell (tok_fold of column_cell's token): base -> its forward-strand code, gap -> 4, '.' -> 10
        const uint32_t folded = !inr[u] ? (uint32_t)TOK_NONE : (isbase[u] ? (h[u].strand ? (code ^ 3u) : code) : 4u);
        if (live[u]) atomicAdd(&J.nd[2 * (uint64_t)h[u].cls + (folded == tw[u] ? 0 : 1)], 1u);
      }
    }
  }
}

// ---- one pileup cell, lane = row, no search ------------------------------------------------------------
// Consecutive lanes own consecutive pileup rows: loads of bitmap words, op entries, query bases and
// qualities are coalesced or broadcast, and there is no divergent control flow.  The op covering
// target-relative position u is entry rank(u)-1 of the overlap's compact M/D table, with
// rank(u) = cum[u>>5] + popc(bits[u>>5] & mask(u&31))  (bitmap of op starts, built by k_ow_stats).
// CellOut: tok = token (inference.rs:23-31), qual = phred+33 (33 '!' where the cell holds no base)

template <bool WITH_QUAL>
__device__ __forceinline__ CellOut column_cell(const JobDev& J, const ColHdr& h, uint32_t o, int32_t p, uint32_t j) {
  const uint32_t uu_raw = (uint32_t)(p - h.off);
  const bool inr = uu_raw < h.t_total;  // also false for p < off (wraps)
  const uint32_t uu = inr ? uu_raw : 0u;
  const uint2 bw = J.bm[(uint64_t)o * J.n_bw + (uu >> 5)];
  const uint32_t rank = bw.y + __popc(bw.x & (0xffffffffu >> (31u - (uu & 31u))));
  const uint4 e = J.md[h.md_off + (inr && h.t_total ? rank - 1u : 0u)];  // t_beg, q_beg, len | M<<31, ins_len
  const bool is_m = (e.z >> 31) != 0;
  const uint32_t len = e.z & 0x7fffffffu;
  const bool last = (uu + 1u == e.x + len);
  // j == 0: M -> query base, D -> gap.  j > 0 (insertion slot j-1 behind u): base iff u is the op's last
  // target base and the following insertion is at least j long (features.rs:173-231)
  const bool isbase = inr && (j == 0 ? is_m : (last && e.w >= j));
  const uint32_t q = j == 0 ? e.y + (uu - e.x) : e.y + (is_m ? len : 0u) + (j - 1u);
  CellOut out;
  out.qual = 33;
  const uint32_t si = isbase ? (uint32_t)(h.sbase + h.sdir * (int32_t)q) : 0u;
  const uint64_t word = J.read_words[h.q_woff + (si >> 5)];
  if (WITH_QUAL) {
    const uint32_t qv = J.read_qual[h.qual_off + si];
    out.qual = isbase ? qv : 33u;
  }
  const uint32_t code = (uint32_t)(word >> ((si & 31u) << 1)) & 3u;
  const uint32_t base_tok = h.strand ? 5u + (code ^ 3u) : code;              // reverse: complement, lower case
  const uint32_t gap_tok = h.strand ? (uint32_t)TOK_GAP_R : (uint32_t)TOK_GAP_F;
  out.tok = !inr ? (uint32_t)TOK_NONE : (isbase ? base_tok : gap_tok);
  return out;
}

// Symbol counters A,C,G,T,* packed as five 12-bit fields (a runtime-indexed register array would
// live in scratch memory); the host caps overlaps per window at 4000.
__device__ __forceinline__ void count_sym(uint64_t& c, uint32_t folded) {
  c += folded < 5u ? 1ull << (12u * folded) : 0ull;
}
// informative row test (features.rs:681-722): >= 2 symbols with count >= thresh.
__device__ __forceinline__ bool supported_from_counts(uint64_t c, uint32_t thresh) {
  uint32_t ns = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) ns += (((uint32_t)(c >> (12 * k)) & 0xfffu) >= thresh) ? 1u : 0u;
  return ns >= 2;
}

// ---- LDS-staged columns of one row tile ---------------------------------------------------------------
// Everything a 256-row tile needs from a column — header, bitmap words, M/D op entries, the 2-bit
// words and the quality bytes of the query stretch it can touch — is copied into LDS by all threads
// at once (a handful of dependent global round trips per workgroup instead of per cell); the per-cell
// loop then reads LDS only.  Columns whose stretch does not fit the staging slots (very dense CIGARs,
// huge insertions) fall back to column_cell() on global memory.
static constexpr int NCOL = 32;   // columns per staging pass
static constexpr int TLD = HERRO_TILE + 4;  // byte-tile row stride for the output transpose (bank spread)
static constexpr int BMW = 10;    // bitmap words per column   (tile spans <= 256 positions -> <= 9 words)
static constexpr int MDS = 16;    // M/D op entries per column (more: the column takes the global-memory path)
static constexpr int WW = 14;     // 2-bit words per column    (<= 448 query bases)
static constexpr int QB = 480;    // quality bytes per column  (multiple of 4)

struct TCol {
  ColHdr h;
  uint32_t ow;      // 0xffffffff: padding column
  uint32_t w0;      // first staged bitmap word
  uint32_t r0;      // md entry staged in slot 0
  uint32_t word0;   // first staged 2-bit word (index into the query read's words)
  uint64_t qg0;     // 4-aligned index into read_qual of the first staged quality byte
  uint32_t fb;      // 1: stretch does not fit, use the global path
  uint32_t nmd, nw, nq; // staged entries / words / quality dwords
};

template <bool WITH_QUAL>
struct TileLds {
  TCol col[NCOL];
  uint2 bm[NCOL * BMW];
  uint4 md[NCOL * MDS];
  uint64_t words[NCOL * WW];
  uint32_t quals[WITH_QUAL ? NCOL * QB / 4 : 1];
};

// cell of staged column c at window position p / insertion ordinal j — LDS only (see column_cell)
template <bool WITH_QUAL>
__device__ __forceinline__ CellOut staged_cell(const JobDev& J, const TileLds<WITH_QUAL>& S, uint32_t c, int32_t p,
                                               uint32_t j) {
  const TCol& t = S.col[c];
  if (t.fb) return column_cell<WITH_QUAL>(J, t.h, t.ow, p, j);  // wave-uniform, rare
  const uint32_t uu_raw = (uint32_t)(p - t.h.off);
  const bool inr = uu_raw < t.h.t_total;
  const uint32_t uu = inr ? uu_raw : (t.w0 << 5);
  const uint2 bw = S.bm[c * BMW + ((uu >> 5) - t.w0)];
  const uint32_t rank = bw.y + __popc(bw.x & (0xffffffffu >> (31u - (uu & 31u))));
  const uint4 e = S.md[c * MDS + (inr ? rank - 1u - t.r0 : 0u)];
  const bool is_m = (e.z >> 31) != 0;
  const uint32_t len = e.z & 0x7fffffffu;
  const bool last = (uu + 1u == e.x + len);
  const bool isbase = inr && (j == 0 ? is_m : (last && e.w >= j));
  const uint32_t q = j == 0 ? e.y + (uu - e.x) : e.y + (is_m ? len : 0u) + (j - 1u);
  const uint32_t si = isbase ? (uint32_t)(t.h.sbase + t.h.sdir * (int32_t)q) : (t.word0 << 5);
  const uint64_t word = S.words[c * WW + ((si >> 5) - t.word0)];
  CellOut out;
  out.qual = 33;
  if (WITH_QUAL) {
    const uint32_t bi = isbase ? (uint32_t)(t.h.qual_off + si - t.qg0) : 0u;
    const uint32_t qv = (S.quals[c * (QB / 4) + (bi >> 2)] >> ((bi & 3u) << 3)) & 0xffu;
    out.qual = isbase ? qv : 33u;
  }
  const uint32_t code = (uint32_t)(word >> ((si & 31u) << 1)) & 3u;
  const uint32_t base_tok = t.h.strand ? 5u + (code ^ 3u) : code;
  const uint32_t gap_tok = t.h.strand ? (uint32_t)TOK_GAP_R : (uint32_t)TOK_GAP_F;
  out.tok = !inr ? (uint32_t)TOK_NONE : (isbase ? base_tok : gap_tok);
  return out;
}

// =====================================================================================================
// k_select_layout — one workgroup per window
// =====================================================================================================
__global__ __launch_bounds__(NT) void k_select_layout(JobDev J) {
  __shared__ uint32_t s_wave[NT / 64];
  extern __shared__ uint32_t s_mi[];  // [mi_idx(window_size)] max insertion behind every position (dynamic, padded: see mi_idx)
  __shared__ uint32_t s_pref[EVCAP];
  __shared__ double s_score[EVCAP];
  __shared__ uint32_t s_sel[32];
  const uint32_t w = blockIdx.x;
  const WinDesc wd = J.win[w];
  const uint32_t n_kept = J.win_nkept[w];
  uint32_t* sel = J.sel_ow + (uint64_t)w * 32;
  if (threadIdx.x < 32) s_sel[threadIdx.x] = 0xffffffffu;
  for (uint32_t p = threadIdx.x; p < wd.win_len; p += NT) s_mi[mi_idx(p)] = 0;
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

// ---- per (tile, column) staging plan --------------------------------------------------------------------
// Which bitmap words / op entries / query words / quality bytes a final-row tile needs from a selected
// column follows from a chain of dependent loads (row map -> selection -> header -> rank directory -> op
// table).  Done inside the tile kernel that chain is paid per workgroup with 30 active lanes; here it is
// one thread per (tile, column), fully parallel, and the tile kernel just reads the 32-byte records.
struct __attribute__((aligned(16))) TPlan {  // 64 B
  uint32_t ow;      // 0xffffffff: padding column
  uint32_t w0, r0, word0;
  uint64_t qg0;
  uint32_t cnt;     // nmd | nw << 8 | outside << 30 | fb << 31
  uint32_t nq;
  // everything else the token kernel needs from the column header, so that its staging is plan -> data
  // (two dependent round trips) instead of plan -> header -> data
  int32_t off;
  uint32_t t_total;
  int32_t sbase;
  uint32_t sdir_strand;  // bit 0: strand, bit 1: sdir < 0
  uint32_t md_off;
  uint32_t pad0;
  uint64_t q_woff;
};
// what a tile kernel needs to know about its tile, gathered once by k_tile_plan (it was a chain of five
// dependent loads at the head of every tile workgroup)
struct __attribute__((aligned(16))) TileHdr {  // 64 B
  uint32_t w, r0, Lf, lub;
  uint64_t row_off, fin_off;
  uint64_t tgt_woff;   // first 2-bit word of the target read
  uint32_t tstart, p_lo, p_hi, pad;
  uint64_t pad2;
};

__global__ __launch_bounds__(NT) void k_tile_plan(JobDev J) {
  const uint32_t idx = blockIdx.x * NT + threadIdx.x;
  const uint32_t tile = idx >> 5, c = idx & 31u;
  if (tile >= J.n_tiles || c >= HERRO_ROWS - 1) return;
  const uint32_t w = J.tile_win[tile], r0 = J.tile_r0[tile];
  const uint32_t Lf = J.win_Lf[w];
  if (r0 >= Lf) {  // the tile lies past the window's last row (tiles are laid out for the upper bound lub)
    if (c == 0) {
      TileHdr th{};
      th.w = w; th.r0 = r0; th.Lf = 0;
      J.thdr[tile] = th;
    }
    return;
  }
  const WinDesc& wd = J.win[w];
  const uint32_t* rowmap = J.rowmap2 + wd.row_off;
  const uint32_t p_lo = rowmap[r0] & 0xffffu, p_hi = rowmap[min(r0 + (uint32_t)HERRO_TILE, Lf) - 1] & 0xffffu;
  if (c == 0) {
    TileHdr th;
    th.w = w; th.r0 = r0; th.Lf = Lf; th.lub = wd.lub;
    th.row_off = wd.row_off; th.fin_off = wd.fin_off;
    th.tgt_woff = J.read_word_off[wd.rid];
    th.tstart = wd.tstart; th.p_lo = p_lo; th.p_hi = p_hi; th.pad = 0; th.pad2 = 0;
    J.thdr[tile] = th;
  }
  TPlan t;
  t.ow = J.sel_ow[(uint64_t)w * 32 + 1 + c];
  t.w0 = 0; t.r0 = 0; t.word0 = 0; t.qg0 = 0; t.cnt = 0; t.nq = 0;
  t.off = 0; t.t_total = 0; t.sbase = 0; t.sdir_strand = 0; t.md_off = 0; t.pad0 = 0; t.q_woff = 0;
  if (t.ow != 0xffffffffu) {
    const ColHdr h = J.chdr[t.ow];
    t.off = h.off; t.t_total = h.t_total; t.sbase = h.sbase; t.sdir_strand = (h.strand ? 1u : 0u) | (h.sdir < 0 ? 2u : 0u);
    t.md_off = h.md_off; t.q_woff = h.q_woff;
    const int32_t ulo = max((int32_t)p_lo - h.off, 0);
    const int32_t uhi = min((int32_t)p_hi - h.off, (int32_t)h.t_total - 1);
    if (ulo > uhi) {
      t.cnt = 1u << 30;  // the tile lies outside the overlap: every cell is '.'
      t.t_total = 0;
    } else {
      const uint2* bm = J.bm + (uint64_t)t.ow * J.n_bw;
      const uint2 blo = bm[ulo >> 5], bhi = bm[uhi >> 5];
      const uint32_t rlo = blo.y + __popc(blo.x & (0xffffffffu >> (31 - (ulo & 31))));
      const uint32_t rhi = bhi.y + __popc(bhi.x & (0xffffffffu >> (31 - (uhi & 31))));
      t.w0 = (uint32_t)ulo >> 5;
      t.r0 = rlo - 1;
      const uint32_t nmd = rhi - rlo + 1;
      const uint4 e0 = J.md[h.md_off + rlo - 1], e1 = J.md[h.md_off + rhi - 1];
      // query stretch [q0, q1) the tile can touch, as stored indices [s0, s1]
      const uint32_t q0 = e0.y, q1 = e1.y + ((e1.z >> 31) ? (e1.z & 0x7fffffffu) : 0u) + e1.w + 1u;
      const int32_t sa = h.sbase + h.sdir * (int32_t)q0, sb = h.sbase + h.sdir * (int32_t)(q1 - 1u);
      const uint32_t s0 = (uint32_t)max(min(sa, sb), 0), s1 = (uint32_t)max(sa, sb);
      t.word0 = s0 >> 5;
      const uint32_t nw = (s1 >> 5) - t.word0 + 1;
      t.qg0 = (h.qual_off + s0) & ~3ull;  // quality bytes are copied as aligned dwords of the global array
      const uint64_t g1 = h.qual_off + s1;
      t.nq = (uint32_t)((g1 - t.qg0) >> 2) + 1u;
      const bool fb = nmd > MDS || nw > WW || g1 - t.qg0 >= QB;
      t.cnt = (fb ? 0u : (nmd | (nw << 8))) | (fb ? 1u << 31 : 0u);
    }
  }
  J.tplan[(uint64_t)tile * 32 + c] = t;
}


// Staging for k_final_tiles from the precomputed plan (see k_tile_plan): 8 threads per column, the
// column's plan and header live in registers, every loop runs over exactly what the tile needs.
template <bool WQ>
__device__ __forceinline__ void stage_from_plan(const JobDev& J, uint32_t tile, uint32_t nspan, TileLds<WQ>& S) {
  const uint32_t c = threadIdx.x >> 3, l8 = threadIdx.x & 7u;
  if (c < HERRO_ROWS - 1) {
    const TPlan p = J.tplan[(uint64_t)tile * 32 + c];
    TCol t;
    t.ow = p.ow; t.w0 = p.w0; t.r0 = p.r0; t.word0 = p.word0; t.qg0 = p.qg0; t.nq = p.nq;
    t.fb = p.cnt >> 31; t.nmd = p.cnt & 0xffu; t.nw = (p.cnt >> 8) & 0xffu;
    if (p.ow != 0xffffffffu) {
      t.h = J.chdr[p.ow];
      if ((p.cnt >> 30) & 1u) t.h.t_total = 0;
      if (t.h.t_total) {
        // all loads of a thread are independent: issue them in batches, then store to LDS
        const uint2* __restrict__ bm = J.bm + (uint64_t)t.ow * J.n_bw + t.w0;
        const uint4* __restrict__ md = J.md + t.h.md_off + t.r0;
        const uint64_t* __restrict__ wsrc = J.read_words + t.h.q_woff + t.word0;
        const uint32_t* __restrict__ qsrc = reinterpret_cast<const uint32_t*>(J.read_qual + t.qg0);  // 8 pad bytes at the end
        const uint32_t nbm = min(min(nspan + 1u, (uint32_t)BMW), J.n_bw - t.w0);
        // small tables: at most 2 items per thread each (BMW <= 16, MDS <= 24 -> 3, WW <= 16)
        uint2 vb[2];
        uint4 vm[3];
        uint64_t vw[2];
#pragma unroll
        for (int k = 0; k < 2; k++) vb[k] = (l8 + 8 * k < nbm) ? bm[l8 + 8 * k] : make_uint2(0, 0);
#pragma unroll
        for (int k = 0; k < 3; k++) vm[k] = (l8 + 8 * k < t.nmd) ? md[l8 + 8 * k] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 2; k++) vw[k] = (l8 + 8 * k < t.nw) ? wsrc[l8 + 8 * k] : 0ull;
        const uint32_t nq = (!WQ || t.fb) ? 0u : t.nq;
        for (uint32_t i0 = 0; i0 < nq; i0 += 32) {  // quality dwords: 4 per thread per round
          uint32_t vq[4];
#pragma unroll
          for (int k = 0; k < 4; k++) vq[k] = (i0 + l8 + 8 * k < nq) ? qsrc[i0 + l8 + 8 * k] : 0u;
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (i0 + l8 + 8 * k < nq) S.quals[c * (QB / 4) + i0 + l8 + 8 * k] = vq[k];
        }
#pragma unroll
        for (int k = 0; k < 2; k++) if (l8 + 8 * k < nbm) S.bm[c * BMW + l8 + 8 * k] = vb[k];
#pragma unroll
        for (int k = 0; k < 3; k++) if (l8 + 8 * k < t.nmd) S.md[c * MDS + l8 + 8 * k] = vm[k];
#pragma unroll
        for (int k = 0; k < 2; k++) if (l8 + 8 * k < t.nw) S.words[c * WW + l8 + 8 * k] = vw[k];
      }
    }
    if (l8 == 0) S.col[c] = t;
  }
  __syncthreads();
}

// =====================================================================================================
// k_final_tiles — one workgroup per 256 final rows
// =====================================================================================================
// WQ = false leaves the quality planes alone: the model only ever looks at the qualities inside the
// receptive fields of informative rows (~1.6 % of the cells; k_rf_quals fills exactly those), and the
// consensus decoder at none.  The full planes are produced on request (herro_job_window_copy).
template <bool WQ>
__global__ __launch_bounds__(NT) void k_final_tiles(JobDev J) {
  __shared__ TileLds<WQ> S;
  const uint32_t w = J.tile_win[blockIdx.x], r0 = J.tile_r0[blockIdx.x];
  const uint32_t Lf = J.win_Lf[w];
  if (r0 >= Lf || (J.dbg & 16u)) return;
  const WinDesc wd = J.win[w];
  const uint32_t* rowmap = J.rowmap2 + wd.row_off;
  const uint32_t p_lo = rowmap[r0] & 0xffffu, p_hi = rowmap[min(r0 + NT, Lf) - 1] & 0xffffu;
  if (!(J.dbg & 2u)) stage_from_plan<WQ>(J, blockIdx.x, ((p_hi - p_lo) >> 5) + 1u, S);
  const uint32_t r = r0 + threadIdx.x;
  const bool valid = r < Lf;
  const uint32_t rm = valid ? rowmap[r] : 0u;
  const int32_t p = (int32_t)(rm & 0xffffu);
  const uint32_t j = rm >> 16;
  // tokens / qualities of this row's 31 cells, packed 4 per register
  uint32_t tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ql[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t cnt = 0;
  if (valid) {
    uint32_t t = TOK_GAP_F, q = 33;
    if (j == 0) {
      t = read_code(J.read_words, J.read_word_off[wd.rid], wd.tstart + (uint32_t)p);
      if (WQ) q = J.read_qual[J.read_qual_off[wd.rid] + wd.tstart + (uint32_t)p];
    }
    tk[0] = t;
    ql[0] = q;
    count_sym(cnt, t);
#pragma unroll
    for (uint32_t c = 1; c < HERRO_ROWS; c++) {
      CellOut co;
      co.tok = TOK_NONE;  // fewer than 30 overlaps: untouched '.' / '!' columns (features.rs:522-525)
      co.qual = 33;
      if (S.col[c - 1].ow != 0xffffffffu && !(J.dbg & 1u)) co = staged_cell<WQ>(J, S, c - 1, p, j);
      tk[c >> 2] |= co.tok << ((c & 3u) << 3);
      if (WQ) ql[c >> 2] |= co.qual << ((c & 3u) << 3);
      count_sym(cnt, tok_fold(co.tok));
    }
    // informative rows of the final [L',31] matrix: thresh = (31 * 0.1) as usize = 3 (features.rs:558,712)
    J.sup_flag[wd.row_off + r] = supported_from_counts(cnt, (uint32_t)((double)HERRO_ROWS * 0.1)) ? 1 : 0;
    // majority vote of the consensus decoder (consensus.rs:178-200) from the same counts: it looks at
    // the first n_alns+1 rows of the pileup, and every row beyond those is '.', which it skips anyway.
    // Two most common symbols by a stable descending sort (ties keep A,C,G,T,* order), target tie-break.
    {
      uint32_t c5[5];
#pragma unroll
      for (int q = 0; q < 5; q++) c5[q] = (uint32_t)(cnt >> (12 * q)) & 0xfffu;
      uint32_t c0 = c5[0], i0 = 0;
#pragma unroll
      for (uint32_t q = 1; q < 5; q++) if (c5[q] > c0) { c0 = c5[q]; i0 = q; }
      uint32_t c1 = 0, i1 = 5;
      bool have = false;
#pragma unroll
      for (uint32_t q = 0; q < 5; q++)
        if (q != i0 && (!have || c5[q] > c1)) { c1 = c5[q]; i1 = q; have = true; }
      const uint32_t tb = tk[0] & 0xffu;
      J.cons_tmp[wd.row_off + r] = (uint8_t)((c0 < 2u || (c0 == c1 && (i0 == tb || i1 == tb))) ? tb : i0);
    }
  }
  // transpose through LDS (the staging area is dead now) so that the [31][rows] planes, whose row axis
  // is contiguous in HBM, are written with 16-byte stores
  __syncthreads();
  uint8_t* tb = reinterpret_cast<uint8_t*>(&S);
  uint8_t* tq = tb + HERRO_ROWS * TLD;
#pragma unroll
  for (uint32_t c = 0; c < HERRO_ROWS; c++) {
    if (J.dbg & 8u) break;
    tb[c * TLD + threadIdx.x] = (uint8_t)(tk[c >> 2] >> ((c & 3u) << 3));
    if (WQ) tq[c * TLD + threadIdx.x] = (uint8_t)(ql[c >> 2] >> ((c & 3u) << 3));
  }
  __syncthreads();
  const uint32_t nseg = (min(wd.lub - r0, (uint32_t)HERRO_TILE) + 15) / 16;
  for (uint32_t it = threadIdx.x; it < ((J.dbg & 4u) ? 0u : HERRO_ROWS * nseg); it += NT) {
    const uint32_t c = it / nseg, sg = it % nseg;
    const uint32_t* lb = reinterpret_cast<const uint32_t*>(tb + c * TLD + sg * 16);
    const uint32_t* lq = reinterpret_cast<const uint32_t*>(tq + c * TLD + sg * 16);
    const uint64_t go = wd.fin_off + (uint64_t)c * wd.lub + r0 + sg * 16;
    *reinterpret_cast<uint4*>(J.fin_b + go) = make_uint4(lb[0], lb[1], lb[2], lb[3]);
    if (WQ) *reinterpret_cast<uint4*>(J.fin_q + go) = make_uint4(lq[0], lq[1], lq[2], lq[3]);
  }
}

// =====================================================================================================
// k_final_tiles_t — the token plane, lean version (what herro_job_featurize runs)
// =====================================================================================================
// Same staging and the same arithmetic as k_final_tiles (kept for the quality planes and as an independent
// cross-check), with the per-cell instruction count cut from ~190 (113 VALU + 74 SALU + 14 LDS; the ISA of
// the generic kernel) to ~50: the column constants are pre-digested into a 32-byte descriptor read by two
// broadcast LDS loads; no branch in the cell path (padding / out-of-range columns have t_total = 0, the
// rare global-memory fallback columns are patched in a second loop); the 2-bit code comes from a byte read
// instead of 64-bit shifts; the stored index is one 24-bit multiply-add; tokens go straight into the LDS
// transpose tile with ds_write_b8 at a compile-time offset; symbol counts are 6-bit fields of one register.
struct __attribute__((aligned(16))) CDesc {
  int32_t off;        // uu = p - off
  uint32_t t_total;   // 0: every cell of this column is '.' in this tile
  uint32_t uu_clamp;  // uu used for cells outside the overlap (keeps every LDS address inside the staged data)
  int32_t bm_base;    // byte offset into S.bm of bitmap word 0 of the overlap
  int32_t md_base;    // byte offset into S.md of op entry with rank 0
  int32_t sbase_rel;  // stored index of query base q, relative to the first staged base: sbase_rel + sdir * q
  int32_t sdir;
  uint32_t tokc;      // strand ? 3 : 0 | (strand ? 5 : 0) << 8 | gap token << 16
};

// One workgroup per tile.  (A persistent, software-pipelined variant — plan of tile i+2 and data of tile i+1
// prefetched into registers under tile i's cell loop — was built and measured 1.7x SLOWER: 156 VGPRs, three
// workgroups per CU and four barriers per tile cost more than the round trips it hid.)
struct TileData {  // what one staging thread contributes to a tile
  uint2 vb[2];
  uint4 vm[3];
  uint64_t vw[2];
  uint32_t rm;
};
__device__ __forceinline__ void tile_fetch(const JobDev& J, const TileHdr& th, const TPlan& pl, uint32_t sc, uint32_t l8,
                                           TileData& d) {
  const bool act = th.r0 < th.Lf;
  const uint32_t r = th.r0 + threadIdx.x;
  d.rm = (act && r < th.Lf) ? J.rowmap2[th.row_off + r] : th.p_lo;  // rows past the window: an in-range position, unused
  const uint32_t nspan = ((th.p_hi - th.p_lo) >> 5) + 1u;
  const bool live = act && pl.ow != 0xffffffffu && !(pl.cnt >> 30) && pl.t_total != 0;
  const uint32_t tw0 = (th.tstart + th.p_lo) >> 5;
  const bool tgt = act && sc == HERRO_ROWS;  // threads 248..255 stage the target's words
  const uint32_t nbm = live ? min(min(nspan + 1u, (uint32_t)BMW), J.n_bw - pl.w0) : 0u;
  const uint32_t nmd = live ? (pl.cnt & 0xffu) : 0u;
  const uint32_t nw = live ? ((pl.cnt >> 8) & 0xffu) : (tgt ? ((th.tstart + th.p_hi) >> 5) - tw0 + 1u : 0u);
  // every load below is UNCONDITIONAL (index clamped into the live range, base pointer of element 0 for columns that
  // stage nothing): predicated loads compiled to one branch + wait each, i.e. seven memory round trips in sequence at the
  // head of every tile instead of one
  const uint2* __restrict__ bm = live ? J.bm + (uint64_t)pl.ow * J.n_bw + pl.w0 : J.bm;
  const uint4* __restrict__ md = live ? J.md + pl.md_off + pl.r0 : J.md;
  const uint64_t* __restrict__ wsrc = tgt ? J.read_words + th.tgt_woff + tw0 : (live ? J.read_words + pl.q_woff + pl.word0 : J.read_words);
#pragma unroll
  for (int k = 0; k < 2; k++) { const uint32_t i = l8 + 8 * k; d.vb[k] = bm[nbm ? min(i, nbm - 1u) : 0u]; }
#pragma unroll
  for (int k = 0; k < 3; k++) { const uint32_t i = l8 + 8 * k; d.vm[k] = md[nmd ? min(i, nmd - 1u) : 0u]; }
#pragma unroll
  for (int k = 0; k < 2; k++) { const uint32_t i = l8 + 8 * k; d.vw[k] = wsrc[nw ? min(i, nw - 1u) : 0u]; }
}

__global__ __launch_bounds__(NT) void k_final_tiles_t(JobDev J) {
  __shared__ uint2 s_bm[(HERRO_ROWS - 1) * BMW];
  __shared__ uint4 s_md[(HERRO_ROWS - 1) * MDS];
  __shared__ uint64_t s_words[HERRO_ROWS * WW];  // 30 columns + the target (slot 30)
  __shared__ CDesc s_cd[32];
  __shared__ __attribute__((aligned(16))) uint8_t s_tb[HERRO_ROWS * TLD];
  __shared__ uint32_t s_fb_ow[32];
  __shared__ uint32_t s_anyfb;
  const uint32_t sc = threadIdx.x >> 3, l8 = threadIdx.x & 7u;  // staging role: 8 threads per column
  // round trip 1: the tile header (uniform) and this thread's column plan (independent of each other)
  const uint32_t tile = blockIdx.x;
  const TileHdr th = J.thdr[tile];
  TPlan pl0;
  pl0.ow = 0xffffffffu; pl0.cnt = 0; pl0.t_total = 0; pl0.w0 = 0; pl0.r0 = 0; pl0.word0 = 0; pl0.md_off = 0; pl0.q_woff = 0;
  pl0.off = 0; pl0.sbase = 0; pl0.sdir_strand = 0;
  if (sc < HERRO_ROWS - 1) pl0 = J.tplan[(uint64_t)tile * 32 + sc];
  if (th.r0 >= th.Lf || (J.dbg & 16u)) return;
  // round trip 2: row map entry, rank directory words, op entries, 2-bit words
  TileData d;
  tile_fetch(J, th, pl0, sc, l8, d);
  const unsigned char* bm_bytes = reinterpret_cast<const unsigned char*>(s_bm);
  const unsigned char* md_bytes = reinterpret_cast<const unsigned char*>(s_md);
  const uint8_t* w_bytes = reinterpret_cast<const uint8_t*>(s_words);
  {
    const bool act = true;
    const uint32_t rm = d.rm;
    if (act) {
      // ---- staging data of this tile: registers -> LDS
      const bool fb = (pl0.cnt >> 31) != 0;
      const bool live = pl0.ow != 0xffffffffu && !(pl0.cnt >> 30) && pl0.t_total != 0;
      const bool tgt = sc == HERRO_ROWS;
      const uint32_t nspan = ((th.p_hi - th.p_lo) >> 5) + 1u;
      const uint32_t nbm = live ? min(min(nspan + 1u, (uint32_t)BMW), J.n_bw - pl0.w0) : 0u;
      const uint32_t nmd = live ? (pl0.cnt & 0xffu) : 0u;
      const uint32_t nw = live ? ((pl0.cnt >> 8) & 0xffu) : (tgt ? ((th.tstart + th.p_hi) >> 5) - ((th.tstart + th.p_lo) >> 5) + 1u : 0u);
      const uint32_t slot = tgt ? 30u : sc;
#pragma unroll
      for (int k = 0; k < 2; k++) if (l8 + 8 * k < nbm) s_bm[sc * BMW + l8 + 8 * k] = d.vb[k];
#pragma unroll
      for (int k = 0; k < 3; k++) if (l8 + 8 * k < nmd) s_md[sc * MDS + l8 + 8 * k] = d.vm[k];
#pragma unroll
      for (int k = 0; k < 2; k++) if (l8 + 8 * k < nw) s_words[slot * WW + l8 + 8 * k] = d.vw[k];
      if (threadIdx.x == 0) s_anyfb = 0;
      if (l8 == 0 && sc < HERRO_ROWS - 1) {
        CDesc cd;
        cd.off = 0; cd.t_total = 0; cd.uu_clamp = 0; cd.bm_base = (int32_t)(sc * BMW * 8); cd.md_base = (int32_t)(sc * MDS * 16);
        cd.sbase_rel = 0; cd.sdir = 1; cd.tokc = (uint32_t)TOK_GAP_F << 16;
        if (live) {
          const bool strand = (pl0.sdir_strand & 1u) != 0;
          cd.off = pl0.off;
          cd.t_total = pl0.t_total;
          cd.uu_clamp = pl0.w0 << 5;
          cd.bm_base = (int32_t)(sc * BMW * 8) - (int32_t)(pl0.w0 * 8);
          cd.md_base = (int32_t)(sc * MDS * 16) - (int32_t)((pl0.r0 + 1u) * 16);
          cd.sbase_rel = pl0.sbase - (int32_t)(pl0.word0 << 5);
          cd.sdir = (pl0.sdir_strand & 2u) ? -1 : 1;
          cd.tokc = strand ? (3u | (5u << 8) | ((uint32_t)TOK_GAP_R << 16)) : ((uint32_t)TOK_GAP_F << 16);
        }
        s_cd[sc] = cd;
        s_fb_ow[sc] = (pl0.ow != 0xffffffffu && fb) ? pl0.ow : 0xffffffffu;
      }
    }
    __syncthreads();
    if (l8 == 0 && sc < HERRO_ROWS - 1 && pl0.ow != 0xffffffffu && (pl0.cnt >> 31)) s_anyfb = 1;  // after thread 0's reset
    if (act) {
      __syncthreads();  // s_anyfb
      const uint32_t r = th.r0 + threadIdx.x;
      const bool valid = r < th.Lf;
      const int32_t p = (int32_t)(rm & 0xffffu);
      const uint32_t j = rm >> 16;
      const bool j0 = j == 0;
      const uint32_t jm1 = j - 1u;
      uint32_t cnt = 0;  // five 6-bit counters A,C,G,T,* (at most 31 columns)
      uint32_t t0tok = TOK_GAP_F;
      {
        const uint32_t si = th.tstart + (uint32_t)p - (((th.tstart + th.p_lo) >> 5) << 5);
        const uint32_t code = (w_bytes[30 * WW * 8 + (si >> 2)] >> ((si & 3u) << 1)) & 3u;
        if (j0) t0tok = code;
      }
      s_tb[threadIdx.x] = (uint8_t)t0tok;
      cnt += 1u << (6u * tok_fold(t0tok));
      // Columns are taken CG at a time, phase by phase (descriptor + rank word -> op entry -> base byte -> token): a cell is
      // a chain of four dependent LDS reads, and column after column that chain ran strictly in sequence (~4 x 64+ cycles
      // per cell with nothing else of the wave in flight); in groups, CG chains overlap.
      // (Measured and rejected: the per-column constants pre-digested by k_tile_plan and read here by scalar loads instead of
      // two broadcast LDS reads per cell: 919 -> 1030 us — scalar-load latency inside the loop and SGPR spills cost more than
      // the LDS cycles they saved, and k_tile_plan paid 20 us for writing them.)
      constexpr uint32_t CG = 5;
      static_assert((HERRO_ROWS - 1) % CG == 0, "column groups");
      if (!(J.dbg & 1u))
#pragma unroll
      for (uint32_t c0 = 1; c0 < HERRO_ROWS; c0 += CG) {
        uint4 d1[CG], e[CG];
        uint2 bw[CG];
        uint32_t uu[CG], byte[CG];
        bool inr[CG], isb[CG];
#pragma unroll
        for (uint32_t g = 0; g < CG; g++) {
          const uint4 d0 = *reinterpret_cast<const uint4*>(&s_cd[c0 + g - 1]);
          d1[g] = *(reinterpret_cast<const uint4*>(&s_cd[c0 + g - 1]) + 1);
          const uint32_t uu_raw = (uint32_t)(p - (int32_t)d0.x);
          inr[g] = uu_raw < d0.y;
          uu[g] = inr[g] ? uu_raw : d0.z;
          bw[g] = *reinterpret_cast<const uint2*>(bm_bytes + (int32_t)d0.w + (int32_t)((uu[g] >> 5) << 3));
        }
#pragma unroll
        for (uint32_t g = 0; g < CG; g++) {
          const uint32_t rank = __popc(bw[g].x & ~(0xfffffffeu << (uu[g] & 31u))) + bw[g].y;
          e[g] = *reinterpret_cast<const uint4*>(md_bytes + (int32_t)d1[g].x + (int32_t)(rank << 4));
        }
#pragma unroll
        for (uint32_t g = 0; g < CG; g++) {
          // all selects, no short-circuit: booleans are combined bitwise so that nothing here becomes a branch
          const uint32_t m_bit = e[g].z >> 31;                      // 1: M op, 0: D op
          const uint32_t len = e[g].z & 0x7fffffffu;
          const uint32_t lenm = m_bit ? len : 0u;
          const bool last = (uu[g] + 1u == e[g].x + len);
          const bool ins_ok = last & (e[g].w >= j);                 // insertion slot j-1 behind uu exists in this read
          isb[g] = inr[g] & (j0 ? (m_bit != 0u) : ins_ok);
          const uint32_t q = e[g].y + (j0 ? uu[g] - e[g].x : lenm + jm1);
          const uint32_t si = isb[g] ? (uint32_t)(__mul24((int32_t)d1[g].z, (int32_t)q) + (int32_t)d1[g].y) : 0u;
          byte[g] = w_bytes[(c0 + g - 1) * WW * 8 + (si >> 2)] >> ((si & 3u) << 1);
        }
#pragma unroll
        for (uint32_t g = 0; g < CG; g++) {
          const uint32_t f = (byte[g] & 3u) ^ (d1[g].w & 0xffu);
          const uint32_t tokb = f + ((d1[g].w >> 8) & 0xffu);
          const uint32_t tokg = d1[g].w >> 16;
          const uint32_t tok = inr[g] ? (isb[g] ? tokb : tokg) : (uint32_t)TOK_NONE;
          s_tb[(c0 + g) * TLD + threadIdx.x] = (uint8_t)tok;
          const uint32_t one = inr[g] ? 1u : 0u;
          cnt += one << (6u * (isb[g] ? f : 4u));
        }
      }
      if (s_anyfb) {  // block-uniform, rare: columns whose stretch did not fit the staging slots
        for (uint32_t c = 1; c < HERRO_ROWS; c++) {
          const uint32_t ow = s_fb_ow[c - 1];
          if (ow == 0xffffffffu) continue;
          const uint32_t tok = column_cell<false>(J, J.chdr[ow], ow, p, j).tok;
          s_tb[c * TLD + threadIdx.x] = (uint8_t)tok;
          const uint32_t f = tok_fold(tok);
          cnt += (f < 5u ? 1u : 0u) << (6u * min(f, 4u));
        }
      }
      uint32_t sup_v = 0, cons_v = 0;
      if (valid) {
        uint32_t c5[5];
#pragma unroll
        for (int q = 0; q < 5; q++) c5[q] = (cnt >> (6 * q)) & 0x3fu;
        // informative rows of the final [L',31] matrix: thresh = (31 * 0.1) as usize = 3 (features.rs:558,712)
        const uint32_t thresh = (uint32_t)((double)HERRO_ROWS * 0.1);
        uint32_t ns = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) ns += c5[q] >= thresh ? 1u : 0u;
        sup_v = ns >= 2 ? 1u : 0u;
        // majority vote of the consensus decoder (consensus.rs:178-200), see k_final_tiles
        uint32_t c0 = c5[0], i0 = 0;
#pragma unroll
        for (uint32_t q = 1; q < 5; q++) if (c5[q] > c0) { c0 = c5[q]; i0 = q; }
        uint32_t c1 = 0, i1 = 5;
        bool have = false;
#pragma unroll
        for (uint32_t q = 0; q < 5; q++)
          if (q != i0 && (!have || c5[q] > c1)) { c1 = c5[q]; i1 = q; have = true; }
        const uint32_t tb0 = t0tok;
        cons_v = (c0 < 2u || (c0 == c1 && (i0 == tb0 || i1 == tb0))) ? tb0 : i0;
      }
      // (measured: holding these two bytes back until after the barrier below — so that the barrier does not have to wait
      // for their acknowledgement — made the kernel 9 % SLOWER, 941 -> 1030 us per 4096 windows: early stores overlap the wait)
      if (valid) {
        J.sup_flag[th.row_off + r] = (uint8_t)sup_v;
        J.cons_tmp[th.row_off + r] = (uint8_t)cons_v;
      }
      __syncthreads();
      const uint32_t nseg = (min(th.lub - th.r0, (uint32_t)HERRO_TILE) + 15) / 16;
      for (uint32_t it = threadIdx.x; it < ((J.dbg & 4u) ? 0u : HERRO_ROWS * nseg); it += NT) {
        const uint32_t c = it / nseg, sg = it % nseg;
        const uint32_t* lb = reinterpret_cast<const uint32_t*>(s_tb + c * TLD + sg * 16);
        *reinterpret_cast<uint4*>(J.fin_b + th.fin_off + (uint64_t)c * th.lub + th.r0 + sg * 16) = make_uint4(lb[0], lb[1], lb[2], lb[3]);
      }
    }
  }
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


// =====================================================================================================
// k_rf_quals — one workgroup per window: the quality bytes the model reads
// =====================================================================================================
// The model is evaluated on the receptive fields of informative rows only (rows sup_row[k] - half ..
// sup_row[k] + half of all 31 columns).  This kernel writes just those cells of the quality planes, straight
// from the read store (column_cell on global memory; neighbouring informative rows recompute shared cells).
__global__ __launch_bounds__(NT) void k_rf_quals(JobDev J, uint32_t half) {
  constexpr uint32_t RCAP = 1024;  // receptive-field rows handled per pass
  __shared__ ColHdr s_h[HERRO_ROWS];
  __shared__ uint32_t s_ow[HERRO_ROWS];
  __shared__ uint32_t s_rm[RCAP];   // row-map entry of each receptive-field row (0xffffffff: outside the window)
  __shared__ uint32_t s_r[RCAP];    // its row
  const uint32_t w = blockIdx.x;
  const uint32_t nsup = J.win_nsup[w], Lf = J.win_Lf[w];
  if (!nsup) return;
  const WinDesc wd = J.win[w];
  const uint32_t span = 2 * half + 1;
  const uint32_t* rowmap = J.rowmap2 + wd.row_off;
  // the 30 selected columns' headers, once per window (they were re-fetched for every cell)
  if (threadIdx.x >= 1 && threadIdx.x < HERRO_ROWS) {
    const uint32_t ow = J.sel_ow[(uint64_t)w * 32 + threadIdx.x];
    s_ow[threadIdx.x] = ow;
    if (ow != 0xffffffffu) s_h[threadIdx.x] = J.chdr[ow];
  }
  const uint64_t tq_off = J.read_qual_off[wd.rid] + wd.tstart;
  const uint32_t kper = max(1u, RCAP / span);  // informative rows per pass
  __shared__ uint32_t s_csafe;
  if (threadIdx.x == 0) s_csafe = 0;
  __syncthreads();
  if (threadIdx.x >= 1 && threadIdx.x < HERRO_ROWS && s_ow[threadIdx.x] != 0xffffffffu) atomicMax(&s_csafe, threadIdx.x);
  __syncthreads();
  const uint32_t c_safe = s_csafe;   // 0: the window has no selected column at all
  for (uint32_t k0 = 0; k0 < nsup; k0 += kper) {
    const uint32_t nk = min(kper, nsup - k0), nrows = nk * span;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nrows; i += NT) {
      const int32_t r = (int32_t)J.sup_row[wd.row_off + k0 + i / span] + (int32_t)(i % span) - (int32_t)half;
      const bool in = r >= 0 && r < (int32_t)Lf;
      s_r[i] = (uint32_t)r;
      s_rm[i] = in ? rowmap[r] : 0xffffffffu;
    }
    __syncthreads();
    // neighbouring lanes: the rows of one receptive field in one column (same header, same ops, adjacent bytes).
    // Four cells per thread and iteration, branch-free: a cell is a chain of four dependent loads (rank directory ->
    // op entry -> 2-bit word / quality byte); one cell at a time, a thread paid that chain ~9 times in sequence (180 us
    // per 4096 windows for 9.5 M cells).  Cells that do not exist (outside the window, padding columns, the target
    // column) run the chain on a safe stand-in column and drop the result.
    const uint32_t total = nrows * HERRO_ROWS;
    if (c_safe == 0) {   // block-uniform: no overlap column at all — target column from the store, '!' everywhere else
      for (uint32_t idx = threadIdx.x; idx < total; idx += NT) {
        const uint32_t k = idx / (span * HERRO_ROWS), rem = idx % (span * HERRO_ROWS), c = rem / span, d = rem % span;
        const uint32_t rm = s_rm[k * span + d];
        if (rm == 0xffffffffu) continue;
        const uint32_t q = (c == 0 && (rm >> 16) == 0) ? (uint32_t)J.read_qual[tq_off + (rm & 0xffffu)] : 33u;
        J.fin_q[wd.fin_off + (uint64_t)c * wd.lub + s_r[k * span + d]] = (uint8_t)q;
      }
      continue;
    }
    for (uint32_t idx0 = threadIdx.x; idx0 < total; idx0 += NT * 4) {
      // phase A: the four cells (LDS only)
      bool in[4], col[4];
      int32_t p[4];
      uint32_t j[4], cidx[4], uu[4], ow[4];
      bool inr[4];
      uint64_t dst[4];
      ColHdr h[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t idx = idx0 + u * NT;
        const bool live = idx < total;
        const uint32_t ic = live ? idx : 0u;
        const uint32_t k = ic / (span * HERRO_ROWS), rem = ic % (span * HERRO_ROWS), c = rem / span, d = rem % span;
        const uint32_t rm = s_rm[k * span + d];
        in[u] = live && rm != 0xffffffffu;
        p[u] = in[u] ? (int32_t)(rm & 0xffffu) : 0;
        j[u] = in[u] ? rm >> 16 : 0u;
        cidx[u] = c;
        col[u] = in[u] && c != 0 && s_ow[c] != 0xffffffffu;
        const uint32_t cc = col[u] ? c : c_safe;   // c_safe >= 1 here: a selected column of this window stands in
        h[u] = s_h[cc];
        ow[u] = s_ow[cc];
        const uint32_t uu_raw = (uint32_t)(p[u] - h[u].off);
        inr[u] = uu_raw < h[u].t_total;
        uu[u] = inr[u] ? uu_raw : 0u;
        dst[u] = wd.fin_off + (uint64_t)c * wd.lub + s_r[k * span + d];
      }
      // phase B: rank directory words — four independent loads in flight (see column_cell for the arithmetic)
      uint2 bw[4];
#pragma unroll
      for (int u = 0; u < 4; u++) bw[u] = J.bm[(uint64_t)ow[u] * J.n_bw + (uu[u] >> 5)];
      // phase C: op entries
      uint4 e[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t rank = bw[u].y + __popc(bw[u].x & (0xffffffffu >> (31u - (uu[u] & 31u))));
        e[u] = J.md[h[u].md_off + (inr[u] && h[u].t_total ? rank - 1u : 0u)];
      }
      // phase D: quality bytes (query cell) and the target's quality byte
      uint32_t qq[4], tq[4];
      bool isbase[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool is_m = (e[u].z >> 31) != 0;
        const uint32_t len = e[u].z & 0x7fffffffu;
        const bool last = (uu[u] + 1u == e[u].x + len);
        isbase[u] = inr[u] && (j[u] == 0 ? is_m : (last && e[u].w >= j[u]));
        const uint32_t q = j[u] == 0 ? e[u].y + (uu[u] - e[u].x) : e[u].y + (is_m ? len : 0u) + (j[u] - 1u);
        const uint32_t si = isbase[u] ? (uint32_t)(h[u].sbase + h[u].sdir * (int32_t)q) : 0u;
        qq[u] = J.read_qual[h[u].qual_off + si];
        tq[u] = J.read_qual[tq_off + (uint32_t)p[u]];   // in range for every position of the window
      }
      // phase E: select and store
#pragma unroll
      for (int u = 0; u < 4; u++) {
        uint32_t q = (col[u] && isbase[u]) ? qq[u] : 33u;
        if (cidx[u] == 0 && j[u] == 0) q = tq[u];
        if (in[u]) J.fin_q[dst[u]] = (uint8_t)q;
      }
    }
  }
}

void launch_rf_quals_old(const JobDev& J, uint32_t half, hipStream_t st, KernelTimer* tm) {
  if (!J.n_win) return;
  KT_BEGIN(tm, "rf_quals", st);
  hipLaunchKernelGGL(k_rf_quals, dim3(J.n_win), dim3(NT), 0, st, J, half);
  KT_END(tm, st);
}

void launch_full_quals_old(const JobDev& J, hipStream_t st) {  // token planes are rewritten with identical contents
  if (J.n_tiles) hipLaunchKernelGGL(k_final_tiles<true>, dim3(J.n_tiles), dim3(NT), 0, st, J);
}

void launch_featurize_old(const JobDev& J, hipStream_t st, KernelTimer* tm) {
  if (J.n_ow) {
    KT_BEGIN(tm, "ow_stats", st);
    hipLaunchKernelGGL(k_ow_stats, dim3((J.n_ow + 3) / 4), dim3(NT), 4 * ow_stats_lds_per_wave(J.window_size), st, J);
    KT_END(tm, st);
  }
  KT_BEGIN(tm, "win_rank", st);
  hipLaunchKernelGGL(k_win_rank, dim3(J.n_win), dim3(NT), 0, st, J);
  KT_END(tm, st);
  KT_BEGIN(tm, "pass1_pos", st);
  {
    const size_t shm = (PG + HCAP) * sizeof(PCol) + (size_t)PG * MDCAP * 16 + (size_t)PG * J.n_bw * 8 + (size_t)PG * PWCAP * 8;
    // counter width: enough bits for the largest threshold floor(0.1 * max(31, columns))
    const uint32_t tmax = (uint32_t)((double)(J.max_cols > 31 ? J.max_cols : 31) * 0.1);
    if (tmax < 4) hipLaunchKernelGGL(k_pass1_pos<2>, dim3(J.n_win), dim3(NT), shm, st, J);
    else if (tmax < 8) hipLaunchKernelGGL(k_pass1_pos<3>, dim3(J.n_win), dim3(NT), shm, st, J);
    else if (tmax < 16) hipLaunchKernelGGL(k_pass1_pos<4>, dim3(J.n_win), dim3(NT), shm, st, J);
    else if (tmax < 64) hipLaunchKernelGGL(k_pass1_pos<6>, dim3(J.n_win), dim3(NT), shm, st, J);
    else hipLaunchKernelGGL(k_pass1_pos<9>, dim3(J.n_win), dim3(NT), shm, st, J);
  }
  KT_END(tm, st);
  KT_BEGIN(tm, "select_layout", st);
  hipLaunchKernelGGL(k_select_layout, dim3(J.n_win), dim3(NT), ((size_t)mi_idx(J.window_size) + 1) * 4, st, J);
  KT_END(tm, st);
  KT_BEGIN(tm, "tile_plan", st);
  hipLaunchKernelGGL(k_tile_plan, dim3((J.n_tiles * 32 + NT - 1) / NT), dim3(NT), 0, st, J);
  KT_END(tm, st);
  KT_BEGIN(tm, "final_tiles", st);
  hipLaunchKernelGGL(k_final_tiles_t, dim3(J.n_tiles), dim3(NT), 0, st, J);
  KT_END(tm, st);
  KT_BEGIN(tm, "sup_compact", st);
  hipLaunchKernelGGL(k_sup_compact, dim3(J.n_win), dim3(NT), 0, st, J);
  KT_END(tm, st);
}

}  // namespace herro
