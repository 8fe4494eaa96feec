// build_dev.hip — windows and descriptors of a job on the device (see build_dev.h).
//
//   k_window_cuts   one thread per alignment: extract_windows (windowing.rs:44-273, restated for cut records in windowing.hpp window_cuts) from the
//                   scan's totals and cut records -> the alignment's (window, slice) records, in window order
//   k_scan_alns     exclusive prefix of the alignments' op counts (their share of the insertion-event scratch)
//   k_win_count     one wave per window: which alignments of its target reach it (lanes = alignments, 64 at a time), the checks build_target makes on
//                   every overlap, the window's counts
//   k_scan_wins     prefixes over the windows (first overlap, event slots, tiles, rows) and the totals the host sizes the arena with
//   k_desc_write    one wave per window again: OwDesc of its overlaps in alignment order, its WinDesc, its tiles
// Every input the reference would panic on, and every text the scan kernel flags, raises BuildTotals::err; the host then builds the job itself.
#include "build_dev.h"

namespace herro {
namespace {

__device__ inline uint64_t wave_sum64(uint64_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ inline uint32_t wave_max32(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
  return v;
}

__global__ __launch_bounds__(256) void k_window_cuts(BuildDev B) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= B.n_aln) return;
  const AlnMeta m = B.am[a];
  AlnHead h{0, 0, 0, 0};
  auto done = [&](uint32_t err) {
    if (err) { atomicOr(&B.tot->err, err); h = AlnHead{0, 0, 0, 0}; }
    B.head[a] = h;
  };
  if (m.flags & 2u) return done(0);
  const CigIn ci = B.in[a];
  const CigOut co = B.out[a];
  if ((co.flags & ~(uint32_t)CIG_INS_PAIR) || co.n_cuts > ci.cut_cap) return done(BLD_SCAN_FLAG);
  const TgtMeta t = B.tm[m.tgt];
  const uint32_t W = B.W, n_windows = t.n_windows;
  if (m.tend < m.tstart || m.qend < m.qstart) return done(BLD_WINDOWING);
  if ((m.tend - m.tstart) < W || (m.qend - m.qstart) < W) return done(0);   // windowing.rs:53-57
  const uint32_t zthr = (uint32_t)(0.1f * (float)W);                        // :65
  if (t.tlen < zthr) return done(BLD_WINDOWING);
  const uint32_t nthr = t.tlen - zthr;
  const uint32_t first_window = m.tstart < zthr ? 0 : (m.tstart + W - 1) / W;   // :75-79
  const uint32_t last_window = m.tend > nthr ? (m.tend - 1) / W + 1 : m.tend / W;   // :81-85
  if (last_window <= first_window) return done(0);   // :106
  // the cuts in op order (the scan found them in whatever order its threads ran; a handful per alignment)
  CigCut* cut = B.cuts + ci.cut_off;
  const uint32_t n_cut = co.n_cuts;
  for (uint32_t i = 1; i < n_cut; i++) {
    const CigCut c = cut[i];
    uint32_t j = i;
    while (j > 0 && cut[j - 1].k > c.k) { cut[j] = cut[j - 1]; j--; }
    if (j != i) cut[j] = c;
  }
  struct P { uint32_t t, q, i; };
  bool started = false;
  uint32_t w_t = 0, w_q = 0, w_op = 0, w_off = 0;
  P w_p{m.tstart, 0, 0};
  uint32_t w_opv = co.op0;
  if (m.tstart % W == 0 || m.tstart < zthr) { started = true; w_t = m.tstart; }   // :120-125
  HowRec* how = B.how + ci.cut_off;
  uint32_t err = 0;
  auto emit = [&](uint32_t widx_plus1, uint32_t qend, uint32_t op_hi, uint32_t end_off, const P& hi, uint32_t last_opv) {
    if (widx_plus1 == 0 || widx_plus1 - 1 >= n_windows) { err |= BLD_WINDOWING; return; }   // windows[] index out of bounds
    const uint32_t win = widx_plus1 - 1;
    if (h.n_how == 0) h.w_first = win;
    if (win != h.w_first + h.n_how || h.n_how >= ci.cut_cap || op_hi < w_op) { err |= BLD_WINDOWING; return; }
    HowRec r;
    r.win = win; r.tstart = w_t; r.qstart = w_q; r.qend = qend; r.op_lo = w_op; r.op_hi = op_hi; r.start_off = w_off; r.end_off = end_off;
    r.st = hi.t - w_p.t; r.sq = hi.q - w_p.q; r.si = hi.i - w_p.i; r.op_first = w_opv; r.op_last = last_opv;
    r.scr_local = h.op_sum; r.pad0 = 0; r.pad1 = 0;
    how[h.n_how] = r;
    h.n_how++;
    h.op_sum += op_hi - w_op;
  };
  const uint32_t n = co.n_ops;
  for (uint32_t ci_ = 0; ci_ < n_cut && !err; ci_++) {
    const CigCut c = cut[ci_];
    const uint32_t k = c.k, tpos = c.t, qpos = c.q;
    const uint32_t ty = op_type(c.o0), l = op_len(c.o0);
    const bool is_m = ty == OP_M;
    const uint32_t tnew = tpos + l, qnew = is_m ? qpos + l : qpos;
    const P p_k{tpos, qpos, c.ins}, p_k1{tnew, qnew, c.ins};
    const uint32_t cur_w = tpos / W, new_w = tnew / W;
    for (uint32_t i = 1; i < new_w - cur_w && !err; i++) {   // windows fully inside this op :150-195
      const uint32_t off = (cur_w + i) * W - tpos;
      const uint32_t qcut = is_m ? qpos + off : qpos;
      if (started) emit(cur_w + i, qcut, k + 1, off, p_k1, c.o0);
      started = true; w_t = tpos + off; w_q = qcut; w_op = k; w_off = off; w_p = p_k; w_opv = c.o0;
    }
    const uint32_t off = new_w * W - tpos;   // :198
    uint32_t qend = is_m ? qpos + off : qpos;
    uint32_t op_hi, end_off, next_op, next_off;
    P p_hi = p_k1, p_next = p_k;
    uint32_t last_opv = c.o0, next_opv = c.o0;
    if (tnew == new_w * W) {   // the op ends exactly on the boundary :210-223
      if (k + 1 < n && op_type(c.o1) == OP_I) {   // a trailing insertion stays with this window
        const uint32_t li = op_len(c.o1);
        qend += li; op_hi = k + 2; end_off = li;
        p_hi = P{tnew, qnew + li, c.ins + li};
        last_opv = c.o1; next_opv = c.o2;
      } else {
        op_hi = k + 1; end_off = l; next_opv = c.o1;
      }
      next_op = op_hi; next_off = 0; p_next = p_hi;
    } else {   // :224-230
      op_hi = k + 1; end_off = off; next_op = k; next_off = off;
    }
    if (started && !err) emit(new_w, qend, op_hi, end_off, p_hi, last_opv);
    started = true; w_t = tpos + off; w_q = qend; w_op = next_op; w_off = next_off; w_p = p_next; w_opv = next_opv;
  }
  if (!err && co.t_end > nthr && co.t_end % W != 0) {   // tail window :261-272
    if (!started || n == 0) err |= BLD_WINDOWING;
    else emit(last_window, co.q_end, n, op_len(co.opn), P{co.t_end, co.q_end, co.ins_end}, co.opn);
  }
  done(err);
}

// one block: exclusive prefix of the alignments' op counts
__global__ __launch_bounds__(1024) void k_scan_alns(BuildDev B) {
  // One workgroup (its latency is the job's).  Round 6: the records are read and written COALESCED — lane t takes record r0 + 1024 j + t — and change hands through the LDS to the
  // thread that scans eight consecutive ones; a thread reading ITS 32 consecutive records made every load instruction 64 separate cache-line requests on one compute unit (50 us
  // for 33 k alignments, three passes of ~15), and the block scan was a 20-barrier ladder.
  constexpr uint32_t E = 8, R = 1024 * E;
  __shared__ uint32_t s_v[R + R / 32];   // record i at i + i / 32 (one pad word per 32: a thread's eight consecutive words against its neighbours')
  __shared__ uint64_t s_wsum[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, n = B.n_aln;
  auto at = [](uint32_t i) { return i + (i >> 5); };
  uint64_t carry = 0;
  for (uint32_t r0 = 0; r0 < n; r0 += R) {
    uint32_t v[E];
#pragma unroll
    for (uint32_t j = 0; j < E; j++) {
      const uint32_t e = r0 + j * 1024u + tid;
      v[j] = e < n ? B.head[e].op_sum : 0u;
    }
#pragma unroll
    for (uint32_t j = 0; j < E; j++) s_v[at(j * 1024u + tid)] = v[j];
    __syncthreads();
    uint32_t x[E];
    uint64_t local = 0;
#pragma unroll
    for (uint32_t k = 0; k < E; k++) { x[k] = s_v[at(tid * E + k)]; local += x[k]; }
    uint64_t inc = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t o = __shfl_up(inc, d, 64);
      if (lane >= (uint32_t)d) inc += o;
    }
    if (lane == 63u) s_wsum[wv] = inc;
    __syncthreads();
    uint64_t base = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) {
      const uint64_t y = s_wsum[w];
      if (w < wv) base += y;
      total += y;
    }
    uint64_t run = carry + base + inc - local;
#pragma unroll
    for (uint32_t k = 0; k < E; k++) { s_v[at(tid * E + k)] = (uint32_t)run; run += x[k]; }   // (a thread's own eight words)
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < E; j++) {
      const uint32_t e = r0 + j * 1024u + tid;
      if (e < n) B.head[e].scr_base = s_v[at(j * 1024u + tid)];
    }
    carry += total;
    __syncthreads();   // s_v and s_wsum go round again
  }
  if (tid == 0) {
    B.tot->scr_ops = carry;
    if (carry > 0xffffffffull) atomicOr(&B.tot->err, BLD_SIZE);
  }
}

// The overlap of alignment a in window wl of its target: every field of OwDesc and every check build_target makes (herro_api.hip; features.rs:585-679,
// 110-237).  false: an input the reference would panic on.
__device__ inline bool make_ow(const BuildDev& B, uint32_t a, const HowRec& x, const AlnMeta& m, const TgtMeta& t, const AlnHead& h, uint32_t w, uint32_t wl,
                               uint32_t win_start, uint32_t win_len, OwDesc& d, uint64_t& tt_out) {
  d.win = w;
  d.qid = m.qid;
  d.cls = m.cls;
  d.tstart = x.tstart;
  d.qlen = x.qend - x.qstart;
  d.strand = m.flags & 1u;
  if (x.win != wl || x.qend < x.qstart) return false;
  if (d.strand == 0) d.qbeg = m.qstart + x.qstart;
  else {
    if (m.qend < x.qend) return false;
    d.qbeg = m.qend - x.qend;
  }
  d.op_begin = B.in[a].op_off + x.op_lo;
  d.op_cnt = x.op_hi - x.op_lo;
  d.start_off = x.start_off;
  d.end_off = x.end_off;
  d.scr_off = h.scr_base + x.scr_local;
  d.wtstart = win_start;
  d.wlen = win_len;
  d.t_woff = B.read_word_off[t.rid];
  d.q_woff = B.read_word_off[m.qid];
  d.q_qual_off = B.read_qual_off[m.qid];
  if (x.op_hi <= x.op_lo) return false;
  if (d.tstart < win_start) return false;
  if (op_type(x.op_first) == OP_I && d.tstart == win_start) return false;   // max_ins[tpos - 1] with tpos == 0
  const uint32_t op_f = x.op_first, op_l = x.op_last;
  uint64_t tt = x.st, qq = x.sq;
  if (d.op_cnt == 1) {
    if (d.end_off <= d.start_off) return false;
    const uint32_t e1 = d.end_off - d.start_off, l1 = op_len(op_f);
    if (op_type(op_f) != OP_I) tt = tt - l1 + e1;
    if (op_type(op_f) != OP_D) qq = qq - l1 + e1;
  } else {
    if (op_len(op_f) <= d.start_off) return false;
    if (d.end_off == 0) return false;
    if (op_type(op_f) != OP_I) tt -= d.start_off;
    if (op_type(op_f) != OP_D) qq -= d.start_off;
    const uint32_t ll = op_len(op_l);
    if (op_type(op_l) != OP_I) tt = tt - ll + d.end_off;
    if (op_type(op_l) != OP_D) qq = qq - ll + d.end_off;
  }
  if ((uint64_t)(d.tstart - win_start) + tt > win_len) return false;   // the slice overruns the target window
  if (qq > d.qlen) return false;                                       // ... the query region
  const uint64_t qlen_read = B.read_qual_off[m.qid + 1] - B.read_qual_off[m.qid];
  if ((uint64_t)d.qbeg + d.qlen > qlen_read) return false;
  tt_out = tt;
  return true;
}

// one wave per window; WRITE: second pass (descriptors), else first pass (counts)
template <bool WRITE>
__global__ __launch_bounds__(256) void k_win_pass(BuildDev B) {
  const uint32_t lane = threadIdx.x & 63u, w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= B.n_win) return;
  const uint32_t tg = B.win_tgt[w];
  const TgtMeta t = B.tm[tg];
  const uint32_t W = B.W, wl = w - t.win0;
  const uint32_t win_start = wl * W, win_len = (wl == t.n_windows - 1) ? t.tlen - wl * W : W;
  const uint64_t lt = (1ull << lane) - 1ull;
  uint32_t cnt = 0, bad = 0;
  uint64_t ev = 0, ins = 0, rd = 0, opb = 0;
  const uint32_t ow0 = WRITE ? B.ow_begin[w] : 0u;
  for (uint32_t base = 0; base < t.n_aln; base += 64) {
    const bool valid = base + lane < t.n_aln;
    const uint32_t a = t.aln0 + min(base + lane, t.n_aln - 1);
    const AlnHead h = B.head[a];
    const bool covers = valid && h.n_how && wl >= h.w_first && wl - h.w_first < h.n_how;
    const uint64_t cm = __ballot(covers);
    if (covers) {
      const HowRec x = B.how[B.in[a].cut_off + (wl - h.w_first)];
      const AlnMeta m = B.am[a];
      OwDesc d{};
      uint64_t tt = 0;
      if (!make_ow(B, a, x, m, t, h, w, wl, win_start, win_len, d, tt)) bad = 1;
      else if (WRITE) B.ow[ow0 + cnt + (uint32_t)__popcll(cm & lt)] = d;
      ev += (uint64_t)d.op_cnt + 2u;
      ins += x.si;
      rd += (uint64_t)d.qlen + (d.qlen + 3) / 4;
      opb += (uint64_t)d.op_cnt * 4;
    }
    cnt += (uint32_t)__popcll(cm);
  }
  if (__ballot(bad != 0)) { if (lane == 0) atomicOr(&B.tot->err, BLD_VALIDATE); }
  ev = wave_sum64(ev); ins = wave_sum64(ins); rd = wave_sum64(rd); opb = wave_sum64(opb);
  const uint32_t lub = (uint32_t)(((uint64_t)win_len + min(ins, (uint64_t)50 * win_len) + 15) & ~15ull);
  if (lane != 0) return;
  if (!WRITE) {
    WinAcc r;
    r.ow_cnt = cnt; r.lub = lub; r.ev = ev;
    r.rd_bytes = rd + (uint64_t)win_len + (win_len + 3) / 4;
    r.op_bytes = opb;
    B.wacc[w] = r;
    return;
  }
  WinDesc wd{};
  wd.rid = t.rid; wd.wid = wl; wd.n_wids = t.n_windows;
  wd.tstart = win_start; wd.win_len = win_len;
  wd.ow_begin = ow0; wd.ow_cnt = cnt; wd.lub = lub;
  wd.col_off = B.tile_off[w];
  wd.row_off = B.row_off[w];
  wd.fin_off = (uint64_t)HERRO_ROWS * wd.row_off;
  wd.pos_off = (uint64_t)w * ((uint64_t)W + 1);
  wd.ev_off = B.ev_off[w];
  B.win[w] = wd;
  uint64_t tile = wd.col_off;
  for (uint32_t r0 = 0; r0 < lub; r0 += HERRO_TILE) { B.tile_win[tile] = w; B.tile_r0[tile] = r0; tile++; }
}

// one block: prefixes over the windows + totals
__global__ __launch_bounds__(1024) void k_scan_wins(BuildDev B, uint32_t* ow_begin, uint64_t* ev_off, uint64_t* tile_off, uint64_t* row_off) {
  __shared__ uint64_t s_p[4][1024];
  __shared__ uint32_t s_mx[1024];
  const uint32_t tid = threadIdx.x, n = B.n_win, per = (n + 1023) / 1024;
  const uint32_t w0 = min(tid * per, n), w1 = min(w0 + per, n);
  uint64_t l[4] = {0, 0, 0, 0}, rd = 0, opb = 0;
  uint32_t mx = 0;
#pragma unroll 4
  for (uint32_t w = w0; w < w1; w++) {
    const WinAcc r = B.wacc[w];
    l[0] += r.ow_cnt; l[1] += r.ev; l[2] += (r.lub + HERRO_TILE - 1) / HERRO_TILE; l[3] += r.lub;
    rd += r.rd_bytes; opb += r.op_bytes;
    mx = max(mx, r.ow_cnt);
  }
  for (int i = 0; i < 4; i++) s_p[i][tid] = l[i];
  s_mx[tid] = mx;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint64_t v[4] = {0, 0, 0, 0};
    uint32_t m2 = 0;
    if (tid >= d) { for (int i = 0; i < 4; i++) v[i] = s_p[i][tid - d]; m2 = s_mx[tid - d]; }
    __syncthreads();
    for (int i = 0; i < 4; i++) s_p[i][tid] += v[i];
    s_mx[tid] = max(s_mx[tid], m2);
    __syncthreads();
  }
  uint64_t run[4];
  for (int i = 0; i < 4; i++) run[i] = s_p[i][tid] - l[i];
  for (uint32_t w = w0; w < w1; w++) {
    const WinAcc r = B.wacc[w];
    ow_begin[w] = (uint32_t)run[0]; ev_off[w] = run[1]; tile_off[w] = run[2]; row_off[w] = run[3];
    run[0] += r.ow_cnt; run[1] += r.ev; run[2] += (r.lub + HERRO_TILE - 1) / HERRO_TILE; run[3] += r.lub;
  }
  // the algorithmic-byte sums are plain totals: one more reduction through the first two scan arrays' storage would do; a wave reduction + atomics is simpler
  rd = wave_sum64(rd); opb = wave_sum64(opb);
  if ((tid & 63u) == 0) { atomicAdd((unsigned long long*)&B.tot->rd_bytes, (unsigned long long)rd); atomicAdd((unsigned long long*)&B.tot->op_bytes, (unsigned long long)opb); }
  if (tid == 1023) {
    BuildTotals* T = B.tot;
    ow_begin[n] = (uint32_t)s_p[0][1023];
    T->n_ow = (uint32_t)s_p[0][1023];
    T->n_tiles = (uint32_t)s_p[2][1023];
    T->row_elems = s_p[3][1023];
    T->fin_bytes = (uint64_t)HERRO_ROWS * s_p[3][1023];
    T->max_cols = max(1u, s_mx[1023] + 1u);
    // 32-bit indices on the device: overlaps, tiles, the event slots (scr_ops + 2 per overlap = the total of `ev`)
    if (s_p[0][1023] > 0xffffffffull || s_p[2][1023] > 0xffffffffull || s_p[1][1023] > 0xffffffffull) atomicOr(&T->err, BLD_SIZE);
  }
}

}  // namespace

void launch_build_phase1(const BuildDev& B, uint32_t* ow_begin, uint64_t* ev_off, uint64_t* tile_off, uint64_t* row_off, hipStream_t st) {
  if (!B.n_aln || !B.n_win) return;
  hipLaunchKernelGGL(k_window_cuts, dim3((B.n_aln + 255) / 256), dim3(256), 0, st, B);
  hipLaunchKernelGGL(k_scan_alns, dim3(1), dim3(1024), 0, st, B);
  hipLaunchKernelGGL(k_win_pass<false>, dim3((B.n_win + 3) / 4), dim3(256), 0, st, B);
  hipLaunchKernelGGL(k_scan_wins, dim3(1), dim3(1024), 0, st, B, ow_begin, ev_off, tile_off, row_off);
}

void launch_build_phase2(const BuildDev& B, hipStream_t st) {
  if (!B.n_aln || !B.n_win) return;
  hipLaunchKernelGGL(k_win_pass<true>, dim3((B.n_win + 3) / 4), dim3(256), 0, st, B);
}

}  // namespace herro
