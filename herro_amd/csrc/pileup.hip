// pileup.hip — pileup feature generation on gfx950, bit-plane formulation (reference src/features.rs:326-583).
//
// Integer / byte work; no MFMA on purpose.  Every kernel is a swarm of small workgroups, each with at most three dependent
// global round trips in front of its arithmetic; k_cols and k_tokens end up bound by VALU issue (a wave64 instruction holds
// its SIMD four cycles on gfx950), the rest by latency (DESIGN.md §4 has the counters).
//
//   k_cols      one WAVE per overlap-window, no barriers: CIGAR ops -> prefix sums (DPP wave scans) -> the
//               overlap's column as three BIT PLANES over the window's target positions (M: a query base is aligned
//               here; lo / hi: its 2-bit code) from the bit-plane copy of the read store: the lane of an M op writes the
//               op's stretch of the word it starts in, the lane that owns a word of 32 positions adds the op that covers
//               the word's first position (a prefix maximum over per-word slots finds it; round 6);
//               insertion events (position, length, first bases).  From the planes: accuracy (features.rs:585-679) =
//               popcounts of M & (query ^ target); long-indel filter (features.rs:315-324).
//   k_win       one workgroup per window: informative positions of pass 1 in bit-sliced counters over the kept
//               columns' planes (features.rs:681-722), match / mismatch tallies per query name (features.rs:461-500),
//               stable accuracy rank (features.rs:386-409).  The reference's [L, 1+n] pass-1 matrix never exists.
//   k_layout    one workgroup per window: haplotype score, stable re-rank, top-30 (features.rs:502-525); row of every
//               position = prefix sum over the selected overlaps' max insertion (features.rs:44-95, 531-556); the column
//               table and the compact event lists its consumers walk (per window for k_rfq, per tile for k_tokens).
//   k_tokens    one workgroup per 1024 rows: the [31][L'] token planes (features.rs:110-266): 16 rows x 1 column per
//               step, four rows per register — bits pulled from the planes, spread to bytes by one multiply, tokens and the
//               position -> row expansion by v_perm_b32; inserted bases patched in from the tile's event list; symbol counts
//               per row -> informative rows (features.rs:558) and the decoder's majority vote (consensus.rs:178-200).
//   k_supgather ordered list of informative positions of a window from its tiles' lists.
//   k_rfq       at infer time: the qualities of the model's receptive fields (five rows around every informative row),
//               compact, 8 bytes per (informative row, column) (features.rs:139-152,197-198,225-226).  Query index of a cell
//               = rank in the M plane + insertion events.
//   k_quals     the same bytes as complete quality planes, on request (herro_job_window_copy with a quality buffer), and the
//               receptive fields of models whose field is wider than 8 rows.
//   k_consensus corrected bases of a window from the logits and the votes (consensus.rs:86-227).
//
// What the formulation buys: the work of a column is proportional to its OPS (~165), not to its 4096 positions; a cell
// of the final matrix costs ~20 ALU operations and no search; HBM traffic is the op arrays and query bit planes in,
// the token planes out, plus 1.5 KB of planes per kept overlap in between.
// Generality: any number of overlaps per window, insertions anywhere the reference accepts them (leading insertion of an
// alignment that starts inside the window, consecutive insertion ops), windows of 16..8192.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>

#include "job_dev.h"
#include "pileup_core.h"

namespace herro {

namespace {

constexpr uint32_t NONE = 0xffffffffu;
constexpr uint32_t RF_LEFT = 0x80000000u;   // JobDev::win_rfbase: the window's receptive-field slots are reserved (low 31 bits) but k_rows did not fill them
constexpr uint32_t ROWCAP = HERRO_TILE;   // rows of the final matrix per k_tokens workgroup (1024)
constexpr uint32_t SCAP = 128;            // overlaps per window whose scores are cached in LDS
constexpr uint32_t QEVCAP = 2048;         // insertion events (position | length, query index) staged in LDS by k_quals

// ---- small helpers ---------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ uint32_t blk_scan(uint32_t v, uint32_t* total, uint32_t* s_wave /*[NT/64]*/) {
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

// bits [lo, hi) of a 32-bit word, clipped
__device__ __forceinline__ uint32_t mask_range(int32_t lo, int32_t hi) {
  lo = max(lo, 0);
  hi = min(hi, 32);
  if (lo >= hi) return 0u;
  const uint32_t m = hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u);
  return m & ~((1u << lo) - 1u);
}

// 32 consecutive plane bits starting at bit index s of a staged plane of npw words (bits outside read 0).  The staged
// words sit between two zero words, pl[-1] and pl[npw]: no bounds branches, two clamps.
__device__ __forceinline__ uint32_t lds_bits(const uint32_t* pl, uint32_t npw, int32_t s) {
  const int32_t w = s >> 5;
  const uint32_t a = pl[min(max(w, -1), (int32_t)npw)], b = pl[min(max(w + 1, -1), (int32_t)npw)];
  return __funnelshift_r(a, b, (uint32_t)s & 31u);
}
// the same from the read store's plane array (word index clamped into the array: wmax = its last word)
__device__ __forceinline__ uint32_t glb_bits(const uint32_t* __restrict__ pl, uint64_t woff, uint64_t wmax, int32_t s) {
  if (s < 0) return s <= -32 ? 0u : (pl[min(woff, wmax)] << (uint32_t)(-s));
  const uint64_t w = woff + ((uint32_t)s >> 5);
  return __funnelshift_r(pl[min(w, wmax)], pl[min(w + 1, wmax)], (uint32_t)s & 31u);
}

struct ColPlanes { uint32_t m, lo, hi, gap; };  // per position bit: query base present / code planes / deletion

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

// dynamic-LDS opt-in is a per-device function attribute
void pileup_opt_in_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(mu);
  if (done.insert({fn, dev}).second) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// ---- wave scans on the DPP network: row_shr inside rows of 16 lanes, then the two row broadcasts -----------
__device__ __forceinline__ uint32_t wscan_incl(uint32_t v) {
#ifdef HERRO_SCAN_SHFL
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
#else
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
#endif
  return v;
}
__device__ __forceinline__ uint32_t wscan_max(uint32_t v) {   // inclusive prefix maximum (unsigned; lanes a step does not reach contribute 0)
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
  return v;
}
__device__ __forceinline__ uint32_t wlast(uint32_t incl) { return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63); }
__device__ __forceinline__ uint32_t wsum(uint32_t v) { return wlast(wscan_incl(v)); }

// =====================================================================================================
// k_cols — one wave per overlap-window, no workgroup barriers
// =====================================================================================================
constexpr int CA_NT = 256, CA_NW = CA_NT / 64;
constexpr uint32_t MDCAP = 192;   // ops per batch (three steps of 64); a slice with more ops runs several batches
__host__ __device__ inline uint32_t cols_qcap(uint32_t nw) { return nw + 40u < 320u ? nw + 40u : 320u; }   // staged query plane words
__host__ __device__ inline uint32_t cols_lds_words(uint32_t nw) { return 2 * MDCAP + nw + 2 * (cols_qcap(nw) + 2) + 3 * nw; }   // 4.9 KB per wave at nw = 128: eight workgroups per compute unit

// QI: 64-word steps that cover the staged query words (cols_qcap(nw) + 2), NWI: plane words per lane (nw / 64, rounded up) — the loops over
// them are unrolled, and with the bounds of the largest window (5, 4) a 4096-base window paid for two empty steps of each (r5)
template <int QI, int NWI>
__global__ __launch_bounds__(CA_NT) void k_cols(JobDev J) {
  extern __shared__ __attribute__((aligned(16))) uint32_t ca_smem[];
  const uint32_t nw = J.nw, qcap = cols_qcap(nw);
  const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t o = blockIdx.x * CA_NW + wave;
  if (o >= J.n_ow) return;
  static_assert(NWI + 1 <= QI, "the target words of a lane's plane words and of the word behind them");
  uint32_t* s_mt = ca_smem + (size_t)wave * cols_lds_words(nw);   // M/D ops of the batch: window position of the first base | length << 14 | M << 31
  uint32_t* s_mq = s_mt + MDCAP;                                  // ... query index of it
  uint32_t* s_cov = s_mq + MDCAP;                                 // [nw] per word: (table slot + 1) << 12 | insertion events in front, of the last op that starts in (32 w - 32, 32 w]
  uint32_t* q0 = s_cov + nw + 1;                                  // [-1 .. qcap] query code planes (alignment orientation) between two zero words
  uint32_t* q1 = q0 + qcap + 2;
  uint32_t* s_pM = q1 + qcap + 1;                                 // [nw] x 3: the plane bits the op lanes write (an M op's stretch inside the word it starts in)
  uint32_t* s_pL = s_pM + nw;
  uint32_t* s_pH = s_pL + nw;
  PROF_BEGIN(J);
  const OwDesc d = J.ow[o];   // carries the window's and the reads' offsets: no further lookups before the data
  const uint32_t cnt_ops = d.op_cnt;
  const uint32_t* __restrict__ ops = J.ops + d.op_begin;
  const int32_t off = (int32_t)(d.tstart - d.wtstart);
  const uint64_t pmax = J.read_n_words + 1;   // last word of the plane arrays
  const uint32_t qw0 = d.qbeg >> 5, nqw = ((d.qbeg + d.qlen) >> 5) - qw0 + 2;
  const bool staged = nqw <= qcap;
  const bool packed_scan = d.qlen < 65536u;   // wave-uniform
  // ---- every global load of the wave is issued here, unconditionally (indices clamped), before anything waits
  uint32_t op_r[3];
#pragma unroll
  for (int i = 0; i < 3; i++) op_r[i] = ops[min(lane + 64u * i, cnt_ops - 1u)];
  uint32_t v0[QI], v1[QI], u0[QI], u1[QI];
  {
    // scalar bases, 32-bit lane offsets clamped once against the end of the plane arrays
    const uint64_t qbase = min(d.q_woff + qw0, pmax), tbase = min(d.t_woff + (d.wtstart >> 5), pmax);
    const uint32_t qlast = (uint32_t)min((uint64_t)(nqw - 1u), pmax - qbase), tlast = (uint32_t)min((uint64_t)(nw + 1u), pmax - tbase);
    const uint32_t* __restrict__ gq0 = J.read_p0 + qbase;
    const uint32_t* __restrict__ gq1 = J.read_p1 + qbase;
    const uint32_t* __restrict__ gt0 = J.read_p0 + tbase;
    const uint32_t* __restrict__ gt1 = J.read_p1 + tbase;
#pragma unroll
    for (int i = 0; i < QI; i++) {
      const uint32_t qi = min(lane + 64u * i, qlast), ti = min(lane + 64u * i, tlast);
      v0[i] = gq0[qi];
      v1[i] = gq1[qi];
      u0[i] = gt0[ti];
      u1[i] = gt1[ti];
    }
  }
#pragma unroll
  for (int i = 0; i < QI; i++) {
    const uint32_t idx = lane + 64u * i;
    // a reverse-strand query is staged complemented and end to end reversed (word nqw - 1 - i = ~brev(word i)): staged bit s0 + k is alignment-orientation
    // base k on either strand, and the reads below carry no strand branch (round 6; ~brev at every read before)
    if (staged && idx < nqw) {
      if (d.strand) { q0[nqw - 1u - idx] = ~__brev(v0[i]); q1[nqw - 1u - idx] = ~__brev(v1[i]); }
      else { q0[idx] = v0[i]; q1[idx] = v1[i]; }
    }
    if (staged && i == 0 && lane < 2) { q0[lane ? (int)nqw : -1] = 0; q1[lane ? (int)nqw : -1] = 0; }
    if (idx < nw) { s_pM[idx] = 0; s_pL[idx] = 0; s_pH[idx] = 0; }
  }
  // the window's target code planes, 32 positions per lane like the planes built below (for the accuracy): lane l holds words l, l + 64, .. of the raw
  // planes from the window's first word; the word behind comes from the next lane (round 6: they were staged in LDS, 1 KB per wave that cost occupancy)
  uint32_t tl[NWI], th[NWI];
  {
    const uint32_t tsh = d.wtstart & 31u;
#pragma unroll
    for (int i = 0; i < NWI; i++) {
      uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u0[i], 0x130, 0xf, 0xf, false);   // wave_shl:1
      uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u1[i], 0x130, 0xf, 0xf, false);
      const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)u0[i + 1]), f1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)u1[i + 1]);
      if (lane == 63u) { n0 = f0; n1 = f1; }
      tl[i] = __funnelshift_r(u0[i], n0, tsh);
      th[i] = __funnelshift_r(u1[i], n1, tsh);
    }
  }
  PROF_MARK(J, 0, 0);
  const int32_t sbase = d.strand ? (int32_t)(d.qbeg + d.qlen - 1u) : (int32_t)d.qbeg;   // stored index of alignment-orientation base 0
  const int32_t rel = -(int32_t)(qw0 << 5);
  const int32_t s0 = d.strand ? (int32_t)(nqw << 5) - 1 - (sbase + rel) : sbase + rel;    // staged bit of alignment-orientation base 0
  // code planes of the 32 alignment-orientation query bases qidx .. qidx + 31 (bits of bases outside the region are
  // meaningless: callers mask).  Reverse strand: base k is the complement of stored base sbase - k (features.rs:128-153).
  auto qbits = [&](int32_t qidx, uint32_t& b0, uint32_t& b1) {
    if (staged) {
      b0 = lds_bits(q0, nqw, s0 + qidx); b1 = lds_bits(q1, nqw, s0 + qidx);
    } else if (d.strand == 0) {
      const int32_t s = sbase + qidx;
      b0 = glb_bits(J.read_p0, d.q_woff, pmax, s); b1 = glb_bits(J.read_p1, d.q_woff, pmax, s);
    } else {
      const int32_t s = sbase - qidx - 31;
      b0 = ~__brev(glb_bits(J.read_p0, d.q_woff, pmax, s)); b1 = ~__brev(glb_bits(J.read_p1, d.q_woff, pmax, s));
    }
  };
  uint32_t carry_t = 0, carry_q = 0, n_ev = 0, isum = 0, dsum = 0, longindel = 0;
  uint4* __restrict__ ev = J.iev + d.scr_off;
  uint32_t pM[NWI], pL[NWI], pH[NWI];
  // directories for k_rfq, per word of 32 positions: alignment-orientation query index of the first base at or behind the word's
  // first position (= M bits + inserted bases in front of it), insertion events in front of it — read off the op that covers
  // the word's first position (no events inside an M / D op); 0 / 0 in front of the overlap, 0 / all events behind it
  uint32_t dQ[NWI], dE[NWI];
#pragma unroll
  for (int i = 0; i < NWI; i++) { pM[i] = 0; pL[i] = 0; pH[i] = 0; dQ[i] = 0; dE[i] = 0xffffffffu; }
  const uint64_t lt = (1ull << lane) - 1ull;

  for (uint32_t b0 = 0; b0 < cnt_ops; b0 += MDCAP) {
    for (uint32_t i = lane; i < nw; i += 64) s_cov[i] = 0;
    uint32_t n_md = 0;
    const int32_t Pb0 = off + (int32_t)carry_t;   // first window position of the batch
#pragma unroll
    for (int s = 0; s < 3; s++) {
      const uint32_t bs = b0 + 64u * s;
      if (bs < cnt_ops) {   // wave-uniform
        const uint32_t k = bs + lane;
        const uint32_t op = b0 == 0 ? op_r[s] : ops[min(k, cnt_ops - 1u)];
        const bool valid = k < cnt_ops;
        const uint32_t ty = op_type(op), len = op_len(op);
        const uint32_t e = valid ? eff_len(op, k, cnt_ops, d.start_off, d.end_off) : 0u;
        const bool isI = valid && ty == OP_I, isM = valid && ty == OP_M, isD = valid && ty == OP_D;
        if ((isI || isD) && len > 50u) longindel = 1;  // untrimmed length (features.rs:317)
        if (isI) isum += e;
        if (isD) dsum += e;
        const uint32_t tadv = (isM || isD) ? e : 0u, qadv = (isM || isI) ? e : 0u;
        // one scan for both where the halves cannot meet: target advance in the low half (a slice consumes <= 8192 target bases), query
        // advance in the high half when the slice's query span (OwDesc::qlen = the sum of all query advances) stays below 65536
        uint32_t it, iq, last_t, last_q;
        if (packed_scan) {
          const uint32_t itq = wscan_incl(tadv | (qadv << 16));
          const uint32_t last = wlast(itq);
          it = itq & 0xffffu; iq = itq >> 16; last_t = last & 0xffffu; last_q = last >> 16;
        } else {
          it = wscan_incl(tadv); iq = wscan_incl(qadv); last_t = wlast(it); last_q = wlast(iq);
        }
        const uint32_t t = carry_t + it - tadv, q = carry_q + iq - qadv;
        // insertion behind window position off + t - 1 (features.rs:77, 219-228): position, trimmed length (bases written),
        // query index of its first base, its first 16 bases, untrimmed length (max_ins)
        const uint64_t imask = __ballot(isI);
        // ONE read of the query planes serves both kinds of lanes (round 6): an insertion's lane wants the bases from q on, an M op's lane the bases that fall
        // into the word of 32 window positions its first base lies in — it ORs that stretch into the LDS planes itself.  The plane walk further down is
        // then left with ONE op per word, the one that covers the word's first position: it used to walk every op that touches the word, each wave as
        // many turns as its busiest lane (~5 at ~40 instructions, twice: the largest phase of the kernel)
        const int32_t Pm = off + (int32_t)t;
        const bool isMs = isM && e != 0u && Pm < (int32_t)d.wlen;
        uint32_t c0 = 0, c1 = 0;
        if (isI || isMs) qbits(isI ? (int32_t)q : (int32_t)q - (Pm & 31), c0, c1);
        if (isMs) {
          const uint32_t wP = (uint32_t)Pm >> 5;
          const uint32_t seg = mask_range(Pm & 31, min((Pm & 31) + (int32_t)e, (int32_t)d.wlen - (int32_t)(wP << 5)));
          atomicOr(&s_pM[wP], seg);
          atomicOr(&s_pL[wP], c0 & seg);
          atomicOr(&s_pH[wP], c1 & seg);
        }
        if (isI) {
          const uint32_t idx = n_ev + (uint32_t)__popcll(imask & lt);
          const int32_t pos = off + (int32_t)t - 1;
          const uint32_t codes = (c0 & 0xffffu) | (c1 << 16);   // the first 16 bases as two bit planes (low code bits | high code bits << 16): the consumers pick single bases, interleaving here cost ~20 instructions per step
          ev[idx] = make_uint4(((uint32_t)pos & 0xffffu) | (min(e, 0xffffu) << 16), q, codes, min(len, 0xffffu));
        }
        const uint32_t ev_before = n_ev + (uint32_t)__popcll(imask & lt);   // insertion events of the slice in front of this op
        n_ev += (uint32_t)__popcll(imask);
        // table of M/D ops + bitmap of their first positions
        const bool isMD = (isM || isD) && e != 0u;
        const uint64_t mdmask = __ballot(isMD);
        if (isMD) {
          const uint32_t idx = n_md + (uint32_t)__popcll(mdmask & lt);
          const uint32_t P = (uint32_t)(off + (int32_t)t);
          s_mt[idx] = min(P, 0x3fffu) | (e << 14) | (isM ? 0x80000000u : 0u);   // P, e <= 8192
          s_mq[idx] = q;
          // the op that covers a word's first position = the last one that starts at or in front of it: table slots grow with the position, and so do the
          // events in front — a maximum per word here, a prefix maximum over the words below (round 6; a bitmap of op starts + a rank directory before)
          const uint32_t slot = (P + 31u) >> 5;
          if (slot < nw) atomicMax(&s_cov[slot], ((idx + 1u) << 12) | min(ev_before, 0xfffu));
        }
        n_md += (uint32_t)__popcll(mdmask);
        carry_t += last_t;
        carry_q += last_q;
      }
    }
    // the wave reads its own LDS writes back: LDS operations of one wave execute in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PROF_MARK(J, 0, 1);
    uint32_t cv[NWI];   // per plane word of this lane: the op that covers its first position (s_cov's format)
    {
      uint32_t carry = 0;
#pragma unroll
      for (int wi_i = 0; wi_i < NWI; wi_i++) {
        const uint32_t i = lane + 64u * wi_i;
        const uint32_t v = max(wscan_max(i < nw ? s_cov[i] : 0u), carry);
        cv[wi_i] = v;
        carry = wlast(v);
      }
    }
    PROF_MARK(J, 0, 2);
    // planes of the batch's positions: a lane owns 32 positions, the op covering the first one is a popcount away; the ops that START inside the word
    // have written their stretches themselves (above)
    const int32_t Pb1 = min(off + (int32_t)carry_t, (int32_t)d.wlen);
#pragma unroll
    for (int wi_i = 0; wi_i < NWI; wi_i++) {
      const uint32_t wi = lane + 64u * wi_i;
      const int32_t ws = (int32_t)(wi << 5);
      if (wi < nw && ws >= Pb0 && ws < Pb1) {
        if (cv[wi_i]) {
          const uint32_t r = (cv[wi_i] >> 12) - 1u;
          const uint32_t ll = s_mt[r];
          const int32_t tP = (int32_t)(ll & 0x3fffu);
          const uint32_t qq = s_mq[r];
          dQ[wi_i] = qq + ((ll >> 31) ? (uint32_t)(ws - tP) : 0u);
          dE[wi_i] = cv[wi_i] & 0xfffu;
          if (ll >> 31) {
            const uint32_t seg = mask_range(0, min(Pb1, tP + (int32_t)((ll >> 14) & 0x3fffu)) - ws);
            uint32_t c0, c1;
            qbits((int32_t)qq + (ws - tP), c0, c1);
            pM[wi_i] |= seg;
            pL[wi_i] |= c0 & seg;
            pH[wi_i] |= c1 & seg;
          }
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int wi_i = 0; wi_i < NWI; wi_i++) {
    const uint32_t wi = min(lane + 64u * wi_i, nw - 1u);
    pM[wi_i] |= s_pM[wi]; pL[wi_i] |= s_pL[wi]; pH[wi_i] |= s_pH[wi];
  }
  PROF_MARK(J, 0, 3);
  const uint32_t t_total = carry_t;
  const bool keep = __ballot(longindel != 0u) == 0ull;
  // accuracy: matches / mismatches over M ops (features.rs:650-665)
  uint32_t mm = 0, ss = 0;
#pragma unroll
  for (int wi_i = 0; wi_i < NWI; wi_i++) {
    if (lane + 64u * wi_i < nw) {
      const uint32_t x = (pL[wi_i] ^ tl[wi_i]) | (pH[wi_i] ^ th[wi_i]);
      ss += __popc(pM[wi_i] & x);
      mm += __popc(pM[wi_i]);
    }
  }
  {
    const uint32_t ms = wsum(mm | (ss << 16));   // both <= 8192
    mm = ms & 0xffffu; ss = ms >> 16;
    isum = wsum(isum); dsum = wsum(dsum);
  }
  if (lane == 0) {
    const uint32_t m_ = mm - ss;
    J.ow_keep[o] = keep ? 1 : 0;
    J.ow_acc[o] = __fdiv_rn((float)m_, (float)(m_ + ss + isum + dsum));   // (m as f32) / ((m+s+i+d) as f32) (features.rs:678)
    J.ow_ttotal[o] = t_total;
    J.ins_cnt[o] = n_ev;
    J.ocol[o] = make_uint4((uint32_t)off, t_total, keep ? 1u : 0u, d.cls);
  }
  if (keep) {
    PlaneRec* __restrict__ g = J.cw + (uint64_t)o * nw;
    uint32_t* __restrict__ gd = J.cwd + (uint64_t)o * nw;
#pragma unroll
    for (int wi_i = 0; wi_i < NWI; wi_i++) {
      const uint32_t wi = lane + 64u * wi_i;
      if (wi < nw) {
        const uint32_t de = dE[wi_i] == 0xffffffffu ? (((int32_t)(wi << 5) < off) ? 0u : n_ev) : dE[wi_i];
        // ONE 16-byte record per word of 32 positions: the three planes and the directory word {query index | events << 20} (round 5; three
        // plane arrays + an 8-byte directory record before: 20 bytes in four places — a slot of k_rfq touched four cache lines where it now
        // touches one, and k_win / k_rows issue one load per column instead of three).  Directory indices that do not fit (2^20 bases /
        // 2^12 events in one overlap-window) are flagged; k_rfq then counts
        g[wi] = PlaneRec{pM[wi_i], pL[wi_i], pH[wi_i]};
        gd[wi] = (dQ[wi_i] < (1u << 20) && de < 0xfffu && !(J.dbg_flags & 1u)) ? (dQ[wi_i] | (de << 20)) : 0xffffffffu;
      }
    }
  }
  PROF_MARK(J, 0, 4);
}

// =====================================================================================================
// k_win — one workgroup per window (blockDim.x >= nw): pass-1 informative positions, tallies, accuracy rank
// =====================================================================================================
constexpr uint32_t RKCAP = 1024;   // overlaps per window whose accuracies are ranked out of LDS

template <int NB>
__global__ __launch_bounds__(256) void k_win(JobDev J) {
  __shared__ float s_acc[RKCAP];
  __shared__ uint8_t s_keep[RKCAP];
  __shared__ uint32_t s_nm2[16], s_ns;  // first 32 columns: matches at informative positions, two columns per word (u | u + 16 << 16: a window has <= 8192 positions); informative positions of the window
  // windows are taken back to front: k_cols has just written the plane records front to back (268 MB per 4096 windows — more than the
  // 256 MB memory-side cache holds), so the last windows' records are the ones still cached; k_rows then walks front to back again (r5)
#ifdef HERRO_FWD_ORDER   // (A/B build only)
  const uint32_t w = blockIdx.x, tid = threadIdx.x, NT = blockDim.x, nw = J.nw;
#else
  const uint32_t w = gridDim.x - 1u - blockIdx.x, tid = threadIdx.x, NT = blockDim.x, nw = J.nw;
#endif
  if (tid < 16) s_nm2[tid] = 0;
  if (tid == 0) s_ns = 0;
  __syncthreads();
  PROF_BEGIN(J);
  const WinDesc wd = J.win[w];
  const uint32_t n = wd.ow_cnt;
  const uint64_t pmax = J.read_n_words + 1;
  const bool active = tid < nw;
  const uint32_t widx = min(tid, nw - 1u);
  const int32_t P = (int32_t)(widx << 5);
  const uint32_t vm = active ? mask_range(0, (int32_t)wd.win_len - P) : 0u;
  const uint64_t t_woff = J.read_word_off[wd.rid];
  const uint32_t tlo = glb_bits(J.read_p0, t_woff, pmax, (int32_t)wd.tstart + P) & vm;
  const uint32_t thi = glb_bits(J.read_p1, t_woff, pmax, (int32_t)wd.tstart + P) & vm;
  SlicedCounters<NB> cnt;
  cnt.clear();
  cnt.add(ColPlanes{vm, tlo, thi, 0u});  // the target column: always a base on these rows
  PROF_MARK(J, 1, 0);
  uint32_t n_kept = 0;
#ifndef HERRO_WIN_UB
#define HERRO_WIN_UB 8
#endif
  static_assert(HERRO_WIN_UB == 8, "the tallies pair column u with column u + 16");
  constexpr int UB = HERRO_WIN_UB;    // columns whose loads are in flight together (16: two round trips instead of four, but 152 VGPRs — measured no faster in r3; -DHERRO_WIN_UB for an A/B build)
  const uint4* __restrict__ ocol = J.ocol;   // read-only here: uniform addresses -> scalar loads
  uint32_t match[4 * UB];  // first 32 columns: positions where the column shows the target's base (kept for the tallies)
#pragma unroll
  for (int u = 0; u < 4 * UB; u++) match[u] = 0;
  uint32_t keepmask = 0;   // kept among the first 32 columns
  auto batch = [&](uint32_t c0, uint32_t* keep_match) {
    uint4 oc[UB];
    uint32_t M[UB], L[UB], H[UB];
#pragma unroll
    for (int u = 0; u < UB; u++) {
      const uint64_t o = wd.ow_begin + min(c0 + u, n - 1u);
      const PlaneRec g = J.cw[o * nw + widx];   // planes of overlaps that were not kept are never written: loaded, ignored
      oc[u] = ocol[o];
      M[u] = g.m; L[u] = g.lo; H[u] = g.hi;
    }
#pragma unroll
    for (int u = 0; u < UB; u++) {
      if (c0 + u < n && oc[u].z) {   // uniform
        n_kept++;
        const int32_t off = (int32_t)oc[u].x;
        const uint32_t inr = mask_range(off - P, off + (int32_t)oc[u].y - P) & vm;
        const uint32_t Mi = M[u] & inr;
        cnt.add(ColPlanes{Mi, L[u] & inr, H[u] & inr, inr & ~M[u]});
        if (keep_match) { keep_match[u] = Mi & ~((L[u] ^ tlo) | (H[u] ^ thi)); keepmask |= 1u << ((c0 + u) & 31u); }
      }
    }
  };
  if (n > 0) batch(0, match);
  if (n > UB) batch(UB, match + UB);
  if (n > 2 * UB) batch(2 * UB, match + 2 * UB);
  if (n > 3 * UB) batch(3 * UB, match + 3 * UB);
  for (uint32_t c0 = 4 * UB; c0 < n; c0 += UB) batch(c0, nullptr);
  PROF_MARK(J, 1, 1);
  // ---- informative positions: at least two symbols reach the threshold (features.rs:681-722)
  uint32_t sup;
  {
    const uint32_t ncols = 1u + (n_kept > 30u ? n_kept : 30u);  // features.rs:282
    const uint32_t thresh = (uint32_t)((double)ncols * 0.1);    // features.rs:712
    uint32_t one = 0, two = 0;
#pragma unroll
    for (int s5 = 0; s5 < 5; s5++) {
      const uint32_t g = cnt.ge(s5, thresh);
      two |= one & g;
      one |= g;
    }
    sup = (thresh == 0 ? 0xffffffffu : two) & vm;
  }
  // ---- tallies (features.rs:478-498): every kept column is scored at every informative target position; anything but
  // the target's base ('.', '*', '#', another base) is a mismatch.  Informative positions are rare: only the lanes whose
  // word holds one take part; the first 32 columns' match masks are still in registers, further columns are read again.
  // Round 6: the sums over the wave's lanes are DPP reductions, one LDS atomic per word from one lane.  (Every lane adding its own count to the column's
  // LDS word: the compiler turns an atomic on a wave-uniform address into a SCALAR loop over the active lanes — 33 of them here, ~13 k of the kernel's 66 k
  // cycles; profiles/r6_k_win_tallies.txt.)
  if (__ballot(sup != 0u)) {   // wave-uniform: all lanes inside
    const uint32_t lane = tid & 63u;
    const uint32_t nsup = (uint32_t)__popc(sup);
    uint32_t acc = 0;
#pragma unroll
    for (int u = 0; u < 2 * UB; u++) {   // match[] of a column that was not kept is 0
      const uint32_t r = wsum((uint32_t)__popc(sup & match[u]) | ((uint32_t)__popc(sup & match[u + 2 * UB]) << 16));   // <= 2048 per half
      if (lane == (uint32_t)u) acc = r;
    }
    const uint32_t ns_w = wsum(nsup);
    if (lane < 2u * UB && acc) atomicAdd(&s_nm2[lane], acc);
    if (lane == 2u * UB) atomicAdd(&s_ns, ns_w);
    for (uint32_t c0 = 4 * UB; c0 < n; c0 += UB) {
      uint4 oc2[UB];
      uint32_t M[UB], L[UB], H[UB];
#pragma unroll
      for (int u = 0; u < UB; u++) {
        const uint64_t o = wd.ow_begin + min(c0 + u, n - 1u);
        const PlaneRec g = J.cw[o * nw + widx];
        oc2[u] = ocol[o];
        M[u] = g.m; L[u] = g.lo; H[u] = g.hi;
      }
#pragma unroll
      for (int u = 0; u < UB; u++) {
        if (c0 + u < n && oc2[u].z) {   // uniform
          const int32_t off = (int32_t)oc2[u].x;
          const uint32_t inr = mask_range(off - P, off + (int32_t)oc2[u].y - P) & vm;
          const uint32_t nm = wsum((uint32_t)__popc(sup & M[u] & inr & ~((L[u] ^ tlo) | (H[u] ^ thi))));
          if (lane == 0u) {
            if (nm) atomicAdd(&J.nd[2 * (uint64_t)oc2[u].w], nm);
            if (ns_w - nm) atomicAdd(&J.nd[2 * (uint64_t)oc2[u].w + 1], ns_w - nm);
          }
        }
      }
    }
  }
  PROF_MARK(J, 1, 2);
  // ---- stable rank of the kept overlaps by descending accuracy: sort_by_key(-acc) (features.rs:386-409)
  const bool in_lds = n <= RKCAP;
  if (in_lds)
    for (uint32_t i = tid; i < n; i += NT) { s_acc[i] = J.ow_acc[wd.ow_begin + i]; s_keep[i] = J.ow_keep[wd.ow_begin + i]; }
  __syncthreads();
  if (tid < 32 && tid < n && ((keepmask >> tid) & 1u) && s_ns) {   // keepmask is uniform
    const uint32_t cls = ocol[wd.ow_begin + tid].w, nm = (s_nm2[tid & 15u] >> (tid & 16u)) & 0xffffu, ns = s_ns;
    if (nm) atomicAdd(&J.nd[2 * (uint64_t)cls], nm);
    if (ns - nm) atomicAdd(&J.nd[2 * (uint64_t)cls + 1], ns - nm);
  }
  // (two instantiations: a choice between an LDS and a global array made per ACCESS — `in_lds ? s_acc[j] : J.ow_acc[..]` — compiles to a select of
  // pointers and a flat load, 2 n^2 of them here; r5)
  auto rank_all = [&](auto lds) {
    constexpr bool LDS = decltype(lds)::value;
    for (uint32_t i = tid; i < n; i += NT) {
      const uint32_t oi = wd.ow_begin + i;
      const uint32_t ki = LDS ? (uint32_t)s_keep[i] : (uint32_t)J.ow_keep[oi];
      if (!ki) continue;
      const float ai = LDS ? s_acc[i] : J.ow_acc[oi];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n; j++) {
        const uint32_t kj = LDS ? (uint32_t)s_keep[j] : (uint32_t)J.ow_keep[wd.ow_begin + j];
        const float aj = LDS ? s_acc[j] : J.ow_acc[wd.ow_begin + j];
        if (kj && (aj > ai || (aj == ai && j < i))) rank++;
      }
      J.slot_ow[wd.ow_begin + rank] = oi;
    }
  };
  if (in_lds) rank_all(std::true_type{}); else rank_all(std::false_type{});
  if (tid == 0) J.win_nkept[w] = n_kept;
  PROF_MARK(J, 1, 3);
}

// =====================================================================================================
// k_layout — one workgroup per window: selection of the 30 columns, row of every position
// =====================================================================================================
constexpr int LY_NT = 256;
constexpr uint32_t TCAP = 2 * LY_NT;   // tiles per window (8192 positions x 51 rows / 1024 = 408)
// the per-position array is walked 16 (W / 256) consecutive positions per thread: one pad word per 16 positions makes the
// lanes' stride 17 instead of 16 (two LDS banks for the whole wave otherwise)
#define MI(p) ((p) + ((p) >> 4))
__host__ __device__ inline size_t layout_lds(uint32_t W) { return (size_t)(W + 1 + ((W + 1) >> 4) + 1) * 4; }

// TILES: the inserted-base runs are listed per tile of 1024 rows, a run once for every tile it reaches into (k_tokens works tile by tile);
// !TILES (the lean path): ONE flat list per window, every run once (k_rows walks them all) — no counting pass, no scan, no second walk.
template <bool TILES>
__global__ __launch_bounds__(LY_NT) void k_layout(JobDev J) {
  extern __shared__ __attribute__((aligned(16))) uint32_t ly_smem[];
  uint32_t* s_mi = ly_smem;   // [W+1] max insertion behind every position, then (in place) the row of every position
  __shared__ double s_score[SCAP];
  __shared__ uint32_t s_sel[32], s_nev[32], s_evoff[32], s_wave[LY_NT / 64], s_maxne, s_nrun;
  __shared__ uint32_t s_tcnt[TCAP], s_toff[TCAP];   // insertion events per tile of the window, first slot of each tile's list
  const uint32_t w = blockIdx.x, tid = threadIdx.x;
  PROF_BEGIN(J);
  const WinDesc wd = J.win[w];
  const uint32_t n_kept = J.win_nkept[w];
  const uint32_t win_len = wd.win_len;
  if (tid < 32) s_sel[tid] = NONE;
  if (tid == 0) { s_maxne = 0; s_nrun = 0; }
  if constexpr (TILES) for (uint32_t t = tid; t < TCAP; t += LY_NT) s_tcnt[t] = 0;
  for (uint32_t p = tid; p <= win_len; p += LY_NT) s_mi[MI(p)] = 0;
  __syncthreads();
  PROF_MARK(J, 2, 5);
  // ---- score n/(n+d)*ln(n+d+1) in f64 (features.rs:505-510); stable descending rank (features.rs:512-513)
  {
    auto score_of = [&](uint32_t k) -> double {
      const uint32_t cls = J.ocol[J.slot_ow[wd.ow_begin + k]].w;
      const uint32_t nn = J.nd[2 * (uint64_t)cls], dd = J.nd[2 * (uint64_t)cls + 1];
      const uint32_t tot = nn + dd;
      if (!tot) return 0.0;
      const double lg = tot < J.ln_table_n ? J.ln_table[tot] : log((double)tot + 1.0);
      return __dmul_rn(__ddiv_rn((double)nn, (double)tot), lg);
    };
    const bool cached = n_kept <= SCAP;
    if (cached) {
      for (uint32_t k = tid; k < n_kept; k += LY_NT) s_score[k] = score_of(k);
      __syncthreads();
      PROF_MARK(J, 2, 6);
    }
    for (uint32_t k = tid; k < n_kept; k += LY_NT) {
      // (round 6: the overlap's query id is requested BEFORE the ranking loop, and the loop reads its LDS scores eight at a time — one dependent LDS round trip per score and two
      // dependent global ones behind the loop were 12 k of the kernel's 46 k cycles)
      const uint32_t o = J.slot_ow[wd.ow_begin + k];
      const uint32_t qid = J.ow[o].qid;
      const double sk = cached ? s_score[k] : score_of(k);
      uint32_t rank = 0;
      if (cached) {
        uint32_t i = 0;
        for (; i + 8 <= n_kept; i += 8) {
          double sv[8];
#pragma unroll
          for (int u = 0; u < 8; u++) sv[u] = s_score[i + u];
#pragma unroll
          for (int u = 0; u < 8; u++) if (sv[u] > sk || (sv[u] == sk && i + u < k)) rank++;
        }
        for (; i < n_kept; i++) {
          const double si = s_score[i];
          if (si > sk || (si == sk && i < k)) rank++;
        }
      } else {
        for (uint32_t i = 0; i < n_kept; i++) {
          const double si = score_of(i);
          if (si > sk || (si == sk && i < k)) rank++;
        }
      }
      J.rank_qid[wd.ow_begin + rank] = qid;
      if (rank < 30u) s_sel[rank + 1] = o;
    }
    __syncthreads();
  }
  PROF_MARK(J, 2, 0);
  // ---- the window's column table: everything the token and quality kernels need to know about a column
  if (tid < 32) {
    CTab t;
    t.off = 0; t.t_total = 0; t.tokc = (uint32_t)TOK_GAP_F << 8; t.ow = NONE; t.n_ev = 0; t.ev_off = 0; t.sbase = 0; t.sdir = 1;
    t.qual_off = 0; t.q_woff = 0;
    if (tid == 0) {   // the target
      t.t_total = win_len;
      t.q_woff = J.read_word_off[wd.rid];
      t.qual_off = J.read_qual_off[wd.rid];
    } else if (s_sel[tid] != NONE) {
      const uint32_t o = s_sel[tid];
      const OwDesc& d = J.ow[o];
      t.off = (int32_t)(d.tstart - d.wtstart);
      t.t_total = J.ow_ttotal[o];
      t.tokc = d.strand ? (5u | ((uint32_t)TOK_GAP_R << 8)) : ((uint32_t)TOK_GAP_F << 8);
      t.ow = o;
      t.n_ev = J.ins_cnt[o];
      t.ev_off = d.scr_off;
      t.sbase = d.strand ? (int32_t)(d.qbeg + d.qlen - 1u) : (int32_t)d.qbeg;
      t.sdir = d.strand ? -1 : 1;
      t.qual_off = d.q_qual_off;
      t.q_woff = d.q_woff;
    }
    // entry 0 (the target) also carries the window's compact event array: first slot (tev), selected events in all
    const uint32_t nev = tid == 0 ? 0u : t.n_ev;
    const uint32_t inc = wscan_incl(nev);   // lanes 0..31 of the first wave
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 31);
    if (tid == 0) {
      t.ev_off = (uint32_t)wd.ev_off;
      t.n_ev = tot;
    }
    J.ctab[(uint64_t)w * 32 + tid] = t;
    J.sel_ow[(uint64_t)w * 32 + tid] = s_sel[tid];
    s_nev[tid] = nev;
    s_evoff[tid] = tid == 0 ? 0u : t.ev_off;
    if (tid == 0) s_evoff[0] = t.ev_off;   // s_evoff[0]: the window's base in tev
    atomicMax(&s_maxne, nev);
  }
  __syncthreads();
  PROF_MARK(J, 2, 1);
  // ---- insertion events of the selected columns, 8 threads per column, 8 events per thread in flight (no search for an
  // event's column).  First use: max insertion behind every position over the SELECTED overlaps — rows where every selected
  // column is a gap are dropped (features.rs:531-556), i.e. the final layout is the row map of the selected overlaps alone.
  const uint32_t lc = tid >> 3, sub = tid & 7u;
  const uint32_t ne = s_nev[lc];
  const uint64_t eo = s_evoff[lc], wbase = s_evoff[0];
  const uint32_t nbatch = (s_maxne + 63u) / 64u;
  uint4 v[8];
  uint32_t hx[8];   // position | length of the event behind it (a later insertion at the same position hides this one's first rows)
  auto load = [&](uint32_t b) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t i = b * 64u + sub + 8u * k;
      v[k] = ne ? J.iev[eo + min(i, ne - 1u)] : make_uint4(0, 0, 0, 0);
      if constexpr (TILES) hx[k] = ne ? J.iev[eo + min(i + 1u, ne - 1u)].x : 0u; else hx[k] = 0u;   // (only the run lists look at the event behind)
    }
  };
  for (uint32_t b = 0; b < nbatch; b++) {
    load(b);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t i = b * 64u + sub + 8u * k;
      if (i < ne) {
        const uint32_t p = v[k].x & 0xffffu;
        if (p < win_len) atomicMax(&s_mi[MI(p)], v[k].w);   // untrimmed length (features.rs:64-79)
      }
    }
  }
  __syncthreads();
  PROF_MARK(J, 2, 2);
  // ---- row of every position = exclusive prefix of (1 + max insertion); first position of every chunk of ROWCAP rows
  {
    const uint32_t ch = (win_len + LY_NT - 1) / LY_NT;
    const uint32_t p0 = min(tid * ch, win_len), p1 = min(p0 + ch, win_len);
    uint32_t local = 0;
    for (uint32_t p = p0; p < p1; p++) local += 1u + s_mi[MI(p)];
    uint32_t Lf;
    uint32_t r = blk_scan<LY_NT>(local, &Lf, s_wave);
    const uint32_t tile0 = (uint32_t)wd.col_off;
    for (uint32_t p = p0; p < p1; p++) {
      const uint32_t rn = r + 1u + s_mi[MI(p)];
      s_mi[MI(p)] = r;
      const uint32_t i = (rn - 1u) / ROWCAP;          // a position has at most 51 rows: it holds at most one chunk start
      if (i * ROWCAP >= r) J.chdr2[tile0 + i] = make_uint2(p, r == i * ROWCAP ? 1u : 0u);
      r = rn;
    }
    if (tid == 0) { s_mi[MI(win_len)] = Lf; J.win_Lf[w] = Lf; }
  }
  __syncthreads();
  PROF_MARK(J, 2, 3);
  for (uint32_t p = tid; p <= win_len; p += LY_NT) J.row_of_pos2[wd.pos_off + p] = s_mi[MI(p)];
  // ---- the events once more, now that rows are known: every tile of ROWCAP rows gets the list of the inserted-base runs that
  // reach into it {position | length << 16, query index, first 16 bases, column | hidden rows << 8} (k_tokens, phase B)
  const uint32_t tile0 = (uint32_t)wd.col_off, n_t = (s_mi[MI(win_len)] + ROWCAP - 1) / ROWCAP;
  auto each_run = [&](uint32_t b, auto&& fn) {   // fn(event, hidden rows, first tile, last tile) for this thread's events of batch b
    if (nbatch > 1) load(b);                     // a single batch is still in registers
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t i = b * 64u + sub + 8u * k;
      if (i >= ne) continue;
      const uint32_t p = v[k].x & 0xffffu, len = v[k].x >> 16;
      if (p >= win_len) continue;
      const uint32_t rp = s_mi[MI(p)], room = s_mi[MI(p + 1)] - rp - 1u;
      uint32_t hide = 0;
      if (i + 1 < ne && (hx[k] & 0xffffu) == p) {
        hide = hx[k] >> 16;
        for (uint32_t e2 = i + 2; e2 < ne; e2++) {
          const uint32_t x3 = J.iev[eo + e2].x;
          if ((x3 & 0xffffu) != p) break;
          hide = max(hide, x3 >> 16);
        }
      }
      const uint32_t lenr = min(len, room);
      if (hide >= lenr) continue;
      fn(v[k], hide, (rp + 1u + hide) / ROWCAP, (rp + lenr) / ROWCAP);
    }
  };
  if constexpr (!TILES) {
    // the lean path lists nothing: k_rows walks the selected columns' events itself (round 6; a flat run list per window until then)
    (void)n_t; (void)tile0; (void)wbase; (void)each_run;
  } else {
  for (uint32_t b = 0; b < nbatch; b++)
    each_run(b, [&](const uint4&, uint32_t, uint32_t t_lo, uint32_t t_hi) {
      for (uint32_t t = t_lo; t <= t_hi && t < TCAP; t++) atomicAdd(&s_tcnt[t], 1u);
    });
  __syncthreads();
  {
    uint32_t tot;
    const uint32_t c0 = s_tcnt[2 * tid], c1 = s_tcnt[2 * tid + 1];
    const uint32_t ex = blk_scan<LY_NT>(c0 + c1, &tot, s_wave);
    s_toff[2 * tid] = ex;
    s_toff[2 * tid + 1] = ex + c0;
    if (2 * tid < n_t) J.tile_ev[tile0 + 2 * tid] = make_uint2(ex, c0);
    if (2 * tid + 1 < n_t) J.tile_ev[tile0 + 2 * tid + 1] = make_uint2(ex + c0, c1);
  }
  __syncthreads();
  for (uint32_t b = 0; b < nbatch; b++)
    each_run(b, [&](const uint4& e, uint32_t hide, uint32_t t_lo, uint32_t t_hi) {
      for (uint32_t t = t_lo; t <= t_hi && t < TCAP; t++)
        J.tev[wbase + atomicAdd(&s_toff[t], 1u)] = make_uint4(e.x, e.y, e.z, lc | (hide << 8) | (t << 16));   // hide <= 50 (a kept overlap has no longer insertion, features.rs:315-324), t < TCAP
    });
  }
  PROF_MARK(J, 2, 4);
}

// =====================================================================================================
// k_tokens — one workgroup per ROWCAP rows of a window's final matrix
// =====================================================================================================
constexpr int TK_NT = 256;
constexpr uint32_t WPAD = 36;    // plane words staged per column: a chunk spans <= 1025 positions

__device__ __forceinline__ uint32_t dpp_quad_add(uint32_t v) {   // sum over the 4 lanes of a quad, in every lane
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ uint32_t dpp_quad_lane0(uint32_t v) {  // value of the quad's first lane
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xf, 0xf, true);  // quad_perm [0,0,0,0]
}

// aux != 0 (the planes path): informative rows and the decoder's votes come out of this kernel too; aux == 0 (planes requested behind a lean
// featurize): token planes and row map only — k_rows' lists and votes are left alone.
__global__ __launch_bounds__(TK_NT, 4) void k_tokens(JobDev J, uint32_t aux) {
  __shared__ __attribute__((aligned(16))) CTab s_ct[32];
  __shared__ __attribute__((aligned(16))) uint16_t s_rowinfo[ROWCAP];   // (position - pa) | base-row flag << 15
  __shared__ uint32_t s_adj[ROWCAP];                                    // per row: inserted A, C, G, T (5 bits each), '*' they replace (bits 20..)
  __shared__ uint32_t s_pl[HERRO_ROWS * 3 * WPAD];                      // [column][plane][word - w_lo]
  __shared__ uint32_t s_rop[ROWCAP + 4];                                // row of positions pa .. pb1
  __shared__ uint32_t s_supbits[ROWCAP / 32];
  const uint32_t tile = blockIdx.x, tid = threadIdx.x, nw = J.nw;
  PROF_BEGIN(J);
  // round trip 1: which rows of which window
  const uint32_t w = J.tile_win[tile], r0 = J.tile_r0[tile];
  // round trip 2: the window, this chunk's first position (and the next chunk's), the column table
  const uint32_t Lf = J.win_Lf[w];
  if (r0 >= Lf) return;
  const WinDesc wd = J.win[w];
  const uint2 ch0 = J.chdr2[tile];
  const bool last = r0 + ROWCAP >= Lf;
  const uint2 ch1 = J.chdr2[last ? tile : tile + 1];
  const uint2 tev_h = J.tile_ev[tile];   // the tile's inserted-base runs: first slot (relative to the window's), count
  if (tid < 32) s_ct[tid] = J.ctab[(uint64_t)w * 32 + tid];
  const uint32_t win_len = wd.win_len;
  const uint32_t r1 = min(r0 + ROWCAP, Lf);
  const uint32_t pa = ch0.x;                                                   // position of row r0
  const uint32_t pb1 = last ? win_len : ch1.x + (ch1.y ? 0u : 1u);            // positions whose base row lies before r1
  const uint32_t pb = pb1 - 1;
  const uint32_t w_lo = pa >> 5, wcnt = (pb >> 5) - w_lo + 2;
  const uint64_t pmax = J.read_n_words + 1;
  const uint32_t nrows = r1 - r0, nrows16 = (nrows + 15u) & ~15u;
  s_adj[tid] = 0; s_adj[tid + TK_NT] = 0; s_adj[tid + 2 * TK_NT] = 0; s_adj[tid + 3 * TK_NT] = 0;
  if (tid < ROWCAP / 32) s_supbits[tid] = 0;
  __syncthreads();
  PROF_MARK(J, 3, 0);
  // round trip 3: rows of the chunk's positions, plane words of the chunk's positions for every column
  {
    const uint32_t npos = pb1 - pa + 1;   // entries pa .. pb1
    uint32_t rv[5];
#pragma unroll
    for (int u = 0; u < 5; u++) rv[u] = J.row_of_pos2[wd.pos_off + pa + min(tid + u * TK_NT, npos - 1u)];
    constexpr int PI = 12;   // 30 columns x 3 planes x 34 words / 256 threads
    constexpr uint32_t WST = 34;   // words staged per plane: a chunk spans <= 1025 positions (constant: the index arithmetic folds to multiplies)
    uint32_t pv[PI];
    constexpr uint32_t items = (HERRO_ROWS - 1) * 3 * WST;
#pragma unroll
    for (int u = 0; u < PI; u++) {
      const uint32_t it = min(tid + u * TK_NT, items - 1u);
      const uint32_t cp = it / WST, k = it - cp * WST, c = 1 + cp / 3, pi = cp - (c - 1) * 3;
      const uint32_t o = s_ct[c].ow;
      const uint32_t wi = min(w_lo + k, nw - 1u);
      pv[u] = reinterpret_cast<const uint32_t*>(J.cw)[((o != NONE ? (uint64_t)o : 0ull) * nw + wi) * 3 + pi];
    }
    if (tid < wcnt) {
      const int32_t P = (int32_t)((w_lo + tid) << 5);
      const uint32_t vm = mask_range(0, (int32_t)win_len - P);
      s_pl[0 * WPAD + tid] = vm;
      s_pl[1 * WPAD + tid] = glb_bits(J.read_p0, s_ct[0].q_woff, pmax, (int32_t)wd.tstart + P) & vm;
      s_pl[2 * WPAD + tid] = glb_bits(J.read_p1, s_ct[0].q_woff, pmax, (int32_t)wd.tstart + P) & vm;
    }
#pragma unroll
    for (int u = 0; u < 5; u++) if (tid + u * TK_NT < npos) s_rop[tid + u * TK_NT] = rv[u];
#pragma unroll
    for (int u = 0; u < PI; u++) {
      const uint32_t it = tid + u * TK_NT;
      if (it < items) {
        const uint32_t cp = it / WST, k = it - cp * WST, c = 1 + cp / 3, pi = cp - (c - 1) * 3;
        const bool live = s_ct[c].ow != NONE && w_lo + k < nw;
        s_pl[(c * 3 + pi) * WPAD + k] = live ? pv[u] : 0u;
      }
    }
  }
  __syncthreads();
  PROF_MARK(J, 3, 1);
  auto rop = [&](uint32_t p) -> uint32_t { return s_rop[p - pa]; };   // p in [pa, pb1]
  // -- A: position (relative to pa) and base-row flag of every row of the chunk, the window's row map
  for (uint32_t p = pa + tid; p <= pb; p += TK_NT) {
    const uint32_t rp = rop(p), nr = rop(p + 1) - rp;
    for (uint32_t j = 0; j < nr; j++) {
      const uint32_t row = rp + j;
      if (row >= r0 && row < r1) {
        s_rowinfo[row - r0] = (uint16_t)((p - pa) | (j == 0 ? 0x8000u : 0u));
        J.rowmap2[wd.row_off + row] = p | (j << 16);
      }
    }
  }
  for (uint32_t i = nrows + tid; i < nrows16; i += TK_NT) s_rowinfo[i] = (uint16_t)(pb - pa);   // rows past the window's last: unused
  __syncthreads();
  PROF_MARK(J, 3, 2);
  // -- C1: 16 rows x 1 column per step, four rows per register.  A quad of lanes shares a row segment, each lane takes 8
  // consecutive columns.  Position space -> row space is a byte permute: a base row takes the byte of its position, an
  // insertion row the default of its position ('*' inside the overlap, '.' outside); inserted bases are patched in by B.
  const uint32_t seg = tid >> 2, cg = tid & 3u;
  const bool seg_live = seg * 16u < nrows;
  uint32_t cAC[4] = {0, 0, 0, 0}, cGT[4] = {0, 0, 0, 0}, cS[4] = {0, 0, 0, 0};   // per row (byte): A | C << 4, G | T << 4, '*' seen in this lane's columns
  uint32_t tgt[4] = {0, 0, 0, 0};
  if (seg_live) {
    const uint4 ri0 = reinterpret_cast<const uint4*>(s_rowinfo)[seg * 2], ri1 = reinterpret_cast<const uint4*>(s_rowinfo)[seg * 2 + 1];
    const uint32_t riw[8] = {ri0.x, ri0.y, ri0.z, ri0.w, ri1.x, ri1.y, ri1.z, ri1.w};
    const uint32_t pf_rel = riw[0] & 0x7fffu;          // first position of the segment, relative to pa
    const uint32_t p_first = pa + pf_rel;
    uint32_t kb[4], selT[4], selS[4];   // per register of four rows: position offset of its first row, byte selectors
#pragma unroll
    for (int d = 0; d < 4; d++) {
      kb[d] = ((riw[2 * d] & 0x7fffu) - pf_rel);
      selT[d] = 0; selS[d] = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t v = (riw[(4 * d + i) >> 1] >> (((4 * d + i) & 1) * 16)) & 0xffffu;
        const uint32_t dk = (v & 0x7fffu) - pf_rel - kb[d];   // 0..3: rows advance at most one position each
        const uint32_t j0 = v >> 15;
        selT[d] |= (j0 ? dk : 4u + dk) << (8 * i);            // tokens: base row <- token of its position, insertion row <- default of its position
        selS[d] |= (j0 ? dk : 0x0cu) << (8 * i);              // base symbols: insertion rows count nothing here (0x0c selects the constant 0)
      }
    }
    const uint32_t wrel = (p_first >> 5) - w_lo, sh = p_first & 31u;
    const uint64_t gseg = wd.fin_off + r0 + seg * 16u;
#pragma unroll 1
    for (uint32_t ci = 0; ci < 8; ci++) {
      const uint32_t c = cg * 8 + ci;   // column 31 (last lane of the quad, ci = 7) is a stand-in: '.', never stored
      const CTab& h = s_ct[c];
      const uint32_t* pl = s_pl + (size_t)min(c, (uint32_t)HERRO_ROWS - 1u) * 3 * WPAD + wrel;
      const bool real = c < HERRO_ROWS;
      const uint32_t rmask = real ? 0xffffffffu : 0u;   // loads unconditional (the pointer is clamped), then masked: a predicated load is a branch and a wait each
      const uint32_t m32 = __funnelshift_r(pl[0], pl[1], sh) & rmask;
      const uint32_t l32 = __funnelshift_r(pl[WPAD], pl[WPAD + 1], sh) & rmask;
      const uint32_t h32 = __funnelshift_r(pl[2 * WPAD], pl[2 * WPAD + 1], sh) & rmask;
      const uint32_t tt = h.t_total & rmask;
      const uint32_t r32 = mask_range(h.off - (int32_t)p_first, h.off + (int32_t)tt - (int32_t)p_first);
      const bool rev = (h.tokc & 0xffu) != 0;                                       // reverse strand: tokens a c g t #
      const uint32_t lut_lo = rev ? 0x09090909u : 0x04040404u;                      // no query base here: gap
      const uint32_t lut_hi = rev ? 0x08070605u : 0x03020100u;                      // query base: its code + strand offset
      const uint32_t tenx = (uint32_t)TOK_NONE * 0x01010101u;
      uint32_t T[4];
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const uint32_t k = kb[d];
        const uint32_t m4 = __builtin_amdgcn_ubfe(m32, k, 4), l4 = __builtin_amdgcn_ubfe(l32, k, 4), h4 = __builtin_amdgcn_ubfe(h32, k, 4),
                       r4 = __builtin_amdgcn_ubfe(r32, k, 4);
        // four positions -> four bytes: bit j of x lands on bit 8 j of x * 0x204081 (the partial products do not overlap)
        const uint32_t idx = ((__umul24(m4, 0x810204u) & 0x04040404u) | (__umul24(h4, 0x408102u) & 0x02020202u)) | (__umul24(l4, 0x204081u) & 0x01010101u);   // per byte: M << 2 | hi << 1 | lo
        const uint32_t Rb = __umul24(r4, 0x204081u) & 0x01010101u;
        uint32_t Rm;                                                                // 0xff in the bytes inside the overlap: (Rb << 8) - Rb
        asm("v_lshlrev_b32 %0, 8, %1" : "=v"(Rm) : "v"(Rb));                        // (kept from being folded into a quarter-rate 32-bit multiply)
        Rm -= Rb;
        const uint32_t tokMP = __builtin_amdgcn_perm(lut_hi, lut_lo, idx);
        const uint32_t tokP = (tokMP & Rm) | (tenx & ~Rm);                          // per position: base / gap / '.'
        const uint32_t defP = (lut_lo & Rm) | (tenx & ~Rm);                         // ... of an insertion row behind it: gap / '.'
        T[d] = __builtin_amdgcn_perm(defP, tokP, selT[d]);
        const uint32_t acP = __builtin_amdgcn_perm(0x00001001u, 0u, idx);           // A -> 0x01, C -> 0x10
        const uint32_t gtP = __builtin_amdgcn_perm(0x10010000u, 0u, idx);           // G -> 0x01, T -> 0x10
        const uint32_t Sb = Rb & ~(idx >> 2);                                       // inside the overlap, no base: '*' (bit 0 of every byte)
        cAC[d] += __builtin_amdgcn_perm(0u, acP, selS[d]);
        cGT[d] += __builtin_amdgcn_perm(0u, gtP, selS[d]);
        cS[d] += __builtin_amdgcn_perm(Rb, Sb & 0x01010101u, selT[d]);              // an insertion row shows '*' wherever the overlap covers its position
      }
      if (c < HERRO_ROWS) *reinterpret_cast<uint4*>(J.fin_b + gseg + (uint64_t)c * wd.lub) = make_uint4(T[0], T[1], T[2], T[3]);
      if (ci == 0) { tgt[0] = T[0]; tgt[1] = T[1]; tgt[2] = T[2]; tgt[3] = T[3]; }   // column 0 sits in the quad's first lane
    }
  }
  __threadfence_block();
  __syncthreads();   // the default tokens are out before the inserted bases go over them
  PROF_MARK(J, 3, 3);
  // -- B: inserted bases of the selected columns (features.rs:213-229): the token byte over the default, the row's symbol
  // counts adjusted in LDS.  The tile's runs were listed by k_layout (a later insertion at the same position overwrites an
  // earlier one from its first row on, as the reference's sequential writes do: those rows are "hidden").
  {
    const uint4* __restrict__ tev = J.tev + s_ct[0].ev_off + tev_h.x;
    for (uint32_t e0 = tid; e0 < tev_h.y; e0 += 2 * TK_NT) {
      uint4 ve[2];
#pragma unroll
      for (int u = 0; u < 2; u++) ve[u] = tev[min(e0 + u * TK_NT, tev_h.y - 1u)];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (e0 + u * TK_NT >= tev_h.y) continue;
        const uint32_t c = ve[u].w & 0xffu, hide = (ve[u].w >> 8) & 0xffu;
        const uint32_t p = ve[u].x & 0xffffu, len = ve[u].x >> 16;
        if (p < pa || p > pb) continue;
        const uint32_t rp = rop(p), room = rop(p + 1) - rp - 1u;
        const uint32_t s5 = s_ct[c].tokc & 0xffu;
        const bool inr = (uint32_t)((int32_t)p - s_ct[c].off) < s_ct[c].t_total;   // the default under it was '*' (counted), not '.'
        for (uint32_t k = hide; k < len && k < room; k++) {
          const uint32_t row = rp + 1u + k;
          if (row < r0 || row >= r1) continue;
          uint32_t code;
          if (k < 16u) code = ((ve[u].z >> k) & 1u) | (((ve[u].z >> (16u + k)) & 1u) << 1);
          else {   // long insertion: bases beyond the 16 carried by the event come from the read store
            const int32_t si = s_ct[c].sbase + s_ct[c].sdir * (int32_t)(ve[u].y + k);
            const uint64_t wi = min(s_ct[c].q_woff + ((uint32_t)si >> 5), pmax);
            code = ((J.read_p0[wi] >> ((uint32_t)si & 31u)) & 1u) | (((J.read_p1[wi] >> ((uint32_t)si & 31u)) & 1u) << 1);
            if (s_ct[c].sdir < 0) code ^= 3u;
          }
          J.fin_b[wd.fin_off + (uint64_t)c * wd.lub + row] = (uint8_t)(s5 + code);
          atomicAdd(&s_adj[row - r0], (1u << (5u * code)) + (inr ? 1u << 20 : 0u));
        }
      }
    }
  }
  __syncthreads();
  PROF_MARK(J, 3, 4);
  // -- C2: counts of every row over all 31 columns; each lane of the quad finishes 4 rows
  if (seg_live) {
    uint32_t cnt8[5][4];   // per symbol: four registers of four row-bytes
#pragma unroll
    for (int d = 0; d < 4; d++) {
      cnt8[0][d] = dpp_quad_add(cAC[d] & 0x0f0f0f0fu);
      cnt8[1][d] = dpp_quad_add((cAC[d] >> 4) & 0x0f0f0f0fu);
      cnt8[2][d] = dpp_quad_add(cGT[d] & 0x0f0f0f0fu);
      cnt8[3][d] = dpp_quad_add((cGT[d] >> 4) & 0x0f0f0f0fu);
      cnt8[4][d] = dpp_quad_add(cS[d]);
    }
    const uint32_t tg0 = dpp_quad_lane0(tgt[0]), tg1 = dpp_quad_lane0(tgt[1]), tg2 = dpp_quad_lane0(tgt[2]), tg3 = dpp_quad_lane0(tgt[3]);
    const uint32_t tgw = cg == 0 ? tg0 : (cg == 1 ? tg1 : (cg == 2 ? tg2 : tg3));
    uint32_t cw[5];
#pragma unroll
    for (int q = 0; q < 5; q++) cw[q] = cg == 0 ? cnt8[q][0] : (cg == 1 ? cnt8[q][1] : (cg == 2 ? cnt8[q][2] : cnt8[q][3]));
    const uint4 adj = *reinterpret_cast<const uint4*>(s_adj + seg * 16u + cg * 4u);
    const uint32_t adjw[4] = {adj.x, adj.y, adj.z, adj.w};
    uint32_t supb = 0, consw = 0;
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) {
      uint32_t c5[5];
#pragma unroll
      for (int q = 0; q < 4; q++) c5[q] = ((cw[q] >> (8 * q4)) & 0xffu) + ((adjw[q4] >> (5 * q)) & 31u);
      c5[4] = ((cw[4] >> (8 * q4)) & 0xffu) - (adjw[q4] >> 20);
      // informative rows of the final [L',31] matrix: thresh = (31 * 0.1) as usize = 3 (features.rs:558,712)
      const uint32_t thresh = (uint32_t)((double)HERRO_ROWS * 0.1);
      uint32_t ns = 0;
#pragma unroll
      for (int q = 0; q < 5; q++) ns += c5[q] >= thresh ? 1u : 0u;
      if (ns >= 2 && seg * 16u + cg * 4u + q4 < nrows) supb |= 1u << q4;
      // majority vote of the consensus decoder (consensus.rs:178-200): it looks at the first n_alns+1 rows of the pileup, and
      // every row beyond those is '.', which it skips anyway.  Two most common symbols by a stable descending sort (ties keep
      // A,C,G,T,* order), target tie-break.
      uint32_t c0 = c5[0], i0 = 0;
#pragma unroll
      for (uint32_t q = 1; q < 5; q++) if (c5[q] > c0) { c0 = c5[q]; i0 = q; }
      uint32_t c1 = 0, i1 = 5;
      bool have = false;
#pragma unroll
      for (uint32_t q = 0; q < 5; q++)
        if (q != i0 && (!have || c5[q] > c1)) { c1 = c5[q]; i1 = q; have = true; }
      const uint32_t tb0 = (tgw >> (q4 * 8)) & 0xffu;
      const uint32_t cons_v = (c0 < 2u || (c0 == c1 && (i0 == tb0 || i1 == tb0))) ? tb0 : i0;
      consw |= cons_v << (q4 * 8);
    }
    if (aux) *reinterpret_cast<uint32_t*>(J.cons_tmp + wd.row_off + r0 + seg * 16u + cg * 4u) = consw;
    if (supb) atomicOr(&s_supbits[(seg * 16u + cg * 4u) >> 5], supb << ((seg * 16u + cg * 4u) & 31u));
  }
  __syncthreads();
  PROF_MARK(J, 3, 5);
  // -- D: the chunk's informative positions, in row order (SupportedPos, features.rs:896-900), by the first wave; the window's
  // list is put together by k_supgather
  if (aux && tid < 64) {
    const uint32_t bits = tid < ROWCAP / 32 ? s_supbits[tid] : 0u;
    const uint32_t inc = wscan_incl((uint32_t)__popc(bits));
    uint32_t k = inc - (uint32_t)__popc(bits);
    for (uint32_t m = bits; m; m &= m - 1u) {
      const uint32_t ric = (tid << 5) + (uint32_t)__ffs((int)m) - 1u;
      const uint32_t p = pa + (s_rowinfo[ric] & 0x7fffu), row = r0 + ric;
      J.sup_row[wd.row_off + r0 + k] = row;
      J.sup_pi[wd.row_off + r0 + k] = p | ((row - rop(p)) << 16);
      k++;
    }
    if (tid == 63) J.tile_nsup[tile] = inc;
  }
  PROF_MARK(J, 3, 6);
}

// =====================================================================================================
// k_supgather — one wave per window: the chunks' lists of informative positions -> one list, chunk order = row order
// =====================================================================================================
__global__ __launch_bounds__(64) void k_supgather(JobDev J) {
  const uint32_t w = blockIdx.x, lane = threadIdx.x;
  const WinDesc wd = J.win[w];
  const uint32_t Lf = J.win_Lf[w], tile0 = (uint32_t)wd.col_off;
  const uint32_t nch = (Lf + ROWCAP - 1) / ROWCAP;
  uint32_t dst = 0;
  for (uint32_t c0 = 0; c0 < nch; c0 += 64) {
    const uint32_t i = c0 + lane;
    const uint32_t cn = i < nch ? J.tile_nsup[tile0 + i] : 0u;
    const uint32_t inc = wscan_incl(cn);
    const uint32_t my_dst = dst + inc - cn;
    for (uint32_t j = 0; j < 64 && c0 + j < nch; j++) {   // chunk by chunk: destinations never run ahead of unread sources
      const uint32_t cnt_j = (uint32_t)__builtin_amdgcn_readlane((int)cn, (int)j), dst_j = (uint32_t)__builtin_amdgcn_readlane((int)my_dst, (int)j);
      const uint32_t src = (c0 + j) * ROWCAP;
      if (src == dst_j) continue;
      for (uint32_t k = lane; k < cnt_j; k += 64) {
        const uint32_t a = J.sup_row[wd.row_off + src + k], b = J.sup_pi[wd.row_off + src + k];
        J.sup_row[wd.row_off + dst_j + k] = a;
        J.sup_pi[wd.row_off + dst_j + k] = b;
      }
    }
    dst += wlast(inc);
  }
  if (lane == 0) J.win_nsup[w] = dst;
}

// =====================================================================================================
// k_rows — one workgroup per window (blockDim.x >= nw): the LEAN path's replacement for k_tokens (round 5)
// =====================================================================================================
// Nobody on the inference path reads the [31][L'] token matrix as a matrix: the model reads the five rows around every informative
// row (~15 of ~4700 rows per window), the decoder reads one vote per row.  Both follow from per-row SYMBOL COUNTS, and the counts of
// a target position's base row are a position-space quantity: bit-sliced counters over the selected columns' planes, 32 positions
// per lane, as k_win does for pass 1 — ~30 bit operations per column and word instead of ~170 instructions per column and 16 rows.
//   * base rows: saturating 2-bit counters per symbol (the threshold is 3, features.rs:558,712) -> informative positions; the
//     decoder's vote (consensus.rs:178-200) is only consulted on rows that are NOT informative, where at most one symbol reaches 3:
//     "the symbol with >= 3", else the first symbol with exactly 2 (stable order A C G T *), with the target tie-break when two
//     symbols have 2, else the target — all expressible on the saturated counters;
//   * insertion rows (~13 % of the rows): '*' wherever the column covers the position (an exact 5-bit bit-sliced count), minus the
//     columns that hold an inserted base there, plus those bases — per-row accumulators in LDS fed from k_layout's run lists, as
//     in k_tokens; the general vote on exact counts.
// Out: the window's informative rows in row order (sup_row / sup_pi / win_nsup — no k_supgather), the base rows' votes as three bit
// planes (vpl), the insertion rows' votes as bytes indexed by insertion-row ordinal (cons_tmp).  No token plane, no row map: the
// receptive fields are gathered by k_rfq from the informative rows' (position, ordinal) and row_of_pos2; k_tokens builds the planes
// when somebody asks for them (herro_job_window_copy, the features writer, models with receptive fields above 8 rows).
#ifndef HERRO_RW_ICAP
#define HERRO_RW_ICAP 2048
#endif
constexpr uint32_t RW_ICAP = HERRO_RW_ICAP;   // insertion rows per pass (accumulators in LDS); a window with more takes several passes (-DHERRO_RW_ICAP for an A/B build)
#define RI(p) ((p) + ((p) >> 5))     // lane i walks positions 32 i ..: one pad word per 32 keeps the lanes on different banks
__host__ __device__ inline size_t rows_lds(uint32_t W) { return (size_t)(W + 2 + ((W + 2) >> 5) + 1) * 4; }

constexpr uint32_t RW_SUPCAP = 256;   // informative rows of a window whose receptive fields k_rows gathers itself (a window with more: k_rfq, for the whole job)
template <int SP>
__device__ __forceinline__ void rf_slot(const JobDev& J, const CTab* __restrict__ s_ct, const WinDesc& wd, uint32_t Lf, uint32_t half, uint32_t c,
                                        uint32_t srow, uint32_t pj, uint32_t nr, bool have_nr, uint4* __restrict__ out);

// NW lanes per half (128: windows up to 4096 positions, 256: up to 8192).  The workgroup is TWO halves of NW threads: both stage the rows and
// walk the run lists; for the symbol counts lane l of half h takes word l of columns 1 + 15 h .. 15 + 15 h (all fifteen records in flight at
// once), half 1 hands its counters over through LDS and half 0 — one lane per word — goes on alone.  (One half doing everything: 30 columns in
// three round trips of ten, 33 staging loads and ~10 runs per thread in a row — the kernel is a chain of latencies, not of arithmetic.)
template <int NW>
__global__ __launch_bounds__(2 * NW) void k_rows(JobDev J) {
  constexpr int NT = 2 * NW;
  extern __shared__ __attribute__((aligned(16))) uint32_t rw_smem[];
  uint32_t* s_rop = rw_smem;                                       // [win_len + 1] row of every position, padded (RI)
  uint32_t* s_mrg = rw_smem + rows_lds(J.window_size) / 4;         // [15][NW] half 1's counters on their way to half 0
  __shared__ __attribute__((aligned(16))) CTab s_ct[32];
  __shared__ uint32_t s_adj[RW_ICAP];                              // per insertion row: inserted A, C, G, T (5 bits each), '*' they replace (bits 20..)
  __shared__ __attribute__((aligned(16))) uint8_t s_iv[RW_ICAP];   // ... its vote | informative << 7
  __shared__ uint32_t s_rfbase;
  __shared__ uint32_t s_sup[3][RW_SUPCAP];                         // row, position | ordinal << 16, neighbours' row counts of the window's informative rows
  __shared__ uint32_t s_wave[NT / 64];
  const uint32_t w = blockIdx.x, tid = threadIdx.x, nw = J.nw;
  const uint32_t half = tid >= (uint32_t)NW ? 1u : 0u, lt = tid - half * NW;   // wave-uniform (NW is a multiple of 64)
  PROF_BEGIN(J);
  const WinDesc wd = J.win[w];
  const uint32_t Lf = J.win_Lf[w], win_len = wd.win_len;
  const uint32_t tile0 = (uint32_t)wd.col_off;
  {
    // every load of the prologue in flight at once (a loop of load -> LDS store pairs was 33 round trips in a row: 37 k of the
    // kernel's 106 k cycles in its first version)
    constexpr int RL = 17;   // NT * RL >= 16 * 2 NW + 1 positions: 256 threads cover windows of 4096, 512 threads 8192
    uint32_t rv[RL];
    const uint32_t* __restrict__ rop = J.row_of_pos2 + wd.pos_off;
#pragma unroll
    for (int u = 0; u < RL; u++) rv[u] = rop[min(tid + (uint32_t)u * NT, win_len)];
    if (tid < 32) s_ct[tid] = J.ctab[(uint64_t)w * 32 + tid];
#pragma unroll
    for (int u = 0; u < RL; u++) {
      const uint32_t p = tid + (uint32_t)u * NT;
      if (p <= win_len) s_rop[RI(p)] = rv[u];
    }
  }
  __syncthreads();
  PROF_MARK(J, 6, 0);
  // ---- 1: symbol counts of the base rows in position space
  const bool active = lt < nw;
  const uint32_t widx = min(lt, nw - 1u);
  const int32_t P = (int32_t)(widx << 5);
  const uint32_t vm = active ? mask_range(0, (int32_t)win_len - P) : 0u;
  const uint64_t pmax = J.read_n_words + 1;
  const uint32_t tlo = glb_bits(J.read_p0, s_ct[0].q_woff, pmax, (int32_t)wd.tstart + P) & vm;
  const uint32_t thi = glb_bits(J.read_p1, s_ct[0].q_woff, pmax, (int32_t)wd.tstart + P) & vm;
  uint32_t c0[5], c1[5];   // saturating 2-bit counters (0, 1, 2, >= 3) of A C G T *
  uint32_t nin[5];         // exact count of the columns that cover the position (target included)
  const uint32_t tsym[4] = {vm & ~tlo & ~thi, tlo & ~thi, thi & ~tlo, tlo & thi};
#pragma unroll
  for (int q = 0; q < 4; q++) { c0[q] = half ? 0u : tsym[q]; c1[q] = 0; }   // the target column is counted by half 0
  c0[4] = 0; c1[4] = 0;
  nin[0] = half ? 0u : vm; nin[1] = 0; nin[2] = 0; nin[3] = 0; nin[4] = 0;
  auto sat_add = [&](int q, uint32_t x) { sat2_add(c0[q], c1[q], x); };
  {
    constexpr int UB = 15;   // this half's columns: all their records in flight together
    const uint32_t cb = 1u + UB * half;
    uint32_t M[UB], L[UB], H[UB];
#pragma unroll
    for (int u = 0; u < UB; u++) {
      const uint32_t o = s_ct[cb + u].ow;
      const PlaneRec g = J.cw[(o != NONE ? (uint64_t)o : 0ull) * nw + widx];
      M[u] = g.m; L[u] = g.lo; H[u] = g.hi;
    }
    PROF_MARK(J, 6, 10);
#pragma unroll
    for (int u = 0; u < UB; u++) {
      const CTab& h = s_ct[cb + u];
      const uint32_t tt = h.ow != NONE ? h.t_total : 0u;
      const uint32_t inr = mask_range(h.off - P, h.off + (int32_t)tt - P) & vm;
      const uint32_t Mi = M[u] & inr, Li = L[u] & Mi, Hi = H[u] & Mi;
      sat_add(0, Mi & ~Li & ~Hi);
      sat_add(1, Li & ~Hi);
      sat_add(2, Hi & ~Li);
      sat_add(3, Li & Hi);
      sat_add(4, inr & ~Mi);
      uint32_t x = inr;
#pragma unroll
      for (int b = 0; b < 5; b++) { const uint32_t cy = nin[b] & x; nin[b] ^= x; x = cy; }
    }
  }
  // half 1's counters join half 0's: saturating sums of the 2-bit counters, a 5-bit ripple add of the cover counts
  if (half) {
#pragma unroll
    for (int q = 0; q < 5; q++) { s_mrg[(q) * NW + lt] = c0[q]; s_mrg[(5 + q) * NW + lt] = c1[q]; s_mrg[(10 + q) * NW + lt] = nin[q]; }
  }
  __syncthreads();
  if (!half) {
#pragma unroll
    for (int q = 0; q < 5; q++) sat2_merge(c0[q], c1[q], s_mrg[q * NW + lt], s_mrg[(5 + q) * NW + lt]);
    uint32_t cy = 0;
#pragma unroll
    for (int b = 0; b < 5; b++) {
      const uint32_t x = nin[b], y = s_mrg[(10 + b) * NW + lt];
      nin[b] = x ^ y ^ cy;
      cy = (x & y) | (cy & (x ^ y));
    }
  }
  const bool lane0 = !half && active;   // the lanes that go on: one per word of the window
  // half 1's counters are merged: their LDS now carries what the insertion rows need (round 6) — the position of every insertion row (written by the
  // threads that walk the runs), the cover counts as five bit planes, and per word of 32 positions which of them have an informative insertion row
  constexpr uint32_t IP_ROWS = (RW_ICAP * 2 + NW * 4 - 1) / (NW * 4);
  static_assert(IP_ROWS + 7 <= 15, "s_mrg holds the insertion rows' positions, five cover planes and two words per lane");
  uint16_t* s_ip = reinterpret_cast<uint16_t*>(s_mrg);          // [RW_ICAP]
  uint32_t* s_nin = s_mrg + IP_ROWS * NW;                       // [5][NW]
  uint32_t* s_insup = s_nin + 5 * NW;                           // [NW] positions with an informative insertion row
  uint32_t* s_nisup = s_insup + NW;                             // [NW] informative insertion rows of the word's positions
  if (!half) {   // (a lane's own column of s_mrg: read above by nobody else)
#pragma unroll
    for (int b = 0; b < 5; b++) s_nin[b * NW + lt] = nin[b];
    s_insup[lt] = 0;
    s_nisup[lt] = 0;
  }
  PROF_MARK(J, 6, 1);
  // informative positions (features.rs:681-722 on the final 31 columns: thresh = (31 * 0.1) as usize = 3) and the votes of the others
  // (the rules and their derivation: base_row_votes, pileup_core.h — shared with the host so that tests/test_vote_planes.py can run them on every count vector)
  const RowVotes rvt = base_row_votes(c0, c1, tsym, vm);
  const uint32_t supb = lane0 ? rvt.sup : 0u, V0 = rvt.v0, V1 = rvt.v1, V2 = rvt.v2;
  // positions of this lane's word that have insertion rows behind them
  uint32_t insmask = 0;
  if (lane0) {
    uint32_t rprev = s_rop[RI((uint32_t)P)];
#pragma unroll 4
    for (uint32_t k = 0; k < 32; k++) {
      const uint32_t p = (uint32_t)P + k;
      const uint32_t rn = s_rop[RI(min(p + 1u, win_len))];
      if (p < win_len && rn - rprev > 1u) insmask |= 1u << k;
      rprev = rn;
    }
    uint32_t* __restrict__ vp = J.vpl + (uint64_t)w * 4 * nw + lt;   // the votes' three planes + the positions with insertion rows (the decoder's emission loop)
    vp[0] = V0; vp[nw] = V1; vp[2 * nw] = V2; vp[3 * nw] = insmask;
  }
  PROF_MARK(J, 6, 2);
  // ---- 2: insertion rows.  Index of an insertion row = its ordinal among the window's insertion rows = row - position - 1.
  const uint32_t n_irows = Lf - win_len;
  uint32_t n_isup = 0, insup = 0;   // informative insertion rows of this lane's positions; positions that have one
  for (uint32_t ch0 = 0; ch0 < n_irows; ch0 += RW_ICAP) {
    if (ch0) __syncthreads();   // the previous pass's votes are out
    for (uint32_t i = tid; i < min(RW_ICAP, n_irows - ch0); i += NT) s_adj[i] = 0;
    __syncthreads();
    PROF_MARK(J, 6, 7);
    // inserted bases of the selected columns (features.rs:213-229), straight from the columns' event lists, NT / 32 threads per column (round 6: k_layout listed the
    // runs of a window for this loop — 19 KB written and read back per window, a fifth of k_layout's time; what it worked out per run, the rows hidden by a later
    // insertion of the same column at the same position, is three instructions here and a loop only where such a pair exists)
    constexpr uint32_t TPC = NT / 32;
    constexpr int EU = 6;   // ~5 events per thread at the bench workload: one round trip
    const uint32_t c = tid / TPC, sub = tid % TPC;   // (column 31 is never a selected overlap)
    const uint32_t ne = (c >= 1u && s_ct[c].ow != NONE) ? s_ct[c].n_ev : 0u;
    const uint4* __restrict__ iev = J.iev + s_ct[c].ev_off;
    for (uint32_t i0 = sub; i0 < ne; i0 += EU * TPC) {
      uint4 ve[EU];
      uint32_t nx[EU];
#pragma unroll
      for (int u = 0; u < EU; u++) {
        const uint32_t i = i0 + u * TPC;
        ve[u] = iev[min(i, ne - 1u)];
        nx[u] = iev[min(i + 1u, ne - 1u)].x;
      }
#pragma unroll
      for (int u = 0; u < EU; u++) {
        const uint32_t i = i0 + u * TPC;
        if (i >= ne) continue;
        const uint32_t p = ve[u].x & 0xffffu, len = ve[u].x >> 16;
        if (p >= win_len) continue;
        const uint32_t rp = s_rop[RI(p)], room = s_rop[RI(p + 1)] - rp - 1u;
        uint32_t hide = 0;   // rows a later insertion of this column at the same position overwrote (features.rs:219-228: the reference's sequential writes)
        if (i + 1u < ne && (nx[u] & 0xffffu) == p) {
          hide = nx[u] >> 16;
          for (uint32_t e2 = i + 2u; e2 < ne; e2++) {
            const uint32_t x3 = iev[e2].x;
            if ((x3 & 0xffffu) != p) break;
            hide = max(hide, x3 >> 16);
          }
        }
        const bool inr = (uint32_t)((int32_t)p - s_ct[c].off) < s_ct[c].t_total;   // the default under it was '*' (counted), not '.'
        for (uint32_t k = hide; k < len && k < room; k++) {
          const uint32_t ir = rp + k - p - ch0;   // row rp + 1 + k, ordinal row - p - 1
          if (ir >= RW_ICAP) continue;
          uint32_t code;
          if (k < 16u) code = ((ve[u].z >> k) & 1u) | (((ve[u].z >> (16u + k)) & 1u) << 1);
          else {   // long insertion: bases beyond the 16 carried by the event come from the read store
            const int32_t si = s_ct[c].sbase + s_ct[c].sdir * (int32_t)(ve[u].y + k);
            const uint64_t wi = min(s_ct[c].q_woff + ((uint32_t)si >> 5), pmax);
            code = ((J.read_p0[wi] >> ((uint32_t)si & 31u)) & 1u) | (((J.read_p1[wi] >> ((uint32_t)si & 31u)) & 1u) << 1);
            if (s_ct[c].sdir < 0) code ^= 3u;
          }
          atomicAdd(&s_adj[ir], (1u << (5u * code)) + (inr ? 1u << 20 : 0u));
          s_ip[ir] = (uint16_t)p;   // (every run that reaches the row writes the same position)
        }
      }
    }
    __syncthreads();
    PROF_MARK(J, 6, 8);
    // every insertion row is evaluated by one thread (round 6; by the lane that owns its position before — the vote is ~80 instructions a row, and the
    // workgroup waited for its busiest lane: 18 k of the kernel's 62 k cycles, profiles/r6_k_rows_phases.txt).  A row no run reached holds no inserted
    // base: '*' whatever the cover count, not informative.
    for (uint32_t i = tid; i < min(RW_ICAP, n_irows - ch0); i += NT) {
      const uint32_t adj = s_adj[i];
      uint32_t cover = 0, p = 0;
      if (adj) {
        p = s_ip[i];
        const uint32_t pw = p >> 5, pb = p & 31u;
        cover = ((s_nin[pw] >> pb) & 1u) | (((s_nin[NW + pw] >> pb) & 1u) << 1) | (((s_nin[2 * NW + pw] >> pb) & 1u) << 2) |
                (((s_nin[3 * NW + pw] >> pb) & 1u) << 3) | (((s_nin[4 * NW + pw] >> pb) & 1u) << 4);
      }
      const uint32_t c5[5] = {adj & 31u, (adj >> 5) & 31u, (adj >> 10) & 31u, (adj >> 15) & 31u, cover - (adj >> 20)};
      uint32_t ns = 0;
#pragma unroll
      for (int q = 0; q < 5; q++) ns += c5[q] >= 3u ? 1u : 0u;
      s_iv[i] = (uint8_t)(vote5(c5, 4u) | ((ns >= 2u ? 1u : 0u) << 7));   // the target shows '*' on an insertion row
      if (ns >= 2u) { atomicOr(&s_insup[p >> 5], 1u << (p & 31u)); atomicAdd(&s_nisup[p >> 5], 1u); }
    }
    __syncthreads();
    PROF_MARK(J, 6, 9);
    if (lane0) { insup = s_insup[lt]; n_isup = s_nisup[lt]; }   // (they accumulate over the passes)
    {
      const uint32_t nb = min(RW_ICAP, n_irows - ch0);
      uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(J.cons_tmp + wd.row_off + ch0);   // row_off and ch0 are multiples of 16
      for (uint32_t i = tid; i * 4u < nb; i += NT) dst[i] = reinterpret_cast<const uint32_t*>(s_iv)[i];
    }
  }
  PROF_MARK(J, 6, 3);
  // ---- 3: the informative rows in row order (SupportedPos, features.rs:896-900)
  const bool iv_lds = n_irows <= RW_ICAP;   // one pass: the flags are still in LDS; else they are read back from the votes just written
  if (!iv_lds) __syncthreads();
  uint32_t total;
  uint32_t kk = blk_scan<NT>((uint32_t)__popc(supb) + n_isup, &total, s_wave);
  for (uint32_t m = supb | insup; m; m &= m - 1u) {
    const uint32_t k = (uint32_t)__ffs((int)m) - 1u, p = (uint32_t)P + k;
    const uint32_t rp = s_rop[RI(p)];
    // rows of the positions around p (6 bits each, 0 outside the window): k_rfq names the cells of the rows around an informative row from these
    uint32_t nr = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int32_t q = (int32_t)p - 2 + i;
      if (q >= 0 && q < (int32_t)win_len) nr |= (s_rop[RI((uint32_t)q + 1u)] - s_rop[RI((uint32_t)q)]) << (6 * i);
    }
    if ((supb >> k) & 1u) {
      J.sup_row[wd.row_off + kk] = rp;
      J.sup_pi[wd.row_off + kk] = p;
      J.sup_nr[wd.row_off + kk] = nr;
      if (kk < RW_SUPCAP) { s_sup[0][kk] = rp; s_sup[1][kk] = p; s_sup[2][kk] = nr; }
      kk++;
    }
    if ((insup >> k) & 1u) {
      const uint32_t n_ins = s_rop[RI(p + 1)] - rp - 1u;
      for (uint32_t j = 0; j < n_ins; j++) {
        const uint32_t ir = rp - p + j;
        uint32_t fl;   // (explicit branches: a select between the two arrays is a flat load)
        if (iv_lds) fl = (uint32_t)s_iv[ir]; else fl = (uint32_t)J.cons_tmp[wd.row_off + ir];
        if (fl & 0x80u) {
          J.sup_row[wd.row_off + kk] = rp + 1u + j;
          J.sup_pi[wd.row_off + kk] = p | ((j + 1u) << 16);
          J.sup_nr[wd.row_off + kk] = nr;
          if (kk < RW_SUPCAP) { s_sup[0][kk] = rp + 1u + j; s_sup[1][kk] = p | ((j + 1u) << 16); s_sup[2][kk] = nr; }
          kk++;
        }
      }
    }
  }
  if (tid == 0) J.win_nsup[w] = total;
  PROF_MARK(J, 6, 4);
  // ---- 4: the model's receptive fields of this window's informative rows (round 5; a kernel of its own before — k_rfq, 150 us per 4096 windows
  // for ~470 records per window: a launch, the informative rows read back, the plane records gone cold).  The records' place in the job's
  // buffer comes from one atomic per window (their order across windows is free: the model finds them through the window's base); a
  // window above RW_SUPCAP rows, or a buffer that turns out too small, leaves the whole job to k_rfq (the host sees both in the counts).
  if (J.rf) {   // wave-uniform
    if (tid == 0) {
      const uint32_t base = atomicAdd(J.rf_alloc, total);
      const bool room = (uint64_t)base + total <= J.rf_cap, ok = room && total <= RW_SUPCAP;
      s_rfbase = ok ? base : NONE;
      // a window above RW_SUPCAP rows keeps the slots it reserved and says so (RF_LEFT): k_rfq fills exactly those windows behind this kernel (round 6, ADVICE r5 — one
      // such window used to send the whole job through k_rfq); only a buffer that is too small (NONE) still does
      J.win_rfbase[w] = ok ? base : (room ? (base | RF_LEFT) : NONE);
    }
    __syncthreads();   // s_sup, s_rfbase
    const uint32_t base = s_rfbase;
    if (base != NONE) {
      const uint32_t nslots = total * HERRO_ROWS;
      for (uint32_t sl = tid; sl < nslots; sl += NT) {
        const uint32_t k = sl / HERRO_ROWS, c = sl - k * HERRO_ROWS;
        uint4* __restrict__ out = reinterpret_cast<uint4*>(J.rf + ((uint64_t)base * HERRO_ROWS + sl) * 16);
        if (J.rf_half <= 2u) rf_slot<5>(J, s_ct, wd, Lf, J.rf_half, c, s_sup[0][k], s_sup[1][k], s_sup[2][k], true, out);   // (uniform)
        else rf_slot<8>(J, s_ct, wd, Lf, J.rf_half, c, s_sup[0][k], s_sup[1][k], s_sup[2][k], true, out);
      }
    }
    PROF_MARK(J, 6, 5);
  }
}

// =====================================================================================================
// k_quals — one workgroup per window
// =====================================================================================================
constexpr int PQ_NT = 256;
constexpr uint32_t RFCAP = 1024;   // receptive-field rows per pass
__host__ __device__ inline size_t quals_lds(uint32_t nw) {
  return (size_t)(HERRO_ROWS - 1) * nw * 4 + (((size_t)(HERRO_ROWS - 1) * nw * 2 + 15) & ~(size_t)15) + (size_t)QEVCAP * 8;
}

// FULL: every cell of the window; otherwise the cells within `half` rows of an informative row (receptive fields wider than the 8 rows of
// k_rfq's compact records), into the quality planes.  Needs the row map (k_tokens).
template <bool FULL>
__global__ __launch_bounds__(PQ_NT) void k_quals(JobDev J, uint32_t half) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pq_smem[];
  const uint32_t nw = J.nw;
  uint32_t* s_M = reinterpret_cast<uint32_t*>(pq_smem);                               // [30][nw] M planes of the selected columns
  uint16_t* s_rk = reinterpret_cast<uint16_t*>(s_M + (size_t)(HERRO_ROWS - 1) * nw);  // [30][nw] M bits in front of the word
  uint2* s_ev = reinterpret_cast<uint2*>(pq_smem + (size_t)(HERRO_ROWS - 1) * nw * 4 + (((size_t)(HERRO_ROWS - 1) * nw * 2 + 15) & ~(size_t)15));
  __shared__ __attribute__((aligned(16))) CTab s_ct[32];
  __shared__ uint32_t s_evl[33];      // first LDS slot of every column's events
  __shared__ uint32_t s_rm[RFCAP];    // row-map entry of each receptive-field row (NONE: outside the window)
  __shared__ uint32_t s_rr[RFCAP];    // its row
  const uint32_t w = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // round trip 1
  const uint32_t nsup = J.win_nsup[w], Lf = J.win_Lf[w];
  const WinDesc wd = J.win[w];
  if (tid < 32) s_ct[tid] = J.ctab[(uint64_t)w * 32 + tid];
  if (!FULL && !nsup) return;
  __syncthreads();
  if (tid < 64) {   // first LDS slot of every column's events
    const uint32_t ne = (tid >= 1 && tid < 32) ? s_ct[tid].n_ev : 0u;   // entry 0 (the target) carries the window's totals
    const uint32_t inc = wscan_incl(ne);
    if (tid < 32) s_evl[tid] = inc - ne;
    if (tid == 31) s_evl[32] = inc;
  }
  __syncthreads();
  const uint32_t n_events = s_evl[32];
  const bool ev_in_lds = n_events <= QEVCAP;
  const uint32_t span = 2 * half + 1;
  const uint32_t kper = max(1u, RFCAP / span);   // informative rows per pass
  // round trip 2: M planes, events, the first pass's informative rows — all loads of a thread issued together
  {
    const uint32_t items = (HERRO_ROWS - 1) * nw;
    constexpr int MI = 16;
    uint32_t srow = 0;
    if (!FULL && tid < min(kper, nsup)) srow = J.sup_row[wd.row_off + tid];   // kper <= 256 whenever span >= 4; larger passes reload below
    constexpr int EI = QEVCAP / PQ_NT;
    uint2 ev[EI];
    if (ev_in_lds) {
#pragma unroll
      for (int u = 0; u < EI; u++) {
        const uint32_t e = min(tid + u * PQ_NT, n_events ? n_events - 1u : 0u);
        uint32_t lo = 0, hi = 32;
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s_evl[mid] <= e) lo = mid; else hi = mid;
        }
        ev[u] = n_events ? *reinterpret_cast<const uint2*>(J.iev + (uint64_t)s_ct[lo].ev_off + (e - s_evl[lo])) : make_uint2(0, 0);
      }
    }
    for (uint32_t it0 = tid; it0 < items; it0 += MI * PQ_NT) {
      uint32_t mv[MI];
#pragma unroll
      for (int u = 0; u < MI; u++) {
        const uint32_t it = min(it0 + u * PQ_NT, items - 1u);
        const uint32_t c = it / nw, wi = it - c * nw;
        const uint32_t o = s_ct[c + 1].ow;
        mv[u] = J.cw[(o != NONE ? (uint64_t)o : 0ull) * nw + wi].m;
      }
#pragma unroll
      for (int u = 0; u < MI; u++) {
        const uint32_t it = it0 + u * PQ_NT;
        if (it < items) s_M[it] = s_ct[it / nw + 1].ow != NONE ? mv[u] : 0u;
      }
    }
    if (ev_in_lds) {
#pragma unroll
      for (int u = 0; u < EI; u++) if (tid + u * PQ_NT < n_events) s_ev[tid + u * PQ_NT] = ev[u];
    }
    if (!FULL && tid < min(kper, nsup)) s_rr[tid] = srow;   // staged through s_rr: rewritten below once the rows are expanded
  }
  __syncthreads();
  uint32_t pre_rm = NONE, pre_r = 0;   // first pass: row-map entries travel while the rank directories are built
  const bool pre = !FULL && kper <= PQ_NT && min(kper, nsup) * span <= PQ_NT * 1u;
  if (pre) {
    const uint32_t nrows = min(kper, nsup) * span;
    if (tid < nrows) {
      const int64_t r = (int64_t)s_rr[tid / span] + (int64_t)(tid % span) - (int64_t)half;
      pre_r = (uint32_t)r;
      if (r >= 0 && r < (int64_t)Lf) pre_rm = J.rowmap2[wd.row_off + (uint32_t)r];
    }
  }
  for (uint32_t c = wave; c < HERRO_ROWS - 1; c += PQ_NT / 64) {   // rank directory of every M plane
    uint32_t carry = 0;
    for (uint32_t b = 0; b < nw; b += 64) {
      const uint32_t i = b + lane;
      const uint32_t pc = i < nw ? (uint32_t)__popc(s_M[c * nw + i]) : 0u;
      const uint32_t inc = wscan_incl(pc);
      if (i < nw) s_rk[c * nw + i] = (uint16_t)(carry + inc - pc);
      carry += wlast(inc);
    }
  }
  __syncthreads();

  const uint64_t tq_off = s_ct[0].qual_off + wd.tstart;
  const uint64_t qmax = J.read_qual_bytes ? J.read_qual_bytes - 1 : 0;
  // address of the quality byte of cell (column c, position p, insertion ordinal j); NONE64: the cell holds no base ('!')
  constexpr uint64_t NONE64 = ~0ull;
  auto cell_addr = [&](auto evlds, uint32_t c, uint32_t p, uint32_t j) -> uint64_t {
    constexpr bool EVL = decltype(evlds)::value;
    if (c == 0) return j == 0 ? min(tq_off + p, qmax) : NONE64;
    const CTab& q = s_ct[c];
    if (q.ow == NONE) return NONE64;
    const uint4* evg = J.iev + q.ev_off;
    const uint32_t el = s_evl[c];
    auto ev_x = [&](uint32_t i) -> uint32_t { return EVL ? s_ev[el + i].x : evg[i].x; };
    auto ev_y = [&](uint32_t i) -> uint32_t { return EVL ? s_ev[el + i].y : evg[i].y; };
    auto first_ge = [&](uint32_t bound) -> uint32_t {   // first event with position >= bound
      uint32_t lo = 0, hi = q.n_ev;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((ev_x(mid) & 0xffffu) < bound) lo = mid + 1; else hi = mid;
      }
      return lo;
    };
    const uint32_t* M = s_M + (size_t)(c - 1) * nw;
    const uint16_t* rk = s_rk + (size_t)(c - 1) * nw;
    auto rank = [&](uint32_t pp) -> uint32_t {   // query bases aligned to positions < pp
      if (pp >= (nw << 5)) return (uint32_t)rk[nw - 1] + (uint32_t)__popc(M[nw - 1]);
      return (uint32_t)rk[pp >> 5] + (uint32_t)__popc(M[pp >> 5] & ((1u << (pp & 31u)) - 1u));
    };
    uint32_t qi;
    if (j == 0) {
      const uint32_t uu = p - (uint32_t)q.off;
      if (uu >= q.t_total || !((M[p >> 5] >> (p & 31u)) & 1u)) return NONE64;
      const uint32_t ub = first_ge(p);          // events strictly before p: [0, ub)
      if (ub == 0) qi = rank(p);
      else {
        const uint32_t ex = ev_x(ub - 1);
        qi = ev_y(ub - 1) + (ex >> 16) + rank(p) - rank((ex & 0xffffu) + 1u);
      }
    } else {
      uint32_t i = first_ge(p + 1);             // events at positions <= p: [0, i)
      bool found = false;
      qi = 0;
      while (i > 0) {
        const uint32_t ex = ev_x(i - 1);
        if ((ex & 0xffffu) != p) break;
        if ((ex >> 16) >= j) { qi = ev_y(i - 1) + j - 1u; found = true; break; }   // the LAST insertion at p that is long enough wrote this row
        i--;
      }
      if (!found) return NONE64;
    }
    const int64_t si = (int64_t)q.sbase + (int64_t)q.sdir * (int64_t)qi;
    return min(q.qual_off + (uint64_t)max(si, (int64_t)0), qmax);
  };
  auto run = [&](auto evlds) {
    if (FULL) {
      const uint64_t total = (uint64_t)Lf * HERRO_ROWS;
      for (uint64_t idx0 = tid; idx0 < total; idx0 += 4 * PQ_NT) {
        uint32_t rm[4];
        uint64_t dst[4], src[4];
        uint32_t cc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint64_t idx = min(idx0 + (uint64_t)u * PQ_NT, total - 1);
          cc[u] = (uint32_t)(idx / Lf);
          const uint32_t r = (uint32_t)(idx - (uint64_t)cc[u] * Lf);
          rm[u] = J.rowmap2[wd.row_off + r];
          dst[u] = wd.fin_off + (uint64_t)cc[u] * wd.lub + r;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) src[u] = cell_addr(evlds, cc[u], rm[u] & 0xffffu, rm[u] >> 16);
        uint32_t qv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) qv[u] = J.read_qual[src[u] == NONE64 ? 0 : src[u]];
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (idx0 + (uint64_t)u * PQ_NT < total) J.fin_q[dst[u]] = (uint8_t)(src[u] == NONE64 ? 33u : qv[u]);
      }
    } else {
      for (uint32_t k0 = 0; k0 < nsup; k0 += kper) {
        const uint32_t nk = min(kper, nsup - k0), nrows = nk * span;
        if (k0 == 0 && pre) {
          if (tid < nrows) { s_rr[tid] = pre_r; s_rm[tid] = pre_rm; }
        } else {
          __syncthreads();
          for (uint32_t i = tid; i < nrows; i += PQ_NT) {
            const int64_t r = (int64_t)J.sup_row[wd.row_off + k0 + i / span] + (int64_t)(i % span) - (int64_t)half;
            const bool in = r >= 0 && r < (int64_t)Lf;
            s_rr[i] = (uint32_t)r;
            s_rm[i] = in ? J.rowmap2[wd.row_off + (uint32_t)r] : NONE;
          }
        }
        __syncthreads();
        // a batch of cells per thread: addresses first (LDS only), then all quality bytes together, then the stores: the 31 columns of a row
        const uint32_t total = nrows * HERRO_ROWS;
        uint8_t* __restrict__ outp = J.fin_q;
        constexpr int CB = 10;
        for (uint32_t idx0 = tid; idx0 < total; idx0 += CB * PQ_NT) {
          uint64_t dst[CB], src[CB];
          bool live[CB];
#pragma unroll
          for (int u = 0; u < CB; u++) {
            const uint32_t idx = min(idx0 + u * PQ_NT, total - 1u);
            const uint32_t slot = idx / HERRO_ROWS, c = idx - slot * HERRO_ROWS;
            dst[u] = wd.fin_off + (uint64_t)c * wd.lub + s_rr[slot];
            const uint32_t rm = s_rm[slot];
            live[u] = idx0 + u * PQ_NT < total && rm != NONE;
            src[u] = live[u] ? cell_addr(evlds, c, rm & 0xffffu, rm >> 16) : NONE64;
          }
          uint32_t qv[CB];
#pragma unroll
          for (int u = 0; u < CB; u++) qv[u] = J.read_qual[src[u] == NONE64 ? 0 : src[u]];
#pragma unroll
          for (int u = 0; u < CB; u++)
            if (live[u]) outp[dst[u]] = (uint8_t)(src[u] == NONE64 ? 33u : qv[u]);
        }
      }
    }
  };
  if (ev_in_lds) run(std::true_type{}); else run(std::false_type{});
}

// =====================================================================================================
// k_rfq — one workgroup per window: the qualities of the model's receptive fields, compact
// =====================================================================================================
// Output [(sup_off[w] + k) * 31 + column][8]: byte i = quality of row sup_row[k] - half + i of that column (span = 2 half + 1 <= 8).
// One thread per (informative row, column) slot.  A cell's quality is the read's byte at the query index of the cell's base
// (features.rs:139-152, 197-198, 225-226): for a base row at position p, the index of the first base at or behind the start of
// p's word (k_cols' directory, which also carries the word of the M plane) + the M bits of the word in front of p + the bases
// inserted between the word's start and p (events from the directory's cursor; skipped when the next word's cursor says none); for row j of an insertion behind p, the event's own query index
// + j - 1 — the LAST event at p that is long enough wrote the row (the reference's sequential writes).  Everything is a gather
// of a few words per slot with three dependent steps; nothing is staged.  (r3's version staged the 30 M planes of the window,
// built rank and event directories for all 4096 positions in LDS — 60 k cycles and 43 KB of LDS per window for ~470 slots,
// 17.7 x the output in HBM traffic; r2 scattered single bytes over the quality planes.)
constexpr int RQ_NT = 512;   // ~470 slots per window: one per thread — a slot is a chain of 5-7 dependent loads, a second one per thread doubled the kernel (r5)

// Round 5: the slot carries the cells' TOKENS as well (bytes 0..7; qualities 8..15) — on the lean path there is no token plane to read
// them from — and the rows of a slot are no longer looked up in a row map: the informative row's (position, ordinal) and the rows
// of the positions around it (row_of_pos2) say which cell each of the slot's rows is.  A base cell's code comes from the column's
// lo / hi planes (two more words per slot), an inserted base's from its event (the first 16 bases travel with it), the target's from
// the read store.
// One receptive-field record: the tokens and qualities of the rows srow - half .. around informative row (pj = position | ordinal << 16) in column c.
// have_nr: nr carries the row counts of the four positions around (k_rows); else the rows come from row_of_pos2.
template <int SP>   // rows of a record the slot fills: 5 (receptive field of half 2, the model's) or 8 — the arrays below are per row, and with 8 of each the kernel held 104 registers (round 6)
__device__ __forceinline__ void rf_slot(const JobDev& J, const CTab* __restrict__ s_ct, const WinDesc& wd, uint32_t Lf, uint32_t half, uint32_t c,
                                        uint32_t srow, uint32_t pj, uint32_t nr, bool have_nr, uint4* __restrict__ out) {
  const uint32_t nw = J.nw;
  const uint32_t span = 2 * half + 1, win_len = wd.win_len;
  const uint64_t tq_off = s_ct[0].qual_off + wd.tstart;
  const uint64_t qmax = J.read_qual_bytes ? J.read_qual_bytes - 1 : 0;
  const uint64_t pmax = J.read_n_words + 1;
  constexpr uint64_t NONE64 = ~0ull;
  const uint32_t* __restrict__ rop = J.row_of_pos2 + wd.pos_off;
  const int32_t pc = (int32_t)(pj & 0xffffu);
  const int64_t row0 = (int64_t)srow - (int64_t)half;
  uint32_t rv[SP];   // first row of positions pc - half + i
  if (have_nr && half == 2) {   // lean path: the neighbours' row counts came with the informative row (k_rows) — one round trip less, no row_of_pos2
    const uint32_t rc = srow - (pj >> 16);                      // row of position pc
    rv[2] = rc;
    rv[1] = rc - ((nr >> 6) & 63u);
    rv[0] = rv[1] - (nr & 63u);
    rv[3] = rc + ((nr >> 12) & 63u);
    rv[4] = rv[3] + ((nr >> 18) & 63u);
    if constexpr (SP > 5) { rv[5] = 0; rv[6] = 0; rv[7] = 0; }
  } else {
#pragma unroll
    for (int i = 0; i < SP; i++) rv[i] = rop[(uint32_t)min(max(pc - (int32_t)half + i, 0), (int32_t)win_len)];
  }
  uint32_t rm[SP];
#pragma unroll
  for (int d = 0; d < SP; d++) {
    const int64_t r = row0 + d;
    rm[d] = NONE;
    if ((uint32_t)d < span && r >= 0 && r < (int64_t)Lf) {
      int32_t pp = 0;
      uint32_t rb = 0;
#pragma unroll
      for (int i = 0; i < SP; i++) {
        const int32_t q = pc - (int32_t)half + i;
        if ((uint32_t)i <= 2 * half && q >= 0 && q < (int32_t)win_len && (int64_t)rv[i] <= r) { pp = q; rb = rv[i]; }
      }
      rm[d] = (uint32_t)pp | (((uint32_t)r - rb) << 16);
    }
  }
  uint64_t addr[SP];
  uint32_t tok[SP];
#pragma unroll
  for (int d = 0; d < SP; d++) { addr[d] = NONE64; tok[d] = TOK_NONE; }
  const CTab& q = s_ct[c];
  if (c == 0) {
    const uint64_t tw = s_ct[0].q_woff;
#pragma unroll
    for (int d = 0; d < SP; d++) {
      if (rm[d] == NONE) continue;
      tok[d] = TOK_GAP_F;   // the target shows '*' on an insertion row
      if ((rm[d] >> 16) == 0) {
        const uint32_t tp = wd.tstart + (rm[d] & 0xffffu);
        const uint64_t wi = min(tw + (tp >> 5), pmax);
        addr[d] = min(tq_off + (rm[d] & 0xffffu), qmax);
        tok[d] = ((J.read_p0[wi] >> (tp & 31u)) & 1u) | (((J.read_p1[wi] >> (tp & 31u)) & 1u) << 1);
      }
    }
  } else if (q.ow != NONE) {
    const uint32_t tbase = q.tokc & 0xffu, tgap = q.tokc >> 8;
    uint32_t p0 = NONE;   // position of the slot's first row inside the window
#pragma unroll
    for (int d = SP - 1; d >= 0; d--) if (rm[d] != NONE) p0 = rm[d] & 0xffffu;
    if (p0 != NONE) {
      const uint32_t w0 = min(p0 >> 5, nw - 1u), w1 = min(w0 + 1u, nw - 1u);
      const PlaneRec* __restrict__ cw = J.cw + (uint64_t)q.ow * nw;
      const uint32_t* __restrict__ cwd = J.cwd + (uint64_t)q.ow * nw;
      const PlaneRec c0 = cw[w0], c1 = cw[w1];   // planes of the slot's (at most two) words, and their directory words — ONE 8-byte read: the word behind the row's last
      uint2 dd;                                  // is the next overlap's first (or the array's spare row), and is not used
      __builtin_memcpy(&dd, cwd + w0, 8);
      const uint2 d0 = make_uint2(c0.m, dd.x), d1 = make_uint2(c1.m, w1 != w0 ? dd.y : dd.x);
      const uint32_t l0 = c0.lo, l1 = c1.lo, h0 = c0.hi, h1 = c1.hi;
      const uint32_t m0 = d0.x, m1 = d1.x;
      const uint32_t n_ev = q.n_ev;
      const uint4* __restrict__ iev = J.iev + q.ev_off;
      uint32_t qw = d0.y & 0xfffffu, e = d0.y >> 20;
      if (d0.y == 0xffffffffu) {   // indices beyond the record's fields: count (M bits and events in front of the word)
        qw = 0; e = 0;
        for (uint32_t i = 0; i < w0; i++) qw += (uint32_t)__popc(cw[i].m);
        while (e < n_ev && (iev[e].x & 0xffffu) < (w0 << 5)) { qw += iev[e].x >> 16; e++; }
      }
      // no insertion event between the two words' first positions and no row beyond: the event list is not needed
      uint32_t pl = p0;
      bool ins_row = false;
#pragma unroll
      for (int d = 0; d < SP; d++) if (rm[d] != NONE) { pl = rm[d] & 0xffffu; ins_row = ins_row || (rm[d] >> 16) != 0; }
      const bool quiet = d0.y != 0xffffffffu && d1.y != 0xffffffffu && w1 != w0 && (d1.y >> 20) == e && (pl >> 5) == w0 && !ins_row;
      // the next four events travel together (one round trip for nearly every slot; a fifth is fetched when the walk gets there)
      const uint32_t e_first = e;
      const uint4 none4 = make_uint4(0xffffffffu, 0, 0, 0);
      // (loads unconditional with clamped indices, values selected afterwards: a select between the array and a constant compiles to a
      // flat load from a stack copy of the constant)
      const uint32_t e_last = n_ev ? n_ev - 1u : 0u;
      uint4 ep0 = iev[min(e, e_last)], ep1 = iev[min(e + 1u, e_last)], ep2 = iev[min(e + 2u, e_last)], ep3 = iev[min(e + 3u, e_last)];
      if (quiet || e >= n_ev) ep0.x = 0xffffffffu;
      if (quiet || e + 1u >= n_ev) ep1.x = 0xffffffffu;
      if (quiet || e + 2u >= n_ev) ep2.x = 0xffffffffu;
      if (quiet || e + 3u >= n_ev) ep3.x = 0xffffffffu;
      auto ev_at = [&](uint32_t i) -> uint4 {   // event i (position 0xffff: none left)
        if (i >= n_ev) return none4;
        const uint32_t o = i - e_first;
        if (o >= 4u) return iev[i];
        const uint4 a = (o & 1u) ? ep1 : ep0, b = (o & 1u) ? ep3 : ep2;
        return (o & 2u) ? b : a;
      };
      uint4 ecur = ep0;
      uint32_t cum = 0;   // bases inserted behind positions [32 w0, p)
#pragma unroll
      for (int d = 0; d < SP; d++) {
        if (rm[d] == NONE) continue;
        const uint32_t p = rm[d] & 0xffffu, j = rm[d] >> 16;
        while ((ecur.x & 0xffffu) < p) {   // events in front of p: [.., e)
          cum += ecur.x >> 16;
          e++;
          ecur = ev_at(e);
        }
        const bool second = (p >> 5) != w0;   // the span is at most 8 rows: two words at most
        const uint32_t mw = second ? m1 : m0;
        const uint32_t below = (uint32_t)__popc(mw & ((1u << (p & 31u)) - 1u)) + (second ? (uint32_t)__popc(m0) : 0u);
        const uint32_t mbit = (mw >> (p & 31u)) & 1u;
        const bool inr = p - (uint32_t)q.off < q.t_total;
        uint32_t qi = NONE;
        tok[d] = inr ? tgap : (uint32_t)TOK_NONE;   // no base here: a gap where the overlap covers the position, '.' outside
        if (j == 0) {
          if (inr && mbit) {
            qi = qw + below + cum;
            tok[d] = tbase + ((((second ? l1 : l0) >> (p & 31u)) & 1u) | ((((second ? h1 : h0) >> (p & 31u)) & 1u) << 1));
          }
        } else if ((ecur.x & 0xffffu) == p) {
          uint4 x = ecur;
          uint32_t code = 0;
          for (uint32_t i = e;;) {
            if ((x.x >> 16) >= j) {   // the LAST insertion at p that is long enough wrote this row
              qi = x.y + j - 1u;
              code = j <= 16u ? (((x.z >> (j - 1u)) & 1u) | (((x.z >> (15u + j)) & 1u) << 1)) : 4u;
            }
            if (++i >= n_ev) break;
            x = ev_at(i);
            if ((x.x & 0xffffu) != p) break;
          }
          if (qi != NONE) {
            if (code == 4u) {   // beyond the 16 bases the event carries: from the read store
              const int32_t si = q.sbase + q.sdir * (int32_t)qi;
              const uint64_t wi = min(q.q_woff + ((uint32_t)si >> 5), pmax);
              code = ((J.read_p0[wi] >> ((uint32_t)si & 31u)) & 1u) | (((J.read_p1[wi] >> ((uint32_t)si & 31u)) & 1u) << 1);
              if (q.sdir < 0) code ^= 3u;
            }
            tok[d] = tbase + code;
          }
        }
        if (qi != NONE) {
          const int64_t si = (int64_t)q.sbase + (int64_t)q.sdir * (int64_t)qi;
          addr[d] = min(q.qual_off + (uint64_t)max(si, (int64_t)0), qmax);
        }
      }
    }
  }
  uint32_t qv[SP];
#pragma unroll
  for (int d = 0; d < SP; d++) qv[d] = J.read_qual[addr[d] == NONE64 ? 0 : addr[d]];
  uint32_t lo = 0, hi = 0, tl = 0, th = 0;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    lo |= (addr[d] == NONE64 ? 33u : qv[d]) << (8 * d);
    tl |= tok[d] << (8 * d);
    if (d + 4 < SP) {   // (compile time)
      hi |= (addr[(d + 4) % SP] == NONE64 ? 33u : qv[(d + 4) % SP]) << (8 * d);
      th |= tok[(d + 4) % SP] << (8 * d);
    } else {
      hi |= 33u << (8 * d);
      th |= (uint32_t)TOK_NONE << (8 * d);
    }
  }
  *out = make_uint4(tl, th, lo, hi);
}

template <int SP>
__global__ __launch_bounds__(RQ_NT) void k_rfq(JobDev J, uint32_t half, const uint64_t* __restrict__ sup_off, uint8_t* __restrict__ rf, uint64_t cap, uint32_t have_nr) {
  __shared__ __attribute__((aligned(16))) CTab s_ct[32];
  const uint32_t w = gridDim.x - 1u - blockIdx.x, tid = threadIdx.x;   // back to front: k_rows has just walked the windows front to back — its last windows' plane records are the cached ones
  PROF_BEGIN(J);
  const uint32_t nsup = J.win_nsup[w], Lf = J.win_Lf[w];
  const bool left_only = (have_nr & 2u) != 0u;   // only the windows k_rows reserved slots for and left (RF_LEFT), at those slots
  have_nr &= 1u;
  uint64_t slot0;
  if (left_only) {
    const uint32_t rb = J.win_rfbase[w];
    if (!nsup || rb == NONE || !(rb & RF_LEFT)) return;
    slot0 = rb & ~RF_LEFT;
  } else {
    if (!nsup || sup_off[w] + nsup > cap) return;   // (a launch in front of the host's count of the informative rows: the buffer was sized by an estimate)
    slot0 = sup_off[w];
  }
  const WinDesc wd = J.win[w];
  if (tid < 32) s_ct[tid] = J.ctab[(uint64_t)w * 32 + tid];
  __syncthreads();
  PROF_MARK(J, 5, 0);
  const uint32_t nslots = nsup * HERRO_ROWS;
  const uint64_t out0 = slot0 * HERRO_ROWS;
  for (uint32_t sl = tid; sl < nslots; sl += RQ_NT) {
    const uint32_t k = sl / HERRO_ROWS, c = sl - k * HERRO_ROWS;
    const uint32_t pj = J.sup_pi[wd.row_off + k], srow = J.sup_row[wd.row_off + k];
    const uint32_t nr = have_nr ? J.sup_nr[wd.row_off + k] : 0u;
    rf_slot<SP>(J, s_ct, wd, Lf, half, c, srow, pj, nr, have_nr != 0u, reinterpret_cast<uint4*>(rf + (out0 + sl) * 16));
    PROF_MARK(J, 5, 1);
  }
}

// =====================================================================================================
// k_consensus — one workgroup per window: corrected bases on the device (consensus.rs:86-227)
// =====================================================================================================
// Per final row: informative -> argmax of the 5 base logits (the LAST maximum wins, NaN is greatest —
// max_by_key(OrderedFloat), consensus.rs:136-141); otherwise the majority vote with the target tie-break
// (consensus.rs:178-200) that k_tokens already derived from its symbol counts.  '*' is dropped.  The window's
// bases are compacted in row order; the host only concatenates windows and splits reads at windows
// with < 2 alignments (consensus.rs:90-111).
constexpr int PC_NT = 256;
__global__ __launch_bounds__(PC_NT) void k_consensus(JobDev J, const uint64_t* sup_off, const float* base_logits) {
  __shared__ uint32_t s_wave[PC_NT / 64];
  const uint32_t w = blockIdx.x;
  const WinDesc wd = J.win[w];
  const uint32_t Lf = J.win_Lf[w], n_kept = J.win_nkept[w];
  const uint32_t n_alns = n_kept < 30u ? n_kept : 30u;
  uint8_t* seq = J.cons_seq + wd.row_off;
  if (n_alns < 2) {  // not corrected: the read is split here
    if (threadIdx.x == 0) J.cons_len[w] = 0;
    return;
  }
  const float* lg = base_logits + sup_off[w] * 5;
  uint8_t* tmp = J.cons_tmp + wd.row_off;  // per-row majority vote, written by k_tokens
  // informative rows: the model decides
  const uint32_t nsup = J.win_nsup[w];
  for (uint32_t k = threadIdx.x; k < nsup; k += PC_NT) {
    const float* l5 = lg + (uint64_t)k * 5;
    uint32_t arg = 0;
    float mx = l5[0];
#pragma unroll
    for (uint32_t c = 1; c < 5; c++) {
      const float v = l5[c];
      const bool ge = (v != v) ? true : ((mx != mx) ? false : v >= mx);
      if (ge) { arg = c; mx = v; }
    }
    tmp[J.sup_row[wd.row_off + k]] = (uint8_t)arg;
  }
  __syncthreads();
  // drop '*' and compact in row order
  const uint32_t ch = (Lf + PC_NT - 1) / PC_NT;
  const uint32_t a = min(threadIdx.x * ch, Lf), b = min(a + ch, Lf);
  uint32_t nout = 0;
  for (uint32_t r = a; r < b; r++) nout += tmp[r] != 4u ? 1u : 0u;
  uint32_t total;
  uint32_t o = blk_scan<PC_NT>(nout, &total, s_wave);
  // the thread's bases leave in aligned 4-byte stores where they can (single bytes at the two ragged ends): a byte store per
  // base was a write transaction per base (PMC: 170 MB written per 4096 windows for 19 MB of bases)
  uint32_t acc = 0, nacc = 0;
  for (uint32_t r = a; r < b; r++) {
    const uint32_t base = tmp[r];
    if (base == 4u) continue;
    const uint32_t ch = ((0x54474341u >> (8u * base)) & 0xffu);
    if ((o & 3u) != 0 && nacc == 0) { seq[o++] = (uint8_t)ch; continue; }   // head: up to the next 4-byte boundary
    acc |= ch << (8 * nacc);
    if (++nacc == 4) { *reinterpret_cast<uint32_t*>(seq + o) = acc; o += 4; acc = 0; nacc = 0; }
  }
  for (uint32_t k = 0; k < nacc; k++) seq[o + k] = (uint8_t)(acc >> (8 * k));   // tail
  if (threadIdx.x == 0) J.cons_len[w] = total;
}

// The same for the lean path (k_rows): the votes arrive in position space — three bit planes for the base rows, a byte per insertion
// row indexed by its ordinal — and so does everything here: a lane owns 32 target positions, counts what they and the insertion rows
// behind them contribute, and writes its stretch of the corrected sequence; the model's calls are patched into the planes / bytes
// first.  No per-row array is read or written.
constexpr uint32_t CP_ICAP = 4096;   // insertion-row votes staged in LDS (a window with more works on the global bytes)
constexpr uint32_t CP_OCAP = 6144;   // corrected bases staged in LDS for coalesced stores (more: byte stores)
// SPW threads per word of 32 positions: each takes a segment of 32 / SPW positions through the (serial, dependent) emission loop — with one lane per
// word that loop was 25 k of the kernel's 40 k cycles (profiles/r5m_phase_cycles.txt)
template <int NT, int SPW>
__global__ __launch_bounds__(NT) void k_consensus_p(JobDev J, const uint64_t* sup_off, const float* base_logits) {
  extern __shared__ __attribute__((aligned(16))) uint32_t cp_smem[];
  uint32_t* s_rop = cp_smem;                                        // [win_len + 1], padded (RI)
  __shared__ uint32_t s_v[3][HERRO_MAX_WINDOW / 32];
  __shared__ uint32_t s_im[HERRO_MAX_WINDOW / 32];                   // positions with insertion rows behind them (k_rows)
  __shared__ __attribute__((aligned(16))) uint8_t s_iv[CP_ICAP];
  __shared__ __attribute__((aligned(16))) uint8_t s_out[CP_OCAP];
  __shared__ uint32_t s_wave[NT / 64];
  const uint32_t w = blockIdx.x, tid = threadIdx.x, nw = J.nw;
  PROF_BEGIN(J);
  const WinDesc wd = J.win[w];
  const uint32_t Lf = J.win_Lf[w], n_kept = J.win_nkept[w], win_len = wd.win_len;
  const uint32_t n_alns = n_kept < 30u ? n_kept : 30u;
  uint8_t* __restrict__ seq = J.cons_seq + wd.row_off;
  if (n_alns < 2) {  // not corrected: the read is split here
    if (tid == 0) J.cons_len[w] = 0;
    return;
  }
  const uint32_t n_irows = Lf - win_len;
  const bool iv_lds = n_irows <= CP_ICAP;
  uint8_t* __restrict__ giv = J.cons_tmp + wd.row_off;
  {
    // all loads of the prologue in flight together (rows of the positions, vote planes, insertion-row votes)
    constexpr int RL = SPW == 4 ? 9 : 17;   // NT * RL > positions: 512 threads x 9 cover windows of 4096 (SPW 4), x 17 of 8192 (SPW 2)
    uint32_t rv[RL];
    const uint32_t* __restrict__ rop = J.row_of_pos2 + wd.pos_off;
#pragma unroll
    for (int u = 0; u < RL; u++) rv[u] = rop[min(tid + (uint32_t)u * NT, win_len)];
    uint32_t vv[4] = {0, 0, 0, 0};
    if (tid < nw) {
      const uint32_t* __restrict__ vp = J.vpl + (uint64_t)w * 4 * nw + tid;
      vv[0] = vp[0]; vv[1] = vp[nw]; vv[2] = vp[2 * nw]; vv[3] = vp[3 * nw];
    }
    constexpr int IL = CP_ICAP / 4 / NT;
    uint32_t iw[IL];
    const uint32_t n_iw = iv_lds ? (n_irows + 3u) / 4u : 0u;
#pragma unroll
    for (int u = 0; u < IL; u++) iw[u] = n_iw ? reinterpret_cast<const uint32_t*>(giv)[min(tid + (uint32_t)u * NT, n_iw - 1u)] : 0u;
#pragma unroll
    for (int u = 0; u < RL; u++) {
      const uint32_t p = tid + (uint32_t)u * NT;
      if (p <= win_len) s_rop[RI(p)] = rv[u];
    }
    if (tid < nw) { s_v[0][tid] = vv[0]; s_v[1][tid] = vv[1]; s_v[2][tid] = vv[2]; s_im[tid] = vv[3]; }
#pragma unroll
    for (int u = 0; u < IL; u++) {
      const uint32_t i = tid + (uint32_t)u * NT;
      if (i < n_iw) reinterpret_cast<uint32_t*>(s_iv)[i] = iw[u];
    }
  }
  __syncthreads();
  PROF_MARK(J, 7, 0);
  // informative rows: the model decides — argmax of the 5 base logits, the LAST maximum wins, NaN is greatest (consensus.rs:136-141)
  const uint32_t nsup = J.win_nsup[w];
  const float* lg = base_logits + sup_off[w] * 5;
  for (uint32_t k = tid; k < nsup; k += NT) {
    const float* l5 = lg + (uint64_t)k * 5;
    uint32_t arg = 0;
    float mx = l5[0];
#pragma unroll
    for (uint32_t c = 1; c < 5; c++) {
      const float v = l5[c];
      const bool ge = (v != v) ? true : ((mx != mx) ? false : v >= mx);
      if (ge) { arg = c; mx = v; }
    }
    const uint32_t pj = J.sup_pi[wd.row_off + k], p = pj & 0xffffu, j = pj >> 16;
    if (j == 0) {
      const uint32_t bit = 1u << (p & 31u);
#pragma unroll
      for (uint32_t b = 0; b < 3; b++) {
        if ((arg >> b) & 1u) atomicOr(&s_v[b][p >> 5], bit); else atomicAnd(&s_v[b][p >> 5], ~bit);
      }
    } else {
      const uint32_t ir = s_rop[RI(p)] - p + j - 1u;
      if (iv_lds) s_iv[ir] = (uint8_t)arg; else giv[ir] = (uint8_t)arg;
    }
  }
  __syncthreads();
  PROF_MARK(J, 7, 1);
  // what every lane's 32 positions contribute: base rows that are not '*', insertion rows behind them that are not '*'
  constexpr uint32_t SEG = 32u / SPW;                       // positions per thread
  const bool active = tid < nw * SPW;
  const uint32_t widx = min(tid / SPW, nw - 1u), sub = tid % SPW, P = widx << 5;   // P: first position of the thread's WORD (bit k of a mask = position P + k)
  const uint32_t segm = (SEG == 32u ? 0xffffffffu : ((1u << SEG) - 1u)) << (sub * SEG);
  const uint32_t vm = active ? mask_range(0, (int32_t)win_len - (int32_t)P) & segm : 0u;
  const uint32_t v0 = s_v[0][widx], v1 = s_v[1][widx], v2 = s_v[2][widx];
  const uint32_t keep = vm & ~(v2 & ~v1 & ~v0);
  uint32_t ia = 0, ib = 0;   // insertion rows [ia, ib) lie behind this thread's positions
  if (vm) {
    const uint32_t ps = P + sub * SEG, pe = min(ps + SEG, win_len);
    ia = s_rop[RI(ps)] - ps;
    ib = s_rop[RI(pe)] - pe;
  }
  uint32_t cnt = (uint32_t)__popc(keep);
  if (iv_lds) { for (uint32_t ir = ia; ir < ib; ir++) cnt += ((uint32_t)s_iv[ir] & 7u) != 4u ? 1u : 0u; }
  else { for (uint32_t ir = ia; ir < ib; ir++) cnt += ((uint32_t)giv[ir] & 7u) != 4u ? 1u : 0u; }
  uint32_t total;
  uint32_t o = blk_scan<NT>(cnt, &total, s_wave);
  PROF_MARK(J, 7, 2);
  const bool out_lds = total <= CP_OCAP;
  // (the letters come out of a register constant: `"ACGT"[code]` is a load from constant memory per base — a dependent global round trip in every
  // iteration of this loop; and the destination / the insertion votes must be LDS or global at COMPILE time: a pointer chosen at run time makes every
  // access a flat one, ~800 cycles per base: 35 k, then 28 k of the kernel's 50 k cycles, profiles/r5k_, r5l_phase_cycles.txt)
  auto emit = [&](auto fast) {
    constexpr bool FAST = decltype(fast)::value;   // corrected bases staged in LDS and the insertion votes read from LDS (nearly every window)
    uint32_t ir = ia;
    const uint32_t im = s_im[widx];
    for (uint32_t m = (keep | im) & vm; m; m &= m - 1u) {   // positions that contribute a base or have insertion rows behind them, in order
      const uint32_t k = (uint32_t)__ffs((int)m) - 1u, p = P + k;
      if ((keep >> k) & 1u) {
        const uint32_t code = ((v0 >> k) & 1u) | (((v1 >> k) & 1u) << 1);   // not '*': A C G T
        const uint8_t ch = (uint8_t)(0x54474341u >> (8u * code));
        if (FAST) s_out[o++] = ch; else if (out_lds) s_out[o++] = ch; else seq[o++] = ch;
      }
      if ((im >> k) & 1u) {   // the rows of a position are consulted only where there are insertion rows (~4 positions of a lane's 32)
        const uint32_t ie = s_rop[RI(p + 1)] - p - 1u;   // insertion rows in front of position p + 1
        for (; ir < ie; ir++) {
          uint32_t vt;
          if (FAST) vt = (uint32_t)s_iv[ir] & 7u; else { if (iv_lds) vt = (uint32_t)s_iv[ir] & 7u; else vt = (uint32_t)giv[ir] & 7u; }
          if (vt != 4u) {
            const uint8_t ch = (uint8_t)(0x54474341u >> (8u * (vt & 3u)));
            if (FAST) s_out[o++] = ch; else if (out_lds) s_out[o++] = ch; else seq[o++] = ch;
          }
        }
      }
    }
  };
  if (cnt) {
    if (out_lds && iv_lds) emit(std::true_type{}); else emit(std::false_type{});
  }
  PROF_MARK(J, 7, 3);
  if (out_lds) {
    __syncthreads();
    uint32_t* __restrict__ sq = reinterpret_cast<uint32_t*>(seq);   // row_off is a multiple of 16
    for (uint32_t i = tid; i * 4u < total; i += NT) sq[i] = reinterpret_cast<const uint32_t*>(s_out)[i];
  }
  if (tid == 0) J.cons_len[w] = total;
  PROF_MARK(J, 7, 4);
}

}  // namespace

void launch_consensus(const JobDev& J, const uint64_t* sup_off, const float* base_logits, bool lean, hipStream_t st, KernelTimer* tm) {
  if (!J.n_win) return;
  KT_BEGIN(tm, "consensus", st);
  if (!lean) hipLaunchKernelGGL(k_consensus, dim3(J.n_win), dim3(PC_NT), 0, st, J, sup_off, base_logits);
  else if (J.nw <= 128) hipLaunchKernelGGL((k_consensus_p<512, 4>), dim3(J.n_win), dim3(512), rows_lds(J.window_size), st, J, sup_off, base_logits);   // four threads per word of 32 positions
  else hipLaunchKernelGGL((k_consensus_p<512, 2>), dim3(J.n_win), dim3(512), rows_lds(J.window_size), st, J, sup_off, base_logits);
  KT_END(tm, st);
}

// sup_off[w] = informative rows of the windows in front of w (n_win + 1 entries): the job-level slot of a window's first row
__global__ __launch_bounds__(256) void k_supoff(JobDev J, uint64_t* __restrict__ sup_off) {
  __shared__ uint32_t s_wave[4];
  const uint32_t n = J.n_win, tid = threadIdx.x, ch = (n + 255) / 256;
  const uint32_t w0 = min(tid * ch, n), w1 = min(w0 + ch, n);
  uint32_t local = 0;
  for (uint32_t w = w0; w < w1; w++) local += J.win_nsup[w];
  uint32_t tot;
  uint64_t run = blk_scan<256>(local, &tot, s_wave);
  for (uint32_t w = w0; w < w1; w++) { sup_off[w] = run; run += J.win_nsup[w]; }
  if (tid == 0) sup_off[n] = tot;
}
void launch_supoff(const JobDev& J, uint64_t* sup_off, hipStream_t st) {
  if (J.n_win) hipLaunchKernelGGL(k_supoff, dim3(1), dim3(256), 0, st, J, sup_off);
}

void launch_rf_quals(const JobDev& J, uint32_t half, const uint64_t* sup_off, uint8_t* rf, uint64_t cap, bool lean, hipStream_t st, KernelTimer* tm, bool left_only) {
  if (!J.n_win) return;
  KT_BEGIN(tm, "rf_quals", st);
  if (rf && 2 * half + 1 <= 8) {
    if (half <= 2u) hipLaunchKernelGGL(k_rfq<5>, dim3(J.n_win), dim3(RQ_NT), 0, st, J, half, sup_off, rf, cap, (lean ? 1u : 0u) | (left_only ? 2u : 0u));
    else hipLaunchKernelGGL(k_rfq<8>, dim3(J.n_win), dim3(RQ_NT), 0, st, J, half, sup_off, rf, cap, (lean ? 1u : 0u) | (left_only ? 2u : 0u));
    KT_END(tm, st);
    return;
  }
  // wider receptive fields: the qualities go into the quality planes (the caller has made sure the row map exists)
  pileup_opt_in_lds(reinterpret_cast<const void*>(k_quals<false>), 96 * 1024);   // windows of 8192: 62 KB dynamic + 10 KB static
  hipLaunchKernelGGL(k_quals<false>, dim3(J.n_win), dim3(PQ_NT), quals_lds(J.nw), st, J, half);
  KT_END(tm, st);
}

void launch_full_quals(const JobDev& J, hipStream_t st) {
  if (!J.n_win) return;
  pileup_opt_in_lds(reinterpret_cast<const void*>(k_quals<true>), 96 * 1024);
  hipLaunchKernelGGL(k_quals<true>, dim3(J.n_win), dim3(PQ_NT), quals_lds(J.nw), st, J, 0u);
}

template <int NB>
static void launch_win(const JobDev& J, hipStream_t st) {
  hipLaunchKernelGGL(k_win<NB>, dim3(J.n_win), dim3(J.nw <= 128 ? 128 : 256), 0, st, J);
}

// HERRO_TRACE=1: name every pileup launch on stderr and wait for it (which kernel faulted)
static void trace_point(const char* name, hipStream_t st) {
  static const bool on = [] { const char* e = getenv("HERRO_TRACE"); return e && atoi(e); }();
  if (!on) return;
  const hipError_t e = hipStreamSynchronize(st);
  fprintf(stderr, "TRACE %s done: %s\n", name, hipGetErrorString(e));
}

void launch_full_tokens(const JobDev& J, hipStream_t st, KernelTimer* tm) {
  if (!J.n_win || !J.n_tiles) return;
  KT_BEGIN(tm, "tokens", st);
  // the lean pass listed the inserted-base runs per window; k_tokens wants them per tile: the layout once more, in that mode (same inputs — the
  // tallies of k_win are still there —, same selection, same rows; only the run lists change shape)
  hipLaunchKernelGGL(k_layout<true>, dim3(J.n_win), dim3(LY_NT), layout_lds(J.window_size), st, J);
  hipLaunchKernelGGL(k_tokens, dim3(J.n_tiles), dim3(TK_NT), 0, st, J, 0u);
  KT_END(tm, st);
  trace_point("tokens (planes on request)", st);
}

void launch_featurize(const JobDev& J, hipStream_t st, KernelTimer* tm, bool lean) {
  if (!J.n_win) return;
  if (J.n_ow) {
    KT_BEGIN(tm, "cols", st);
    if (J.nw <= 128) hipLaunchKernelGGL((k_cols<3, 2>), dim3((J.n_ow + CA_NW - 1) / CA_NW), dim3(CA_NT), (size_t)CA_NW * cols_lds_words(J.nw) * 4, st, J);   // cols_qcap(128) + 2 = 170 <= 192
    else hipLaunchKernelGGL((k_cols<5, 4>), dim3((J.n_ow + CA_NW - 1) / CA_NW), dim3(CA_NT), (size_t)CA_NW * cols_lds_words(J.nw) * 4, st, J);
    KT_END(tm, st);
  trace_point("cols", st);
  }
  KT_BEGIN(tm, "win", st);
  if (J.n_cls) (void)hipMemsetAsync(J.nd, 0, (size_t)J.n_cls * 8, st);   // match / mismatch tallies start from zero
  {
    // counter width: enough bits for the largest threshold floor(0.1 * max(31, columns))
    const uint32_t tmax = (uint32_t)((double)(J.max_cols > 31 ? J.max_cols : 31) * 0.1);
    if (tmax < 4) launch_win<2>(J, st);
    else if (tmax < 8) launch_win<3>(J, st);
    else if (tmax < 16) launch_win<4>(J, st);
    else if (tmax < 64) launch_win<6>(J, st);
    else if (tmax < 512) launch_win<9>(J, st);
    else launch_win<14>(J, st);
  }
  KT_END(tm, st);
  trace_point("win", st);
  KT_BEGIN(tm, "layout", st);
  if (lean) hipLaunchKernelGGL(k_layout<false>, dim3(J.n_win), dim3(LY_NT), layout_lds(J.window_size), st, J);
  else hipLaunchKernelGGL(k_layout<true>, dim3(J.n_win), dim3(LY_NT), layout_lds(J.window_size), st, J);
  KT_END(tm, st);
  trace_point("layout", st);
  if (lean) {
    KT_BEGIN(tm, "rows", st);
    if (J.nw <= 128) {
      pileup_opt_in_lds(reinterpret_cast<const void*>(k_rows<128>), 96 * 1024);
      hipLaunchKernelGGL(k_rows<128>, dim3(J.n_win), dim3(256), rows_lds(J.window_size) + (size_t)15 * 128 * 4, st, J);
    } else {
      pileup_opt_in_lds(reinterpret_cast<const void*>(k_rows<256>), 96 * 1024);
      hipLaunchKernelGGL(k_rows<256>, dim3(J.n_win), dim3(512), rows_lds(J.window_size) + (size_t)15 * 256 * 4, st, J);
    }
    KT_END(tm, st);
    trace_point("rows", st);
    return;
  }
  KT_BEGIN(tm, "tokens", st);
  if (J.n_tiles) hipLaunchKernelGGL(k_tokens, dim3(J.n_tiles), dim3(TK_NT), 0, st, J, 1u);
  KT_END(tm, st);
  trace_point("tokens", st);
  KT_BEGIN(tm, "supgather", st);
  hipLaunchKernelGGL(k_supgather, dim3(J.n_win), dim3(64), 0, st, J);
  KT_END(tm, st);
  trace_point("supgather", st);
}

}  // namespace herro
