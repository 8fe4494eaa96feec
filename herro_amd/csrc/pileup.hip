// pileup.hip — pileup feature generation on gfx950, bit-plane formulation (reference src/features.rs:326-583).
//
// Integer / byte work, HBM- and LDS-bound; no MFMA on purpose.  Three launches per job, one workgroup per WINDOW:
//
//   k_pass1   per overlap (one wave each): CIGAR ops -> prefix sums -> the overlap's column as three BIT PLANES over the
//             window's target positions (M: a query base is aligned here; lo / hi: its 2-bit code), written op by op with
//             LDS atomics from the bit-plane copy of the read store; insertion events (position, length, first bases).
//             From the planes: accuracy (features.rs:585-679) = popcounts of M & (query ^ target); long-indel filter
//             (features.rs:315-324).  Per window: stable accuracy rank (features.rs:386-409), informative positions of
//             pass 1 in bit-sliced counters (features.rs:681-722), match / mismatch tallies per query name
//             (features.rs:461-500).  The reference's [L, 1+n] pass-1 matrix never exists.
//   k_final   haplotype score, stable re-rank, top-30 (features.rs:502-525); row layout = prefix sum over the selected
//             overlaps' max insertion (features.rs:44-95, 531-556); the [31][L'] token planes (features.rs:110-266) in
//             chunks of <= 2048 rows: 16 rows x 1 column per step, bits pulled from the planes, inserted bases from a small
//             LDS tile filled by the insertion events, one 16-byte store per step; symbol counts per row in registers ->
//             informative rows (features.rs:558) and the decoder's majority vote (consensus.rs:178-200); ordered list of
//             informative positions.
//   k_quals   quality bytes (features.rs:139-152,197-198,225-226): the receptive fields of informative rows (what the model
//             reads), or the complete planes on request.  Query index of a cell = rank in the M plane + insertion events.
//
// What the formulation buys: the work of a column is proportional to its OPS (~165), not to its 4096 positions; a cell
// of the final matrix costs ~15 ALU operations and no search; HBM traffic is the op arrays and query bit planes in,
// the token planes out, plus 1.5 KB of planes per kept overlap between the first two kernels.
// Generality: any number of overlaps per window (columns in groups), insertions anywhere the reference accepts them
// (leading insertion of an alignment that starts inside the window, consecutive insertion ops), windows of 16..8192.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <set>

#include "job_dev.h"
#include "pileup_core.h"

namespace herro {

namespace {

constexpr int PA_NT = 512, PA_NW = PA_NT / 64;   // k_pass1
constexpr int PB_NT = 512;                       // k_final
constexpr int PQ_NT = 256;                       // k_quals
constexpr uint32_t NONE = 0xffffffffu;
constexpr uint32_t ROWCAP = 2048;   // rows of the final matrix per chunk (k_final)
constexpr uint32_t INSCAP = 448;    // insertion rows per chunk whose tokens fit the LDS tile
constexpr uint32_t SCAP = 128;      // overlaps per window whose scores are cached in LDS
constexpr uint32_t PCAP = 8;        // plane words an op-lane writes itself; longer M runs are written by the whole wave
constexpr uint32_t QEVCAP = 1024;   // insertion events staged in LDS by k_quals

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

__device__ __forceinline__ uint64_t wscan64(uint64_t v, uint64_t* total) {  // exclusive, within the wave
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
__device__ __forceinline__ uint64_t wsum64(uint64_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// bits [lo, hi) of a 32-bit word, clipped
__device__ __forceinline__ uint32_t mask_range(int32_t lo, int32_t hi) {
  lo = max(lo, 0);
  hi = min(hi, 32);
  if (lo >= hi) return 0u;
  const uint32_t m = hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u);
  return m & ~((1u << lo) - 1u);
}

// 32 consecutive plane bits starting at bit index s of a staged plane of npw words (bits outside read 0)
__device__ __forceinline__ uint32_t lds_bits(const uint32_t* pl, uint32_t npw, int32_t s) {
  if (s < 0) return s <= -32 ? 0u : (pl[0] << (uint32_t)(-s));
  const uint32_t w = (uint32_t)s >> 5;
  const uint32_t a = w < npw ? pl[w] : 0u, b = w + 1 < npw ? pl[w + 1] : 0u;
  return __funnelshift_r(a, b, (uint32_t)s & 31u);
}
// the same from the read store's plane array (word index clamped into the array: wmax = its last word)
__device__ __forceinline__ uint32_t glb_bits(const uint32_t* __restrict__ pl, uint64_t woff, uint64_t wmax, int32_t s) {
  if (s < 0) return s <= -32 ? 0u : (pl[min(woff, wmax)] << (uint32_t)(-s));
  const uint64_t w = woff + ((uint32_t)s >> 5);
  return __funnelshift_r(pl[min(w, wmax)], pl[min(w + 1, wmax)], (uint32_t)s & 31u);
}

// bits 0..15 of x -> even bit positions 0, 2, .. 30
__device__ __forceinline__ uint32_t spread16(uint32_t x) {
  x &= 0xffffu;
  x = (x | (x << 8)) & 0x00ff00ffu;
  x = (x | (x << 4)) & 0x0f0f0f0fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
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
void pileup_opt_in_lds(const void* fn) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(mu);
  if (done.insert({fn, dev}).second) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// =====================================================================================================
// k_pass1
// =====================================================================================================
struct ACol { int32_t off; uint32_t t_total, keep, pad; };

__host__ __device__ inline uint32_t pass1_qcap(uint32_t nw) { return nw + 40u < 320u ? nw + 40u : 320u; }   // staged query plane words per wave
__host__ __device__ inline uint32_t pass1_group(uint32_t nw) { return nw <= 128u ? 32u : (nw <= 256u ? 16u : 8u); }  // columns resident in LDS
__host__ __device__ inline size_t pass1_lds(uint32_t nw) {
  const uint32_t G = pass1_group(nw);
  return ((size_t)(3 * G + 5) * nw + 16) * 4 + (size_t)G * sizeof(ACol) + (size_t)PA_NW * 2 * pass1_qcap(nw) * 4;
}

template <int NB>
__global__ __launch_bounds__(PA_NT) void k_pass1(JobDev J) {
  extern __shared__ __attribute__((aligned(16))) uint32_t pa_smem[];
  const uint32_t nw = J.nw, G = pass1_group(nw), qcap = pass1_qcap(nw);
  uint32_t* s_pl = pa_smem;                              // [G][3][nw] column planes M, lo, hi
  uint32_t* s_tp = s_pl + (size_t)G * 3 * nw;            // [2][nw]    target code planes
  uint32_t* s_tv = s_tp + 2 * nw;                        // [nw]       positions inside the window
  uint32_t* s_sup = s_tv + nw;                           // [nw]       informative positions (pass 1)
  uint32_t* s_list = s_sup + nw;                         // [nw]       words with informative positions
  uint32_t* s_misc = s_list + nw;                        // [16]       0: kept overlaps, 1: entries of s_list
  ACol* s_col = reinterpret_cast<ACol*>(s_misc + 16);    // [G]
  uint32_t* s_wq = reinterpret_cast<uint32_t*>(s_col + G);  // [PA_NW][2][qcap] query planes of the overlap a wave works on

  const uint32_t w = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WinDesc wd = J.win[w];
  const uint32_t n = wd.ow_cnt;
  const uint64_t pmax = J.read_n_words + 1;   // last word of the plane arrays
  if (tid < nw) {
    const uint64_t t_woff = J.read_word_off[wd.rid];
    const int32_t P = (int32_t)(tid << 5);
    const uint32_t vm = mask_range(0, (int32_t)wd.win_len - P);
    s_tv[tid] = vm;
    s_tp[tid] = glb_bits(J.read_p0, t_woff, pmax, (int32_t)wd.tstart + P) & vm;
    s_tp[nw + tid] = glb_bits(J.read_p1, t_woff, pmax, (int32_t)wd.tstart + P) & vm;
  }
  if (tid < 16) s_misc[tid] = 0;
  __syncthreads();

  SlicedCounters<NB> cnt;
  cnt.clear();
  if (tid < nw) cnt.add(ColPlanes{s_tv[tid], s_tp[tid], s_tp[nw + tid], 0u});  // the target column: always a base on these rows

  for (uint32_t g0 = 0; g0 < n; g0 += G) {
    const uint32_t ng = min(G, n - g0);
    // ---- phase 1: one wave per overlap
    for (uint32_t ci = wave; ci < ng; ci += PA_NW) {
      const uint32_t o = wd.ow_begin + g0 + ci;
      const OwDesc d = J.ow[o];
      uint32_t* pl = s_pl + (size_t)ci * 3 * nw;
      for (uint32_t i = lane; i < 3 * nw; i += 64) pl[i] = 0;
      const uint32_t cnt_ops = d.op_cnt;
      const uint32_t* __restrict__ ops = J.ops + d.op_begin;
      const int32_t off = (int32_t)(d.tstart - d.wtstart);
      const uint32_t qw0 = d.qbeg >> 5, nqw = ((d.qbeg + d.qlen) >> 5) - qw0 + 2;
      const bool staged = nqw <= qcap;
      uint32_t* q0 = s_wq + (size_t)wave * 2 * qcap;
      uint32_t* q1 = q0 + qcap;
      // every global load of the overlap is issued here, unconditionally (indices clamped), before anything waits
      uint32_t op_r[3];
#pragma unroll
      for (int i = 0; i < 3; i++) op_r[i] = ops[min(lane + 64u * i, cnt_ops - 1u)];
      constexpr int QI = 5;
      uint32_t v0[QI], v1[QI];
      const uint64_t qbase = d.q_woff + qw0;
#pragma unroll
      for (int i = 0; i < QI; i++) {
        const uint64_t gi = min(qbase + min(lane + 64u * i, nqw - 1u), pmax);
        v0[i] = J.read_p0[gi];
        v1[i] = J.read_p1[gi];
      }
      if (staged) {
#pragma unroll
        for (int i = 0; i < QI; i++) {
          const uint32_t idx = lane + 64u * i;
          if (idx < nqw) { q0[idx] = v0[i]; q1[idx] = v1[i]; }
        }
      }
      const int32_t sbase = d.strand ? (int32_t)(d.qbeg + d.qlen - 1u) : (int32_t)d.qbeg;   // stored index of alignment-orientation base 0
      const int32_t rel = -(int32_t)(qw0 << 5);
      // code planes of the 32 alignment-orientation query bases qidx .. qidx + 31 (bits of bases outside the region are
      // meaningless: callers mask).  Reverse strand: base k is the complement of stored base sbase - k (features.rs:128-153).
      auto qbits = [&](int32_t qidx, uint32_t& b0, uint32_t& b1) {
        if (d.strand == 0) {
          const int32_t s = sbase + qidx;
          if (staged) { b0 = lds_bits(q0, nqw, s + rel); b1 = lds_bits(q1, nqw, s + rel); }
          else { b0 = glb_bits(J.read_p0, d.q_woff, pmax, s); b1 = glb_bits(J.read_p1, d.q_woff, pmax, s); }
        } else {
          const int32_t s = sbase - qidx - 31;
          if (staged) { b0 = ~__brev(lds_bits(q0, nqw, s + rel)); b1 = ~__brev(lds_bits(q1, nqw, s + rel)); }
          else { b0 = ~__brev(glb_bits(J.read_p0, d.q_woff, pmax, s)); b1 = ~__brev(glb_bits(J.read_p1, d.q_woff, pmax, s)); }
        }
      };
      // one plane word of an M run that starts at window position P0 (query index q) and is e long
      auto emit = [&](uint32_t wi, int32_t P0, uint32_t e, uint32_t q) {
        const int32_t ws = (int32_t)(wi << 5);
        const uint32_t seg = mask_range(P0 - ws, P0 + (int32_t)e - ws);
        uint32_t b0, b1;
        qbits((int32_t)q + (ws - P0), b0, b1);
        atomicOr(&pl[wi], seg);
        atomicOr(&pl[nw + wi], b0 & seg);
        atomicOr(&pl[2 * nw + wi], b1 & seg);
      };
      uint32_t carry_t = 0, carry_q = 0, n_ev = 0, isum = 0, dsum = 0, longindel = 0;
      uint4* __restrict__ ev = J.iev + d.scr_off;
      auto step = [&](uint32_t base, uint32_t op) {
        const uint32_t k = base + lane;
        const bool valid = k < cnt_ops;
        const uint32_t ty = op_type(op), len = op_len(op);
        const uint32_t e = valid ? eff_len(op, k, cnt_ops, d.start_off, d.end_off) : 0u;
        const bool isI = valid && ty == OP_I, isM = valid && ty == OP_M, isD = valid && ty == OP_D;
        if ((isI || isD) && len > 50u) longindel = 1;  // untrimmed length (features.rs:317)
        if (isI) isum += e;
        if (isD) dsum += e;
        const uint32_t tadv = (isM || isD) ? e : 0u, qadv = (isM || isI) ? e : 0u;
        uint64_t tot;
        const uint64_t ex = wscan64((uint64_t)tadv | ((uint64_t)qadv << 32), &tot);
        const uint32_t t = carry_t + (uint32_t)ex, q = carry_q + (uint32_t)(ex >> 32);
        // insertion behind window position off + t - 1 (features.rs:77, 219-228): position, trimmed length (bases written),
        // query index of its first base, its first 16 bases, untrimmed length (max_ins)
        const uint64_t imask = __ballot(isI);
        if (isI) {
          const uint32_t idx = n_ev + (uint32_t)__popcll(imask & ((1ull << lane) - 1ull));
          const int32_t pos = off + (int32_t)t - 1;
          uint32_t b0, b1;
          qbits((int32_t)q, b0, b1);
          const uint32_t codes = spread16(b0) | (spread16(b1) << 1);
          ev[idx] = make_uint4(((uint32_t)pos & 0xffffu) | (min(e, 0xffffu) << 16), q, codes, min(len, 0xffffu));
        }
        n_ev += (uint32_t)__popcll(imask);
        // plane words of an M run
        int32_t P0 = 0;
        uint32_t ee = 0, w0 = 0, nwords = 0;
        if (isM) {
          P0 = off + (int32_t)t;
          ee = (uint32_t)P0 < d.wlen ? min(e, d.wlen - (uint32_t)P0) : 0u;
          if (ee) { w0 = (uint32_t)P0 >> 5; nwords = (((uint32_t)P0 + ee - 1u) >> 5) - w0 + 1u; }
        }
        const bool longop = nwords > PCAP;
        if (!longop)
          for (uint32_t i = 0; i < nwords; i++) emit(w0 + i, P0, ee, q);
        uint64_t lm = __ballot(longop);
        while (lm) {
          const int src = __ffsll((unsigned long long)lm) - 1;
          lm &= lm - 1;
          const int32_t P0s = __shfl(P0, src, 64);
          const uint32_t es = __shfl(ee, src, 64), qs = __shfl(q, src, 64), w0s = __shfl(w0, src, 64), nws = __shfl(nwords, src, 64);
          for (uint32_t i = lane; i < nws; i += 64) emit(w0s + i, P0s, es, qs);
        }
        carry_t += (uint32_t)tot;
        carry_q += (uint32_t)(tot >> 32);
      };
#pragma unroll
      for (int i = 0; i < 3; i++)
        if (64u * i < cnt_ops) step(64u * i, op_r[i]);   // wave-uniform
      for (uint32_t base = 192; base < cnt_ops; base += 64) step(base, ops[min(base + lane, cnt_ops - 1u)]);
      const uint32_t t_total = carry_t;
      // the wave reads its own LDS atomics back: LDS operations of one wave execute in order
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // accuracy: matches / mismatches over M ops (features.rs:650-665)
      uint32_t mm = 0, ss = 0;
      for (uint32_t i = lane; i < nw; i += 64) {
        const uint32_t M = pl[i];
        const uint32_t x = (pl[nw + i] ^ s_tp[i]) | (pl[2 * nw + i] ^ s_tp[nw + i]);
        ss += __popc(M & x);
        mm += __popc(M);
      }
      const uint64_t t1 = wsum64((uint64_t)mm | ((uint64_t)ss << 32));
      const uint64_t t2 = wsum64((uint64_t)isum | ((uint64_t)dsum << 32));
      const bool keep = __ballot(longindel != 0u) == 0ull;
      if (lane == 0) {
        const uint32_t s_ = (uint32_t)(t1 >> 32), m_ = (uint32_t)t1 - s_, i_ = (uint32_t)t2, d_ = (uint32_t)(t2 >> 32);
        J.ow_keep[o] = keep ? 1 : 0;
        J.ow_acc[o] = __fdiv_rn((float)m_, (float)(m_ + s_ + i_ + d_));   // (m as f32) / ((m+s+i+d) as f32) (features.rs:678)
        J.ow_ttotal[o] = t_total;
        J.ins_cnt[o] = n_ev;
        ACol c;
        c.off = off; c.t_total = t_total; c.keep = keep ? 1u : 0u; c.pad = 0;
        s_col[ci] = c;
        if (keep) atomicAdd(&s_misc[0], 1u);
      }
      if (keep) {
        uint32_t* __restrict__ g = J.cpl + (uint64_t)o * 3 * nw;
        for (uint32_t i = lane; i < 3 * nw; i += 64) g[i] = pl[i];
      }
    }
    __syncthreads();
    // ---- phase 2: symbol counts per position over the kept columns of the group
    if (tid < nw) {
      const int32_t ws = (int32_t)(tid << 5);
      for (uint32_t ci = 0; ci < ng; ci++) {
        const ACol c = s_col[ci];
        if (!c.keep) continue;
        const uint32_t* pl = s_pl + (size_t)ci * 3 * nw;
        const uint32_t M = pl[tid];
        const uint32_t inr = mask_range(c.off - ws, c.off + (int32_t)c.t_total - ws) & s_tv[tid];
        cnt.add(ColPlanes{M, pl[nw + tid], pl[2 * nw + tid], inr & ~M});
      }
    }
    if (g0 + G < n) __syncthreads();
  }
  __syncthreads();
  const uint32_t n_kept = s_misc[0];
  // ---- informative positions: at least two symbols reach the threshold (features.rs:681-722)
  if (tid < nw) {
    const uint32_t ncols = 1u + (n_kept > 30u ? n_kept : 30u);  // features.rs:282
    const uint32_t thresh = (uint32_t)((double)ncols * 0.1);    // features.rs:712
    uint32_t one = 0, two = 0;
#pragma unroll
    for (int s5 = 0; s5 < 5; s5++) {
      const uint32_t g = cnt.ge(s5, thresh);
      two |= one & g;
      one |= g;
    }
    const uint32_t sup = (thresh == 0 ? 0xffffffffu : two) & s_tv[tid];
    s_sup[tid] = sup;
    if (sup) s_list[atomicAdd(&s_misc[1], 1u)] = tid;
  }
  __syncthreads();
  // ---- tallies (features.rs:478-498): every kept column is scored at every informative target position; anything but
  // the target's base ('.', '*', '#', another base) is a mismatch.  Informative positions are rare: only the plane words
  // that hold one are looked at.
  {
    const uint32_t nl = s_misc[1];
    const bool in_lds = n <= G;
    for (uint32_t item = tid; item < nl * n; item += PA_NT) {
      const uint32_t li = item / n, c = item - li * n;
      const uint32_t o = wd.ow_begin + c;
      if (!J.ow_keep[o]) continue;
      const uint32_t wi = s_list[li], sup = s_sup[wi];
      const uint32_t* src = in_lds ? (const uint32_t*)(s_pl + (size_t)c * 3 * nw) : (const uint32_t*)(J.cpl + (uint64_t)o * 3 * nw);
      const uint32_t M = src[wi], lo = src[nw + wi], hi = src[2 * nw + wi];
      const uint32_t match = sup & M & ~((lo ^ s_tp[wi]) | (hi ^ s_tp[nw + wi]));
      const uint32_t nm = __popc(match), cls = J.ow[o].cls;
      if (nm) atomicAdd(&J.nd[2 * (uint64_t)cls], nm);
      atomicAdd(&J.nd[2 * (uint64_t)cls + 1], (uint32_t)__popc(sup) - nm);
    }
  }
  // ---- stable rank of the kept overlaps by descending accuracy: sort_by_key(-acc) (features.rs:386-409)
  for (uint32_t i = tid; i < n; i += PA_NT) {
    const uint32_t oi = wd.ow_begin + i;
    if (!J.ow_keep[oi]) continue;
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
  if (tid == 0) J.win_nkept[w] = n_kept;
}

// =====================================================================================================
// k_final
// =====================================================================================================
struct __attribute__((aligned(16))) BCol {
  int32_t off;       // window position where the overlap starts
  uint32_t t_total;  // target bases it covers (0: padding column)
  uint32_t tokc;     // token offset of a base (0 forward, 5 reverse) | gap token << 8
  uint32_t ow;
};

struct FinalLds {   // byte offsets into the dynamic LDS block
  uint32_t hdr, uni, pl, rbase, rop16, misc, sel, nev, pref, supbits, wave, total;
};
__host__ __device__ inline FinalLds final_lds(uint32_t W, uint32_t nw) {
  FinalLds L;
  uint32_t cur = 0;
  auto take = [&](uint32_t bytes) { const uint32_t o = cur; cur = (cur + bytes + 15u) & ~15u; return o; };
  L.hdr = take(32 * sizeof(BCol));
  const uint32_t a = W * 4u, b = ROWCAP * 2u + 32u * INSCAP, c = SCAP * 8u;   // max insertion per position | row info + insertion tile | scores
  L.uni = take(a > b ? (a > c ? a : c) : (b > c ? b : c));
  L.pl = take(HERRO_ROWS * 3u * (nw + 1u) * 4u);
  L.rbase = take((nw + 1u) * 4u);
  L.rop16 = take((W + 2u) * 2u);
  L.misc = take(32 * 4);
  L.sel = take(32 * 4);
  L.nev = take(32 * 4);
  L.pref = take(32 * 4);
  L.supbits = take((ROWCAP / 32) * 4);
  L.wave = take((PB_NT / 64) * 4);
  L.total = cur;
  return L;
}

__device__ __forceinline__ uint32_t dpp_quad_add(uint32_t v) {   // sum over the 4 lanes of a quad, in every lane
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ uint32_t dpp_quad_lane0(uint32_t v) {  // value of the quad's first lane
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xf, 0xf, true);  // quad_perm [0,0,0,0]
}

__global__ __launch_bounds__(PB_NT) void k_final(JobDev J) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pb_smem[];
  const uint32_t nw = J.nw, nwp = nw + 1, W = J.window_size;
  const FinalLds LO = final_lds(W, nw);
  BCol* s_hdr = reinterpret_cast<BCol*>(pb_smem + LO.hdr);
  uint32_t* s_mi = reinterpret_cast<uint32_t*>(pb_smem + LO.uni);          // phase 2-3
  double* s_score = reinterpret_cast<double*>(pb_smem + LO.uni);           // phase 0
  uint16_t* s_rowinfo = reinterpret_cast<uint16_t*>(pb_smem + LO.uni);     // chunks: position | first-row flag << 15
  uint8_t* s_ins = pb_smem + LO.uni + ROWCAP * 2;                          // chunks: [32][INSCAP] tokens of inserted bases, 0xff: none
  uint32_t* s_pl = reinterpret_cast<uint32_t*>(pb_smem + LO.pl);           // [31][3][nwp]
  uint32_t* s_rbase = reinterpret_cast<uint32_t*>(pb_smem + LO.rbase);     // [nwp] row of position 32 i
  uint16_t* s_rop16 = reinterpret_cast<uint16_t*>(pb_smem + LO.rop16);     // [W+1] row of position p - s_rbase[p >> 5]
  uint32_t* s_misc = reinterpret_cast<uint32_t*>(pb_smem + LO.misc);
  uint32_t* s_sel = reinterpret_cast<uint32_t*>(pb_smem + LO.sel);
  uint32_t* s_nev = reinterpret_cast<uint32_t*>(pb_smem + LO.nev);
  uint32_t* s_pref = reinterpret_cast<uint32_t*>(pb_smem + LO.pref);
  uint32_t* s_supbits = reinterpret_cast<uint32_t*>(pb_smem + LO.supbits);
  uint32_t* s_wave = reinterpret_cast<uint32_t*>(pb_smem + LO.wave);

  const uint32_t w = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
  const WinDesc wd = J.win[w];
  const uint32_t n_kept = J.win_nkept[w];
  const uint32_t win_len = wd.win_len;
  const uint64_t pmax = J.read_n_words + 1;
  if (tid < 32) { s_sel[tid] = NONE; s_misc[tid] = 0; }
  __syncthreads();

  // ---- phase 0: score n/(n+d)*ln(n+d+1) in f64 (features.rs:505-510); stable descending rank (features.rs:512-513)
  {
    auto score_of = [&](uint32_t k) -> double {
      const uint32_t cls = J.ow[J.slot_ow[wd.ow_begin + k]].cls;
      const uint32_t nn = J.nd[2 * (uint64_t)cls], dd = J.nd[2 * (uint64_t)cls + 1];
      const uint32_t tot = nn + dd;
      if (!tot) return 0.0;
      const double lg = tot < J.ln_table_n ? J.ln_table[tot] : log((double)tot + 1.0);
      return __dmul_rn(__ddiv_rn((double)nn, (double)tot), lg);
    };
    const bool cached = n_kept <= SCAP;
    if (cached) {
      for (uint32_t k = tid; k < n_kept; k += PB_NT) s_score[k] = score_of(k);
      __syncthreads();
    }
    for (uint32_t k = tid; k < n_kept; k += PB_NT) {
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
    __syncthreads();   // also: nobody reads s_score any more
  }
  // ---- phase 1: headers and planes of the selected columns (column 0 = the target), max-insertion array cleared
  if (tid < 32) {
    J.sel_ow[(uint64_t)w * 32 + tid] = s_sel[tid];
    BCol h;
    h.off = 0; h.t_total = 0; h.tokc = (uint32_t)TOK_GAP_F << 8; h.ow = NONE;
    uint32_t nev = 0;
    if (tid == 0) h.t_total = win_len;
    else if (s_sel[tid] != NONE) {
      const uint32_t o = s_sel[tid];
      const OwDesc& d = J.ow[o];
      h.off = (int32_t)(d.tstart - d.wtstart);
      h.t_total = J.ow_ttotal[o];
      h.tokc = d.strand ? (5u | ((uint32_t)TOK_GAP_R << 8)) : ((uint32_t)TOK_GAP_F << 8);
      h.ow = o;
      nev = J.ins_cnt[o];
    }
    s_hdr[tid] = h;
    s_nev[tid] = nev;
  }
  {
    const uint32_t per = 3 * nw, total = (HERRO_ROWS - 1) * per;
    for (uint32_t it = tid; it < total; it += PB_NT) {
      const uint32_t c = it / per, rem = it - c * per, pi = rem / nw, wi = rem - pi * nw;
      const uint32_t o = s_sel[c + 1];
      s_pl[((c + 1) * 3 + pi) * nwp + wi] = o != NONE ? J.cpl[(uint64_t)o * per + rem] : 0u;
    }
    if (tid < nw) {
      const uint64_t t_woff = J.read_word_off[wd.rid];
      const int32_t P = (int32_t)(tid << 5);
      const uint32_t vm = mask_range(0, (int32_t)win_len - P);
      s_pl[0 * nwp + tid] = vm;
      s_pl[1 * nwp + tid] = glb_bits(J.read_p0, t_woff, pmax, (int32_t)wd.tstart + P) & vm;
      s_pl[2 * nwp + tid] = glb_bits(J.read_p1, t_woff, pmax, (int32_t)wd.tstart + P) & vm;
    }
    if (tid < HERRO_ROWS * 3) s_pl[tid * nwp + nw] = 0;   // pad word of every plane
    for (uint32_t p = tid; p < W; p += PB_NT) s_mi[p] = 0;
  }
  __syncthreads();
  // ---- phase 2: max insertion behind every position over the SELECTED overlaps: rows where every selected column is a
  // gap are dropped (features.rs:531-556), i.e. the final layout is the row map of the selected overlaps alone
  if (tid == 0) {
    uint32_t acc = 0;
    for (uint32_t c = 0; c < 32; c++) { s_pref[c] = acc; acc += s_nev[c]; }
    s_misc[1] = acc;
  }
  __syncthreads();
  const uint32_t n_events = s_misc[1];
  auto event_col = [&](uint32_t e) -> uint32_t {   // column of flattened event e: largest c with s_pref[c] <= e
    uint32_t lo = 0, hi = 32;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s_pref[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
  };
  for (uint32_t e = tid; e < n_events; e += PB_NT) {
    const uint32_t c = event_col(e);
    const uint4 v = J.iev[J.ow[s_hdr[c].ow].scr_off + (e - s_pref[c])];
    const uint32_t p = v.x & 0xffffu;
    if (p < win_len) atomicMax(&s_mi[p], v.w);   // untrimmed length (features.rs:64-79)
  }
  __syncthreads();
  // ---- phase 3: row of every position = exclusive prefix of (1 + max insertion)
  uint32_t Lf;
  {
    uint32_t ch = 1;
    while (ch * PB_NT < win_len) ch <<= 1;       // consecutive positions per thread: a power of two <= 32
    const uint32_t p0 = tid * ch, p1 = min(p0 + ch, win_len);
    uint32_t local = 0;
    for (uint32_t p = p0; p < p1; p++) local += 1u + s_mi[p];
    const uint32_t ex = blk_scan<PB_NT>(local, &Lf, s_wave);
    const uint32_t per_blk = 32u / ch;                           // threads per block of 32 positions (same wave)
    const uint32_t rb = __shfl(ex, (int)(lane & ~(per_blk - 1u)), 64);
    if (p0 < win_len && (p0 & 31u) == 0) s_rbase[p0 >> 5] = ex;
    uint32_t r = ex;
    for (uint32_t p = p0; p < p1; p++) {
      s_rop16[p] = (uint16_t)(r - rb);
      r += 1u + s_mi[p];
    }
    if (p0 < win_len && p1 == win_len) {   // the entry behind the last position: row count
      if ((win_len & 31u) == 0) { s_rbase[win_len >> 5] = r; s_rop16[win_len] = 0; }
      else s_rop16[win_len] = (uint16_t)(r - rb);
    }
    if (tid == 0) J.win_Lf[w] = Lf;
  }
  __syncthreads();   // s_mi is dead from here on (its memory becomes the chunk buffers)
  auto rop = [&](uint32_t p) -> uint32_t { return s_rbase[p >> 5] + s_rop16[p]; };
  auto lower = [&](uint32_t r) -> uint32_t {   // first p in [0, win_len] with rop(p) >= r (win_len + 1: none)
    uint32_t lo = 0, hi = win_len + 1;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (rop(mid) < r) lo = mid + 1; else hi = mid;
    }
    return lo;
  };

  // ---- phase 4: the token planes, in chunks of rows [r0, r1)
  const uint32_t seg = tid >> 2, cg = tid & 3u;
  uint32_t n_sup_total = 0;   // meaningful in wave 0
  for (uint32_t r0 = 0; r0 < Lf;) {
    const uint32_t pa = lower(r0 + 1) - 1;           // position of row r0
    const uint32_t f0 = rop(pa) == r0 ? 1u : 0u;     // its base row lies in the chunk
    uint32_t rows = ROWCAP, r1c, pb1;
    for (;;) {
      r1c = min(r0 + rows, Lf);
      pb1 = lower(r1c);                              // positions with a base row before r1c
      const uint32_t nj0 = pb1 - (f0 ? pa : pa + 1);
      const uint32_t nins = (r1c - r0) - nj0;
      if (nins <= INSCAP || rows <= 16) break;
      rows = max(16u, (rows >> 1) & ~15u);
    }
    const uint32_t pb = pb1 - 1;                      // last position with a row in the chunk (>= pa)
    const uint32_t nrows = r1c - r0, nrows16 = (nrows + 15u) & ~15u;
    // -- A: row info, row map, cleared insertion tile
    for (uint32_t p = pa + tid; p <= pb; p += PB_NT) {
      const uint32_t rp = rop(p), nr = rop(p + 1) - rp;
      for (uint32_t j = 0; j < nr; j++) {
        const uint32_t row = rp + j;
        if (row >= r0 && row < r1c) {
          s_rowinfo[row - r0] = (uint16_t)(p | (j == 0 ? 0x8000u : 0u));
          J.rowmap2[wd.row_off + row] = p | (j << 16);
        }
      }
    }
    for (uint32_t i = nrows + tid; i < nrows16; i += PB_NT) s_rowinfo[i] = (uint16_t)pb;   // rows past the window's last: unused
    for (uint32_t i = tid; i < 32 * INSCAP / 4; i += PB_NT) reinterpret_cast<uint32_t*>(s_ins)[i] = 0xffffffffu;
    if (tid < ROWCAP / 32) s_supbits[tid] = 0;
    __syncthreads();
    // -- B: inserted bases of the selected columns (features.rs:213-229).  A later insertion at the same position
    // overwrites an earlier one from its first row on, as the reference's sequential writes do.
    for (uint32_t e = tid; e < n_events; e += PB_NT) {
      const uint32_t c = event_col(e), ei = e - s_pref[c], ne = s_nev[c];
      const uint32_t o = s_hdr[c].ow;
      const uint4* __restrict__ evs = J.iev + J.ow[o].scr_off;
      const uint4 v = evs[ei];
      const uint32_t p = v.x & 0xffffu, len = v.x >> 16;
      if (p < pa || p > pb || p >= win_len) continue;
      const uint32_t rp = rop(p), room = rop(p + 1) - rp - 1u;
      uint32_t hide = 0;   // rows [0, hide) are overwritten by later insertions at the same position
      for (uint32_t e2 = ei + 1; e2 < ne; e2++) {
        const uint4 v2 = evs[e2];
        if ((v2.x & 0xffffu) != p) break;
        hide = max(hide, v2.x >> 16);
      }
      const uint32_t s5 = s_hdr[c].tokc & 0xffu;
      for (uint32_t k = hide; k < len && k < room; k++) {
        const uint32_t row = rp + 1u + k;
        if (row < r0 || row >= r1c) continue;
        uint32_t code;
        if (k < 16u) code = (v.z >> (2u * k)) & 3u;
        else {   // long insertion: bases beyond the 16 carried by the event come from the read store
          const OwDesc& d = J.ow[o];
          const uint32_t qi = v.y + k;
          const uint32_t si = d.strand ? d.qbeg + d.qlen - 1u - qi : d.qbeg + qi;
          const uint64_t wi = min(d.q_woff + (si >> 5), pmax);
          code = ((J.read_p0[wi] >> (si & 31u)) & 1u) | (((J.read_p1[wi] >> (si & 31u)) & 1u) << 1);
          if (d.strand) code ^= 3u;
        }
        const uint32_t idx = (row - r0) - (p - pa + f0);
        if (idx < INSCAP) s_ins[c * INSCAP + idx] = (uint8_t)(s5 + code);
      }
    }
    __syncthreads();
    // -- C: 16 rows x 1 column per step; a quad of lanes shares a row segment, each lane a quarter of the columns
    if (seg * 16u < nrows) {
      const uint4 ri0 = reinterpret_cast<const uint4*>(s_rowinfo)[seg * 2], ri1 = reinterpret_cast<const uint4*>(s_rowinfo)[seg * 2 + 1];
      const uint32_t riw[8] = {ri0.x, ri0.y, ri0.z, ri0.w, ri1.x, ri1.y, ri1.z, ri1.w};
      const uint32_t p_first = riw[0] & 0x7fffu;
      uint32_t kj[16], cnt[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const uint32_t v = (riw[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
        kj[i] = ((v & 0x7fffu) - p_first) | ((v >> 15) << 5);   // bits 0..4: position - p_first (< 16), bit 5: base row
        cnt[i] = 0;
      }
      const int32_t C = (int32_t)(seg * 16u) - (int32_t)(p_first - pa + f0);   // insertion-tile index of row i with offset k: C - k + i
      const uint32_t wi = p_first >> 5, sh = p_first & 31u;
      const uint64_t gseg = wd.fin_off + r0 + seg * 16u;
      uint32_t tgt[4] = {0, 0, 0, 0};
      for (uint32_t c = cg; c < HERRO_ROWS; c += 4) {
        const BCol h = s_hdr[c];
        const uint32_t* pl = s_pl + (size_t)c * 3 * nwp + wi;
        const uint32_t m16 = __funnelshift_r(pl[0], pl[1], sh);
        const uint32_t l16 = __funnelshift_r(pl[nwp], pl[nwp + 1], sh);
        const uint32_t h16 = __funnelshift_r(pl[2 * nwp], pl[2 * nwp + 1], sh);
        const uint32_t r16 = mask_range(h.off - (int32_t)p_first, h.off + (int32_t)h.t_total - (int32_t)p_first);
        const uint32_t s5 = h.tokc & 0xffu, gapt = h.tokc >> 8;
        const uint8_t* insrow = s_ins + c * INSCAP;
        uint32_t T[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const uint32_t k = kj[i];
          const uint32_t j0 = __builtin_amdgcn_ubfe(k, 5, 1);
          const uint32_t m = __builtin_amdgcn_ubfe(m16, k, 1) & j0;
          const uint32_t code = __builtin_amdgcn_ubfe(l16, k, 1) | (__builtin_amdgcn_ubfe(h16, k, 1) << 1);
          uint32_t r = __builtin_amdgcn_ubfe(r16, k, 1);
          uint32_t fidx = m ? code : 4u;
          uint32_t tok = m ? code + s5 : gapt;
          tok = r ? tok : (uint32_t)TOK_NONE;
          if (!j0) {   // insertion row: a base if this column inserts here (it counts even in front of the overlap's first base)
            const int32_t ii = C - (int32_t)(k & 31u) + i;
            const uint32_t b = insrow[min((uint32_t)max(ii, 0), INSCAP - 1u)];
            if (b != 0xffu) { tok = b; fidx = b >= 5u ? b - 5u : b; r = 1u; }
          }
          T[i >> 2] |= tok << ((i & 3) * 8);
          cnt[i] += r << (5u * fidx);
        }
        *reinterpret_cast<uint4*>(J.fin_b + gseg + (uint64_t)c * wd.lub) = make_uint4(T[0], T[1], T[2], T[3]);
        if (c == 0) { tgt[0] = T[0]; tgt[1] = T[1]; tgt[2] = T[2]; tgt[3] = T[3]; }
      }
      // counts of the row over all 31 columns, then each lane of the quad finishes 4 rows
#pragma unroll
      for (int i = 0; i < 16; i++) cnt[i] = dpp_quad_add(cnt[i]);
      const uint32_t tg0 = dpp_quad_lane0(tgt[0]), tg1 = dpp_quad_lane0(tgt[1]), tg2 = dpp_quad_lane0(tgt[2]), tg3 = dpp_quad_lane0(tgt[3]);
      const uint32_t tgw = cg == 0 ? tg0 : (cg == 1 ? tg1 : (cg == 2 ? tg2 : tg3));
      uint32_t supb = 0, consw = 0;
#pragma unroll
      for (int q4 = 0; q4 < 4; q4++) {
        // row 4 cg + q4: select its counter word without indexing the register array dynamically
        const uint32_t cw = cg == 0 ? cnt[q4] : (cg == 1 ? cnt[4 + q4] : (cg == 2 ? cnt[8 + q4] : cnt[12 + q4]));
        uint32_t c5[5];
#pragma unroll
        for (int q = 0; q < 5; q++) c5[q] = (cw >> (5 * q)) & 31u;
        // informative rows of the final [L',31] matrix: thresh = (31 * 0.1) as usize = 3 (features.rs:558,712)
        const uint32_t thresh = (uint32_t)((double)HERRO_ROWS * 0.1);
        uint32_t ns = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) ns += c5[q] >= thresh ? 1u : 0u;
        const uint32_t row_in_chunk = seg * 16u + cg * 4u + q4;
        if (ns >= 2 && row_in_chunk < nrows) supb |= 1u << q4;
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
      *reinterpret_cast<uint32_t*>(J.cons_tmp + wd.row_off + r0 + seg * 16u + cg * 4u) = consw;
      if (supb) atomicOr(&s_supbits[(seg * 16u + cg * 4u) >> 5], supb << ((seg * 16u + cg * 4u) & 31u));
    }
    __syncthreads();
    // -- D: ordered list of informative positions (SupportedPos, features.rs:896-900), by the first wave
    if (tid < 64) {
      const uint32_t bits = s_supbits[tid];
      uint64_t tot;
      const uint32_t ex = (uint32_t)wscan64((uint64_t)__popc(bits), &tot);
      uint32_t k = n_sup_total + ex;
      for (uint32_t m = bits; m; m &= m - 1u) {
        const uint32_t ric = (tid << 5) + (uint32_t)__ffs((int)m) - 1u;
        const uint32_t p = s_rowinfo[ric] & 0x7fffu, row = r0 + ric;
        J.sup_row[wd.row_off + k] = row;
        J.sup_pi[wd.row_off + k] = p | ((row - rop(p)) << 16);
        k++;
      }
      n_sup_total += (uint32_t)tot;
    }
    __syncthreads();
    r0 += nrows16;   // r0 stays a multiple of 16; only the window's last chunk has nrows % 16 != 0
  }
  if (tid == 0) J.win_nsup[w] = n_sup_total;
}

// =====================================================================================================
// k_quals
// =====================================================================================================
struct QCol {
  int32_t off;
  uint32_t t_total;
  int32_t sbase, sdir;   // stored index of alignment-orientation query base q: sbase + sdir * q
  uint32_t n_ev, ev_lds; // events: count, first slot in the LDS copy
  uint64_t qual_off;     // first quality byte of the query read
  uint64_t ev_glb;       // first event in J.iev
};
__host__ __device__ inline size_t quals_lds(uint32_t nw) {
  return 32 * sizeof(QCol) + (size_t)(HERRO_ROWS - 1) * nw * 4 + (((size_t)(HERRO_ROWS - 1) * nw * 2 + 15) & ~(size_t)15) + (size_t)QEVCAP * 16 + 64;
}

// FULL: every cell of the window; otherwise the cells within `half` rows of an informative row (the model's receptive fields).
template <bool FULL>
__global__ __launch_bounds__(PQ_NT) void k_quals(JobDev J, uint32_t half) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pq_smem[];
  const uint32_t nw = J.nw;
  QCol* s_col = reinterpret_cast<QCol*>(pq_smem);
  uint32_t* s_M = reinterpret_cast<uint32_t*>(pq_smem + 32 * sizeof(QCol));          // [30][nw]
  uint16_t* s_rk = reinterpret_cast<uint16_t*>(s_M + (size_t)(HERRO_ROWS - 1) * nw);  // [30][nw] M bits before the word
  uint4* s_ev = reinterpret_cast<uint4*>(pq_smem + 32 * sizeof(QCol) + (size_t)(HERRO_ROWS - 1) * nw * 4 + (((size_t)(HERRO_ROWS - 1) * nw * 2 + 15) & ~(size_t)15));
  uint32_t* s_misc = reinterpret_cast<uint32_t*>(s_ev + QEVCAP);

  const uint32_t w = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t nsup = J.win_nsup[w], Lf = J.win_Lf[w];
  if (!FULL && !nsup) return;
  const WinDesc wd = J.win[w];
  if (tid == 0) {
    uint32_t acc = 0;
    for (uint32_t c = 1; c < HERRO_ROWS; c++) {
      const uint32_t o = J.sel_ow[(uint64_t)w * 32 + c];
      QCol q;
      q.off = 0; q.t_total = 0; q.sbase = 0; q.sdir = 1; q.n_ev = 0; q.ev_lds = acc; q.qual_off = 0; q.ev_glb = 0;
      if (o != NONE) {
        const OwDesc& d = J.ow[o];
        q.off = (int32_t)(d.tstart - d.wtstart);
        q.t_total = J.ow_ttotal[o];
        q.sbase = d.strand ? (int32_t)(d.qbeg + d.qlen - 1u) : (int32_t)d.qbeg;
        q.sdir = d.strand ? -1 : 1;
        q.n_ev = J.ins_cnt[o];
        q.qual_off = d.q_qual_off;
        q.ev_glb = d.scr_off;
      }
      s_col[c] = q;
      acc += q.n_ev;
    }
    s_misc[0] = acc;
  }
  __syncthreads();
  const bool ev_in_lds = s_misc[0] <= QEVCAP;
  // M planes of the selected columns, their rank directories, the events
  for (uint32_t it = tid; it < (HERRO_ROWS - 1) * nw; it += PQ_NT) {
    const uint32_t c = it / nw, wi = it - c * nw;
    const uint32_t o = J.sel_ow[(uint64_t)w * 32 + c + 1];
    s_M[it] = o != NONE ? J.cpl[(uint64_t)o * 3 * nw + wi] : 0u;
  }
  if (ev_in_lds) {
    for (uint32_t c = 1; c < HERRO_ROWS; c++) {
      const QCol& q = s_col[c];
      for (uint32_t i = tid; i < q.n_ev; i += PQ_NT) s_ev[q.ev_lds + i] = J.iev[q.ev_glb + i];
    }
  }
  __syncthreads();
  for (uint32_t c = wave; c < HERRO_ROWS - 1; c += PQ_NT / 64) {
    uint32_t carry = 0;
    for (uint32_t b = 0; b < nw; b += 64) {
      const uint32_t i = b + lane;
      const uint32_t pc = i < nw ? (uint32_t)__popc(s_M[c * nw + i]) : 0u;
      uint64_t tot;
      const uint32_t ex = (uint32_t)wscan64(pc, &tot);
      if (i < nw) s_rk[c * nw + i] = (uint16_t)(carry + ex);
      carry += (uint32_t)tot;
    }
  }
  __syncthreads();

  const uint64_t tq_off = J.read_qual_off[wd.rid] + wd.tstart;
  // quality of cell (column c >= 1, position p, insertion ordinal j): '!' unless the cell holds a query base
  auto cell = [&](uint32_t c, uint32_t p, uint32_t j) -> uint32_t {
    const QCol& q = s_col[c];
    if (q.t_total == 0 && q.n_ev == 0) return 33u;
    const uint4* evg = J.iev + q.ev_glb;
    auto event = [&](uint32_t i) -> uint4 { return ev_in_lds ? s_ev[q.ev_lds + i] : evg[i]; };
    // first event with position >= bound
    auto first_ge = [&](uint32_t bound) -> uint32_t {
      uint32_t lo = 0, hi = q.n_ev;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((event(mid).x & 0xffffu) < bound) lo = mid + 1; else hi = mid;
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
      if (uu >= q.t_total || !((M[p >> 5] >> (p & 31u)) & 1u)) return 33u;
      const uint32_t ub = first_ge(p);          // events strictly before p: [0, ub)
      if (ub == 0) qi = rank(p);
      else {
        const uint4 e = event(ub - 1);
        qi = e.y + (e.x >> 16) + rank(p) - rank((e.x & 0xffffu) + 1u);
      }
    } else {
      uint32_t i = first_ge(p + 1);             // events at positions <= p: [0, i)
      bool found = false;
      qi = 0;
      while (i > 0) {
        const uint4 e = event(i - 1);
        if ((e.x & 0xffffu) != p) break;
        if ((e.x >> 16) >= j) { qi = e.y + j - 1u; found = true; break; }   // the LAST insertion at p that is long enough wrote this row
        i--;
      }
      if (!found) return 33u;
    }
    const int64_t si = (int64_t)q.sbase + (int64_t)q.sdir * (int64_t)qi;
    const uint64_t gi = q.qual_off + (uint64_t)max(si, (int64_t)0);
    return J.read_qual[min(gi, J.read_qual_bytes ? J.read_qual_bytes - 1 : 0)];
  };
  if (FULL) {
    const uint64_t total = (uint64_t)Lf * HERRO_ROWS;
    for (uint64_t idx = tid; idx < total; idx += PQ_NT) {
      const uint32_t c = (uint32_t)(idx / Lf), r = (uint32_t)(idx - (uint64_t)c * Lf);
      const uint32_t rm = J.rowmap2[wd.row_off + r];
      const uint32_t p = rm & 0xffffu, j = rm >> 16;
      uint32_t qv;
      if (c == 0) qv = j == 0 ? (uint32_t)J.read_qual[tq_off + p] : 33u;
      else qv = cell(c, p, j);
      J.fin_q[wd.fin_off + (uint64_t)c * wd.lub + r] = (uint8_t)qv;
    }
  } else {
    const uint32_t span = 2 * half + 1;
    const uint64_t total = (uint64_t)nsup * span * HERRO_ROWS;
    for (uint64_t idx = tid; idx < total; idx += PQ_NT) {
      // neighbouring lanes: the rows of one receptive field in one column
      const uint32_t k = (uint32_t)(idx / (span * HERRO_ROWS)), rem = (uint32_t)(idx - (uint64_t)k * span * HERRO_ROWS), c = rem / span, dd = rem - c * span;
      const int64_t r = (int64_t)J.sup_row[wd.row_off + k] + (int64_t)dd - (int64_t)half;
      if (r < 0 || r >= (int64_t)Lf) continue;
      const uint32_t rm = J.rowmap2[wd.row_off + (uint32_t)r];
      const uint32_t p = rm & 0xffffu, j = rm >> 16;
      uint32_t qv;
      if (c == 0) qv = j == 0 ? (uint32_t)J.read_qual[tq_off + p] : 33u;
      else qv = cell(c, p, j);
      J.fin_q[wd.fin_off + (uint64_t)c * wd.lub + (uint32_t)r] = (uint8_t)qv;
    }
  }
}

// =====================================================================================================
// k_consensus — one workgroup per window: corrected bases on the device (consensus.rs:86-227)
// =====================================================================================================
// Per final row: informative -> argmax of the 5 base logits (the LAST maximum wins, NaN is greatest —
// max_by_key(OrderedFloat), consensus.rs:136-141); otherwise the majority vote with the target tie-break
// (consensus.rs:178-200) that k_final already derived from its symbol counts.  '*' is dropped.  The window's
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
  uint8_t* tmp = J.cons_tmp + wd.row_off;  // per-row majority vote, written by k_final
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
  for (uint32_t r = a; r < b; r++) {
    const uint32_t base = tmp[r];
    if (base != 4u) seq[o++] = (uint8_t)"ACGT"[base];
  }
  if (threadIdx.x == 0) J.cons_len[w] = total;
}

}  // namespace

void launch_consensus(const JobDev& J, const uint64_t* sup_off, const float* base_logits, hipStream_t st, KernelTimer* tm) {
  if (!J.n_win) return;
  KT_BEGIN(tm, "consensus", st);
  hipLaunchKernelGGL(k_consensus, dim3(J.n_win), dim3(PC_NT), 0, st, J, sup_off, base_logits);
  KT_END(tm, st);
}

void launch_rf_quals(const JobDev& J, uint32_t half, hipStream_t st, KernelTimer* tm) {
  if (!J.n_win) return;
  KT_BEGIN(tm, "rf_quals", st);
  pileup_opt_in_lds(reinterpret_cast<const void*>(k_quals<false>));
  hipLaunchKernelGGL(k_quals<false>, dim3(J.n_win), dim3(PQ_NT), quals_lds(J.nw), st, J, half);
  KT_END(tm, st);
}

void launch_full_quals(const JobDev& J, hipStream_t st) {
  if (!J.n_win) return;
  pileup_opt_in_lds(reinterpret_cast<const void*>(k_quals<true>));
  hipLaunchKernelGGL(k_quals<true>, dim3(J.n_win), dim3(PQ_NT), quals_lds(J.nw), st, J, 0u);
}

size_t pileup_lds_bytes(uint32_t W, int which) {   // for DESIGN / diagnostics
  const uint32_t nw = (W + 31) / 32;
  return which == 0 ? pass1_lds(nw) : (which == 1 ? (size_t)final_lds(W, nw).total : quals_lds(nw));
}

template <int NB>
static void launch_pass1(const JobDev& J, hipStream_t st) {
  pileup_opt_in_lds(reinterpret_cast<const void*>(k_pass1<NB>));
  hipLaunchKernelGGL(k_pass1<NB>, dim3(J.n_win), dim3(PA_NT), pass1_lds(J.nw), st, J);
}

void launch_featurize(const JobDev& J, hipStream_t st, KernelTimer* tm) {
  if (!J.n_win) return;
  KT_BEGIN(tm, "pass1", st);
  if (J.n_cls) (void)hipMemsetAsync(J.nd, 0, (size_t)J.n_cls * 8, st);   // match / mismatch tallies start from zero
  {
    // counter width: enough bits for the largest threshold floor(0.1 * max(31, columns))
    const uint32_t tmax = (uint32_t)((double)(J.max_cols > 31 ? J.max_cols : 31) * 0.1);
    if (tmax < 4) launch_pass1<2>(J, st);
    else if (tmax < 8) launch_pass1<3>(J, st);
    else if (tmax < 16) launch_pass1<4>(J, st);
    else if (tmax < 64) launch_pass1<6>(J, st);
    else if (tmax < 512) launch_pass1<9>(J, st);
    else launch_pass1<14>(J, st);
  }
  KT_END(tm, st);
  KT_BEGIN(tm, "final", st);
  pileup_opt_in_lds(reinterpret_cast<const void*>(k_final));
  hipLaunchKernelGGL(k_final, dim3(J.n_win), dim3(PB_NT), final_lds(J.window_size, J.nw).total, st, J);
  KT_END(tm, st);
}

}  // namespace herro
