// cigar_dev.hip — CIGAR text -> binary ops + cut records on the GPU, one workgroup per alignment.
//
// herro_job_create's host cost was ~95 % text decoding (5 000 ops per 4096-bp window of 32 overlaps, ~4 ns each on the
// host: 20 us of CPU per window, which on a host with few usable cores caps the end-to-end rate at half the device rate).
// The text is byte work with a prefix-sum structure, i.e. GPU work: the host stages the bytes (one memcpy, one H2D), this
// kernel writes the ops where the pileup kernels will read them and returns only what the windowing needs — per
// alignment the totals and the few ops that reach a window boundary ("cuts", windowing.hpp) — and the host cuts windows
// from those records without ever seeing an op (window_cuts).
//
// Per alignment, ONE WAVE sweeps the text in steps of 1024 bytes (16 per lane, one 16-byte load each; round 6 — a workgroup of 256 threads per alignment
// with block scans until round 5):
//   letters    bit 6 of a byte separates 'A'..'Z' from '0'..'9': a 16-bit letter mask per lane
//   scan 1     exclusive max over "position of my last letter": where the digits of my first op start
//   pass 1     decode my ops (<= 8) from the wave's LDS copy of the step: per-lane op count and target / query / insertion totals
//   scan 2     exclusive sums of those four (DPP)
//   pass 2     the same ops (kept in registers: at most eight per lane), now with op index and running totals known: write the op, test for a window boundary
//              (the next step's 1 KB is requested before the current one is looked at)
// Malformed text (what CigarIter panics on, aligners.rs:252-293) only raises a flag; the host re-reads that one text
// for the message.  Reference: extract_windows walks the same ops one by one on a feature thread (windowing.rs:44-273).
#include "cigar_dev.h"

#include "pileup_core.h"

namespace herro {
namespace {

constexpr uint32_t CW = 4;            // alignments (waves) per workgroup
constexpr uint32_t CB = 16;           // text bytes per lane and step
constexpr uint32_t CHUNK = 64 * CB;   // ... per wave and step

// Wave scans on the DPP network (row_shr inside rows of 16 lanes, then the two row broadcasts): six vector instructions each.  The kernel's first version ran
// five scans per chunk over __shfl_up — a round trip through the LDS crossbar per step, thirty dependent ones per chunk — inside a 256-thread workgroup per
// alignment with three block barriers; the texts of the bench's alignments are 1.5 KB, 96 lanes' worth.  Round 6: one WAVE per alignment, no barrier.
__device__ inline uint32_t wave_incl_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ inline uint32_t wave_incl_max(uint32_t v) {   // (values are >= 0: lanes a shift does not reach read 0)
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
  return v;
}
__device__ inline uint32_t lane63(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
// the wave reads back what its other lanes wrote to the LDS: LDS operations of one wave execute in order
__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// letter flags of four text bytes as a 4-bit mask (bit 6 of each byte, gathered by one multiply)
__device__ inline uint32_t letter_nibble(uint32_t w) { return ((((w >> 6) & 0x01010101u) * 0x01020408u) >> 24) & 0xfu; }

struct Op { uint32_t op; uint32_t bad; };

// the op whose letter sits at text position p, its digits starting at `start`; lb(pos) reads a text byte (pos >= base - 16)
template <class LB>
__device__ inline Op decode(uint32_t p, uint32_t start, LB lb) {
  Op r{(1u << 2) | OP_M, 0};
  const uint32_t nd = p - start;
  if (nd - 1u > 9u) { r.bad = 1; return r; }   // no digits, or more than ten
  uint32_t val = 0, big = 0;   // 32-bit arithmetic (round 6): a value that is about to leave the 30 bits an op holds is flagged in front of the multiply that could wrap
  for (uint32_t j = 0; j < nd; j++) {
    const uint32_t d = (uint32_t)lb(start + j) - (uint32_t)'0';
    if (d > 9u) r.bad = 1;
    big |= val > 0x3fffffffu / 10u ? 1u : 0u;
    val = val * 10u + d;
  }
  const uint32_t c = lb(p);
  const uint32_t ty = c == 'M' ? OP_M : c == 'I' ? OP_I : c == 'D' ? OP_D : 3u;
  if (ty == 3u || val == 0 || big || val > 0x3fffffffu) { r.bad = 1; return r; }
  if (!r.bad) r.op = (val << 2) | ty;
  return r;
}

__global__ __launch_bounds__(CW * 64) void k_cigar_scan(const uint8_t* __restrict__ txt, const CigIn* __restrict__ in, CigOut* __restrict__ out,
                                                        CigCut* __restrict__ cuts, uint32_t* __restrict__ ops, uint32_t W, uint32_t n_aln) {
  __shared__ uint4 s_txt_all[CW][1 + 64];   // per wave: [0] the 16 bytes in front of the step's text, [1 + l] lane l's bytes
  __shared__ uint32_t s_nc[CW], s_fl[CW];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t ai = blockIdx.x * CW + wv;
  if (ai >= n_aln) return;   // (wave-uniform; no block barrier anywhere below)
  uint4* s_txt = s_txt_all[wv];
  const CigIn a = in[ai];
  const uint8_t* s = txt + a.txt_off;
  uint32_t* o = ops + a.op_off;
  CigCut* cut = cuts + a.cut_off;
  // positions count from the aligned 16 bytes the text starts in: `skip` bytes in front of it belong to whatever lies there in the caller's
  // buffer (the zero-copy path of herro_job_create copies a range of the caller's memory as it is) and are masked out
  const uint32_t skip = a.skip & 15u, len = a.len + skip, room = a.len / 2 + 1;
  if (lane == 0) { s_txt[0] = make_uint4(0, 0, 0, 0); s_nc[wv] = 0; s_fl[wv] = 0; }
  uint32_t k_c = 0, t_c = a.tstart, q_c = 0, i_c = 0, prev1_c = skip;   // carries: ops so far, running totals, position + 1 of the last letter (the first op's digits start at `skip`)
  uint32_t flags = 0;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (lane * CB < len) v = *reinterpret_cast<const uint4*>(s + lane * CB);
  for (uint32_t base = 0; base < len; base += CHUNK) {
    const uint32_t my = base + lane * CB;
    // the next step's bytes are requested before this step's are looked at (round 6: a text of the bench's alignments is two steps — the second load used to wait for the first step's
    // two decoding passes and five scans)
    uint4 vn = make_uint4(0, 0, 0, 0);
    if (my + CHUNK < len) vn = *reinterpret_cast<const uint4*>(s + my + CHUNK);
    s_txt[1 + lane] = v;
    wave_lds_sync();
    auto lb = [&](uint32_t pos) -> uint32_t { return reinterpret_cast<const uint8_t*>(s_txt)[16u + pos - base]; };   // pos >= base - 16 (wraps correctly in u32)
    uint32_t mask = letter_nibble(v.x) | letter_nibble(v.y) << 4 | letter_nibble(v.z) << 8 | letter_nibble(v.w) << 12;
    const uint32_t nvalid = my < len ? min(CB, len - my) : 0u;
    mask &= (1u << nvalid) - 1u;
    if (my < skip) mask &= ~((1u << (skip - my)) - 1u);   // (lane 0 of the first step)
    const uint32_t cnt = __popc(mask);
    if (cnt > 8u) flags |= CIG_MALFORMED;   // more than eight letters in sixteen bytes: two of them are neighbours (an op without digits) — the passes below hold eight ops per lane
    const uint32_t last1 = mask ? my + (31u - __clz(mask)) + 1u : 0u;
    // ---- scan 1: position + 1 of the last letter in front of my bytes
    const uint32_t mx = wave_incl_max(last1);
    uint32_t prev1 = prev1_c;
    {
      const uint32_t up = __builtin_amdgcn_update_dpp(0, (int)mx, 0x138, 0xf, 0xf, false);   // wave_shr:1
      if (lane) prev1 = max(prev1, up);
    }
    const uint32_t chunk_last1 = lane63(mx);
    // ---- pass 1: my totals
    uint32_t st = 0, sq = 0, si = 0;
    uint32_t ro[8];   // my ops (16 bytes hold at most eight), decoded once: pass 2 reads them back (it decoded the text a second time until round 6)
    {
      uint32_t m = mask, start = prev1;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        ro[i] = 0;
        if (m) {
          const uint32_t p = my + (uint32_t)__ffs(m) - 1u;
          m &= m - 1;
          const Op r = decode(p, start, lb);
          flags |= r.bad ? CIG_MALFORMED : 0u;
          ro[i] = r.op;
          const uint32_t ty = op_type(r.op), l = op_len(r.op);
          st += ty != OP_I ? l : 0u;
          sq += ty != OP_D ? l : 0u;
          si += ty == OP_I ? l : 0u;
          start = p + 1;
        }
      }
    }
    // ---- scan 2: op index and running totals in front of my first op
    const uint32_t in_n = wave_incl_sum(cnt), in_t = wave_incl_sum(st), in_q = wave_incl_sum(sq), in_i = wave_incl_sum(si);
    uint32_t k = k_c + in_n - cnt, t = t_c + in_t - st, q = q_c + in_q - sq, ins = i_c + in_i - si;
    // ---- pass 2: emit
    if (mask) {
      unsigned long long wnext = ((unsigned long long)(t / W) + 1ull) * W;   // first window boundary above my running target position
      uint32_t prev_i = 0;
      if (prev1 > skip && prev1 + 15u >= base) prev_i = lb(prev1 - 1u) == 'I';   // the op in front of mine (further back than the 16-byte halo: its successor has 16+ digits and is malformed anyway)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if ((uint32_t)i >= cnt) break;
        struct { uint32_t op; } r{ro[i]};
        const uint32_t ty = op_type(r.op), l = op_len(r.op);
        if (k < room) o[k] = r.op; else flags |= CIG_MALFORMED;
        const uint32_t is_i = ty == OP_I;
        if (is_i & prev_i) flags |= CIG_INS_PAIR;
        const uint32_t tnew = t + (is_i ? 0u : l);
        if (!is_i && (unsigned long long)tnew >= wnext) {
          const uint32_t slot = atomicAdd(&s_nc[wv], 1u);
          if (slot < a.cut_cap) {
            CigCut c{k, t, q, ins, r.op, 0, 0, 0};
            cut[slot] = c;
          } else {
            flags |= CIG_CUT_OVERFLOW;
          }
          wnext = ((unsigned long long)(tnew / W) + 1ull) * W;
        }
        t = tnew;
        q += ty != OP_D ? l : 0u;
        ins += is_i ? l : 0u;
        prev_i = is_i;
        k++;
      }
    }
    k_c += lane63(in_n); t_c += lane63(in_t); q_c += lane63(in_q); i_c += lane63(in_i);
    prev1_c = max(prev1_c, chunk_last1);
    // every read of this step's text is issued (LDS operations of a wave execute in order): the next step's halo goes in
    if (lane == 63) s_txt[0] = v;
    v = vn;
  }
  if (len && prev1_c != len) flags |= CIG_MALFORMED;   // text ends inside an op
  if (flags) atomicOr(&s_fl[wv], flags);
  // The lanes read back ops and cuts their neighbours wrote to global memory.  Release by the writers, acquire by the readers, at workgroup scope
  // (an agent-scope fence writes the XCD's whole L2 back on gfx950, once per wave — measured 8.5 ms per 4096 alignments instead of 0.1).
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // ---- the two ops behind every cut, first / last op, totals
  const uint32_t n_ops = min(k_c, room);
  auto ld = [&](uint32_t idx) { return __hip_atomic_load(o + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); };
  const uint32_t n_cut_all = __hip_atomic_load(&s_nc[wv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const uint32_t n_cut = min(n_cut_all, a.cut_cap);
  for (uint32_t c = lane; c < n_cut; c += 64) {
    const uint32_t kk = __hip_atomic_load(&cut[c].k, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    cut[c].o1 = kk + 1 < n_ops ? ld(kk + 1) : 0u;
    cut[c].o2 = kk + 2 < n_ops ? ld(kk + 2) : 0u;
  }
  if (lane == 0) {
    CigOut r;
    r.n_ops = n_ops; r.t_end = t_c; r.q_end = q_c; r.ins_end = i_c;
    r.n_cuts = n_cut_all; r.flags = __hip_atomic_load(&s_fl[wv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    r.op0 = n_ops ? ld(0) : 0u;
    r.opn = n_ops ? ld(n_ops - 1) : 0u;
    out[ai] = r;
  }
}

}  // namespace

void launch_cigar_scan(const uint8_t* d_txt, const CigIn* d_in, CigOut* d_out, CigCut* d_cuts, uint32_t* d_ops, uint32_t n_aln,
                       uint32_t W, hipStream_t st) {
  if (n_aln == 0) return;
  hipLaunchKernelGGL(k_cigar_scan, dim3((n_aln + CW - 1) / CW), dim3(CW * 64), 0, st, d_txt, d_in, d_out, d_cuts, d_ops, W, n_aln);
}

}  // namespace herro
