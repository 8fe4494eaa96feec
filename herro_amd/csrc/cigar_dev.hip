// cigar_dev.hip — CIGAR text -> binary ops + cut records on the GPU, one workgroup per alignment.
//
// herro_job_create's host cost was ~95 % text decoding (5 000 ops per 4096-bp window of 32 overlaps, ~4 ns each on the
// host: 20 us of CPU per window, which on a host with few usable cores caps the end-to-end rate at half the device rate).
// The text is byte work with a prefix-sum structure, i.e. GPU work: the host stages the bytes (one memcpy, one H2D), this
// kernel writes the ops where the pileup kernels will read them and returns only what the windowing needs — per
// alignment the totals and the few ops that reach a window boundary ("cuts", windowing.hpp) — and the host cuts windows
// from those records without ever seeing an op (window_cuts).
//
// Per alignment, 256 threads sweep the text in chunks of 4096 bytes (16 per thread, one 16-byte load each):
//   letters    bit 6 of a byte separates 'A'..'Z' from '0'..'9': a 16-bit letter mask per thread
//   scan 1     exclusive max over "position of my last letter": where the digits of my first op start
//   pass 1     decode my ops (<= 8) from LDS: per-thread op count and target / query / insertion totals
//   scan 2     exclusive sums of those four
//   pass 2     decode again, now with op index and running totals known: write the op, test for a window boundary
// Malformed text (what CigarIter panics on, aligners.rs:252-293) only raises a flag; the host re-reads that one text
// for the message.  Reference: extract_windows walks the same ops one by one on a feature thread (windowing.rs:44-273).
#include "cigar_dev.h"

#include "pileup_core.h"

namespace herro {
namespace {

constexpr uint32_t CT = 256;          // threads per alignment
constexpr uint32_t CB = 16;           // text bytes per thread and chunk
constexpr uint32_t CHUNK = CT * CB;

__device__ inline uint32_t wave_incl_sum(uint32_t v) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint32_t u = __shfl_up(v, d);
    if (lane >= d) v += u;
  }
  return v;
}
__device__ inline uint32_t wave_incl_max(uint32_t v) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint32_t u = __shfl_up(v, d);
    if (lane >= d) v = max(v, u);
  }
  return v;
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
  unsigned long long val = 0;
  for (uint32_t j = 0; j < nd; j++) {
    const uint32_t d = (uint32_t)lb(start + j) - (uint32_t)'0';
    if (d > 9u) r.bad = 1;
    val = val * 10 + d;
  }
  const uint32_t c = lb(p);
  const uint32_t ty = c == 'M' ? OP_M : c == 'I' ? OP_I : c == 'D' ? OP_D : 3u;
  if (ty == 3u || val == 0 || val > 0x3fffffffull) { r.bad = 1; return r; }
  if (!r.bad) r.op = ((uint32_t)val << 2) | ty;
  return r;
}

__global__ __launch_bounds__(CT) void k_cigar_scan(const uint8_t* __restrict__ txt, const CigIn* __restrict__ in, CigOut* __restrict__ out,
                                                   CigCut* __restrict__ cuts, uint32_t* __restrict__ ops, uint32_t W) {
  __shared__ uint4 s_txt[1 + CT];      // [0]: the 16 bytes in front of the chunk; [1 + t]: thread t's bytes
  __shared__ uint32_t s_w[5][4];       // wave totals of the block scans
  __shared__ uint32_t s_ncut, s_flags;
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  const CigIn a = in[blockIdx.x];
  const uint8_t* s = txt + a.txt_off;
  uint32_t* o = ops + a.op_off;
  CigCut* cut = cuts + a.cut_off;
  // positions count from the aligned 16 bytes the text starts in: `skip` bytes in front of it belong to whatever lies there in the caller's
  // buffer (the zero-copy path of herro_job_create copies a range of the caller's memory as it is) and are masked out
  const uint32_t skip = a.skip & 15u, len = a.len + skip, room = a.len / 2 + 1;
  if (tid == 0) { s_txt[0] = make_uint4(0, 0, 0, 0); s_ncut = 0; s_flags = 0; }
  uint32_t k_c = 0, t_c = a.tstart, q_c = 0, i_c = 0, prev1_c = skip;   // carries: ops so far, running totals, position + 1 of the last letter (the first op's digits start at `skip`)
  uint32_t flags = 0;
  for (uint32_t base = 0; base < len; base += CHUNK) {
    const uint32_t my = base + tid * CB;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (my < len) v = *reinterpret_cast<const uint4*>(s + my);
    s_txt[1 + tid] = v;
    __syncthreads();
    auto lb = [&](uint32_t pos) -> uint32_t { return reinterpret_cast<const uint8_t*>(s_txt)[16u + pos - base]; };   // pos >= base - 16 (wraps correctly in u32)
    uint32_t mask = letter_nibble(v.x) | letter_nibble(v.y) << 4 | letter_nibble(v.z) << 8 | letter_nibble(v.w) << 12;
    const uint32_t nvalid = my < len ? min(CB, len - my) : 0u;
    mask &= (1u << nvalid) - 1u;
    if (my < skip) mask &= ~((1u << (skip - my)) - 1u);   // (thread 0 of the first chunk)
    const uint32_t cnt = __popc(mask);
    const uint32_t last1 = mask ? my + (31u - __clz(mask)) + 1u : 0u;
    // ---- scan 1: position + 1 of the last letter in front of my bytes
    uint32_t mx = wave_incl_max(last1);
    if (lane == 63) s_w[4][wave] = mx;
    __syncthreads();
    uint32_t prev1 = prev1_c;
    for (uint32_t w = 0; w < wave; w++) prev1 = max(prev1, s_w[4][w]);
    {
      const uint32_t up = __shfl_up(mx, 1);
      if (lane) prev1 = max(prev1, up);
    }
    const uint32_t chunk_last1 = max(max(s_w[4][0], s_w[4][1]), max(s_w[4][2], s_w[4][3]));
    // ---- pass 1: my totals
    uint32_t st = 0, sq = 0, si = 0;
    {
      uint32_t m = mask, start = prev1;
      while (m) {
        const uint32_t p = my + (uint32_t)__ffs(m) - 1u;
        m &= m - 1;
        const Op r = decode(p, start, lb);
        flags |= r.bad ? CIG_MALFORMED : 0u;
        const uint32_t ty = op_type(r.op), l = op_len(r.op);
        st += ty != OP_I ? l : 0u;
        sq += ty != OP_D ? l : 0u;
        si += ty == OP_I ? l : 0u;
        start = p + 1;
      }
    }
    // ---- scan 2: op index and running totals in front of my first op
    const uint32_t in_n = wave_incl_sum(cnt), in_t = wave_incl_sum(st), in_q = wave_incl_sum(sq), in_i = wave_incl_sum(si);
    if (lane == 63) { s_w[0][wave] = in_n; s_w[1][wave] = in_t; s_w[2][wave] = in_q; s_w[3][wave] = in_i; }
    __syncthreads();
    uint32_t k = k_c + in_n - cnt, t = t_c + in_t - st, q = q_c + in_q - sq, ins = i_c + in_i - si;
    uint32_t tot_n = 0, tot_t = 0, tot_q = 0, tot_i = 0;
    for (uint32_t w = 0; w < 4; w++) {
      if (w < wave) { k += s_w[0][w]; t += s_w[1][w]; q += s_w[2][w]; ins += s_w[3][w]; }
      tot_n += s_w[0][w]; tot_t += s_w[1][w]; tot_q += s_w[2][w]; tot_i += s_w[3][w];
    }
    // ---- pass 2: emit
    if (mask) {
      unsigned long long wnext = ((unsigned long long)(t / W) + 1ull) * W;   // first window boundary above my running target position
      uint32_t m = mask, start = prev1;
      uint32_t prev_i = 0;
      if (prev1 > skip && prev1 + 15u >= base) prev_i = lb(prev1 - 1u) == 'I';   // the op in front of mine (further back than the 16-byte halo: its successor has 16+ digits and is malformed anyway)
      while (m) {
        const uint32_t p = my + (uint32_t)__ffs(m) - 1u;
        m &= m - 1;
        const Op r = decode(p, start, lb);
        const uint32_t ty = op_type(r.op), l = op_len(r.op);
        if (k < room) o[k] = r.op; else flags |= CIG_MALFORMED;
        const uint32_t is_i = ty == OP_I;
        if (is_i & prev_i) flags |= CIG_INS_PAIR;
        const uint32_t tnew = t + (is_i ? 0u : l);
        if (!is_i && (unsigned long long)tnew >= wnext) {
          const uint32_t slot = atomicAdd(&s_ncut, 1u);
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
        start = p + 1;
      }
    }
    k_c += tot_n; t_c += tot_t; q_c += tot_q; i_c += tot_i;
    prev1_c = max(prev1_c, chunk_last1);
    __syncthreads();   // every read of this chunk's text and of the scan totals is done
    if (tid == CT - 1) s_txt[0] = v;
  }
  if (len && prev1_c != len) flags |= CIG_MALFORMED;   // text ends inside an op
  if (flags) atomicOr(&s_flags, flags);
  // Workgroup scope is all that is needed (the readers below are waves of this workgroup, behind the same L1), and all
  // that is affordable: an agent-scope fence writes the XCD's whole L2 back on gfx950, once per wave — measured 8.5 ms
  // per 4096 alignments instead of 0.1.
  // Release by the writers, acquire by the readers, both at workgroup scope: what the memory model asks for whatever the
  // workgroup's placement (threadgroup-split mode included); on gfx950 in the default mode neither costs an instruction
  // beyond the barrier's own wait.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // ---- the two ops behind every cut, first / last op, totals
  const uint32_t n_ops = min(k_c, room);
  auto ld = [&](uint32_t idx) { return __hip_atomic_load(o + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); };
  const uint32_t n_cut = min(s_ncut, a.cut_cap);
  for (uint32_t c = tid; c < n_cut; c += CT) {
    const uint32_t kk = __hip_atomic_load(&cut[c].k, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    cut[c].o1 = kk + 1 < n_ops ? ld(kk + 1) : 0u;
    cut[c].o2 = kk + 2 < n_ops ? ld(kk + 2) : 0u;
  }
  if (tid == 0) {
    CigOut r;
    r.n_ops = n_ops; r.t_end = t_c; r.q_end = q_c; r.ins_end = i_c;
    r.n_cuts = s_ncut; r.flags = s_flags;
    r.op0 = n_ops ? ld(0) : 0u;
    r.opn = n_ops ? ld(n_ops - 1) : 0u;
    out[blockIdx.x] = r;
  }
}

}  // namespace

void launch_cigar_scan(const uint8_t* d_txt, const CigIn* d_in, CigOut* d_out, CigCut* d_cuts, uint32_t* d_ops, uint32_t n_aln,
                       uint32_t W, hipStream_t st) {
  if (n_aln == 0) return;
  hipLaunchKernelGGL(k_cigar_scan, dim3(n_aln), dim3(CT), 0, st, d_txt, d_in, d_out, d_cuts, d_ops, W);
}

}  // namespace herro
