// model_h.hip — the correction-model forward with f16 MFMA operands (precision modes 4 and 5).
//
// Why a second operand format.  The logits contract is |error| <= 1e-3 (BASELINE.json north_star).  Plain bf16
// operands miss it (6e-3); the bf16 hi/lo split of model.hip meets it with 3 MFMAs per product (1e-5).  f16 keeps
// 11 mantissa bits per operand, and the error budget is not spent evenly over the model (tools/precision_study.py,
// CPU emulation of this dataflow, max |logit error| on 481 tokens, every other GEMM in bf16x3):
//     conv2 7.8e-5, FC 1.0e-4, Q.K^T 4.0e-5, P.V 1.8e-4  with single f16 operands (1 MFMA per product)
//     QKV 2.0e-4, proj 2.2e-4, FF1 1.8e-4, FF2 1.8e-4    with the activation split in two f16 terms (2 MFMAs)
//     heads 5.7e-4 single -> kept at three terms (a 256 x 16 GEMM: nothing to gain)
// so precision 4 runs conv2 / FC / attention on single f16 operands (v_mfma_f32_16x16x32_f16), the heads on three terms, and
// of the four big GEMMs of an encoder layer proj / FF1 / FF2 on `activation hi + lo` x `weight` (2 MFMAs).  QKV runs on the hi
// term alone: Q, K and V are rounded to single f16 fragments for the attention anyway, and the whole-model error does not see
// its lo term (same emulation, 1020 tokens, three weight seeds, max | rms of the logit error:
//     all four split 3.9e-4 | 1.02e-4, 5.6e-4 | 1.57e-4, 4.2e-4 | 0.93e-4      QKV single 4.1e-4 | 1.05e-4, 5.3e-4 | 1.58e-4, 3.3e-4 | 0.95e-4
//     proj single 5.3e-4, 7.0e-4, 4.2e-4    FF1 + FF2 single 6.2e-4, 7.0e-4, 5.5e-4    everything single 6.1e-4, 8.4e-4, 5.7e-4)
// — an eighth of the stack's MFMAs and the LayerNorm-1 lo plane for nothing.  Precision 5 drops the activation lo term
// everywhere (1 MFMA everywhere but the heads); it is measured and reported, not the default.  Accumulation is f32 in every mode.
//
// Kernels (same dataflow as model.hip, re-derived for one 2-byte plane per operand):
//   k_conv_m   embedding + quality + conv1 as a K = 96 GEMM -> (registers) -> conv2 (K = 192) -> y2 as ONE f16 plane [N*31][128]
//   k_fc_r     y2[N][3968] . Wfc -> x[N][256]; (64 .. 128) x 256 tiles, weights L2 -> registers, activations through the LDS, 2 workgroups per CU
//   k_layers_p the whole encoder stack per tile of <= 64 (or 32) tokens: residual stream in registers from the FC output to
//              the logits (positional encoding added on the way in), weight fragments one half GEMM call ahead;
//              <., 4, true>: the same grid headed by SIBLING tiles — a window of 65 .. 512 informative rows on ceil(rows / 64) tiles
//              that exchange the K / V fragments of their heads through L2 / HBM layer by layer (softmax accumulated block by block)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "model_dev.h"

namespace herro {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float norm_qual_h(uint32_t q) {  // inference.rs:16-21,153 — f32, two roundings
  const float QS = (float)(2.0 / 93.0);
  const float QO = (float)(2.0 * 33.0 / 93.0 + 1.0);
  return __fsub_rn(__fmul_rn(QS, (float)q), QO);
}

// (a, b) -> packed f16 pair, round to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
  const float2v v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, half2v));
}
// (a, b) -> hi, lo packed f16 pairs with a ~= hi.x + lo.x (22 mantissa bits)
__device__ __forceinline__ void split_h2(float a, float b, uint32_t& hi, uint32_t& lo) {
  const float2v v = {a, b};
  const half2v h = __builtin_convertvector(v, half2v);
  hi = __builtin_bit_cast(uint32_t, h);
  const float2v hf = __builtin_convertvector(h, float2v);
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - hf, half2v));
}
__device__ __forceinline__ half8 as_half8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  return __builtin_bit_cast(half8, make_uint4(a, b, c, d));
}
__device__ __forceinline__ half8 pack_h8(const float (&v)[8]) {
  return as_half8(pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7]));
}
__device__ __forceinline__ void split_h8(const float (&v)[8], half8& hi, half8& lo) {
  uint4 h, l;
  split_h2(v[0], v[1], h.x, l.x);
  split_h2(v[2], v[3], h.y, l.y);
  split_h2(v[4], v[5], h.z, l.z);
  split_h2(v[6], v[7], h.w, l.w);
  hi = __builtin_bit_cast(half8, h);
  lo = __builtin_bit_cast(half8, l);
}
// max(x, 0) on eight packed halves, then a lane mask
__device__ __forceinline__ half8 relu_mask_h8(half8 x, uint32_t mask) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  uint4 u = __builtin_bit_cast(uint4, x);
  const h2 z = {(_Float16)0, (_Float16)0};
  auto mx = [&](uint32_t v) -> uint32_t { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(h2, v), z)) & mask; };
  return __builtin_bit_cast(half8, make_uint4(mx(u.x), mx(u.y), mx(u.z), mx(u.w)));
}
// v + (v of lane ^ 16) + (v of lane ^ 32) + (v of lane ^ 48) in every lane, on the vector unit: v_permlane16_swap / v_permlane32_swap
// (gfx950) exchange rows of 16 / halves of 32 lanes between two registers; with both holding v, their sum is v + shfl_xor(v, 16 / 32).
// (__shfl_xor compiles to ds_bpermute_b32: a round trip through the LDS crossbar per step, 48 of them per layer in the LayerNorms and
// the softmax.  The clang builtin of this LLVM returns the same register for both results, hence the asm; tools/probes/permlane_check.hip.)
__device__ __forceinline__ float fg_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  a += b; b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
// max(a, b) for values that are never NaN, in ONE instruction: fmaxf compiles to a canonicalising v_max of each operand in front of the v_max (three per maximum:
// 48 instead of 16 per query block in the softmax); the median of (a, b, +inf) is the same number and has no such prefix
__device__ __forceinline__ float fmax_nc(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_huge_valf()); }
__device__ __forceinline__ float fmax_asm(float a, float b) {   // ... and where the operands come out of inline asm (the compiler canonicalises those too)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fg_max(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  a = fmax_asm(a, b); b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmax_asm(a, b);
}
__device__ __forceinline__ f32x4 mma(half8 a, half8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// ---- precision 6: the activation remainder (the `lo` term of precision 4) as OCP e4m3 bytes, multiplied with an e4m3 copy of the weight by
// v_mfma_scale_f32_16x16x128_f8f6f4 — K = 128 per instruction at twice the f16 rate per product (measured, tools/probes/mx_probe.hip: 32 cycles per
// instruction against 4 x 18 for the same K in f16).  The remainder is <= 2^-11 of the activation, so its 4 significant bits cost 2^-15
// relative; it is scaled by 2^13 into the e4m3 range (clamped at 448: v_cvt_pk_fp8_f32 turns larger values into NaN) and the weight by a
// power of two chosen per matrix at load (Weight::s8); both scales go back out through the instruction's E8M0 scale operands.  Operand
// layout (checked bit-exact against a host product by the probe): lane l holds row / column l % 16 and the 32 k-values 32 (l / 16) + j in
// byte order — the same for A and B.
typedef int v8i __attribute__((ext_vector_type(8)));
constexpr int LO8_SHIFT = 13;
constexpr int LO8_SCALE_B = (127 - LO8_SHIFT) * 0x01010101;
__device__ __forceinline__ f32x4 mma8(v8i a, v8i b, f32x4 c, int scale_a) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, LO8_SCALE_B);
}
// eight values -> f16 (round to nearest even) + their remainders as eight e4m3 bytes
__device__ __forceinline__ void split_h8_f8(const float (&v)[8], half8& hi, uint2& lo8) {
  uint32_t h[4];
  float2v r[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float2v x = {v[2 * i], v[2 * i + 1]};
    const half2v hh = __builtin_convertvector(x, half2v);
    h[i] = __builtin_bit_cast(uint32_t, hh);
    const float2v d = (x - __builtin_convertvector(hh, float2v)) * (float)(1 << LO8_SHIFT);
    r[i] = float2v{__builtin_amdgcn_fmed3f(d.x, -448.f, 448.f), __builtin_amdgcn_fmed3f(d.y, -448.f, 448.f)};
  }
  int w[2];
  w[0] = __builtin_amdgcn_cvt_pk_fp8_f32(r[0].x, r[0].y, 0, false);
  w[0] = __builtin_amdgcn_cvt_pk_fp8_f32(r[1].x, r[1].y, w[0], true);
  w[1] = __builtin_amdgcn_cvt_pk_fp8_f32(r[2].x, r[2].y, 0, false);
  w[1] = __builtin_amdgcn_cvt_pk_fp8_f32(r[3].x, r[3].y, w[1], true);
  hi = as_half8(h[0], h[1], h[2], h[3]);
  lo8 = make_uint2((uint32_t)w[0], (uint32_t)w[1]);
}

__device__ __forceinline__ void glds16_h(const void* gsrc, uint32_t lds_dst) {  // 16 B per lane, global -> LDS, no VGPR staging
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_h() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// dynamic-LDS opt-in is a per-device function attribute: one context per GPU in one process must set it on each
void opt_in_lds(const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(mu);
  if (done.insert({fn, dev}).second) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

constexpr int HC2 = 128;

// ---------------------------------------------------------------------------------------------------
// k_conv_m — conv1 AND conv2 on the MFMA pipe, chained through registers (round 4).
// Rounds 2-3 (k_conv_h, deleted) generated conv1 on the vector unit: 1400 VALU instructions per thread and tile of 128 pairs
// for 96 MFMAs (MFMA pipe 14 % busy, 282 us per 4096 windows; this kernel 198).  But conv1 of a (token, read row) pair is a
// tiny GEMM: y1[j][c] = relu(b1[c] + sum_tap T1[tap][tok(j + tap)][c] + wq[tap][c] q(j + tap)) = W1g . a, with a = per tap a
// one-hot of the cell's token (13 values: 12 tokens + "outside"), the normalised quality and a constant 1 for the bias.  To
// keep conv1 as exact as the f32 VALU version (the table was not rounded to f16 anywhere), table, quality weight and bias
// enter as f16 hi + lo in separate K slots — the one-hot is exact, so hi and lo rows simply add up — and the quality as
// hi + lo too: 32 slots per tap, K = 96, 3 k-steps:
//     slots 0..12 one-hot -> T1 hi | 13 q_hi -> wq hi | 14 q_lo -> wq hi | 15 one -> b1 hi (tap 0) | 16..28 one-hot -> T1 lo |
//     29 q_hi -> wq lo | 30 unused | 31 one -> b1 lo (tap 0)                       (Weight M.conv1g, built at load)
// The MFMA's result layout (channel rows, pair columns; two row-interleaved MFMAs give a lane 8 consecutive channels of its
// pair) IS the B-operand layout of conv2's k-step, so y1 never leaves the registers: a wave takes a block of 16 pairs through
// conv1 (36 MFMAs) and conv2 (48, W2 fragments from LDS) and stores y2 as 16-byte pieces.  No activation in LDS, no barrier in
// the loop; the waves of a workgroup only share the 48 KB of conv2 fragments and three small tables (one-hot fragments by
// token, normalised quality by byte).  The five receptive-field rows of a token are resolved once per token (TokCv, written by
// k_build_tokens_h) and fetched two steps (record) / one step (cells) ahead.  What bounds it (r4 counters, timing variants with
// cache-hot cells / one fragment, profiles/r4_conv_*): instruction issue — 84 MFMAs against ~340 other instructions per block
// (a 16-cycle MFMA hides three) — and, for a fifth of the time, the latency of the scattered cell loads; NOT the LDS (28 % busy).
// A spill of even a few registers triples the time (scratch behind every load): check ScratchSize after any change.
// ---------------------------------------------------------------------------------------------------
constexpr int CM_NT = 256;
#ifndef HERRO_CONV_DBG
#define HERRO_CONV_DBG 0   // timing experiments (wrong results): 1 every wave fetches the same cells, 2 one conv2 fragment for all
#endif
constexpr size_t CONV_M_SHM = (size_t)48 * 1024 + 128 * 4 + 32 * 16 + 256 * 4;

template <int NB, int WPS, bool RF>   // RF: the cells come as 16-byte receptive-field records (B.rf_q); else from the token / quality planes
__global__ __launch_bounds__(CM_NT, WPS) void k_conv_m(ModelDev M, BatchDev B, ModelScratch S, uint32_t n_rows, uint32_t n_units, uint32_t unit0) {   // units [unit0, n_units) of 16 pairs; n_rows: pairs of the launch group
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* s_w2 = reinterpret_cast<uint4*>(smem);                     // conv2 weights in fragment order: fragment f, lane l at [f * 64 + l]
  float* s_b2 = reinterpret_cast<float*>(smem + 48 * 1024);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t fr = lane & 15, fg = lane >> 4;
  {
    const uint4* src = reinterpret_cast<const uint4*>(M.conv2.ph16);
    for (uint32_t i = tid; i < 48 * 64; i += CM_NT) s_w2[i] = src[i];
    for (uint32_t i = tid; i < 128; i += CM_NT) s_b2[i] = M.conv2.bias[i];
  }
  half8 w1[3][2][2];   // [tap][slab][jt]
  {
    const uint16_t* p = M.conv1g.ph16 + (uint64_t)lane * 8;
#pragma unroll
    for (int sl = 0; sl < 2; sl++)
#pragma unroll
      for (int jt = 0; jt < 2; jt++)
#pragma unroll
        for (int tp = 0; tp < 3; tp++) w1[tp][sl][jt] = *reinterpret_cast<const half8*>(p + (uint64_t)(((sl * 2 + jt) * 3 + tp) * 64) * 8);
  }
  constexpr bool rfq = RF;
  // The receptive field of pair m: rows tok_row - 2 .. + 2 of read row (m % 31) of the token's window; outside [0, lmax): token 12
  // (zero table row) and no quality; inside the batch padding [len, lmax): pad token and quality 126 (inference.rs:86-97).
  // Two dependent loads (the token's record, then its cells) run two and one unit ahead of the arithmetic.
  struct Raw { uint32_t t[5], q[5]; uint4 rf; };
  struct Cells { uint32_t tok[2], q[2], ok; };   // five token bytes, five quality bytes (0xff: none), validity of the three conv1 positions
  auto load_meta = [&](uint32_t m) -> TokCv {
#if HERRO_CONV_DBG & 1
    m = lane;   // timing experiment: every wave fetches the same few cells (cache hits)
#endif
    return S.tok_cv[min(m, n_rows - 1u) / HERRO_ROWS];
  };
  auto load_raw = [&](const TokCv& tm, uint32_t m) -> Raw {
    Raw r;
#if HERRO_CONV_DBG & 1
    m = lane;
#endif
    const uint32_t rr = min(m, n_rows - 1u) % HERRO_ROWS;
    if constexpr (RF) {   // one 16-byte record: the pair's five tokens and five qualities (k_rfq) — no token plane exists on the lean path
      r.rf = *reinterpret_cast<const uint4*>(B.rf_q + ((uint64_t)tm.rf_idx * HERRO_ROWS + rr) * 16);
#pragma unroll
      for (int i = 0; i < 5; i++) { r.t[i] = 0; r.q[i] = 0; }
      return r;
    } else {
    const uint32_t ld = tm.ld_d1 & 0xffffu, trow = tm.row_ok & 0xffffu;
    const uint8_t* pb = B.planes_b + tm.plane_off + (uint64_t)rr * ld;
    r.rf = make_uint4(0, 0, 0, 0);
    const uint8_t* pq = B.planes_q + tm.plane_off + (uint64_t)rr * ld;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const uint32_t row = i < 2 ? (uint32_t)max((int32_t)trow - 2 + i, 0) : trow - 2 + i;   // a row behind the plane's last is read (valid memory) and masked out
      r.t[i] = pb[row];
      r.q[i] = (uint32_t)pq[row];
    }
    return r;
    }
  };
  auto finish = [&](const Raw& r, const TokCv& tm) -> Cells {
    Cells c;
    const uint32_t t0 = rfq ? r.rf.x : (r.t[0] | (r.t[1] << 8) | (r.t[2] << 16) | (r.t[3] << 24)), t1 = rfq ? r.rf.y : r.t[4];
    const uint32_t q0 = rfq ? r.rf.z : (r.q[0] | (r.q[1] << 8) | (r.q[2] << 16) | (r.q[3] << 24)), q1 = rfq ? r.rf.w : r.q[4];
    const uint32_t mk1 = tm.row_ok >> 24;
    c.tok[0] = (t0 & tm.mk0) | tm.dt0; c.tok[1] = (t1 & mk1) | ((tm.ld_d1 >> 16) & 0xffu);
    c.q[0] = (q0 & tm.mk0) | tm.dq0; c.q[1] = (q1 & mk1) | (tm.ld_d1 >> 24);
    c.ok = (tm.row_ok >> 16) & 7u;
    return c;
  };
  // one-hot operand fragments by (token, lane parity): lanes fg 0 / 2 hold slots 0..7, lanes fg 1 / 3 slots 8..15 (+ one in slot 15)
  uint4* s_oh = reinterpret_cast<uint4*>(s_b2 + 128);
  uint32_t* s_ql = reinterpret_cast<uint32_t*>(s_oh + 32);   // quality byte -> normalised quality as f16 hi | lo << 16 (255: no cell -> 0)
  {
    uint32_t qh = 0, ql = 0;
    if (tid < 255) split_h2(norm_qual_h(tid), 0.f, qh, ql);
    s_ql[tid] = (qh & 0xffffu) | (ql << 16);
  }
  if (tid < 32) {
    const uint32_t tok = tid >> 1, od = tid & 1u, t = tok - 8u * od;
    uint32_t r[4] = {0, 0, 0, od ? 0x3c000000u : 0u};
    if (tok < 13 && (od ? t < 5u : t < 8u)) r[t >> 1] |= 0x3c00u << ((t & 1u) * 16u);
    s_oh[tid] = make_uint4(r[0], r[1], r[2], r[3]);
  }
  __syncthreads();

  const uint32_t odd = fg & 1u, mq_hi = odd ? 0xffffu : 0u, mq_lo = fg == 1u ? 0xffffu : 0u;
  const uint32_t stride = gridDim.x * (CM_NT / 64);
  uint32_t unit = unit0 + blockIdx.x * (CM_NT / 64) + wave;
  TokCv mt1[NB], mt2[NB];   // records of the pairs of unit + stride, unit + 2 stride
  Cells cur[NB];
  Raw raw[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const TokCv m0 = load_meta(unit * (16 * NB) + b * 16 + fr);
    cur[b] = finish(load_raw(m0, unit * (16 * NB) + b * 16 + fr), m0);
    mt1[b] = load_meta((unit + stride) * (16 * NB) + b * 16 + fr);
  }
  for (; unit < n_units; unit += stride) {
#pragma unroll
    for (int b = 0; b < NB; b++) {
      mt2[b] = load_meta((unit + 2 * stride) * (16 * NB) + b * 16 + fr);
      raw[b] = load_raw(mt1[b], (unit + stride) * (16 * NB) + b * 16 + fr);
    }
    half8 y1f[NB][6];   // [block][conv2 k-step = position * 2 + slab]
#pragma unroll
    for (int b = 0; b < NB; b++) {
      // operand fragments of the five receptive-field rows: lanes fg 0 / 2 slots 0..7 (one-hot of tokens 0..7), lanes fg 1 / 3
      // slots 8..15 (tokens 8..12, q_hi, q_lo | 0, one)
      half8 rf[5];
#pragma unroll
      for (int i = 0; i < 5; i++) {
        const uint32_t tok = ((i < 4 ? cur[b].tok[0] >> (8 * i) : cur[b].tok[1]) & 0xffu);
        const uint32_t qb = ((i < 4 ? cur[b].q[0] >> (8 * i) : cur[b].q[1]) & 0xffu);
        uint4 r = s_oh[tok * 2u + odd];
        const uint32_t qhl = s_ql[qb];
        r.z |= (qhl & mq_hi) << 16;                               // slot 13 (29): q_hi
        r.w |= (qhl >> 16) & mq_lo;                               // slot 14: q_lo (30: unused)
        rf[i] = as_half8(r.x, r.y, r.z, r.w);
      }
#pragma unroll
      for (int j = 0; j < 3; j++) {
        f32x4 a[2][2];
#pragma unroll
        for (int sl = 0; sl < 2; sl++)
#pragma unroll
          for (int jt = 0; jt < 2; jt++) a[sl][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < 3; tp++)
#pragma unroll
          for (int sl = 0; sl < 2; sl++)
#pragma unroll
            for (int jt = 0; jt < 2; jt++) a[sl][jt] = mma(w1[tp][sl][jt], rf[j + tp], a[sl][jt]);
        const uint32_t okm = ((cur[b].ok >> j) & 1u) ? 0xffffffffu : 0u;   // conv2 sees zeros beyond the (padded) sequence ends
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; q++) v[q] = a[sl][q >> 2][q & 3];
          y1f[b][j * 2 + sl] = relu_mask_h8(pack_h8(v), okm);   // ReLU after the rounding: the same number
        }
      }
    }
    // conv2: 64 output channels at a time, both blocks against one read of every weight fragment
    uint16_t* y2 = S.y2_hi;
#pragma unroll
    for (int hs = 0; hs < 2; hs++) {
      f32x4 a2[NB][2][2];   // [block][slab][jt]
#pragma unroll
      for (int sl = 0; sl < 2; sl++)
#pragma unroll
        for (int jt = 0; jt < 2; jt++) {
          const float4 bv = *reinterpret_cast<const float4*>(s_b2 + (hs * 2 + sl) * 32 + 8 * fg + 4 * jt);
#pragma unroll
          for (int b = 0; b < NB; b++) a2[b][sl][jt] = f32x4{bv.x, bv.y, bv.z, bv.w};
        }
      half8 wf[2][2], wn[2][2];   // the fragments of one k-step, read one k-step ahead of their MFMAs
      auto rdw = [&](int ks, half8 (&w)[2][2]) {
#pragma unroll
        for (int sl = 0; sl < 2; sl++)
#pragma unroll
          for (int jt = 0; jt < 2; jt++) w[sl][jt] = __builtin_bit_cast(half8, s_w2[(HERRO_CONV_DBG & 2) ? lane : ((((hs * 2 + sl) * 2 + jt) * 6 + ks) * 64) + lane]);
      };
      rdw(0, wf);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 6; ks++) {
        if (ks < 5) rdw(ks + 1, wn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sl = 0; sl < 2; sl++)
#pragma unroll
          for (int jt = 0; jt < 2; jt++)
#pragma unroll
            for (int b = 0; b < NB; b++) a2[b][sl][jt] = mma(wf[sl][jt], y1f[b][ks], a2[b][sl][jt]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sl = 0; sl < 2; sl++)
#pragma unroll
          for (int jt = 0; jt < 2; jt++) wf[sl][jt] = wn[sl][jt];
      }
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const uint32_t m = unit * (16 * NB) + b * 16 + fr;
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; q++) v[q] = a2[b][sl][q >> 2][q & 3];
          if (m < n_rows) *reinterpret_cast<half8*>(y2 + (uint64_t)m * HC2 + (hs * 2 + sl) * 32 + 8 * fg) = relu_mask_h8(pack_h8(v), 0xffffffffu);
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
      cur[b] = finish(raw[b], mt1[b]);
      mt1[b] = mt2[b];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// k_fc_r — FC: x[M][256] = y2[M][K] . W^T + bias, y2 and W single f16 planes; the weights streamed L2 -> registers in fragment order (round 4).
// Its predecessor k_fc_h (rounds 2-3, deleted in round 5: git show 4ba82b6:herro_amd/csrc/model_h.hip) moved both operands through the LDS by LDS-DMA; its
// counters and depth experiments said (r4): a k-step costs ~1.4 us where its MFMAs need 0.25; deeper LDS-DMA prefetch with one workgroup per CU is SLOWER
// (208 / 201 us against 182 with three buffers and two workgroups): the time is the fixed cost of a k-step — three LDS-DMA pieces per wave (100-185
// cycles of issue each beside MFMAs and ds_reads), eight ds_reads, the barrier.  Here the weights (the same for every workgroup, L2-resident) never
// touch the LDS: a wave owns 32 output columns of all 128 rows, its two fragments per k-step come straight from Weight::ph16 into registers one
// macro-step ahead; only the activation tile goes through the LDS, by plain loads (one macro-step ahead, in registers) and ds_write_b128 — one barrier
// per 64 k.  Per 64 k and wave: 32 MFMAs, 16 ds_read_b128, 4 fragment loads, 2 row loads + 2 ds_writes.
// ---------------------------------------------------------------------------------------------------
constexpr int FR_ROWS = 128;                               // rows of an LDS buffer; a tile takes 32 * G of them
constexpr size_t FC_R_SHM = (size_t)2 * FR_ROWS * 64 * 2;   // two buffers [128 rows][64 k] f16

// NB: 16-row blocks per tile, i.e. tiles of 16 * NB rows (4 .. 8: 64 .. 128 rows) — chosen per launch so that the busiest compute unit holds the fewest
// rows (two workgroups fit a unit): 38.7 k rows in 128-row tiles are 303 workgroups, two on 47 of the 256 compute units and one on the others; in 96-row
// tiles 404, and the two-workgroup units carry 192 rows; in 80-row tiles (round 6, last change: the tile height was 32 * G, G = 3 | 4, before) 484 — at most
// two per unit, 160 rows where the even share would be 151.  The row blocks go through the k-step in pairs (one read of the weight fragments' registers per
// pair); an odd NB leaves one block on its own.  The order of the k-steps per output element does not depend on NB: every height gives the same bits.
// pe_tab (round 6): the positional encoding of a token's row is added HERE, from the table of herro_load_model (ModelDev::pe_tab), for rows inside the table — the stack's
// prologue, where every compute unit fetches its tile at the same moment, reads 64 KB per tile instead of 128 (and ran 16 sincosf per lane before the table existed);
// k_layers_p adds the encoding of the rows BEYOND the table itself (the same rule on both sides: row < pe_rows)
template <int NB>
__global__ __launch_bounds__(512, 4) void k_fc_r(const uint16_t* __restrict__ A, uint32_t lda, Weight W, float* C, uint32_t ldc, uint32_t M,
                                                 const uint32_t* __restrict__ tok_row, const float* __restrict__ pe_tab, uint32_t pe_rows) {
  constexpr int FR_TM = 16 * NB, G = (NB + 1) / 2;   // G: pairs of row blocks (the last one single when NB is odd)
  static_assert(NB >= 3 && NB <= 8, "tiles of 48 .. 128 rows");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_fc[];
  uint16_t* s_a = reinterpret_cast<uint16_t*>(smem_fc);   // [2][128][64]: row r at r * 128 B, 16-byte chunk c at (c ^ (r & 7))
  const uint32_t nks = W.K >> 5, nms = W.K >> 6;
  // tiles are taken from the END of the activation: k_conv_m has just written y2 front to back, and what it wrote last is what the
  // memory-side cache (256 MB) still holds — reading in the writer's order would evict the tail before reaching it (r5)
#ifdef HERRO_FWD_ORDER   // (A/B build only)
  const uint32_t m0 = blockIdx.x * FR_TM;
#else
  const uint32_t m0 = (gridDim.x - 1u - blockIdx.x) * FR_TM;
#endif
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t fr = lane & 15, fg = lane >> 4;
  f32x4 acc[NB][2];
#pragma unroll
  for (int pt = 0; pt < NB; pt++)
#pragma unroll
    for (int jt = 0; jt < 2; jt++) acc[pt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // activation rows: thread t stages row (t >> 2), 16-byte chunks (t & 3) and (t & 3) + 4 of every 64-k tile
  const uint32_t srow = tid >> 2, sch = tid & 3;
  const uint32_t lrow = srow < (uint32_t)FR_TM ? srow : srow - (uint32_t)(FR_ROWS - FR_TM);   // below 128 rows: the threads of the rows beyond the tile repeat its last rows (same lines, no new traffic) into LDS rows nobody reads
  const uint16_t* ga = A + (uint64_t)min(m0 + lrow, M - 1) * lda + sch * 8;
  const uint32_t sdst0 = srow * 64 + ((sch ^ (srow & 7u)) << 3), sdst1 = srow * 64 + (((sch + 4) ^ (srow & 7u)) << 3);
  // weights: fragment (jt, ks) of this wave's 32-column slab
  const uint16_t* gw = W.ph16 + ((uint64_t)(wave * 2) * nks * 64 + lane) * 8;
  auto wfrag = [&](uint32_t jt, uint32_t ks) -> half8 { return *reinterpret_cast<const half8*>(gw + (uint64_t)(jt * nks + ks) * 512); };
  // DEEP (tiles of <= 96 rows, whose accumulators leave the registers for it): weight fragments four k-steps and activation tiles two macro-steps ahead of their use —
  // 123-128 -> 120-121 us at the driver's size (profiles/r6_ab_fc_prefetch_depth.txt; -DHERRO_FC_SHALLOW builds the other side of that A/B).  Worth 3 %: load latency is not
  // what the ~1.6 us per macro-step are made of (nor LDS latency, nor the barrier count: DESIGN §5, profiles/r6_probe_mfma_lds_vmem_overlap.txt)
#ifdef HERRO_FC_SHALLOW
  constexpr bool DEEP = false;
#else
  constexpr bool DEEP = NB <= 6;
#endif
  constexpr int WD = DEEP ? 4 : 2;
  half8 wr[WD][2];      // ring: k-step & (WD - 1) (a slot is refilled with the k-step WD ahead as soon as its MFMAs are issued)
  uint4 as0, as1;       // the activation tile of the next macro-step, in flight (DEEP: of an odd macro-step)
  uint4 bs0, bs1;       // DEEP: the tile of an even macro-step, in flight
  auto load_a = [&](uint32_t ms, uint4& r0, uint4& r1) {
    r0 = *reinterpret_cast<const uint4*>(ga + (uint64_t)ms * 64);
    r1 = *reinterpret_cast<const uint4*>(ga + (uint64_t)ms * 64 + 32);
  };
  auto put_a = [&](uint32_t buf, const uint4& r0, const uint4& r1) {
    *reinterpret_cast<uint4*>(s_a + buf * (FR_ROWS * 64) + sdst0) = r0;   // (all 128 rows are staged whatever G: rows beyond the tile are read for nothing)
    *reinterpret_cast<uint4*>(s_a + buf * (FR_ROWS * 64) + sdst1) = r1;
  };
  // a quarter k-step = 2 row blocks x 2 column blocks = 4 MFMAs; the fragments of the next quarter are read under them
  auto rd = [&](uint32_t buf, uint32_t kk, uint32_t g, half8 (&x)[2]) {
    const uint16_t* t = s_a + buf * (FR_ROWS * 64);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if (g * 2 + i >= (uint32_t)NB) continue;   // (compile time: the loops around are unrolled)
      const uint32_t r = (g * 2 + i) * 16 + fr;
      x[i] = *reinterpret_cast<const half8*>(t + r * 64 + (((kk * 4 + fg) ^ (r & 7u)) << 3));
    }
  };
  auto mm = [&](uint32_t g, const half8 (&x)[2], const half8 (&w)[2]) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int jt = 0; jt < 2; jt++) if (g * 2 + i < (uint32_t)NB) acc[g * 2 + i][jt] = mma(w[jt], x[i], acc[g * 2 + i][jt]);
  };
  // prologue: tile 0 into LDS, tile 1 in flight, weights of k-steps 0, 1
  {
    uint4 t0, t1;
    load_a(0, t0, t1);
    load_a(min(1u, nms - 1), as0, as1);
    if constexpr (DEEP) load_a(min(2u, nms - 1), bs0, bs1); else { bs0 = make_uint4(0, 0, 0, 0); bs1 = bs0; }
#pragma unroll
    for (int k = 0; k < WD; k++)
#pragma unroll
      for (int jt = 0; jt < 2; jt++) wr[k][jt] = wfrag(jt, min((uint32_t)k, nks - 1));
    put_a(0, t0, t1);
  }
  for (uint32_t ms = 0; ms < nms; ms += 2) {   // nms is even (K % 128 == 0)
#pragma unroll
    for (int u = 0; u < 2; u++) {   // macro-step ms + u reads buffer u; its successor's tile arrives in as0 / as1
      const uint32_t s_ = ms + u;
      // tile s_ is complete, buffer (u + 1) & 1 is no longer read.  NOT __syncthreads(): its fence waits for vmcnt(0) — every
      // weight fragment and activation row in flight — at each of the 62 barriers, which is the latency the prefetch is there to hide.
      // Everything below is unconditional (indices clamped; the last tile is written once more into the buffer nobody reads): with
      // branches in the loop the compiler's wait-count bookkeeping falls back to vmcnt(0) in front of the first use of a load.
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      half8 x[2], xn[2];
      rd(u, 0, 0, x);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2 * G; q++) {   // q = G kk + g
        if (q < 2 * G - 1) rd(u, (q + 1) / G, (q + 1) % G, xn);
        __builtin_amdgcn_sched_barrier(0);
        const int ws = DEEP ? u * 2 + q / G : q / G;   // ring slot of this k-step (a constant once the loops are unrolled)
        mm(q % G, x, wr[ws]);
        __builtin_amdgcn_sched_barrier(0);
        if (q % G == G - 1) {   // the k-step's fragments have been issued: the slot takes the k-step WD ahead
          const uint32_t kn = min(s_ * 2 + (q / G) + WD, nks - 1);
#pragma unroll
          for (int jt = 0; jt < 2; jt++) wr[ws][jt] = wfrag(jt, kn);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (q == G + 1) {   // the next tile (in flight since the previous macro-step; DEEP: since the one before) goes into the other buffer; its successor is requested
          if (DEEP && u == 1) {   // (compile time) tile s_ + 1 is even
            put_a(0, bs0, bs1);
            __builtin_amdgcn_sched_barrier(0);
            load_a(min(s_ + 3, nms - 1), bs0, bs1);
          } else {
            put_a((u + 1) & 1, as0, as1);
            __builtin_amdgcn_sched_barrier(0);
            load_a(min(s_ + (DEEP ? 3 : 2), nms - 1), as0, as1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) x[i] = xn[i];
      }
    }
  }
  float bs[8];
  {
    const float* bp = W.bias + wave * 32 + 8 * fg;
#pragma unroll
    for (int q = 0; q < 8; q++) bs[q] = W.bias ? bp[q] : 0.f;
  }
  uint32_t prow[NB];
#pragma unroll
  for (int pt = 0; pt < NB; pt++) prow[pt] = pe_tab ? tok_row[min(m0 + pt * 16 + fr, M - 1)] : 0xffffffffu;
#pragma unroll
  for (int pt = 0; pt < NB; pt++) {
    const uint32_t m = m0 + pt * 16 + fr;
    if (m < M) {
      float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0;
      if (prow[pt] < pe_rows) {
        const float* pp = pe_tab + (uint64_t)prow[pt] * ldc + wave * 32 + 8 * fg;
        p0 = *reinterpret_cast<const float4*>(pp);
        p1 = *reinterpret_cast<const float4*>(pp + 4);
      }
      float* cp = C + (uint64_t)m * ldc + wave * 32 + 8 * fg;
      // (bias first, then the encoding: the sum the stack's prologue used to form)
      *reinterpret_cast<float4*>(cp) = make_float4((acc[pt][0][0] + bs[0]) + p0.x, (acc[pt][0][1] + bs[1]) + p0.y, (acc[pt][0][2] + bs[2]) + p0.z, (acc[pt][0][3] + bs[3]) + p0.w);
      *reinterpret_cast<float4*>(cp + 4) = make_float4((acc[pt][1][0] + bs[4]) + p1.x, (acc[pt][1][1] + bs[5]) + p1.y, (acc[pt][1][2] + bs[6]) + p1.z, (acc[pt][1][3] + bs[7]) + p1.w);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The encoder stack in one kernel, f16 operands.  Organisation as k_layers (model.hip): a tile of whole windows
// (<= 64 tokens) per workgroup, 8 waves, wave = head for attention and = 32-channel slab of every GEMM output;
// LayerNorm output / attention output / FF hidden chunk are f16 planes in LDS (hi and, with TERMS = 2, lo), weights
// stream L2 -> registers in MFMA fragment order (Weight::ph16).  Differences that the smaller operands buy:
//   * the residual stream x never leaves the registers (the bf16x3 kernel parked it in HBM around every GEMM phase
//     to make room for 128 VGPRs of weight fragments; here a full-K weight batch is 64);
//   * Q, K, V, P are single f16 fragments (attention's error share is 4e-5 / 1.8e-4);
//   * the positional encoding is added while x is fetched (no k_add_pe launch, no extra pass over x).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hlsw(uint32_t row, uint32_t chunk) { return row * 256 + ((chunk ^ (row & 15u)) << 3); }
// the e4m3 remainder plane [tokens][256 bytes]: byte offset of 16-byte chunk `chunk` (k / 16) of a row, same XOR swizzle
__device__ __forceinline__ uint32_t lsw8(uint32_t row, uint32_t chunk) { return row * 256 + ((chunk ^ (row & 15u)) << 4); }

// ---------------------------------------------------------------------------------------------------
// k_layers_p — the stack with its three exposed latencies taken off the critical path (an un-pipelined first version,
// all 16 fragment loads of a GEMM call issued at its head, took 1196 us per 4096 windows; this one 959; with x parked
// in L2 across the GEMM phases instead of held in registers: 1068):
//   * weights run one half-call AHEAD across GEMM calls: a call enters with the fragments of its first four k-steps
//     already in registers (loaded under the previous call's MFMAs), issues the loads of its last four at once, and
//     under those MFMAs fetches the first four of the NEXT call (the call sequence Q, K, V, proj, [FF1, FF2] x chunks,
//     next layer's Q ... is static) — no L2 round trip at the head of each of the 48 calls per tile, with both waves
//     of every SIMD waiting at the same time;
//   * LDS activation fragments are read one half k-step (8 MFMAs, >= 128 cycles) ahead of their use instead of one
//     MFMA pair ahead;
//   * bias vectors and LayerNorm parameters are requested before the MFMAs / reductions they follow, not after.
// ---------------------------------------------------------------------------------------------------
struct WStream {  // per-lane fragment pointer of one GEMM call: fragment (k, jt) at p + (jt * nks + k) * 512
  const uint16_t* p;
  uint32_t nks;
  const uint8_t* p8;   // precision 6: e4m3 fragment (s = k / 128, jt, half) at p8 + ((jt * (nks / 4) + s) * 2 + half) * 1024 (Weight::p8)
  int s8;              // its E8M0 scale in all four bytes
};
template <bool F8 = false>
__device__ __forceinline__ WStream wstream(const Weight& W, uint32_t cb, uint32_t kofs, uint32_t lane) {
  const uint32_t nks = W.K >> 5;
  WStream s{W.ph16 + ((uint64_t)((cb >> 5) * 2 * nks + (kofs >> 5)) * 64 + lane) * 8, nks, nullptr, 0};
  if (F8) {
    s.p8 = W.p8 + ((uint64_t)((cb >> 5) * 2 * (nks >> 2) + (kofs >> 7)) * 2 * 64 + lane) * 16;
    s.s8 = (int)(W.s8 * 0x01010101u);
  }
  return s;
}
__device__ __forceinline__ void wload8(const WStream& s, uint32_t step, v8i (&w)[2]) {
#pragma unroll
  for (int jt = 0; jt < 2; jt++) {
    const uint8_t* q = s.p8 + (uint64_t)((jt * (s.nks >> 2) + step) * 2) * 1024;
    const uint4 a = *reinterpret_cast<const uint4*>(q), b = *reinterpret_cast<const uint4*>(q + 1024);
    w[jt] = v8i{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w};
  }
}
__device__ __forceinline__ void wload4(const WStream& s, uint32_t k0, half8 (&w)[4][2]) {
#pragma unroll
  for (int jt = 0; jt < 2; jt++)
#pragma unroll
    for (int k = 0; k < 4; k++) w[k][jt] = *reinterpret_cast<const half8*>(s.p + (uint64_t)(jt * s.nks + k0 + k) * 512);
}

#ifndef HERRO_LP_DBG
#define HERRO_LP_DBG 0   // timing experiments (wrong results): 1 every second activation fragment read from LDS is skipped, 2 every second k-step's MFMAs, 4 the weight loads inside a call,
                         // 8 (precision 6) no K = 128 instructions, 16 (precision 6) no e4m3 conversion of the remainders (the byte planes stay as they are)
#endif
template <bool SWAP, int TERMS, int PT>
__device__ __forceinline__ void tile_gemm_p(const WStream& cur, half8 (&wa)[4][2], const WStream& nxt, const uint16_t* sh,
                                            const uint16_t* sl, uint32_t fr, uint32_t fg, f32x4 (&acc)[PT][2]) {
  half8 wb[4][2];
  if (HERRO_LP_DBG & 4) {
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int jt = 0; jt < 2; jt++) wb[k][jt] = wa[k][jt];
  } else wload4(cur, 4, wb);
  static_assert(!(SWAP && TERMS == 3), "the e4m3 term is written for weights as the A operand");
  v8i w8a[2], w8b[2], xb[PT];   // TERMS 3: the e4m3 weight fragments of the call's two K = 128 steps; the remainder fragments of the first
  auto rd8 = [&](int step, int pt) -> v8i {
    const uint8_t* s8p = reinterpret_cast<const uint8_t*>(sl);
    const uint32_t row = (uint32_t)pt * 16 + fr, c = (uint32_t)step * 8 + fg * 2;
    const uint4 a = *reinterpret_cast<const uint4*>(s8p + lsw8(row, c)), b = *reinterpret_cast<const uint4*>(s8p + lsw8(row, c + 1));
    return v8i{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w};
  };
  half8 xh[PT], xl[PT], xn[PT];
#pragma unroll
  for (int pt = 0; pt < PT; pt++) { xl[pt] = half8{}; xn[pt] = half8{}; }
  auto rd = [&](const uint16_t* plane, int k, half8 (&x)[PT]) {
#pragma unroll
    for (int pt = 0; pt < PT; pt++) x[pt] = *reinterpret_cast<const half8*>(plane + hlsw(pt * 16 + fr, k * 4 + fg));
  };
  auto mm8 = [&](const half8 (&x)[PT], const half8 (&w)[2]) {
#pragma unroll
    for (int pt = 0; pt < PT; pt++)
#pragma unroll
      for (int jt = 0; jt < 2; jt++) acc[pt][jt] = SWAP ? mma(x[pt], w[jt], acc[pt][jt]) : mma(w[jt], x[pt], acc[pt][jt]);
  };
  // TERMS 2 reads the next k-step's hi fragments between its two MFMA groups — half a step (8 MFMAs) ahead of their use.  TERMS 3 has one group per k-step and
  // reads TWO steps ahead into a third buffer (710 -> 705 us per 4096 windows; for TERMS 1 — QKV — the same change measured nothing, 694 -> 693, and is not made:
  // the two waves of a SIMD take turns on the MFMA pipe, one's wait is the other's issue slot).
  constexpr bool AHEAD2 = TERMS == 3 && !(HERRO_LP_DBG & 3);
  half8 xm[PT];
  rd(sh, 0, xh);
  if (AHEAD2) rd(sh, 1, xn);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    constexpr bool dbg_rd = (HERRO_LP_DBG & 1) != 0, dbg_mm = (HERRO_LP_DBG & 2) != 0;
    if (TERMS == 3 && k == 0) { wload8(cur, 0, w8a); __builtin_amdgcn_sched_barrier(0); }   // used behind the f16 k-steps
    if (k == 4 && !(HERRO_LP_DBG & 4)) {  // wa's last reader (k = 3) is behind us: refill it with the head of the next call
      wload4(nxt, 0, wa);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (TERMS == 3 && k == 6) { wload8(cur, 1, w8b); __builtin_amdgcn_sched_barrier(0); }   // into the registers wb's first two k-steps have left
    if (TERMS == 3 && k == 7) {   // the remainder fragments of the first K = 128 step, under the last f16 k-step
#pragma unroll
      for (int pt = 0; pt < PT; pt++) xb[pt] = rd8(0, pt);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (AHEAD2) {
      if (k + 2 < 8) rd(sh, k + 2, xm);
      __builtin_amdgcn_sched_barrier(0);
      if (k < 4) mm8(xh, wa[k]); else mm8(xh, wb[k - 4]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < PT; pt++) { xh[pt] = xn[pt]; xn[pt] = xm[pt]; }
      continue;
    }
    if (TERMS == 2 && !(dbg_rd && (k & 1))) rd(sl, k, xl);
    __builtin_amdgcn_sched_barrier(0);
    if (!(dbg_mm && (k & 1))) { if (k < 4) mm8(xh, wa[k]); else mm8(xh, wb[k - 4]); }
    __builtin_amdgcn_sched_barrier(0);
    if (k < 7 && !(dbg_rd && !(k & 1))) rd(sh, k + 1, xn);
    __builtin_amdgcn_sched_barrier(0);
    if (TERMS == 2 && !(dbg_mm && (k & 1))) {
      if (k < 4) mm8(xl, wa[k]); else mm8(xl, wb[k - 4]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int pt = 0; pt < PT; pt++) xh[pt] = xn[pt];
  }
  if constexpr (TERMS == 3) {   // + remainder (e4m3, plane `sl` read as bytes) x weight (e4m3): 2 steps of K = 128, 2 * PT * 2 instructions
    // Fenced pair by pair: left alone, the scheduler pulls the second step of the LAST row block right behind its first (it wants that block's result for the
    // code that follows the call) — two dependent K = 128 instructions back to back, each waiting out the other's full latency (seen in the ISA: s_nop 9 / 11).
    // The order of these pure instructions is pinned through their data: the second step's weight fragments pass through (empty) asm statements together with
    // the accumulators of the first step, so every instruction of the second step follows all of the first (nothing is emitted, nothing waits).
    v8i xc[PT];
#pragma unroll
    for (int pt = 0; pt < PT; pt++) xc[pt] = rd8(1, pt);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pt = 0; pt < PT; pt++) {
#pragma unroll
      for (int jt = 0; jt < 2; jt++) if (!(HERRO_LP_DBG & 8)) acc[pt][jt] = mma8(w8a[jt], xb[pt], acc[pt][jt], cur.s8);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int pt = 0; pt < PT; pt++) asm volatile("" : "+v"(acc[pt][0]), "+v"(acc[pt][1]), "+v"(w8b[0]), "+v"(w8b[1]));
#pragma unroll
    for (int pt = 0; pt < PT; pt++) {
#pragma unroll
      for (int jt = 0; jt < 2; jt++) if (!(HERRO_LP_DBG & 8)) acc[pt][jt] = mma8(w8b[jt], xc[pt], acc[pt][jt], cur.s8);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// LDS parameter block of one layer (floats): LayerNorm 1 / 2 gain and bias, then the four bias vectors
constexpr int PAR_LN1G = 0, PAR_LN1B = 256, PAR_LN2G = 512, PAR_LN2B = 768, PAR_BQKV = 1024, PAR_BPROJ = 1792, PAR_BFF2 = 2048, PAR_BFF1 = 2304;
constexpr int PAR_MAX_FF = 2048;                 // d_ff supported by the parameter block
constexpr int PAR_FLOATS = PAR_BFF1 + PAR_MAX_FF;

// Phase timer of the stack, compiled only into HERRO_PROF_BUILD libraries (HERRO_PROF_BUILD=1 python -m herro_amd.build; run with
// HERRO_PROF=1): thread 0 of one tile in 16 adds the shader cycles since its previous mark to its phase's counter
// (0 prologue, 1 LayerNorm 1, 2 QKV, 3 attention, 4 proj, 5 LayerNorm 2, 6 the wait behind FF1's epilogue, 7 the wait behind FF2, 8 final LayerNorm + heads, 9 FF1's GEMM, 10 its epilogue, 11 FF2's GEMM; 15 tiles
// sampled).  Release kernels carry nothing of it.
#ifdef HERRO_PROF_BUILD
__device__ unsigned long long g_lp_prof[16];
#define LP_BEGIN() unsigned long long _lp_t = (tid == 0 && (blockIdx.x & 15u) == 0) ? __builtin_readcyclecounter() : 0ull
#define LP_MARK(ph)                                                                      \
  do {                                                                                   \
    if (tid == 0 && (blockIdx.x & 15u) == 0) {                                           \
      const unsigned long long _n = __builtin_readcyclecounter();                        \
      atomicAdd(&g_lp_prof[ph], _n - _lp_t);                                             \
      if ((ph) == 0) atomicAdd(&g_lp_prof[15], 1ull);                                    \
      _lp_t = __builtin_readcyclecounter();                                              \
    }                                                                                    \
  } while (0)
#else
#define LP_BEGIN() do { } while (0)
#define LP_MARK(ph) do { } while (0)
#endif

// TERMS: MFMA terms of the two-term candidates proj / FF1 + FF2 — 1, 2, 3 (f16 + e4m3 remainder) for all three, or mixed (round 6, the tiers the load-time
// calibration chooses from): 21 = proj on two terms, FF1 / FF2 on one ("FF single", precision 7); 12 = proj on one, FF on two ("proj single", precision 8)
template <int TERMS, int PT, bool SIB = false>
__global__ __launch_bounds__(512) void k_layers_p(ModelDev M, BatchDev B, ModelScratch S) {
  static_assert(!SIB || PT == 4, "sibling tiles are 64-token tiles");
  constexpr int TP = TERMS == 21 ? 2 : (TERMS == 12 ? 1 : TERMS);   // proj
  constexpr int TF = TERMS == 21 ? 1 : (TERMS == 12 ? 2 : TERMS);   // FF1, FF2
  // SIB: the grid starts with the batch's B.n_tiles_b sibling tiles (windows above 64 informative rows), the ordinary tiles follow
  const bool big = SIB && blockIdx.x < B.n_tiles_b;
  const uint32_t bt = SIB ? (big ? blockIdx.x : blockIdx.x - B.n_tiles_b) : blockIdx.x;
  constexpr int HLT = 16 * PT;   // tokens per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* s_hh = reinterpret_cast<uint16_t*>(smem);
  uint16_t* s_hl = s_hh + HLT * 256;
  uint16_t* s_ah = s_hl + HLT * 256;
  uint16_t* s_al = s_ah + HLT * 256;
  float* s_red = reinterpret_cast<float*>(s_al + HLT * 256);       // [2 passes][8 waves][tokens]
  uint32_t* s_win = reinterpret_cast<uint32_t*>(s_red + 2 * 8 * HLT);
  float* s_par = reinterpret_cast<float*>(s_win + HLT);            // this layer's LayerNorm parameters and biases
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  LP_BEGIN();
  uint32_t fr = lane & 15, fg = lane >> 4;
  const uint32_t* tile_tok0 = PT == 4 ? (big ? B.tile_tok0_b : B.tile_tok0) : B.tile_tok0_q;
  const uint32_t t0 = tile_tok0[bt], nt = tile_tok0[bt + 1] - t0;
  const uint32_t grp = big ? B.tile_grp[bt] : 0u;   // the window's sibling tiles: first | count << 24
  const uint32_t cw = wave * 32;
  const float eps = M.h.ln_eps;
  const uint32_t n_layers = M.h.n_layers, d_ff = M.h.d_ff;
  // Everything below is address arithmetic on fr / fg (LDS swizzles, fragment pointers, parameter offsets).  Left alone, the
  // compiler hoists ~170 such loop-invariant values out of the layer loop and then SPILLS them (168 scratch stores before the
  // loop, ~200 scratch loads per layer, each with its own vmcnt wait).  RELAUNDER() makes fr / fg opaque at the points where
  // it is called, so the values are recomputed (1-2 VALU each) instead of kept.
#define RELAUNDER() asm volatile("" : "+v"(fr), "+v"(fg))

  if (tid < HLT) s_win[tid] = tid < nt ? S.tok_win[t0 + tid] : 0xffffff00u + tid;
  f32x4 xr[PT][2];   // the residual stream, in the accumulator layout of the GEMM calls (channel cw + 8 fg + q of token pt * 16 + fr at [pt][q / 4][q % 4]): proj and FF2 accumulate straight into it
#define XQ(pt, q) xr[pt][(q) >> 2][(q) & 3]
  half8 wa[4][2];  // the first four k-steps of the next GEMM call, always one call ahead
  wload4(wstream(M.layer[0].qkv, cw, 0, lane), 0, wa);

  auto lds8 = [&](uint32_t o, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(s_par + o), b = *reinterpret_cast<const float4*>(s_par + o + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  };
  // one coalesced pass over the layer's small parameter vectors -> LDS (the GEMM epilogues and LayerNorms then read
  // them with LDS latency instead of one L2 round trip each).  Visible after the next barrier.
  auto stage_params = [&](const LayerW& L) {
    uint32_t tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));   // (opaque: the addresses below are recomputed here instead of being kept — and spilled — across the layer body)
    const uint32_t tid = tid_;
    for (uint32_t e = tid; e < 256; e += 512) {
      s_par[PAR_LN1G + e] = L.ln1_g[e]; s_par[PAR_LN1B + e] = L.ln1_b[e];
      s_par[PAR_LN2G + e] = L.ln2_g[e]; s_par[PAR_LN2B + e] = L.ln2_b[e];
      s_par[PAR_BPROJ + e] = L.proj.bias[e]; s_par[PAR_BFF2 + e] = L.ff2.bias[e];
    }
    for (uint32_t e = tid; e < 768; e += 512) s_par[PAR_BQKV + e] = L.qkv.bias[e];
    for (uint32_t e = tid; e < d_ff; e += 512) s_par[PAR_BFF1 + e] = L.ff1.bias[e];
  };
  // LayerNorm over the 256 channels of the register-resident x, two passes (mean, then centred sum of squares).  (The one-barrier
  // form of k_layers_q — per-wave mean and M2 merged exactly — measured no faster here: 726 vs 726 us per 4096 windows.)
  auto layer_norm = [&](uint32_t og, uint32_t ob, const float* __restrict__ gp, const float* __restrict__ bp, int want_lo) {   // want_lo: 0 none, 1 an f16 plane, 2 e4m3 bytes
    float mean[PT], rstd[PT];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      float* red = s_red + pass * 8 * HLT;   // separate arrays per pass: one barrier less per LayerNorm
#pragma unroll
      for (int pt = 0; pt < PT; pt++) {
        float sm = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const float d = pass == 0 ? XQ(pt, q) : XQ(pt, q) - mean[pt];
          sm += pass == 0 ? d : d * d;
        }
        sm = fg_sum(sm);
        if (fg == 0) red[wave * HLT + pt * 16 + fr] = sm;
      }
      __syncthreads();
#pragma unroll
      for (int pt = 0; pt < PT; pt++) {
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < 8; w++) sm += red[w * HLT + pt * 16 + fr];
        if (pass == 0) mean[pt] = sm * (1.0f / 256.f);
        else rstd[pt] = __builtin_amdgcn_rsqf(sm * (1.0f / 256.f) + eps);   // v_rsq_f32 (1 ulp) instead of a correctly rounded square root and division: ~25 instructions per row block less, 100 per LayerNorm
      }
    }
    float gg[8], bb[8];
    if (gp) {  // final LayerNorm: parameters straight from global memory
      const float4 g0 = *reinterpret_cast<const float4*>(gp + cw + 8 * fg), g1 = *reinterpret_cast<const float4*>(gp + cw + 8 * fg + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(bp + cw + 8 * fg), b1 = *reinterpret_cast<const float4*>(bp + cw + 8 * fg + 4);
      gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w; gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w;
      bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
    } else {
      lds8(og + cw + 8 * fg, gg);
      lds8(ob + cw + 8 * fg, bb);
    }
#pragma unroll
    for (int pt = 0; pt < PT; pt++) {
      float y[8];
#pragma unroll
      for (int q = 0; q < 8; q++) y[q] = (XQ(pt, q) - mean[pt]) * rstd[pt] * gg[q] + bb[q];
      const uint32_t o = hlsw(pt * 16 + fr, wave * 4 + fg);
      if (want_lo == 2 && (HERRO_LP_DBG & 16)) {
        *reinterpret_cast<half8*>(s_hh + o) = pack_h8(y);
      } else if (want_lo == 2) {
        half8 hi;
        uint2 lo8;
        split_h8_f8(y, hi, lo8);
        *reinterpret_cast<half8*>(s_hh + o) = hi;
        *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(s_hl) + lsw8(pt * 16 + fr, wave * 2 + (fg >> 1)) + (fg & 1u) * 8) = lo8;
      } else if (want_lo) {
        half8 hi, lo;
        split_h8(y, hi, lo);
        *reinterpret_cast<half8*>(s_hh + o) = hi;
        *reinterpret_cast<half8*>(s_hl + o) = lo;
      } else {
        *reinterpret_cast<half8*>(s_hh + o) = pack_h8(y);
      }
    }
    __syncthreads();  // planes complete; also: everybody is past its reads of both s_red arrays
  };
  auto zero = [](f32x4 (&a)[PT][2]) {
#pragma unroll
    for (int pt = 0; pt < PT; pt++)
#pragma unroll
      for (int jt = 0; jt < 2; jt++) a[pt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto store_act = [&](auto terms_c, uint16_t* ph, uint16_t* pl, uint32_t o, const float (&v)[8]) {   // o = hlsw(row, wave * 4 + fg); terms_c: the terms of the GEMM that reads the plane
    constexpr int TA = decltype(terms_c)::value;
    if (TA == 3 && (HERRO_LP_DBG & 16)) {
      *reinterpret_cast<half8*>(ph + o) = pack_h8(v);
    } else if (TA == 3) {
      half8 hi;
      uint2 lo8;
      split_h8_f8(v, hi, lo8);
      *reinterpret_cast<half8*>(ph + o) = hi;
      // the same (row, chunk) in the byte plane: 16-byte chunk wave * 2 + fg / 2 of the row (hlsw's 8-half chunk wave * 4 + fg, halved), XORed alike
      const uint32_t row = o >> 8, ch8 = ((o >> 3) & 31u) ^ (row & 15u);
      *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(pl) + lsw8(row, ch8 >> 1) + (ch8 & 1u) * 8) = lo8;
    } else if (TA == 2) {
      half8 hi, lo;
      split_h8(v, hi, lo);
      *reinterpret_cast<half8*>(ph + o) = hi;
      *reinterpret_cast<half8*>(pl + o) = lo;
    } else {
      *reinterpret_cast<half8*>(ph + o) = pack_h8(v);
    }
  };

  stage_params(M.layer[0]);
  {  // x = FC output + positional encoding: the residual stream stays in registers from here to the heads
    const float4 pdv = *reinterpret_cast<const float4*>(M.pe_div + ((cw + 8 * fg) >> 1));
    const float pd[4] = {pdv.x, pdv.y, pdv.z, pdv.w};
#pragma unroll
    for (int pt = 0; pt < PT; pt++) {
      const uint32_t tok = pt * 16 + fr;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      float row = 0.f;
      uint32_t rowu = 0;
      if (tok < nt) {
        const float* xp = S.x + (uint64_t)(t0 + tok) * 256 + cw + 8 * fg;
        a = *reinterpret_cast<const float4*>(xp);
        b = *reinterpret_cast<const float4*>(xp + 4);
        rowu = S.tok_row[t0 + tok];
        row = (float)rowu;
      }
      xr[pt][0] = f32x4{a.x, a.y, a.z, a.w};
      xr[pt][1] = f32x4{b.x, b.y, b.z, b.w};
      if (tok < nt) {
        if (rowu >= M.pe_rows) {   // rows inside the table got their encoding in k_fc_r's epilogue (the table holds exactly what this branch computes: launch_pe_table)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const float ang = __fmul_rn(row, pd[j]);
            float sn, cs;
            sincosf(ang, &sn, &cs);   // one range reduction for both
            xr[pt][j >> 1][(2 * j) & 3] += sn;
            xr[pt][j >> 1][(2 * j + 1) & 3] += cs;
          }
        }
      }
    }
  }
  __syncthreads();  // s_win, s_par

  const float scale = 1.4426950408889634f / sqrtf(32.f);   // 1 / sqrt(head dim) and log2(e): the softmax runs on v_exp_f32 (2^x) without a multiply per score
  for (uint32_t li = 0; li < n_layers; li++) {
    const LayerW& L = M.layer[li];
    const LayerW& Ln = M.layer[li + 1 < n_layers ? li + 1 : 0];  // after the last layer: a harmless re-read of layer 0
    RELAUNDER();
    LP_MARK(li ? 7 : 0);
    layer_norm(PAR_LN1G, PAR_LN1B, nullptr, nullptr, 0);   // Q, K, V read the hi plane only (see the header: QKV single)
    LP_MARK(1);
    RELAUNDER();
    {  // ---- attention, head = wave
      half8 qh[PT], kh[PT], vh[2][PT / 2];
      {
        f32x4 a[PT][2];
        float bq[8];
        zero(a);
        tile_gemm_p<false, 1, PT>(wstream(L.qkv, cw, 0, lane), wa, wstream(L.qkv, 256 + cw, 0, lane), s_hh, s_hl, fr, fg, a);
        lds8(PAR_BQKV + cw + 8 * fg, bq);
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; q++) v[q] = (a[pt][q >> 2][q & 3] + bq[q]) * scale;
          qh[pt] = pack_h8(v);
        }
        zero(a);
        tile_gemm_p<false, 1, PT>(wstream(L.qkv, 256 + cw, 0, lane), wa, wstream(L.qkv, 512 + cw, 0, lane), s_hh, s_hl, fr, fg, a);
        lds8(PAR_BQKV + 256 + cw + 8 * fg, bq);
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; q++) v[q] = a[pt][q >> 2][q & 3] + bq[q];
          kh[pt] = pack_h8(v);
        }
        zero(a);
        tile_gemm_p<true, 1, PT>(wstream(L.qkv, 512 + cw, 0, lane), wa, wstream(L.proj, cw, 0, lane), s_hh, s_hl, fr, fg, a);   // (a call only reads the f16 head of its successor)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
          const float bv = s_par[PAR_BQKV + 512 + cw + 8 * (fr >> 2) + 4 * ct + (fr & 3)];
#pragma unroll
          for (int kk = 0; kk < PT / 2; kk++) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = a[2 * kk + (e >> 2)][ct][e & 3] + bv;
            vh[ct][kk] = pack_h8(v);
          }
        }
      }
      LP_MARK(2);
      uint32_t wj[PT][4];
#pragma unroll
      for (int pj = 0; pj < PT; pj++)
#pragma unroll
        for (int r = 0; r < 4; r++) wj[pj][r] = s_win[pj * 16 + 4 * fg + r];
      if (!SIB || !big) {
#pragma unroll
      for (int pi = 0; pi < PT; pi++) {
        const uint32_t wi = s_win[pi * 16 + fr];
        f32x4 st[PT];
        float m = -INFINITY;
#pragma unroll
        for (int pj = 0; pj < PT; pj++) {
          st[pj] = mma(kh[pj], qh[pi], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
          for (int r = 0; r < 4; r++) {
            st[pj][r] = wj[pj][r] == wi ? st[pj][r] : -INFINITY;
            m = fmax_nc(m, st[pj][r]);
          }
        }
        m = fg_max(m);
        float l = 0.f;
#pragma unroll
        for (int pj = 0; pj < PT; pj++) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float pexp = __builtin_amdgcn_exp2f(st[pj][r] - m);
            st[pj][r] = pexp;
            l += pexp;
          }
        }
        l = fg_sum(l);
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kk = 0; kk < PT / 2; kk++) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = st[2 * kk + (e >> 2)][e & 3];
          const half8 ph = pack_h8(v);
#pragma unroll
          for (int ct = 0; ct < 2; ct++) o[ct] = mma(vh[ct][kk], ph, o[ct]);
        }
        const float inv = __builtin_amdgcn_rcpf(l);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = o[e >> 2][e & 3] * inv;
        store_act(std::integral_constant<int, TP>{}, s_ah, s_al, hlsw(pi * 16 + fr, wave * 4 + fg), v);
      }
      } else {
        // ---- a window above 64 informative rows: its tiles (siblings) hold 64 of its tokens each.  Every tile publishes the K / V
        // fragments of its heads in REGISTER layout (a sibling's wave loads them straight into MFMA operands), attends to its own
        // keys from registers, then walks the siblings' blocks with a running maximum / sum (the softmax over all keys of the window,
        // accumulated block by block).  Two buffers by layer parity: a sibling is at most one layer ahead (it needs OUR keys of its
        // layer before it can finish it).  Co-residency: the siblings are consecutive workgroups at the HEAD of the grid and workgroups
        // are dispatched in index order, so the lowest unfinished group always has all its tiles on the chip (and in a batch's first
        // round they all start together: no sibling waits for a compute unit while the others hold theirs).
        {
          uint16_t* mine = S.sib_kv + ((((uint64_t)bt * 2 + (li & 1u)) * 8 + wave) * 8 * 64 + lane) * 8;
#pragma unroll
          for (int pt = 0; pt < PT; pt++) *reinterpret_cast<half8*>(mine + pt * 512) = kh[pt];
#pragma unroll
          for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int kk = 0; kk < PT / 2; kk++) *reinterpret_cast<half8*>(mine + (PT + ct * (PT / 2) + kk) * 512) = vh[ct][kk];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the siblings run on other XCDs, behind their own L2: write back, once per wave
          __syncthreads();
          if (tid == 0) __hip_atomic_store(S.sib_flag + bt, li + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float mrun[PT], lrun[PT];
        f32x4 orun[PT][2];
#pragma unroll
        for (int pi = 0; pi < PT; pi++) {   // own keys first: every query finds itself there, the running maximum is finite from here on
          const uint32_t wi = s_win[pi * 16 + fr];
          f32x4 st[PT];
          float m = -INFINITY;
#pragma unroll
          for (int pj = 0; pj < PT; pj++) {
            st[pj] = mma(kh[pj], qh[pi], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int r = 0; r < 4; r++) {
              st[pj][r] = wj[pj][r] == wi ? st[pj][r] : -INFINITY;
              m = fmax_nc(m, st[pj][r]);
            }
          }
          m = fg_max(m);
          float l = 0.f;
#pragma unroll
          for (int pj = 0; pj < PT; pj++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const float pexp = __builtin_amdgcn_exp2f(st[pj][r] - m);
              st[pj][r] = pexp;
              l += pexp;
            }
          mrun[pi] = m;
          lrun[pi] = fg_sum(l);
          orun[pi][0] = f32x4{0.f, 0.f, 0.f, 0.f};
          orun[pi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < PT / 2; kk++) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = st[2 * kk + (e >> 2)][e & 3];
            const half8 ph = pack_h8(v);
#pragma unroll
            for (int ct = 0; ct < 2; ct++) orun[pi][ct] = mma(vh[ct][kk], ph, orun[pi][ct]);
          }
        }
        const uint32_t g0 = grp & 0xfffffu, gk = (grp >> 20) & 15u, nlast = grp >> 24;   // the group; the window's tokens in its last tile (small windows may fill the rest)
        const uint32_t nbig = bt == g0 + gk - 1 ? nlast : (uint32_t)HLT;                   // ... in this tile: queries beyond them belong to other windows
#pragma unroll 1
        for (uint32_t sb = g0; sb < g0 + gk; sb++) {
          if (sb == bt) continue;
          uint32_t spins = 0;
          // poll without cache maintenance (an acquiring load invalidates caches on every turn, for the whole XCD); ONE acquire when the flag is up
          while (__hip_atomic_load(S.sib_flag + sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= li) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1u << 21)) {   // ~ seconds: the sibling is not coming (it would be a planner bug) — give up LOUDLY instead of hanging the device
              if (lane == 0) atomicExch(S.sib_err, 1u + li);
              break;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          const uint32_t nts = sb == g0 + gk - 1 ? nlast : (uint32_t)HLT;   // the sibling's tokens of this window
          const uint16_t* src = S.sib_kv + ((((uint64_t)sb * 2 + (li & 1u)) * 8 + wave) * 8 * 64 + lane) * 8;
          half8 kf[PT], vf[2][PT / 2];
#pragma unroll
          for (int pt = 0; pt < PT; pt++) kf[pt] = *reinterpret_cast<const half8*>(src + pt * 512);
#pragma unroll
          for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int kk = 0; kk < PT / 2; kk++) vf[ct][kk] = *reinterpret_cast<const half8*>(src + (PT + ct * (PT / 2) + kk) * 512);
#pragma unroll
          for (int pi = 0; pi < PT; pi++) {
            f32x4 st[PT];
            float bm = -INFINITY;
#pragma unroll
            for (int pj = 0; pj < PT; pj++) {
              st[pj] = mma(kf[pj], qh[pi], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
              for (int r = 0; r < 4; r++) {
                st[pj][r] = ((uint32_t)(pj * 16 + 4 * fg + r) < nts && (uint32_t)(pi * 16) + fr < nbig) ? st[pj][r] : -INFINITY;   // this window's keys, for this window's queries
                bm = fmax_nc(bm, st[pj][r]);
              }
            }
            const float mn = fmax_nc(mrun[pi], fg_max(bm));
            const float alpha = __builtin_amdgcn_exp2f(mrun[pi] - mn);
            float l = 0.f;
#pragma unroll
            for (int pj = 0; pj < PT; pj++)
#pragma unroll
              for (int r = 0; r < 4; r++) {
                const float pexp = __builtin_amdgcn_exp2f(st[pj][r] - mn);
                st[pj][r] = pexp;
                l += pexp;
              }
            lrun[pi] = lrun[pi] * alpha + fg_sum(l);
            mrun[pi] = mn;
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
#pragma unroll
              for (int r = 0; r < 4; r++) orun[pi][ct][r] *= alpha;
#pragma unroll
            for (int kk = 0; kk < PT / 2; kk++) {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; e++) v[e] = st[2 * kk + (e >> 2)][e & 3];
              const half8 ph = pack_h8(v);
#pragma unroll
              for (int ct = 0; ct < 2; ct++) orun[pi][ct] = mma(vf[ct][kk], ph, orun[pi][ct]);
            }
          }
        }
#pragma unroll
        for (int pi = 0; pi < PT; pi++) {
          const float inv = __builtin_amdgcn_rcpf(lrun[pi]);
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = orun[pi][e >> 2][e & 3] * inv;
          store_act(std::integral_constant<int, TP>{}, s_ah, s_al, hlsw(pi * 16 + fr, wave * 4 + fg), v);
        }
      }
    }
    __syncthreads();
    LP_MARK(3);
    RELAUNDER();
    {  // ---- output projection + residual
      float bp[8];
      lds8(PAR_BPROJ + cw + 8 * fg, bp);
#pragma unroll
      for (int pt = 0; pt < PT; pt++)
#pragma unroll
        for (int q = 0; q < 8; q++) XQ(pt, q) += bp[q];
      tile_gemm_p<false, TP, PT>(wstream<TP == 3>(L.proj, cw, 0, lane), wa, wstream(L.ff1, cw, 0, lane), s_ah, s_al, fr, fg, xr);   // x += attention . Wproj
    }
    RELAUNDER();
    LP_MARK(4);
    layer_norm(PAR_LN2G, PAR_LN2B, nullptr, nullptr, TF == 3 ? 2 : (TF == 2 ? 1 : 0));
    LP_MARK(5);
    {  // ---- feed-forward, 256 hidden channels at a time
      {  // FF2 accumulates into the residual stream itself (32 accumulator registers less through the FF loop), which takes the FF2 bias first: no
         // parameter is read after the loop's last barrier, which is what lets the next layer's parameters be staged right behind it
        float b2[8];
        lds8(PAR_BFF2 + cw + 8 * fg, b2);
#pragma unroll
        for (int pt = 0; pt < PT; pt++)
#pragma unroll
          for (int q = 0; q < 8; q++) XQ(pt, q) += b2[q];
      }
      // Round 6, measured on one box against the round-5 library (profiles/r6_ab_layers.txt): (i) the epilogue of chunk c issued in slices INSIDE FF1(c + 1)
      // (call order FF1(0), [FF1(c + 1), FF2(c)] ..., two accumulator sets) — the matrix pipe was meant to stay busy while a wave splits and stores; every
      // wave at k-steps 0..3 or the two waves of a SIMD in different halves of the call: +5 % / +6 % kernel time, every phase of the layer slower, not only
      // the feed-forward; (ii) the epilogues of Q / K inside the calls of K / V: no change; (iii) key blocks that share no window with a query block skipped
      // in the attention: the phase 8 % shorter, the kernel the same.  None of them is kept (git show 2a15d40 has (i) and (ii)).  What is kept: with ONE term in
      // FF1 / FF2 (precisions 5, 7) nobody uses the lo planes during the loop, so the hidden chunk alternates between s_ah and s_al and the barrier BEHIND
      // FF2(c) goes: a wave that is through its FF2 starts FF1(c + 1) while its SIMD partner still streams — the skew a barrier removes is what overlaps one
      // wave's epilogue with the other's MFMAs.
      constexpr bool TWO_BUF = TF == 1;
      for (uint32_t c = 0; c < d_ff; c += 256) {
        RELAUNDER();
        uint16_t* hid = (TWO_BUF && ((c >> 8) & 1u)) ? s_al : s_ah;   // (read as an f16 hi plane in either case: TF == 1 has no lo plane)
        f32x4 a1[PT][2];
        {  // the accumulators start from the bias (the moves that would zero them carry it)
          float b1[8];
          lds8(PAR_BFF1 + c + cw + 8 * fg, b1);
#pragma unroll
          for (int pt = 0; pt < PT; pt++)
#pragma unroll
            for (int jt = 0; jt < 2; jt++) a1[pt][jt] = f32x4{b1[4 * jt], b1[4 * jt + 1], b1[4 * jt + 2], b1[4 * jt + 3]};
        }
        tile_gemm_p<false, TF, PT>(wstream<TF == 3>(L.ff1, c + cw, 0, lane), wa, wstream(L.ff2, cw, c, lane), s_hh, s_hl, fr, fg, a1);
        LP_MARK(9);
#pragma unroll
        for (int pt = 0; pt < PT; pt++) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; q++) v[q] = __builtin_amdgcn_fmed3f(a1[pt][q >> 2][q & 3], 0.f, __builtin_huge_valf());   // ReLU in ONE instruction (fmaxf: a canonicalising v_max in front of the v_max)
          store_act(std::integral_constant<int, TF>{}, hid, s_al, hlsw(pt * 16 + fr, wave * 4 + fg), v);
        }
        LP_MARK(10);
        __syncthreads();   // the hidden chunk is complete; TWO_BUF: ... and every wave is through FF2(c - 1), whose buffer the NEXT epilogue writes
        LP_MARK(6);
        const bool more = c + 256 < d_ff;
        const WStream nx = more ? wstream(L.ff1, c + 256 + cw, 0, lane) : wstream(Ln.qkv, cw, 0, lane);
        tile_gemm_p<false, TF, PT>(wstream<TF == 3>(L.ff2, cw, c, lane), wa, nx, hid, s_al, fr, fg, xr);
        LP_MARK(11);
        if (!TWO_BUF || !more) __syncthreads();   // (the loop's last barrier stays: the next layer's parameters are staged behind it)
        LP_MARK(7);
      }
    }
    // every wave is past its last read of this layer's parameters (the barrier that closed the FF loop)
    if (li + 1 < n_layers) stage_params(Ln);   // visible after the first barrier of the next LayerNorm
  }
  RELAUNDER();
  // where the logits of this lane's token go, and the heads' bias: requested in front of the last LayerNorm, used behind the heads
  const uint64_t out_o = (wave < PT && wave * 16 + fr < nt) ? S.tok_out[t0 + wave * 16 + fr] : 0ull;
  const float4 hb = *reinterpret_cast<const float4*>(M.heads.bias + 4 * fg);
  layer_norm(0, 0, M.lnf_g, M.lnf_b, 1);
#undef RELAUNDER
#undef XQ
  if (wave < PT) {  // heads, three terms
    const uint32_t pt = wave;
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    const Weight& W = M.heads;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      const half8 wh = *reinterpret_cast<const half8*>(W.h16 + (uint64_t)fr * 256 + ks * 32 + fg * 8);
      const half8 wl = *reinterpret_cast<const half8*>(W.l16 + (uint64_t)fr * 256 + ks * 32 + fg * 8);
      const uint32_t o = hlsw(pt * 16 + fr, ks * 4 + fg);
      const half8 xh = *reinterpret_cast<const half8*>(s_hh + o);
      const half8 xl = *reinterpret_cast<const half8*>(s_hl + o);
      a = mma(wl, xh, a);
      a = mma(wh, xl, a);
      a = mma(wh, xh, a);
    }
    const uint32_t tok = pt * 16 + fr;
    if (tok < nt) {
      const uint64_t o = out_o;
      const float hbv[4] = {hb.x, hb.y, hb.z, hb.w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t ch = 4 * fg + r;
        const float v = a[r] + hbv[r];
        if (ch == 0) B.out_info[o] = v;
        else if (ch < 6) B.out_base[o * 5 + (ch - 1)] = v;
      }
    }
  }
  LP_MARK(8);
}
constexpr size_t layers_p_shm(int tokens) { return (size_t)4 * tokens * 256 * 2 + 2 * 8 * tokens * 4 + tokens * 4 + (size_t)PAR_FLOATS * 4; }

// ---------------------------------------------------------------------------------------------------
// Tiles of 32 tokens (k_layers_p<TERMS, 2>): the same organisation — 8 waves, wave = head and 32-channel slab, weights one half
// call ahead — over two row blocks instead of four: half the MFMAs and half the LDS traffic for the same weight stream.  A launch
// runs in rounds of one 64-token tile per compute unit; when its last round would fill at most half of the chip, the windows of
// that round go into 32-token tiles instead (plan_tiles, herro_api.hip).
//
// What round 4 measured before settling on this (profiles/r4_layers_*): a SEPARATE design for 32-token tiles, k_layers_q — four
// waves per workgroup, wave = 64 channels (two heads), a ring of four weight k-steps, one-barrier LayerNorm, 80 KB of LDS so that
// TWO workgroups share a compute unit and one's LayerNorm / softmax / epilogue phases sit under the other's MFMAs — the
// restructuring VERDICT r3 asked for.  Parity was green and both workgroups were resident (7.5 waves per CU, L2 hit rate 97 %),
// but it was SLOWER: 780 us per 4096 windows against 710.  Every weight fragment then feeds two MFMAs per term instead of four,
// i.e. the 6.3 MB weight pass of a tile goes through the CU's vector-memory path (64 B/clk) once per 32 tokens: 98 k cycles per
// workgroup, 197 k for the pair, against 176 k cycles of MFMA issue — QKV (one term) ran at exactly that bound (46.6 k cycles vs
// 49 k), and the denser body clocked ~10 % lower (1.79 vs 1.98 GHz).  The phase overlap is real (339 k cycles per 64 tokens
// instead of 375 k) but the path it needs is the one this kernel already uses at 50 % when its MFMAs run at full rate.  Alone on
// a CU a k_layers_q tile took 0.8 of a 64-token tile's time (one wave per SIMD hides nothing); this variant takes ~0.45.
// ---------------------------------------------------------------------------------------------------
__global__ void k_build_tokens_h(BatchDev B, ModelScratch S) {
  const uint32_t b = blockIdx.x;
  const uint32_t t0 = B.tok_off[b], t1 = B.tok_off[b + 1];
  for (uint32_t n = t0 + threadIdx.x; n < t1; n += blockDim.x) {
    const uint32_t row = B.sup_row[B.sup_off[b] + (n - t0)];
    S.tok_win[n] = b;
    S.tok_row[n] = row;
    S.tok_out[n] = B.out_off[b] + (n - t0);
    TokMeta tm;
    tm.plane_off = B.plane_off[b];
    tm.plane_ld = B.plane_ld[b];
    tm.tok_row = row;
    tm.len = B.len[b];
    tm.lmax = B.lmax[b];
    tm.rf_idx = (uint32_t)((B.rf_base ? B.rf_base[b] : B.out_off[b]) + (n - t0));
    tm.pad1 = 0;
    S.tok_meta[n] = tm;
    uint32_t mk[2] = {0, 0}, dt[2] = {0, 0}, dq[2] = {0, 0}, ok = 0;
    for (int i = 0; i < 5; i++) {
      const int32_t r = (int32_t)row - 2 + i;
      const bool in = r >= 0 && r < (int32_t)tm.lmax, real = in && r < (int32_t)tm.len;
      const uint32_t sh = 8u * (i & 3);
      if (real) mk[i >> 2] |= 0xffu << sh;
      else { dt[i >> 2] |= (in ? (uint32_t)TOK_PAD : 12u) << sh; dq[i >> 2] |= (in ? 126u : 0xffu) << sh; }
    }
    for (int j = 0; j < 3; j++) {
      const int32_t pos = (int32_t)row + j - 1;
      if (pos >= 0 && pos < (int32_t)tm.lmax) ok |= 1u << j;
    }
    TokCv cv;
    cv.plane_off = tm.plane_off;
    cv.ld_d1 = tm.plane_ld | (dt[1] << 16) | (dq[1] << 24);
    cv.row_ok = row | (ok << 16) | (mk[1] << 24);
    cv.rf_idx = tm.rf_idx;
    cv.mk0 = mk[0]; cv.dt0 = dt[0]; cv.dq0 = dq[0];
    S.tok_cv[n] = cv;
  }
}

// the positional-encoding table of the stack's prologue: the same two instructions per angle the kernel's own branch runs (a product rounded once, one sincosf)
__global__ void k_pe_table(const float* __restrict__ pe_div, float* __restrict__ tab, uint32_t rows, uint32_t half_d) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * half_d) return;
  const uint32_t row = i / half_d, j = i % half_d;
  const float ang = __fmul_rn((float)row, pe_div[j]);
  float sn, cs;
  sincosf(ang, &sn, &cs);
  *reinterpret_cast<float2*>(tab + (uint64_t)row * (2 * half_d) + 2 * j) = make_float2(sn, cs);
}

}  // namespace

void launch_pe_table(const float* pe_div, float* tab, uint32_t rows, uint32_t d_model, hipStream_t st) {
  const uint32_t n = rows * (d_model / 2);
  hipLaunchKernelGGL(k_pe_table, dim3((n + 255) / 256), dim3(256), 0, st, pe_div, tab, rows, d_model / 2);
}

bool model_h_supported(const ModelDev& M) {
  const ModelHyper& h = M.h;
  if (!(model_default_variant(M) && h.d_model == 256 && h.n_heads == 8 && h.d_ff % 256 == 0 && h.d_ff <= (uint32_t)PAR_MAX_FF && h.rows == HERRO_ROWS && M.heads.h16 && M.heads.l16)) return false;
  for (uint32_t l = 0; l < h.n_layers; l++)
    if (!M.layer[l].qkv.ph16 || !M.layer[l].proj.ph16 || !M.layer[l].ff1.ph16 || !M.layer[l].ff2.ph16) return false;
  return true;
}

// the f16 conv / FC kernels are written for ONE conv stack (kw 3, 64 -> 128 channels); any other member of the family with the encoder shapes above takes
// the bf16x3 front end of model.hip (launch_front_generic) and the f16 stack behind it (round 6: until then such a model ran entirely on the generic kernels)
bool model_h_conv_supported(const ModelDev& M) {
  const ModelHyper& h = M.h;
  return h.kw == 3 && h.c1 == 64 && h.c2 == HC2 && M.conv1g.ph16 && M.conv2.ph16 && M.fc.h16 && M.fc.ph16 && M.fc.K % 128 == 0;
}

bool model_h_f8_supported(const ModelDev& M) {
  if (!model_h_supported(M)) return false;
  for (uint32_t l = 0; l < M.h.n_layers; l++)
    if (!M.layer[l].proj.p8 || !M.layer[l].ff1.p8 || !M.layer[l].ff2.p8) return false;
  return true;
}

// HERRO_LAYERS_Q: 0 keeps every tile at 64 tokens, 2 puts every window of <= 32 rows into 32-token tiles (both for A/B);
// default 1: 32-token tiles take the short last round of a launch (plan_tiles, herro_api.hip)
int model_h_half_tiles(const ModelDev& M) {
  static const int mode = std::max(0, std::min(2, ab_env("HERRO_LAYERS_Q", 1)));   // (A/B builds only)
  return model_h_supported(M) ? mode : 0;
}

// B must be tileable (windows of <= 64 informative rows in B.tile_tok0 / _q, windows of 65 .. 64 * 8 rows on sibling tiles, B.tile_tok0_b) —
// herro_job_infer sends still larger windows through the layer-by-layer kernels of model.hip.  terms: 2 (precision 4), 1 (precision 5), 3 (6), 21 (7: FF single), 12 (8: proj single).
void launch_model_h(const ModelDev& M, const BatchDev& B, const ModelScratch& S, int terms, hipStream_t st, KernelTimer* tm) {
  const uint32_t N = B.n_tok;
  if (N == 0 || B.n_tiles + B.n_tiles_q + B.n_tiles_b == 0) return;
  const ModelHyper& h = M.h;
  const bool f16_front = model_h_conv_supported(M);
  ModelDev M_enc = M;
  if (!f16_front) {
    launch_front_generic(M, B, S, st, tm);   // token table, conv stack, projection, position term: S.x complete
    M_enc.pe_rows = 0xffffffffu;             // (the stack's prologue adds the encoding of rows >= pe_rows itself: nothing left to add)
  } else {
  KT_BEGIN(tm, "build_tokens", st);
  hipLaunchKernelGGL(k_build_tokens_h, dim3(B.n_win), dim3(64), 0, st, B, S);
  KT_END(tm, st);
  // conv and FC over the whole launch group, or (A/B builds, HERRO_CONV_CHUNK = tokens per chunk, a multiple of 16) chunk by chunk — conv(k), FC(k), conv(k + 1) ... —
  // so that a chunk's y2 (7936 B per token) is read back while it can still sit in the 256 MB memory-side cache (measured: profiles/r5_ab_runs.json r5p)
  static const uint32_t chunk_tok = (uint32_t)std::max(0, ab_env("HERRO_CONV_CHUNK", 0)) & ~15u;
  static const int force_g = ab_env("HERRO_FC_G", 0);
  static const uint32_t n_cu = [] { int dev = 0, c = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev); return (uint32_t)std::max(c, 1); }();
  const uint32_t n_rows = N * HERRO_ROWS;
  auto conv = [&](uint32_t t0, uint32_t t1) {   // tokens [t0, t1): t0 a multiple of 16 (so that its first pair opens a unit of 16)
    const uint32_t unit0 = t0 * HERRO_ROWS / 16, n_units = (std::min(t1 * HERRO_ROWS, n_rows) + 15) / 16, nu = n_units - unit0;
    // one block of 16 pairs per step, three workgroups per compute unit
    if (B.rf_q) {
      opt_in_lds(reinterpret_cast<const void*>(k_conv_m<1, 3, true>), CONV_M_SHM);
      hipLaunchKernelGGL((k_conv_m<1, 3, true>), dim3(std::min<uint32_t>((nu + 3) / 4, 768u)), dim3(CM_NT), CONV_M_SHM, st, M, B, S, n_rows, n_units, unit0);
    } else {
      opt_in_lds(reinterpret_cast<const void*>(k_conv_m<1, 3, false>), CONV_M_SHM);
      hipLaunchKernelGGL((k_conv_m<1, 3, false>), dim3(std::min<uint32_t>((nu + 3) / 4, 768u)), dim3(CM_NT), CONV_M_SHM, st, M, B, S, n_rows, n_units, unit0);
    }
  };
  auto fc = [&](uint32_t t0, uint32_t t1) {
    // tile height (16-row blocks, 4 .. 8): the one that leaves the busiest compute unit the fewest rows — two workgroups fit a unit, a third would wait for a slot —, the taller
    // one on a tie (every workgroup streams the whole weight matrix: fewer of them, less L2 traffic).  HERRO_FC_G forces a height in A/B builds (3 | 4: the 96 | 128 rows of
    // the kernel's first version, 5 .. 8 and 14 (= 4): blocks)
    const uint32_t n = t1 - t0, lda = HERRO_ROWS * h.c2;
    auto busiest = [&](uint32_t rows) { const uint32_t wg = (n + rows - 1) / rows; return (uint64_t)((wg + n_cu - 1) / n_cu) * rows + (wg > 2 * n_cu ? 1u << 20 : 0u); };
    int nb = 8;
    for (int b = 7; b >= 4; b--) if (busiest(16u * b) < busiest(16u * nb)) nb = b;
    if (force_g == 3) nb = 6; else if (force_g == 4) nb = 8; else if (force_g == 14) nb = 4; else if (force_g >= 5 && force_g <= 8) nb = force_g;
    const uint16_t* A = S.y2_hi + (uint64_t)t0 * lda;
    float* C = S.x + (uint64_t)t0 * h.d_model;
    const uint32_t* rows = S.tok_row + t0;
    const dim3 grid((n + 16u * nb - 1) / (16u * nb));
    switch (nb) {
      case 4: hipLaunchKernelGGL(k_fc_r<4>, grid, dim3(512), FC_R_SHM, st, A, lda, M.fc, C, h.d_model, n, rows, M.pe_tab, M.pe_rows); break;
      case 5: hipLaunchKernelGGL(k_fc_r<5>, grid, dim3(512), FC_R_SHM, st, A, lda, M.fc, C, h.d_model, n, rows, M.pe_tab, M.pe_rows); break;
      case 6: hipLaunchKernelGGL(k_fc_r<6>, grid, dim3(512), FC_R_SHM, st, A, lda, M.fc, C, h.d_model, n, rows, M.pe_tab, M.pe_rows); break;
      case 7: hipLaunchKernelGGL(k_fc_r<7>, grid, dim3(512), FC_R_SHM, st, A, lda, M.fc, C, h.d_model, n, rows, M.pe_tab, M.pe_rows); break;
      default: hipLaunchKernelGGL(k_fc_r<8>, grid, dim3(512), FC_R_SHM, st, A, lda, M.fc, C, h.d_model, n, rows, M.pe_tab, M.pe_rows);
    }
  };
  if (chunk_tok && N > chunk_tok) {
    KT_BEGIN(tm, "conv_fused", st);   // (the span carries both kernels of every chunk)
    for (uint32_t t0 = 0; t0 < N; t0 += chunk_tok) { conv(t0, std::min(N, t0 + chunk_tok)); fc(t0, std::min(N, t0 + chunk_tok)); }
    KT_END(tm, st);
  } else {
    KT_BEGIN(tm, "conv_fused", st);
    conv(0, N);
    KT_END(tm, st);
    KT_BEGIN(tm, "fc_gemm", st);
    fc(0, N);
    KT_END(tm, st);
  }
  }
  KT_BEGIN(tm, "layers_fused", st);   // one span: the 64-token tiles (windows of 33..64 informative rows and what shares their tiles), then the 32-token ones
  auto launch = [&](auto kern, uint32_t n_tiles, int tokens) {
    opt_in_lds(reinterpret_cast<const void*>(kern), layers_p_shm(tokens));
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(512), layers_p_shm(tokens), st, M_enc, B, S);
  };
  auto launch5 = [&](auto k1, auto k2, auto k3, auto k21, auto k12, uint32_t n_tiles, int tokens) {
    switch (terms) {
      case 3: launch(k3, n_tiles, tokens); break;
      case 2: launch(k2, n_tiles, tokens); break;
      case 21: launch(k21, n_tiles, tokens); break;
      case 12: launch(k12, n_tiles, tokens); break;
      default: launch(k1, n_tiles, tokens);
    }
  };
  if (B.n_tiles_b) {   // sibling tiles of the windows above 64 informative rows at the head of ONE grid with the ordinary 64-token tiles
    (void)hipMemsetAsync(S.sib_flag, 0, (size_t)B.n_tiles_b * 4, st);
    launch5(k_layers_p<1, 4, true>, k_layers_p<2, 4, true>, k_layers_p<3, 4, true>, k_layers_p<21, 4, true>, k_layers_p<12, 4, true>, B.n_tiles_b + B.n_tiles, 64);
  } else if (B.n_tiles) launch5(k_layers_p<1, 4>, k_layers_p<2, 4>, k_layers_p<3, 4>, k_layers_p<21, 4>, k_layers_p<12, 4>, B.n_tiles, 64);
  if (B.n_tiles_q) launch5(k_layers_p<1, 2>, k_layers_p<2, 2>, k_layers_p<3, 2>, k_layers_p<21, 2>, k_layers_p<12, 2>, B.n_tiles_q, 32);
  KT_END(tm, st);
}

#ifdef HERRO_PROF_BUILD
void model_h_prof_dump() {
  unsigned long long h[16] = {0};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lp_prof), sizeof h) != hipSuccess || !h[15]) return;
  // (6 / 7 are the waits at the barriers behind FF1's epilogue / FF2 as wave 0 sees them since round 5's finer marks: 9 FF1's GEMM call, 10 its epilogue, 11 FF2's GEMM call)
  static const char* name[12] = {"prologue", "LN1", "QKV", "attention", "proj", "LN2", "FF1 barrier wait", "FF2 barrier wait", "final LN + heads", "FF1 gemm", "FF1 epilogue", "FF2 gemm"};
  fprintf(stderr, "PROF k_layers_p (%llu tiles sampled), shader cycles per tile by phase:", h[15]);
  for (int p = 0; p < 12; p++) fprintf(stderr, " %s %.0f", name[p], (double)h[p] / (double)h[15]);
  fprintf(stderr, "\n");
}
#endif

}  // namespace herro
