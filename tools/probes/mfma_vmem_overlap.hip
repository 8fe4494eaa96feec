// Probe (round 6): do MFMAs and L2 -> register weight loads of one compute unit OVERLAP, or do their times add?  k_fc_r and the GEMM calls of k_layers_p both measure
// "matrix-pipe cycles + vector-memory cycles" per step, as if the two were serialised.  One workgroup of 8 waves per compute unit (the stack's shape) or two
// (k_fc_r's); per iteration a wave issues NM MFMAs (16x16x32 f16, register operands) and / or NL loads of 16 B per lane (1 KB per wave and load) from an L2-resident 2 MB
// buffer in fragment order.  Modes: 0 MFMAs only, 1 loads only, 2 both in every wave (loads of iteration i + 1 requested before the MFMAs of iteration i), 3 MFMAs in waves
// 0..3 and loads in waves 4..7 (twice the per-wave counts, the same totals per unit).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_vmem_overlap mfma_vmem_overlap.hip ; run: ./mfma_vmem_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NM, int NL>
__global__ __launch_bounds__(512) void k_probe(const uint4* __restrict__ buf, uint32_t n_frag, uint32_t iters, float* out) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  half8 a, b;
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * (float)(lane + i)); b[i] = (_Float16)(0.002f * (float)(lane ^ i)); }
  uint4 sink = make_uint4(0, 0, 0, 0);
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
  const bool do_l = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
  constexpr int RM = MODE == 3 ? 2 * NM : NM, RL = MODE == 3 ? 2 * NL : NL;
  uint32_t f = (wave * 37u + blockIdx.x * 11u) % n_frag;
  uint4 v[RL];
#pragma unroll
  for (int j = 0; j < RL; j++) v[j] = make_uint4(0, 0, 0, 0);
  if (do_l) {
#pragma unroll
    for (int j = 0; j < RL; j++) { v[j] = buf[(uint64_t)f * 64 + lane]; f = f + 1 < n_frag ? f + 1 : 0; }
  }
  for (uint32_t it = 0; it < iters; it++) {
    uint4 vn[RL];
    if (do_l) {   // (wave-uniform)
#pragma unroll
      for (int j = 0; j < RL; j++) { vn[j] = buf[(uint64_t)f * 64 + lane]; f = f + 1 < n_frag ? f + 1 : 0; }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (do_m) {
#pragma unroll
      for (int m = 0; m < RM; m++) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 3], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (do_l) {
#pragma unroll
      for (int j = 0; j < RL; j++) { sink.x ^= v[j].x; sink.y ^= v[j].y; sink.z ^= v[j].z; sink.w ^= v[j].w; v[j] = vn[j]; }
    }
  }
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if (s == 12345.678f || sink.x == 0x12345678u) out[blockIdx.x * 512 + tid] = s + (float)sink.y;
}

template <int MODE, int NM, int NL>
float run(const uint4* buf, uint32_t n_frag, float* out, int blocks, uint32_t iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k_probe<MODE, NM, NL>), dim3(blocks), dim3(512), 0, 0, buf, n_frag, iters, out);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_probe<MODE, NM, NL>), dim3(blocks), dim3(512), 0, 0, buf, n_frag, iters, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f;
}

template <int NM, int NL>
void sweep(const uint4* buf, uint32_t n_frag, float* out) {
  const uint32_t iters = 2000;
  for (int blocks : {256, 512}) {
    const float t0 = run<0, NM, NL>(buf, n_frag, out, blocks, iters), t1 = run<1, NM, NL>(buf, n_frag, out, blocks, iters);
    const float t2 = run<2, NM, NL>(buf, n_frag, out, blocks, iters), t3 = run<3, NM, NL>(buf, n_frag, out, blocks, iters);
    const double wgs = blocks / 256.0;
    printf("%d MFMAs + %d loads per wave and iteration, %d workgroups of 8 waves: MFMAs only %7.1f us (%.0f cycles of pipe per SIMD and iteration), loads only %7.1f us "
           "(%.1f B/clk per unit @2.4 GHz), both in every wave %7.1f us (max %.1f, sum %.1f), split across waves %7.1f us\n",
           NM, NL, blocks, t0, 16.0 * NM * 2 * wgs, t1, 8.0 * NL * 1024 * iters * wgs / (t1 * 1e-6) / 2.4e9, t2, t0 > t1 ? t0 : t1, t0 + t1, t3);
  }
}

int main() {
  const uint32_t n_frag = 2048;   // 2 MB of 1 KB fragments
  uint4* buf; float* out;
  hipMalloc(&buf, (size_t)n_frag * 1024); hipMemset(buf, 1, (size_t)n_frag * 1024); hipMalloc(&out, 1024 * 512 * 4);
  sweep<16, 2>(buf, n_frag, out);   // the stack's GEMM calls in the single-term tier: 16 MFMAs per 2 KB of fragments and wave
  sweep<24, 4>(buf, n_frag, out);   // k_fc_r at 96 rows: 24 MFMAs, 4 fragments (+ 2 activation rows) per macro-step and wave
  sweep<32, 4>(buf, n_frag, out);   // ... at 128 rows
  sweep<32, 2>(buf, n_frag, out);   // twice the rows per weight fetch
  return 0;
}
