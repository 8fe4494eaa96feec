// Probe (round 6): which of a compute unit's three streams serialise — MFMAs (register operands), LDS fragment reads (ds_read_b128, conflict-free) and L2 -> register loads
// (1 KB per wave and load)?  k_fc_r needs ~1300-1600 cycles of each per round of its sixteen waves and takes ~4500.  Per iteration a wave issues NM MFMAs,
// ND ds_read_b128 and NL loads; bit mask MODE: 1 MFMAs, 2 LDS reads, 4 loads.  Two workgroups of 8 waves per unit (k_fc_r's shape).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_vmem_overlap mfma_lds_vmem_overlap.hip ; run: ./mfma_lds_vmem_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NM, int ND, int NL>
__global__ __launch_bounds__(512, 4) void k_probe(const uint4* __restrict__ buf, uint32_t n_frag, uint32_t iters, float* out) {
  __shared__ uint4 s_buf[2048];   // 32 KB
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (uint32_t i = tid; i < 2048; i += 512) s_buf[i] = make_uint4(i, i + 1, i + 2, i + 3);
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  half8 a, b;
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * (float)(lane + i)); b[i] = (_Float16)(0.002f * (float)(lane ^ i)); }
  uint4 sink = make_uint4(0, 0, 0, 0);
  uint32_t f = (wave * 37u + blockIdx.x * 11u) % n_frag;
  uint32_t dpos = (wave * 64u + lane) & 2047u;
  for (uint32_t it = 0; it < iters; it++) {
    uint4 vn[NL > 0 ? NL : 1], dn[ND > 0 ? ND : 1];
    if (MODE & 4) {
#pragma unroll
      for (int j = 0; j < NL; j++) { vn[j] = buf[(uint64_t)f * 64 + lane]; f = f + 1 < n_frag ? f + 1 : 0; }
    }
    if (MODE & 2) {
#pragma unroll
      for (int j = 0; j < ND; j++) { dn[j] = s_buf[dpos]; dpos = (dpos + 64u) & 2047u; }   // consecutive lanes, consecutive 16 bytes: conflict-free
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE & 1) {
#pragma unroll
      for (int m = 0; m < NM; m++) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 3], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE & 4) {
#pragma unroll
      for (int j = 0; j < NL; j++) { sink.x ^= vn[j].x; sink.y ^= vn[j].y; sink.z ^= vn[j].z; sink.w ^= vn[j].w; }
    }
    if (MODE & 2) {
#pragma unroll
      for (int j = 0; j < ND; j++) { sink.x ^= dn[j].x; sink.y ^= dn[j].y; sink.z ^= dn[j].z; sink.w ^= dn[j].w; }
    }
  }
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if (s == 12345.678f || sink.x == 0x12345678u) out[blockIdx.x * 512 + tid] = s + (float)sink.y;
}

template <int MODE, int NM, int ND, int NL>
float run(const uint4* buf, uint32_t n_frag, float* out, uint32_t iters) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k_probe<MODE, NM, ND, NL>), dim3(512), dim3(512), 0, 0, buf, n_frag, iters, out);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k_probe<MODE, NM, ND, NL>), dim3(512), dim3(512), 0, 0, buf, n_frag, iters, out);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters * 2400.f;   // cycles @2.4 GHz per iteration (= per round of the unit's sixteen waves)
}

template <int NM, int ND, int NL>
void sweep(const uint4* buf, uint32_t n_frag, float* out) {
  const uint32_t iters = 2000;
  const float m = run<1, NM, ND, NL>(buf, n_frag, out, iters), d = run<2, NM, ND, NL>(buf, n_frag, out, iters), l = run<4, NM, ND, NL>(buf, n_frag, out, iters);
  const float md = run<3, NM, ND, NL>(buf, n_frag, out, iters), ml = run<5, NM, ND, NL>(buf, n_frag, out, iters), dl = run<6, NM, ND, NL>(buf, n_frag, out, iters);
  const float all = run<7, NM, ND, NL>(buf, n_frag, out, iters);
  printf("%2d MFMAs, %2d ds_read_b128, %d loads per wave and round, 2 workgroups of 8 waves per unit — cycles @2.4 GHz per round: MFMAs %5.0f | LDS %5.0f | loads %5.0f | "
         "MFMAs + LDS %5.0f | MFMAs + loads %5.0f | LDS + loads %5.0f | all three %5.0f (sum %5.0f)\n", NM, ND, NL, m, d, l, md, ml, dl, all, m + d + l);
}

int main() {
  const uint32_t n_frag = 2048;
  uint4* buf; float* out;
  (void)hipMalloc(&buf, (size_t)n_frag * 1024); (void)hipMemset(buf, 1, (size_t)n_frag * 1024); (void)hipMalloc(&out, 1024 * 512 * 4);
  sweep<20, 10, 6>(buf, n_frag, out);   // k_fc_r at 80 rows: per 64-k macro-step 20 MFMAs, 10 fragment reads, 4 weight fragments + 2 activation pieces
  sweep<32, 16, 6>(buf, n_frag, out);   // ... at 128 rows
  sweep<32, 8, 12>(buf, n_frag, out);   // the 2 x 4 wave grid at 128 rows
  sweep<16, 8, 2>(buf, n_frag, out);    // a GEMM call of the encoder stack per k-step pair
  return 0;
}
