#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
// per iteration: NF16 groups of 8 f16 MFMAs, then NMX groups of 8 MX MFMAs on the same accumulators; out[wave] = cycles per iteration as seen by each wave
template <int NF16, int NMX>
__global__ void k_mix(float* out, int iters, int sa, int sb) {
  half8 a, b;
  for (int d = 0; d < 8; d++) { a[d] = (_Float16)(0.001f * threadIdx.x); b[d] = (_Float16)0.25f; }
  v8i a8, b8;
  for (int d = 0; d < 8; d++) { a8[d] = 0x38383838 + (threadIdx.x & 3); b8[d] = 0x30303030; }
  f32x4 c[8];
  for (int i = 0; i < 8; i++) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < NF16; g++) {
#pragma unroll
      for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = 0; g < NMX; g++) {
#pragma unroll
      for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c[i], 0, 0, 0, sa, 0, sb);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; i++) s += c[i][0];
  if ((threadIdx.x & 63) == 0) { out[(threadIdx.x >> 6) * 2] = (float)(t1 - t0) / (float)iters; out[(threadIdx.x >> 6) * 2 + 1] = s; }
}
template <int NF16, int NMX>
void run(const char* what, int threads, int sa, int sb, int blocks = 1, int iters = 500) {
  float* d; float h[16] = {0};
  hipMalloc(&d, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_mix<NF16, NMX><<<blocks, threads>>>(d, iters, sa, sb);   // warm
  hipEventRecord(e0);
  k_mix<NF16, NMX><<<blocks, threads>>>(d, iters, sa, sb);   // (every block writes the same slots: the last one's figures stay)
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  printf("%-18s %4d blocks x %d waves: counter ticks per iteration, wave 0..: ", what, blocks, threads / 64);
  for (int w = 0; w < threads / 64; w++) printf("%.0f ", h[2 * w]);
  printf("| kernel %.3f ms = %.1f ns per iteration of the slowest wave -> %.2f ticks per ns\n", ms, 1e6 * ms / iters, h[2 * (threads / 64 - 1)] / (1e6 * ms / iters));
  hipFree(d);
}
int main() {
  run<8, 0>("64 f16", 64, 0, 0);
  run<8, 0>("64 f16", 512, 0, 0);
  run<16, 0>("128 f16", 512, 0, 0);
  run<0, 2>("16 mx", 64, 0x7f7f7f7f, 0x7f7f7f7f);
  run<0, 2>("16 mx", 512, 0x7f7f7f7f, 0x7f7f7f7f);
  run<0, 2>("16 mx", 512, 0x78787878, 0x72727272);
  run<8, 2>("64 f16 + 16 mx", 64, 0x78787878, 0x72727272);
  run<8, 2>("64 f16 + 16 mx", 512, 0x78787878, 0x72727272);
  run<8, 2>("64 f16 + 16 mx", 512, 0x7f7f7f7f, 0x7f7f7f7f);
  // the whole chip busy: does the counter follow the shader clock, and does the clock hold?
  run<16, 0>("128 f16", 512, 0, 0, 1, 20000);
  run<16, 0>("128 f16", 512, 0, 0, 256, 20000);
  run<16, 0>("128 f16", 512, 0, 0, 1024, 5000);
  return 0;
}
