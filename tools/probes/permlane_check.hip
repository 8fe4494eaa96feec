#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float add_xor16(float v) {   // v + shfl_xor(v, 16)
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float add_xor32(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k(float* o, const float* in) {
  float v = in[threadIdx.x];
  o[threadIdx.x] = add_xor32(add_xor16(v));
  float w = v + __shfl_xor(v, 16, 64);
  o[64 + threadIdx.x] = w + __shfl_xor(w, 32, 64);
}
int main() {
  float h[64], *d, *o, r[128];
  for (int i = 0; i < 64; i++) h[i] = (float)(i * i % 37) + 0.25f * i;
  hipMalloc(&d, 256); hipMalloc(&o, 512);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d);
  hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; i++) if (r[i] != r[64 + i]) bad++;
  printf("permlane check: %d mismatches (r0 %f vs %f)\n", bad, r[0], r[64]);
  return bad;
}
