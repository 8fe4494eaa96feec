#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_ref(float a, float b, uint32_t& hi, uint32_t& lo) {
  const float2v v = {a, b};
  const half2v h = __builtin_convertvector(v, half2v);
  hi = __builtin_bit_cast(uint32_t, h);
  const float2v hf = __builtin_convertvector(h, float2v);
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - hf, half2v));
}
__device__ __forceinline__ void split_mix(float a, float b, uint32_t& hi, uint32_t& lo) {
  const float2v v = {a, b};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, half2v));
  uint32_t l = 0;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(b));
  lo = l;
}
__global__ void k(const float* in, uint32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    uint32_t h0, l0, h1, l1;
    split_ref(in[2 * i], in[2 * i + 1], h0, l0);
    split_mix(in[2 * i], in[2 * i + 1], h1, l1);
    out[4 * i] = h0; out[4 * i + 1] = l0; out[4 * i + 2] = h1; out[4 * i + 3] = l1;
  }
}
int main() {
  const int n = 1 << 20;
  float* h = (float*)malloc(8 * n);
  srand(5);
  for (int i = 0; i < 2 * n; i++) {
    const int e = rand() % 40 - 28;   // magnitudes 2^-28 .. 2^11: remainders deep in the f16 denormal range included
    float x = ldexpf((float)rand() / RAND_MAX + 1.0f, e);
    if (rand() & 1) x = -x;
    if (i % 97 == 0) x = 0.f;
    if (i % 101 == 0) x = 65504.f * ((rand() & 1) ? 1 : -1);
    h[i] = x;
  }
  float* d; uint32_t* o; uint32_t* ho = (uint32_t*)malloc(16 * n);
  hipMalloc(&d, 8 * n); hipMalloc(&o, 16 * n);
  hipMemcpy(d, h, 8 * n, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(d, o, n);
  hipMemcpy(ho, o, 16 * n, hipMemcpyDeviceToHost);
  long bad_hi = 0, bad_lo = 0; int shown = 0;
  for (int i = 0; i < n; i++) {
    if (ho[4 * i] != ho[4 * i + 2]) bad_hi++;
    if (ho[4 * i + 1] != ho[4 * i + 3]) { bad_lo++; if (shown++ < 5) printf("x = %g, %g: hi %08x lo ref %08x mix %08x\n", h[2 * i], h[2 * i + 1], ho[4 * i], ho[4 * i + 1], ho[4 * i + 3]); }
  }
  printf("pairs %d: hi differs %ld, lo differs %ld\n", n, bad_hi, bad_lo);
  return 0;
}
