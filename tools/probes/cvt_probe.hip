#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef short v2s __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, const float* sc, uint32_t* out, int n) {
  int i = threadIdx.x;
  if (i < n) {
    v2s old = {0, 0};
    v2s r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, in[i], -in[i], sc[i], false);
    v2h hh = {(_Float16)in[i], (_Float16)(-in[i])};
    v2s r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, hh, sc[i], false);
    out[2 * i] = (uint32_t)(uint16_t)r[0];
    out[2 * i + 1] = (uint32_t)(uint16_t)r2[0];
  }
}
// rate of the conversions: dependent-free streams
__global__ void k_rate(float* out, int iters, float s) {
  float a = threadIdx.x * 0.01f, b = a + 1.f;
  int w0 = 0, w1 = 0, w2 = 0, w3 = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w0, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(b, a, w1, true);
    w2 = __builtin_amdgcn_cvt_pk_fp8_f32(a, a, w2, false);
    w3 = __builtin_amdgcn_cvt_pk_fp8_f32(b, b, w3, true);
    a += s; b += s;
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = (float)(w0 ^ w1 ^ w2 ^ w3); out[1] = (float)(t1 - t0) / (float)iters; }
}
__global__ void k_rate2(float* out, int iters, float s) {
  float a = threadIdx.x * 0.01f, b = a + 1.f;
  float c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    c0 = __builtin_amdgcn_fmed3f(a, -448.f, c0);
    c1 = __builtin_amdgcn_fmed3f(b, -448.f, c1);
    c2 = __builtin_amdgcn_fmed3f(a, c2, 448.f);
    c3 = __builtin_amdgcn_fmed3f(b, c3, 448.f);
    a += s; b += s;
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = c0 + c1 + c2 + c3; out[1] = (float)(t1 - t0) / (float)iters; }
}
int main() {
  const int n = 14;
  float hv[n] = {1.f, 1.f, 1.f, 3.3f, 448.f, 500.f, 1000.f, 1e6f, 0.001f, 0.1f, 100.f, 100.f, INFINITY, 2.0f};
  float hs[n] = {1.f, 2.f, 0.5f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.125f, 0.125f, 8.f, 1.f, 0.000122070312f};
  float *dv, *ds; uint32_t* dout; uint32_t ho[2 * n];
  hipMalloc(&dv, 4 * n); hipMalloc(&ds, 4 * n); hipMalloc(&dout, 8 * n);
  hipMemcpy(dv, hv, 4 * n, hipMemcpyHostToDevice); hipMemcpy(ds, hs, 4 * n, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dv, ds, dout, n);
  hipMemcpy(ho, dout, 8 * n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; i++) printf("scalef32 x %g scale %g -> f32 src: +0x%02x -0x%02x | f16 src: +0x%02x -0x%02x\n", hv[i], hs[i], ho[2 * i] & 0xff, (ho[2 * i] >> 8) & 0xff, ho[2 * i + 1] & 0xff, (ho[2 * i + 1] >> 8) & 0xff);
  float* dr; float hr[2]; hipMalloc(&dr, 8);
  k_rate<<<1, 64>>>(dr, 4000, 0.001f); hipMemcpy(hr, dr, 8, hipMemcpyDeviceToHost); printf("4 cvt_pk_fp8_f32 + 2 add: %.1f cycles per iteration (one wave)\n", hr[1]);
  k_rate2<<<1, 64>>>(dr, 4000, 0.001f); hipMemcpy(hr, dr, 8, hipMemcpyDeviceToHost); printf("4 med3 + 2 add: %.1f cycles per iteration (one wave)\n", hr[1]);
  return 0;
}
