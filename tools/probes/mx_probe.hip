#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_cvt(const float* in, uint32_t* out, int n) {
  int i = threadIdx.x;
  if (i < n) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], -in[i], 0, false);
    out[i] = (uint32_t)w;
  }
}
// A[16][128] row-major floats, B[128][16] (k-major) floats -> D[16][16]; lane l: row/col l%16, k = (l/16)*32 + j, byte j of the lane's 32 bytes
__global__ void k_mx(const float* A, const float* B, float* D, int sa, int sb) {
  const int l = threadIdx.x, rc = l & 15, kb = l >> 4;
  v8i a, b;
  for (int d = 0; d < 8; d++) {
    int wa = 0, wb = 0;
    const int k = kb * 32 + d * 4;
    wa = __builtin_amdgcn_cvt_pk_fp8_f32(A[rc * 128 + k], A[rc * 128 + k + 1], wa, false);
    wa = __builtin_amdgcn_cvt_pk_fp8_f32(A[rc * 128 + k + 2], A[rc * 128 + k + 3], wa, true);
    wb = __builtin_amdgcn_cvt_pk_fp8_f32(B[k * 16 + rc], B[(k + 1) * 16 + rc], wb, false);
    wb = __builtin_amdgcn_cvt_pk_fp8_f32(B[(k + 2) * 16 + rc], B[(k + 3) * 16 + rc], wb, true);
    a[d] = wa; b[d] = wb;
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa * 0x01010101, 0, sb * 0x01010101);
  for (int r = 0; r < 4; r++) D[(4 * kb + r) * 16 + rc] = c[r];   // row = 4 * (l / 16) + r (A's index), col = l % 16 (B's index)
}
// throughput: one wave, N dependent-free MFMAs
__global__ void k_rate(float* out, int iters) {
  v8i a, b;
  for (int d = 0; d < 8; d++) { a[d] = 0x38383838 + threadIdx.x; b[d] = 0x38383838; }
  f32x4 c[8];
  for (int i = 0; i < 8; i++) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; i++) s += c[i][0];
  if (threadIdx.x == 0) { out[0] = s; out[1] = (float)(t1 - t0) / (8.f * iters); }
}
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
__global__ void k_rate16(float* out, int iters) {
  half8 a, b;
  for (int d = 0; d < 8; d++) { a[d] = (_Float16)(0.5f + threadIdx.x); b[d] = (_Float16)0.25f; }
  f32x4 c[8];
  for (int i = 0; i < 8; i++) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; i++) s += c[i][0];
  if (threadIdx.x == 0) { out[0] = s; out[1] = (float)(t1 - t0) / (8.f * iters); }
}

int main() {
  float hv[16] = {0.f, 1.f, 0.5f, 1.75f, 448.f, 449.f, 480.f, 500.f, 1e6f, 0.001953125f, 0.0009765625f, 0.0176f, 3.3f, 0.1f, 240.f, INFINITY};
  float* dv; uint32_t* dout; uint32_t ho[16];
  hipMalloc(&dv, 64); hipMalloc(&dout, 64);
  hipMemcpy(dv, hv, 64, hipMemcpyHostToDevice);
  k_cvt<<<1, 64>>>(dv, dout, 16);
  hipMemcpy(ho, dout, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; i++) printf("cvt %g -> +0x%02x -0x%02x\n", hv[i], ho[i] & 0xff, (ho[i] >> 8) & 0xff);
  static float A[16 * 128], B[128 * 16], D[256], R[256];
  for (int i = 0; i < 16; i++) for (int k = 0; k < 128; k++) A[i * 128 + k] = (float)((i * 7 + k * 3) % 5 - 2) * 0.5f;
  for (int k = 0; k < 128; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = (float)((k * 5 + j * 11) % 7 - 3) * 0.25f;
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dD, sizeof D);
  hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
  const int cases[3][2] = {{127, 127}, {124, 127}, {127, 130}};
  for (auto& cs : cases) {
    k_mx<<<1, 64>>>(dA, dB, dD, cs[0], cs[1]);
    hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
    const float sc = ldexpf(1.f, cs[0] - 127 + cs[1] - 127);
    double worst = 0; int bad = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
      float r = 0; for (int k = 0; k < 128; k++) r += A[i * 128 + k] * B[k * 16 + j];
      R[i * 16 + j] = r * sc;
      const double e = fabs(D[i * 16 + j] - R[i * 16 + j]); if (e > worst) worst = e; if (e > 1e-6) bad++;
    }
    printf("mx scale_a %d scale_b %d: worst abs err %g, bad %d / 256; D[3][5] %g ref %g; D[5][3] %g ref %g\n", cs[0], cs[1], worst, bad, D[3 * 16 + 5], R[3 * 16 + 5], D[5 * 16 + 3], R[5 * 16 + 3]);
  }
  float* dr; float hr[2]; hipMalloc(&dr, 8);
  k_rate<<<1, 64>>>(dr, 2000); hipMemcpy(hr, dr, 8, hipMemcpyDeviceToHost); printf("mx fp8 16x16x128: %.1f cycles per MFMA (one wave)\n", hr[1]);
  k_rate16<<<1, 64>>>(dr, 2000); hipMemcpy(hr, dr, 8, hipMemcpyDeviceToHost); printf("f16 16x16x32: %.1f cycles per MFMA (one wave)\n", hr[1]);
  k_rate<<<1, 128>>>(dr, 2000); hipMemcpy(hr, dr, 8, hipMemcpyDeviceToHost); printf("mx fp8, 2 waves in the workgroup: %.1f cycles per MFMA per wave\n", hr[1]);
  k_rate16<<<1, 512>>>(dr, 2000); hipMemcpy(hr, dr, 8, hipMemcpyDeviceToHost); printf("f16, 8 waves (2 per SIMD): %.1f cycles per MFMA per wave\n", hr[1]);
  k_rate<<<1, 512>>>(dr, 2000); hipMemcpy(hr, dr, 8, hipMemcpyDeviceToHost); printf("mx fp8, 8 waves (2 per SIMD): %.1f cycles per MFMA per wave\n", hr[1]);
  return 0;
}
