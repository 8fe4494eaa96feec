// Probe: how fast can ONE CU pull L2-resident data into registers, as a function of the number of
// 16-byte-per-lane loads each wave keeps in flight?  Every workgroup streams the same 3 MiB buffer
// (the fused transformer stack's situation: all CUs read the same layer weights).
// build: hipcc --offload-arch=gfx950 -O3 -o l2_stream l2_stream.hip ; run: ./l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int DEPTH>
__global__ __launch_bounds__(512) void k_stream(const uint4* __restrict__ buf, uint32_t n16, uint32_t passes, uint4* out) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  const uint32_t tid = threadIdx.x;
  for (uint32_t p = 0; p < passes; p++)
    for (uint32_t i = tid; i + (DEPTH - 1) * 512 < n16; i += DEPTH * 512) {
      uint4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; d++) v[d] = buf[i + d * 512];
#pragma unroll
      for (int d = 0; d < DEPTH; d++) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
    }
  if (acc.x == 0x12345678u) out[blockIdx.x * 512 + tid] = acc;
}
template <int DEPTH>
void run(const uint4* buf, uint32_t n16, uint4* out, int blocks) {
  const uint32_t passes = 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_stream<DEPTH>, dim3(blocks), dim3(512), 0, 0, buf, n16, passes, out);
  hipEventRecord(a);
  hipLaunchKernelGGL(k_stream<DEPTH>, dim3(blocks), dim3(512), 0, 0, buf, n16, passes, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)n16 * 16 * passes;  // per workgroup
  printf("blocks %4d depth %2d: %8.1f us  per-CU %6.1f GB/s = %5.1f B/clk @2.4GHz  aggregate %6.2f TB/s\n", blocks, DEPTH, ms * 1e3,
         bytes / (ms * 1e-3) / 1e9 * (blocks > 256 ? blocks / 256.0 : 1.0), bytes / (ms * 1e-3) / 2.4e9 * (blocks > 256 ? blocks / 256.0 : 1.0),
         bytes * blocks / (ms * 1e-3) / 1e12);
}
int main() {
  const uint32_t n16 = 3u << 16;  // 3 MiB
  uint4 *buf, *out;
  hipMalloc(&buf, n16 * 16); hipMemset(buf, 1, n16 * 16); hipMalloc(&out, 1024 * 512 * 16);
  for (int blocks : {256, 512}) {
    run<1>(buf, n16, out, blocks); run<2>(buf, n16, out, blocks); run<4>(buf, n16, out, blocks);
    run<8>(buf, n16, out, blocks); run<16>(buf, n16, out, blocks); run<32>(buf, n16, out, blocks);
  }
  return 0;
}
