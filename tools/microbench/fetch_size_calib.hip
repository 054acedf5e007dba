// FETCH_SIZE calibration for the search kernel's access pattern: every quad of lanes reads one
// 3072-byte row in 48 steps of 64 contiguous bytes (16 B per lane), rows visited once each in a
// scattered order, 4.03 GB in total out of a 4 GiB buffer (>> the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(64) void rows_once(const float* __restrict__ buf, uint32_t nrows, uint32_t iters, float* out) {
  const uint32_t lane = threadIdx.x, quad = lane >> 2, s4 = (lane & 3) * 4;
  float acc = 0.f;
  for (uint32_t it = 0; it < iters; ++it) {
    const uint64_t idx = ((uint64_t)blockIdx.x * iters + it) * 16 + quad;
    const uint32_t row = (uint32_t)((idx * 104729ull) % nrows);
    const float* rp = buf + (uint64_t)row * 768 + s4;
#pragma unroll 12
    for (int st = 0; st < 48; ++st) {
      float4 x = *reinterpret_cast<const float4*>(rp + 16 * st);
      acc += x.x + x.y + x.z + x.w;
    }
  }
  out[blockIdx.x * 64 + lane] = acc;
}
int main() {
  const uint32_t grid = 4096, iters = 20, nrows = grid * iters * 16;
  float *buf, *out;
  if (hipMalloc(&buf, (size_t)nrows * 3072 + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMalloc(&out, grid * 64 * 4);
  (void)hipMemset(buf, 0, (size_t)nrows * 3072);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rows_once, dim3(grid), dim3(64), 0, 0, buf, nrows, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("rows_once: %.3f GB in %.3f ms = %.1f GB/s\n", nrows * 3072.0 / 1e9, ms, nrows * 3072.0 / ms / 1e6);
  }
  return 0;
}
