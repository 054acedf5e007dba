// FETCH_SIZE calibration and access-pattern ceiling for the search kernel's row reads: every
// quad of lanes reads one 3072-byte row of a 4 GiB buffer (>> the 256 MiB Infinity Cache), rows
// visited once each in a scattered order.  MODE 0: 48 steps of 64 contiguous bytes per quad
// (16 B per lane, what direct_group does); MODE 1: 24 steps of 128 contiguous bytes per quad
// (two adjacent 16-byte loads per lane); MODE 2: 12 steps of 256 bytes (four loads per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(64) void rows_once(const float* __restrict__ buf, uint32_t nrows, uint32_t iters, float* out) {
  const uint32_t lane = threadIdx.x, quad = lane >> 2, s = lane & 3;
  float acc = 0.f;
  constexpr int PER = MODE == 0 ? 1 : (MODE == 1 ? 2 : 4);  // float4 loads per lane and step
  for (uint32_t it = 0; it < iters; ++it) {
    const uint64_t idx = ((uint64_t)blockIdx.x * iters + it) * 16 + quad;
    const uint32_t row = (uint32_t)((idx * 104729ull) % nrows);
    const float* rp = buf + (uint64_t)row * 768 + s * 4 * PER;
#pragma unroll 12
    for (int st = 0; st < 48 / PER; ++st) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        float4 x = *reinterpret_cast<const float4*>(rp + 16 * PER * st + 4 * u);
        acc += x.x + x.y + x.z + x.w;
      }
    }
  }
  out[blockIdx.x * 64 + lane] = acc;
}
template <int MODE> void run(const float* buf, uint32_t nrows, uint32_t grid, uint32_t iters, float* out, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rows_once<MODE>, dim3(grid), dim3(64), 0, 0, buf, nrows, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s grid %5u: %.3f GB in %.3f ms = %.1f GB/s\n", name, grid, nrows * 3072.0 / 1e9, ms, nrows * 3072.0 / ms / 1e6);
  }
}
int main() {
  const uint32_t nrows = 4096 * 20 * 16;
  float *buf, *out;
  if (hipMalloc(&buf, (size_t)nrows * 3072 + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMalloc(&out, 16384 * 64 * 4);
  (void)hipMemset(buf, 0, (size_t)nrows * 3072);
  (void)hipDeviceSynchronize();
  for (uint32_t grid : {2048u, 4096u, 8192u}) {
    const uint32_t iters = nrows / 16 / grid;
    run<0>(buf, nrows, grid, iters, out, "64 B per quad and step");
    run<1>(buf, nrows, grid, iters, out, "128 B per quad and step");
    run<2>(buf, nrows, grid, iters, out, "256 B per quad and step");
  }
  return 0;
}
