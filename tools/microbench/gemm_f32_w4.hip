// The float32 LDS-DMA GEMM's 256 x 256 tile on FOUR waves (128 x 128 per wave, 16 accumulator blocks of 32 x 32 = 256
// accumulator registers, one wave per SIMD) against the product's eight waves (64 x 128 per wave), in ONE process, rounds
// interleaved, outputs compared bit for bit (both add an element's products in ascending k order):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I islands_amd/csrc tools/microbench/gemm_f32_w4.hip -o tools/microbench/gemm_f32_w4
//   tools/microbench/gemm_f32_w4 [M N K]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_f32.hip.h"
using namespace isl_gemm;

template <int ACT, bool RES, int WM, int WN, int MF, int NF>
float run(const float* A, const float* W, const float* bias, const float* R, float* C, uint32_t M, uint32_t N, uint32_t K, int reps) {
  auto kern = gemm_tn_f32_dma<ACT, RES, WM, WN, MF, NF, 0>;
  constexpr size_t lds = 2 * (256 + 256) * FBK * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t ntn = (N + 255) / 256, ntiles = ((M + 255) / 256) * ntn, grid = ntiles > 256 ? 256 : ntiles;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, 0, A, W, bias, R, C, M, N, K, ntn, (uint64_t)N, ntiles);
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, 0, A, W, bias, R, C, M, N, K, ntn, (uint64_t)N, ntiles);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

__global__ void count_diff(const uint32_t* a, const uint32_t* b, uint64_t n, unsigned long long* out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long d = 0;
  for (; i < n; i += (uint64_t)gridDim.x * blockDim.x) d += a[i] != b[i];
  if (d) atomicAdd(out, d);
}
static unsigned long long differing(const float* a, const float* b, uint64_t n) {
  unsigned long long* d;
  (void)hipMalloc(&d, 8);
  (void)hipMemset(d, 0, 8);
  hipLaunchKernelGGL(count_diff, dim3(4096), dim3(256), 0, 0, (const uint32_t*)a, (const uint32_t*)b, n, d);
  unsigned long long h = 0;
  (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  return h;
}

int main(int argc, char** argv) {
  const uint32_t M = argc > 1 ? atoi(argv[1]) : 524288, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
  float *A, *W, *bias, *R, *C, *C2;
  (void)hipMalloc(&A, (size_t)M * K * 4);
  (void)hipMalloc(&W, (size_t)N * K * 4);
  (void)hipMalloc(&bias, N * 4);
  (void)hipMalloc(&R, (size_t)M * N * 4);
  (void)hipMalloc(&C, (size_t)M * N * 4);
  (void)hipMalloc(&C2, (size_t)M * N * 4);
  std::vector<float> h((size_t)M * K);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
  (void)hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(bias, h.data(), (size_t)N * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(R, h.data(), std::min<size_t>((size_t)M * N, h.size()) * 4, hipMemcpyHostToDevice);
  const double fl = 2.0 * M * N * K;
  for (int round = 0; round < 3; ++round) {
    const float t8 = run<0, false, 2, 4, 4, 2>(A, W, bias, nullptr, C, M, N, K, 5);
    const float t4 = run<0, false, 2, 2, 4, 4>(A, W, bias, nullptr, C2, M, N, K, 5);
    const float r8 = run<0, true, 2, 4, 4, 2>(A, W, bias, R, C, M, N, K, 5);
    const float r4 = run<0, true, 2, 2, 4, 4>(A, W, bias, R, C2, M, N, K, 5);
    const float g8 = run<1, false, 2, 4, 4, 2>(A, W, bias, nullptr, C, M, N, K, 5);
    const float g4 = run<1, false, 2, 2, 4, 4>(A, W, bias, nullptr, C2, M, N, K, 5);
    printf("M=%u N=%u K=%u round %d: bias: 8 waves %.3f ms %.1f TF | 4 waves %.3f ms %.1f TF || bias + residual: %.1f | %.1f TF || bias + GELU: %.1f | %.1f TF\n",
           M, N, K, round, t8, fl / t8 / 1e9, t4, fl / t4 / 1e9, fl / r8 / 1e9, fl / r4 / 1e9, fl / g8 / 1e9, fl / g4 / 1e9);
    fflush(stdout);
  }
  (void)run<0, false, 2, 4, 4, 2>(A, W, bias, nullptr, C, M, N, K, 1);
  (void)run<0, false, 2, 2, 4, 4>(A, W, bias, nullptr, C2, M, N, K, 1);
  printf("bias-only outputs, differing elements of %llu: %llu\n", (unsigned long long)M * N, differing(C, C2, (uint64_t)M * N));
  (void)run<0, true, 2, 4, 4, 2>(A, W, bias, R, C, M, N, K, 1);
  (void)run<0, true, 2, 2, 4, 4>(A, W, bias, R, C2, M, N, K, 1);
  printf("bias + residual outputs, differing elements of %llu: %llu\n", (unsigned long long)M * N, differing(C, C2, (uint64_t)M * N));
  (void)run<1, false, 2, 4, 4, 2>(A, W, bias, nullptr, C, M, N, K, 1);
  (void)run<1, false, 2, 2, 4, 4>(A, W, bias, nullptr, C2, M, N, K, 1);
  printf("bias + GELU outputs, differing elements: %llu\n", differing(C, C2, (uint64_t)M * N));
  return 0;
}
