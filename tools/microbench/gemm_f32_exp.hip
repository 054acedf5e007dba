// A/B of the float32 LDS-DMA GEMM's pipeline variants in ONE process (devices differ by several
// per cent on MFMA-dense loops, so variants are only comparable on the same card, interleaved):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I islands_amd/csrc tools/microbench/gemm_f32_exp.hip -o tools/microbench/gemm_f32_exp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_f32.hip.h"
using namespace isl_gemm;

// EXP bit 0: per-element epilogue stores, bit 1: no staggered start; persistent: grid = CUs instead of one
// workgroup per tile
template <int EXP>
float run(bool persistent, const float* A, const float* W, const float* bias, float* C, uint32_t M, uint32_t N, uint32_t K, int reps) {
  auto kern = gemm_tn_f32_dma<0, false, 2, 4, 4, 2, EXP>;
  constexpr size_t lds = 2 * (256 + 256) * FBK * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t ntn = (N + 255) / 256, ntiles = ((M + 255) / 256) * ntn, grid = persistent && ntiles > 256 ? 256 : ntiles;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, A, W, bias, nullptr, C, M, N, K, ntn, (uint64_t)N, ntiles);
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, A, W, bias, nullptr, C, M, N, K, ntn, (uint64_t)N, ntiles);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

float run_round1(const float* A, const float* W, const float* bias, float* C, uint32_t M, uint32_t N, uint32_t K, int reps) {
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((gemm_tn_f32<0, false>), grid, dim3(256), 0, 0, A, W, bias, nullptr, C, M, N, K);
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((gemm_tn_f32<0, false>), grid, dim3(256), 0, 0, A, W, bias, nullptr, C, M, N, K);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const uint32_t M = argc > 1 ? atoi(argv[1]) : 524288, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
  float *A, *W, *bias, *C, *C2;
  hipMalloc(&A, (size_t)M * K * 4);
  hipMalloc(&W, (size_t)N * K * 4);
  hipMalloc(&bias, N * 4);
  hipMalloc(&C, (size_t)M * N * 4);
  hipMalloc(&C2, (size_t)M * N * 4);
  std::vector<float> h((size_t)M * K);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
  hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4);
  const double fl = 2.0 * M * N * K;
  for (int round = 0; round < 3; ++round) {
    const float tr = run_round1(A, W, bias, C2, M, N, K, 5);  // (a different k order: its bits differ)
    const float t0 = run<3>(true, A, W, bias, C, M, N, K, 5);
    const float t1 = run<2>(true, A, W, bias, C2, M, N, K, 5);
    const float t2 = run<1>(true, A, W, bias, C2, M, N, K, 5);
    const float t3 = run<0>(true, A, W, bias, C2, M, N, K, 5);
    printf("round-1 kernel (register-staged 128x128) %.3f ms %.1f TF\n", tr, fl / tr / 1e9);
    printf("M=%u N=%u K=%u persistent: scalar-epi %.3f ms %.1f TF | wide-epi %.3f ms %.1f TF | staggered scalar %.3f ms %.1f TF | staggered wide %.3f ms %.1f TF\n",
           M, N, K, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, t3, fl / t3 / 1e9);
  }
  // same bits from every variant
  std::vector<float> c1(1 << 20), c2(1 << 20);
  hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost);
  size_t diff = 0;
  for (size_t i = 0; i < c1.size(); ++i) diff += c1[i] != c2[i];
  printf("differing elements among the first 2^20: %zu\n", diff);
  return 0;
}
