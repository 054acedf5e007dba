// A/B of the bf16 LDS-DMA GEMM's variants in ONE process on one card (config-5 shape by default):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I islands_amd/csrc tools/microbench/gemm_bf16_exp.hip -o tools/microbench/gemm_bf16_exp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_bf16.hip.h"
using namespace isl_gemm;

static const float* g_rn = nullptr;  // |w_n|^2 per column / |a_m|^2 per row for the cosine epilogue
static const float* g_qn = nullptr;
template <int EXP, int EPI = EPI_DOT>
float run(const __bf16* A, const __bf16* W, float* C, uint32_t M, uint32_t N, uint32_t K, int reps, uint64_t* dbg) {
  auto kern = gemm_tn_bf16_dma<EPI, false, false, 2, 4, 4, 2, EXP>;
  constexpr size_t lds = 2 * (256 + 256) * HBK * 2;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t ntn = (N + 255) / 256, ntiles = ((M + 255) / 256) * ntn;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(ntiles), dim3(512), lds, 0, A, W, g_rn, g_qn, C, M, N, K, ntn, (uint64_t)N, dbg);
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(512), lds, 0, A, W, g_rn, g_qn, C, M, N, K, ntn, (uint64_t)N, dbg);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const uint32_t M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 65536, K = argc > 3 ? atoi(argv[3]) : 4096;
  __bf16 *A, *W;
  float *C, *C2;
  uint64_t* dbg;
  (void)hipMalloc(&A, (size_t)M * K * 2);
  (void)hipMalloc(&W, (size_t)N * K * 2);
  (void)hipMalloc(&C, (size_t)M * N * 4);
  (void)hipMalloc(&C2, (size_t)M * N * 4);
  (void)hipMalloc(&dbg, 256);
  (void)hipMemset(dbg, 0, 256);
  std::vector<uint16_t> h((size_t)N * K);
  uint32_t s = 12345;
  // uniform in (-1, 1), truncated to bf16: exponents and mantissas vary like real rows (the clock a
  // chip holds under MFMA load depends on the operand bits)
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    float f = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
    if (argc > 4) f *= 0.0156f * 1.7f;  // the scale of L2-normalised rows of d = 4096
    uint32_t u;
    memcpy(&u, &f, 4);
    v = (uint16_t)(u >> 16);
  }
  (void)hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  const double fl = 2.0 * M * N * K;
  float *rn, *qn;
  (void)hipMalloc(&rn, (size_t)N * 4);
  (void)hipMalloc(&qn, (size_t)M * 4);
  {
    std::vector<float> ones(N > M ? N : M, (float)K / 3.0f);
    (void)hipMemcpy(rn, ones.data(), (size_t)N * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(qn, ones.data(), (size_t)M * 4, hipMemcpyHostToDevice);
  }
  g_rn = rn;
  g_qn = qn;
  for (int round = 0; round < 3; ++round) {
    const float t0 = run<1>(A, W, C, M, N, K, 40, dbg);
    const float t1 = run<8>(A, W, C2, M, N, K, 40, dbg);
    const float t2 = run<0>(A, W, C2, M, N, K, 40, dbg);
    const float t3 = run<12>(A, W, C2, M, N, K, 40, dbg);

    printf("M=%u N=%u K=%u  all behind the barrier %.3f ms %.1f TF | waves 4..7 at k-step 2 %.3f ms %.1f TF | at 1 %.3f ms %.1f TF | at 3 %.3f ms %.1f TF\n", M, N, K,
           t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, t3, fl / t3 / 1e9);
  }
  auto stamps = [&](const char* what) {
    uint64_t hd[16];
    (void)hipMemcpy(hd, dbg, sizeof(hd), hipMemcpyDeviceToHost);
    printf("%s: workgroup 300, mean s_memtime ticks per slab\n", what);
    for (int w = 0; w < 2; ++w) {
      const double n = hd[w * 8 + 4] ? (double)hd[w * 8 + 4] : 1.0;
      printf("  wave %d: wait vmcnt %.0f | barrier %.0f | DMA issue behind the barrier %.0f | ds_read + MFMA (+ late DMA) issue %.0f\n", w * 4, hd[w * 8] / n,
             hd[w * 8 + 1] / n, hd[w * 8 + 2] / n, hd[w * 8 + 3] / n);
    }
  };
  (void)run<129>(A, W, C2, M, N, K, 2, dbg);
  stamps("all behind the barrier");
  (void)run<128>(A, W, C2, M, N, K, 2, dbg);
  stamps("waves 4..7 at k-step 1 (the product's arrangement)");

  std::vector<float> c1(1 << 20), c2(1 << 20);
  (void)hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost);
  size_t diff = 0;
  for (size_t i = 0; i < c1.size(); ++i) diff += c1[i] != c2[i];
  printf("differing elements among the first 2^20: %zu\n", diff);
  return 0;
}
