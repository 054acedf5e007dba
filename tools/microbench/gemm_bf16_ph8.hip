// The bf16 GEMM on the eight-phase schedule (gemm_tn_bf16_ph8) against the product's 8-wave LDS-DMA kernel
// (gemm_tn_bf16_dma<..,2,4,4,2>) in ONE process on one card, rounds interleaved, plus a bit-for-bit
// comparison of every output element (the two accumulate each element in the same order):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I islands_amd/csrc tools/microbench/gemm_bf16_ph8.hip -o tools/microbench/gemm_bf16_ph8
//   tools/microbench/gemm_bf16_ph8 [M N K]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_bf16.hip.h"
using namespace isl_gemm;

static const float* g_rn = nullptr;
static const float* g_qn = nullptr;
constexpr size_t kLds = 2 * (256 + 256) * HBK * 2;

// the persistent form (gemm_tn_bf16_ph8p): one workgroup per CU walks the tiles
template <int EPI, int VAR = 0>
static float run_p(const __bf16* A, const __bf16* W, float* C, uint32_t M, uint32_t N, uint32_t K, int reps) {
  const uint32_t ntn = (N + 255) / 256, ntiles = ((M + 255) / 256) * ntn;
  constexpr size_t lds = kLds + 8 * 8 * 72 * 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto kern = gemm_tn_bf16_ph8p<EPI, false, VAR>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  auto launch = [&] {
    hipLaunchKernelGGL(kern, dim3(ntiles < 256 ? ntiles : 256), dim3(512), lds, 0, A, W, g_rn, g_qn, C, M, N, K, ntn, (uint64_t)N, ntiles);
  };
  launch();
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

template <int EPI, bool PH8, int VAR = 0>
static float run(const __bf16* A, const __bf16* W, float* C, uint32_t M, uint32_t N, uint32_t K, int reps) {
  const uint32_t ntn = (N + 255) / 256, ntiles = ((M + 255) / 256) * ntn;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto launch = [&] {
    if constexpr (PH8) {
      auto kern = gemm_tn_bf16_ph8<EPI, false, false, VAR>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
      hipLaunchKernelGGL(kern, dim3(ntiles), dim3(512), kLds, 0, A, W, g_rn, g_qn, C, M, N, K, ntn, (uint64_t)N);
    } else {
      auto kern = gemm_tn_bf16_dma<EPI, false, false, 2, 4, 4, 2, 0>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
      hipLaunchKernelGGL(kern, dim3(ntiles), dim3(512), kLds, 0, A, W, g_rn, g_qn, C, M, N, K, ntn, (uint64_t)N, (uint64_t*)nullptr);
    }
  };
  launch();
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
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

static void fill(std::vector<uint16_t>& h, uint32_t seed, float scale) {
  uint32_t s = seed;
  for (auto& v : h) {  // uniform in (-scale, scale), truncated to bf16
    s = s * 1664525u + 1013904223u;
    float f = (((s >> 8) & 0xFFFF) / 32768.0f - 1.0f) * scale;
    uint32_t u;
    memcpy(&u, &f, 4);
    v = (uint16_t)(u >> 16);
  }
}

int main(int argc, char** argv) {
  uint32_t M = argc > 3 ? atoi(argv[1]) : 4096, N = argc > 3 ? atoi(argv[2]) : 65536, K = argc > 3 ? atoi(argv[3]) : 4096;
  const int reps = argc > 4 ? atoi(argv[4]) : 40;
  // edge shapes first (ragged last tiles, one K-tile, two, an odd number): correctness only
  {
    const uint32_t shapes[][3] = {{300, 1000, 64}, {257, 513, 128}, {256, 256, 192}, {1000, 300, 448}, {512, 768, 4096},
                                  {257, 516, 128}, {2100, 70000, 192}, {4096, 66048, 128}, {3000, 70004, 320}};
    for (auto& sh : shapes) {
      const uint32_t m = sh[0], n = sh[1], k = sh[2];
      __bf16 *a, *w; float *c0, *c1, *rn, *qn;
      (void)hipMalloc(&a, (size_t)m * k * 2); (void)hipMalloc(&w, (size_t)n * k * 2);
      (void)hipMalloc(&c0, (size_t)m * n * 4); (void)hipMalloc(&c1, (size_t)m * n * 4);
      (void)hipMalloc(&rn, (size_t)n * 4); (void)hipMalloc(&qn, (size_t)m * 4);
      std::vector<uint16_t> ha((size_t)m * k), hw((size_t)n * k);
      fill(ha, 7 + m, 1.0f); fill(hw, 11 + n, 1.0f);
      (void)hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
      (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
      std::vector<float> ones(n > m ? n : m, (float)k / 3.0f);
      (void)hipMemcpy(rn, ones.data(), (size_t)n * 4, hipMemcpyHostToDevice);
      (void)hipMemcpy(qn, ones.data(), (size_t)m * 4, hipMemcpyHostToDevice);
      g_rn = rn; g_qn = qn;
      (void)hipMemset(c0, 0xFF, (size_t)m * n * 4); (void)hipMemset(c1, 0xFF, (size_t)m * n * 4);
      (void)run<EPI_COSINE, false>(a, w, c0, m, n, k, 1);
      (void)run<EPI_COSINE, true>(a, w, c1, m, n, k, 1);
      const hipError_t e = hipDeviceSynchronize();
      printf("shape %u x %u x %u: %llu differing elements (%s)\n", m, n, k, differing(c0, c1, (uint64_t)m * n), hipGetErrorString(e));
      if (n % 4 == 0) {
        (void)hipMemset(c1, 0xFF, (size_t)m * n * 4);
        (void)run<EPI_COSINE, true, 4>(a, w, c1, m, n, k, 1);
        const hipError_t e3 = hipDeviceSynchronize();
        printf("   8-phase kernel, LDS-transposed epilogue: %llu differing elements (%s)\n", differing(c0, c1, (uint64_t)m * n), hipGetErrorString(e3));
        (void)hipMemset(c1, 0xFF, (size_t)m * n * 4);
        (void)run<EPI_COSINE, true, 12>(a, w, c1, m, n, k, 1);
        const hipError_t e4 = hipDeviceSynchronize();
        printf("   8-phase kernel, VAR 12: %llu differing elements (%s)\n", differing(c0, c1, (uint64_t)m * n), hipGetErrorString(e4));
      }
      if (k >= 128 && n % 4 == 0) {
        (void)hipMemset(c1, 0xFF, (size_t)m * n * 4);
        (void)run_p<EPI_COSINE>(a, w, c1, m, n, k, 1);
        const hipError_t e2 = hipDeviceSynchronize();
        printf("   persistent kernel: %llu differing elements (%s)\n", differing(c0, c1, (uint64_t)m * n), hipGetErrorString(e2));
      }
      (void)hipFree(a); (void)hipFree(w); (void)hipFree(c0); (void)hipFree(c1); (void)hipFree(rn); (void)hipFree(qn);
    }
  }
  __bf16 *A, *W; float *C0, *C1, *rn, *qn;
  (void)hipMalloc(&A, (size_t)M * K * 2); (void)hipMalloc(&W, (size_t)N * K * 2);
  (void)hipMalloc(&C0, (size_t)M * N * 4); (void)hipMalloc(&C1, (size_t)M * N * 4);
  (void)hipMalloc(&rn, (size_t)N * 4); (void)hipMalloc(&qn, (size_t)M * 4);
  {
    std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
    fill(ha, 12345, 0.0156f * 1.7f);  // the scale of L2-normalised rows of d = 4096
    fill(hw, 54321, 0.0156f * 1.7f);
    (void)hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> ones(N > M ? N : M, 1.0f);
    (void)hipMemcpy(rn, ones.data(), (size_t)N * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(qn, ones.data(), (size_t)M * 4, hipMemcpyHostToDevice);
  }
  g_rn = rn; g_qn = qn;
  const double fl = 2.0 * M * N * K;
  for (int round = 0; round < 4; ++round) {
    const float t0 = run<EPI_DOT, false>(A, W, C0, M, N, K, reps);
    const float t1 = run<EPI_DOT, true>(A, W, C1, M, N, K, reps);
    const float v1 = run<EPI_DOT, true, 1>(A, W, C1, M, N, K, reps), v3 = run<EPI_DOT, true, 3>(A, W, C1, M, N, K, reps);
    const float t2 = run<EPI_COSINE, false>(A, W, C0, M, N, K, reps);
    const float t3 = run<EPI_COSINE, true>(A, W, C1, M, N, K, reps);
    printf("M=%u N=%u K=%u round %d: dot epilogue: 8-wave DMA %.3f ms %.1f TF | 8-phase %.3f ms %.1f TF || cosine: %.3f ms %.1f TF | %.3f ms %.1f TF\n",
           M, N, K, round, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, t3, fl / t3 / 1e9);
    printf("   8-phase, dot: no setprio %.1f TF | static priority for waves 4..7 only %.1f TF\n", fl / v1 / 1e9, fl / v3 / 1e9);
    const float p0 = run_p<EPI_DOT>(A, W, C1, M, N, K, reps), p1 = run_p<EPI_DOT, 1>(A, W, C1, M, N, K, reps), p2 = run_p<EPI_DOT, 2>(A, W, C1, M, N, K, reps);
    const float p3 = run_p<EPI_COSINE>(A, W, C1, M, N, K, reps), p4 = run_p<EPI_COSINE, 1>(A, W, C1, M, N, K, reps), p5 = run_p<EPI_COSINE, 2>(A, W, C1, M, N, K, reps);
    const float e0 = run<EPI_DOT, true, 4>(A, W, C1, M, N, K, reps), e1 = run<EPI_COSINE, true, 4>(A, W, C1, M, N, K, reps);
    printf("   8-phase with the LDS-transposed epilogue: dot %.3f ms %.1f TF | cosine %.3f ms %.1f TF\n", e0, fl / e0 / 1e9, e1, fl / e1 / 1e9);
    const float e2 = run<EPI_COSINE, true, 12>(A, W, C1, M, N, K, reps), e3 = run<EPI_EUCLIDEAN, true, 12>(A, W, C1, M, N, K, reps), e4 = run<EPI_EUCLIDEAN, true, 4>(A, W, C1, M, N, K, reps);
    printf("   ... + row norms through LDS, no load between the stores (VAR 12): cosine %.3f ms %.1f TF | euclidean %.3f ms %.1f TF (VAR 4: %.1f TF)\n",
           e2, fl / e2 / 1e9, e3, fl / e3 / 1e9, fl / e4 / 1e9);
    printf("   persistent: dot %.3f ms %.1f TF (non-temporal stores %.1f, per-element epilogue %.1f) | cosine %.3f ms %.1f TF (non-temporal stores %.1f, per-element epilogue %.1f)\n",
           p0, fl / p0 / 1e9, fl / p1 / 1e9, fl / p2 / 1e9, p3, fl / p3 / 1e9, fl / p4 / 1e9, fl / p5 / 1e9);
    fflush(stdout);
  }
  (void)run<EPI_COSINE, false>(A, W, C0, M, N, K, 1);
  (void)run<EPI_COSINE, true>(A, W, C1, M, N, K, 1);
  printf("cosine outputs, differing elements of %llu: %llu\n", (unsigned long long)M * N, differing(C0, C1, (uint64_t)M * N));
  (void)hipMemset(C1, 0xFF, (size_t)M * N * 4);
  (void)run<EPI_COSINE, true, 4>(A, W, C1, M, N, K, 1);
  printf("cosine outputs of the 8-phase kernel with the LDS-transposed epilogue, differing elements: %llu\n", differing(C0, C1, (uint64_t)M * N));
  (void)hipMemset(C1, 0xFF, (size_t)M * N * 4);
  (void)run<EPI_COSINE, true, 12>(A, W, C1, M, N, K, 1);
  printf("cosine outputs of the 8-phase kernel, VAR 12 (row norms through LDS, branch-free block), differing elements: %llu\n", differing(C0, C1, (uint64_t)M * N));
  {  // tiny norms in one corner: that block must take the careful branch, every bit as before
    std::vector<float> hq(M, 1.0f), hr(N, 1.0f);
    for (uint32_t i = 0; i < 40 && i < M; ++i) hq[i] = (i % 3 == 0) ? 0.0f : 1e-30f;
    for (uint32_t i = 100; i < 180 && i < N; ++i) hr[i] = (i % 5 == 0) ? 0.0f : 3e-12f;
    (void)hipMemcpy(qn, hq.data(), (size_t)M * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(rn, hr.data(), (size_t)N * 4, hipMemcpyHostToDevice);
    (void)run<EPI_COSINE, false>(A, W, C0, M, N, K, 1);
    (void)hipMemset(C1, 0xFF, (size_t)M * N * 4);
    (void)run<EPI_COSINE, true, 12>(A, W, C1, M, N, K, 1);
    printf("   the same with zero / tiny norms in rows 0..39 and columns 100..179: %llu\n", differing(C0, C1, (uint64_t)M * N));
    std::vector<float> ones(N > M ? N : M, 1.0f);
    (void)hipMemcpy(rn, ones.data(), (size_t)N * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(qn, ones.data(), (size_t)M * 4, hipMemcpyHostToDevice);
    (void)run<EPI_COSINE, false>(A, W, C0, M, N, K, 1);
  }
  (void)hipMemset(C1, 0xFF, (size_t)M * N * 4);
  (void)run_p<EPI_COSINE>(A, W, C1, M, N, K, 1);
  printf("cosine outputs of the persistent kernel, differing elements: %llu\n", differing(C0, C1, (uint64_t)M * N));
  (void)hipMemset(C1, 0xFF, (size_t)M * N * 4);
  (void)run_p<EPI_DOT>(A, W, C1, M, N, K, 1);
  (void)run<EPI_DOT, false>(A, W, C0, M, N, K, 1);
  printf("dot outputs of the persistent kernel, differing elements: %llu\n", differing(C0, C1, (uint64_t)M * N));
  {
    unsigned long long badp = 0;
    for (int it = 0; it < 20; ++it) {
      (void)run_p<EPI_DOT>(A, W, C1, M, N, K, 1);
      badp += differing(C0, C1, (uint64_t)M * N);
    }
    printf("race screen, 20 further runs of the persistent kernel: %llu differing elements in all\n", badp);
  }
  (void)run<EPI_DOT, false>(A, W, C0, M, N, K, 1);
  (void)run<EPI_DOT, true>(A, W, C1, M, N, K, 1);
  printf("dot outputs, differing elements: %llu\n", differing(C0, C1, (uint64_t)M * N));
  // a race screen: the 8-phase kernel twenty more times against the first result
  unsigned long long bad = 0;
  for (int it = 0; it < 20; ++it) {
    (void)run<EPI_DOT, true>(A, W, C1, M, N, K, 1);
    bad += differing(C0, C1, (uint64_t)M * N);
  }
  printf("race screen, 20 further runs of the 8-phase kernel: %llu differing elements in all\n", bad);
  return 0;
}
