// A/B of the bf16 LDS-DMA GEMM's variants in ONE process on one card (config-5 shape by default):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I islands_amd/csrc tools/microbench/gemm_bf16_exp.hip -o tools/microbench/gemm_bf16_exp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
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

// ---- experiment: 32-deep slabs in four LDS buffers, every wave weaves its 4 DMA pieces of slab kt + 3
// between its MFMAs of slab kt (two slabs of lead: no piece has to land within the slab it is issued in),
// one barrier per 32-deep slab with a counted vmcnt.  Same MFMA sequence per accumulator as the product
// kernel: bit-identical results.
template <int ACT, int PREFETCH>
__global__ __launch_bounds__(512) void gemm_tn_bf16_ring(const __bf16* __restrict__ A, const __bf16* __restrict__ W,
                                                         const float* __restrict__ bias, const float* __restrict__ R,
                                                         float* __restrict__ C, uint32_t M, uint32_t N, uint32_t K,
                                                         uint32_t ntn, uint64_t ldc) {
  constexpr uint32_t TM = 256, TN = 256, NW = 8, SBK = 32, MF = 4, NF = 2, WN = 4;
  constexpr uint32_t ABYTES = TM * SBK * 2, BUF = (TM + TN) * SBK * 2;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // [4][A slab | W slab], rows of 64 bytes
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t nwg = gridDim.x, orig = blockIdx.x;
  const uint32_t q8 = nwg / 8, r8 = nwg % 8, xcd = orig % 8;
  const uint32_t wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + orig / 8;
  const uint32_t ntm = nwg / ntn;
  constexpr uint32_t GM = 8;
  const uint32_t group = wgid / (GM * ntn), in_group = wgid % (GM * ntn);
  const uint32_t gm = ntm - group * GM < GM ? ntm - group * GM : GM;
  const uint64_t m0 = (uint64_t)(group * GM + in_group % gm) * TM, n0 = (uint64_t)(in_group / gm) * TN;
  const uint32_t wm = (wave / WN) * (32 * MF), wn = (wave % WN) * (32 * NF);
  floatx16 acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  // piece q of an operand image = rows 16 q .. 16 q + 15 (1 KiB); LDS chunk c of row r holds the
  // row's 16-byte chunk c ^ ((r >> 2) & 3).  Wave w moves A pieces w, w + 8 and W pieces w, w + 8.
  const __bf16* src[4];
  uint32_t dst[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const uint32_t q = wave + 8u * (p & 1), chunk = q * 64u + lane, row = chunk >> 2, c = (chunk & 3u) ^ ((row >> 2) & 3u);
    if (p < 2) {
      const uint64_t ra = m0 + row < M ? m0 + row : (uint64_t)M - 1;
      src[p] = A + ra * K + c * 8u;
      dst[p] = q * 1024u;
    } else {
      const uint64_t rw = n0 + row < N ? n0 + row : (uint64_t)N - 1;
      src[p] = W + rw * K + c * 8u;
      dst[p] = ABYTES + q * 1024u;
    }
  }
  auto piece = [&](int p, uint32_t kt) {
    __builtin_amdgcn_global_load_lds((isl_glb_void*)(src[p] + (uint64_t)kt * SBK), (isl_lds_void*)(lds + (kt & 3u) * BUF + dst[p]), 16, 0, 0);
  };
  const uint32_t kh = lane >> 5, c32 = lane & 31;
  const uint32_t nk = K / SBK;
#pragma unroll
  for (int s0 = 0; s0 < 3; ++s0)
    if ((uint32_t)s0 < nk) {
#pragma unroll
      for (int p = 0; p < 4; ++p) piece(p, s0);
    }
  auto frags_b = [&](const unsigned char* wb, uint32_t ks, bf16x8 (&b)[NF]) {
    const uint32_t cl = 2 * ks + kh;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const uint32_t rb = wn + 32 * j + c32;
      b[j] = *reinterpret_cast<const bf16x8*>(wb + rb * 64u + ((cl ^ ((rb >> 2) & 3u)) << 4));
    }
  };
  auto frag_a = [&](const unsigned char* ab, uint32_t ks, int i) {
    const uint32_t cl = 2 * ks + kh, ra = wm + 32 * i + c32;
    return *reinterpret_cast<const bf16x8*>(ab + ra * 64u + ((cl ^ ((ra >> 2) & 3u)) << 4));
  };
  bf16x8 pb[NF], pa[MF];  // PREFETCH: the fragments of k-step 0 of the next slab, read before its barrier
  if (PREFETCH) {
    if (nk >= 3) __builtin_amdgcn_s_waitcnt(0x0F78);  // slab 0 has landed: at most slabs 1 and 2 outstanding
    else if (nk == 2) __builtin_amdgcn_s_waitcnt(0x0F74);
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    frags_b(lds + ABYTES, 0, pb);
#pragma unroll
    for (int i = 0; i < MF; ++i) pa[i] = frag_a(lds, 0, i);
  }
  for (uint32_t kt = 0; kt < nk; ++kt) {
    if (!PREFETCH) {
      // slab kt has landed when at most the pieces of the later slabs are outstanding
      const uint32_t later = (kt + 2 < nk ? kt + 2 : nk - 1) - kt;  // slabs issued beyond kt: 0, 1 or 2
      if (later >= 2) __builtin_amdgcn_s_waitcnt(0x0F78);
      else if (later == 1) __builtin_amdgcn_s_waitcnt(0x0F74);
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
    const bool more = kt + 3 < nk;
    const unsigned char* ab = lds + (kt & 3u) * BUF;
    const unsigned char* wb = ab + ABYTES;
#pragma unroll
    for (uint32_t ks = 0; ks < 2; ++ks) {
      bf16x8 b[NF];
      if (PREFETCH && ks == 0) {
#pragma unroll
        for (int j = 0; j < NF; ++j) b[j] = pb[j];
      } else {
        frags_b(wb, ks, b);
      }
      if (PREFETCH && ks == 1 && kt + 1 < nk) {
        // slab kt + 1 must have landed for everyone before its first fragments are read: the barrier
        // of slab kt + 1, taken one k-step early
        // (outstanding then: the pieces of slab kt + 2, issued behind the previous such barrier; those of
        // slab kt + 3 go out behind this one, into the buffer slab kt - 1 has just left)
        if (kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0F74);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
      }
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const bf16x8 a = (PREFETCH && ks == 0) ? pa[i] : frag_a(ab, ks, i);
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc[i][j], 0, 0, 0);
        if (PREFETCH) {
          if (ks == 1 && more) piece(i, kt + 3);
        } else if ((i & 1) == 1 && more) {
          piece((int)ks * 2 + i / 2, kt + 3);
        }
      }
      if (PREFETCH && ks == 1 && kt + 1 < nk) {
        const unsigned char* nab = lds + ((kt + 1) & 3u) * BUF;
        frags_b(nab + ABYTES, 0, pb);
#pragma unroll
        for (int i = 0; i < MF; ++i) pa[i] = frag_a(nab, 0, i);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const uint64_t n = n0 + wn + j * 32 + c32;
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint64_t m = m0 + wm + i * 32 + 8 * (r / 4) + 4 * kh + (r % 4);
        if (m >= M) continue;
        C[m * ldc + n] = -acc[i][j][r];  // EPI_DOT
      }
    }
}

// four waves per 256 x 256 tile, 128 x 128 per wave (hipBLASLt's decomposition), compiler-scheduled
float run_w4(const __bf16* A, const __bf16* W, float* C, uint32_t M, uint32_t N, uint32_t K, int reps) {
  auto kern = gemm_tn_bf16_dma<EPI_DOT, false, false, 2, 2, 4, 4, 0>;
  constexpr size_t lds = 2 * (256 + 256) * HBK * 2;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t ntn = (N + 255) / 256, ntiles = ((M + 255) / 256) * ntn;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), lds, 0, A, W, nullptr, nullptr, C, M, N, K, ntn, (uint64_t)N, (uint64_t*)nullptr);
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), lds, 0, A, W, nullptr, nullptr, C, M, N, K, ntn, (uint64_t)N, (uint64_t*)nullptr);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

template <int PREFETCH>
float run_ring(const __bf16* A, const __bf16* W, float* C, uint32_t M, uint32_t N, uint32_t K, int reps) {
  auto kern = gemm_tn_bf16_ring<EPI_DOT, PREFETCH>;
  constexpr size_t lds = 4 * (256 + 256) * 32 * 2;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t ntn = (N + 255) / 256, ntiles = ((M + 255) / 256) * ntn;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(ntiles), dim3(512), lds, 0, A, W, nullptr, nullptr, C, M, N, K, ntn, (uint64_t)N);
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(512), lds, 0, A, W, nullptr, nullptr, C, M, N, K, ntn, (uint64_t)N);
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

  float* C3m;
  (void)hipMalloc(&C3m, (size_t)M * N * 4);
  for (int round = 0; round < 3; ++round) {
    const float t0 = run<65>(A, W, C, M, N, K, 40, dbg);
    const float t1 = run<72>(A, W, C2, M, N, K, 40, dbg);
    const float t2 = run<64>(A, W, C2, M, N, K, 40, dbg);
    const float t3 = run<76>(A, W, C2, M, N, K, 40, dbg);
    const float tm = run<0>(A, W, C3m, M, N, K, 40, dbg);
    const float tl1 = run<4>(A, W, C2, M, N, K, 40, dbg), tl2 = run<8>(A, W, C2, M, N, K, 40, dbg), tl3 = run<12>(A, W, C2, M, N, K, 40, dbg), tl0 = run<1>(A, W, C2, M, N, K, 40, dbg);
    printf("16x16x32, late waves issue after MFMA group 1 / 2 / 4 (product) / 6 of 16, or with the others: %.1f / %.1f / %.1f / %.1f / %.1f TF\n", fl / tl1 / 1e9, fl / tl2 / 1e9,
           fl / tm / 1e9, fl / tl3 / 1e9, fl / tl0 / 1e9);
    const float tcos = run<0, EPI_COSINE>(A, W, C2, M, N, K, 40, dbg);
    const float tcos32 = run<64, EPI_COSINE>(A, W, C2, M, N, K, 40, dbg);
    printf("cosine epilogue: product kernel %.3f ms %.1f TF | 32x32x16 %.3f ms %.1f TF\n", tcos, fl / tcos / 1e9, tcos32, fl / tcos32 / 1e9);
    printf("16x16x32 MFMAs (the product kernel; the variants below use 32x32x16): %.3f ms %.1f TF\n", tm, fl / tm / 1e9);

    printf("M=%u N=%u K=%u  all behind the barrier %.3f ms %.1f TF | waves 4..7 at k-step 2 %.3f ms %.1f TF | at 1 %.3f ms %.1f TF | at 3 %.3f ms %.1f TF\n", M, N, K,
           t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, t3, fl / t3 / 1e9);
  }
  {
    float* C3;
    (void)hipMalloc(&C3, (size_t)M * N * 4);
    for (int round = 0; round < 3; ++round) {
      const float tw4 = run_w4(A, W, C3, M, N, K, 40);
      printf("four waves per tile, 128 x 128 per wave, 16x16x32: %.3f ms %.1f TF\n", tw4, fl / tw4 / 1e9);
      const float tr0 = run_ring<0>(A, W, C3, M, N, K, 40);
      const float tr1 = run_ring<1>(A, W, C3, M, N, K, 40);
      printf("ring (4 x 32-deep buffers, pieces woven): %.3f ms %.1f TF | + first fragments read before the barrier: %.3f ms %.1f TF\n", tr0,
             fl / tr0 / 1e9, tr1, fl / tr1 / 1e9);
    }
    (void)run_ring<0>(A, W, C3, M, N, K, 1);
    std::vector<float> c1(1 << 20), c3(1 << 20);
    (void)hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c3.data(), C3, c3.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0;
    for (size_t i = 0; i < c1.size(); ++i) diff += c1[i] != c3[i];
    (void)run_ring<1>(A, W, C3, M, N, K, 1);
    (void)hipMemcpy(c3.data(), C3, c3.size() * 4, hipMemcpyDeviceToHost);
    size_t diff1 = 0;
    for (size_t i = 0; i < c1.size(); ++i) diff1 += c1[i] != c3[i];
    printf("ring vs product kernel, differing elements among the first 2^20: %zu / %zu (prefetch variant)\n", diff, diff1);
  }
  {
    std::vector<float> c1(1 << 20), c3(1 << 20);
    (void)hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c3.data(), C3m, c3.size() * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (size_t i = 0; i < c1.size(); ++i) { md = std::max(md, (double)fabsf(c1[i] - c3[i])); mx = std::max(mx, (double)fabsf(c1[i])); }
    printf("16x16x32 vs 32x32x16: max |difference| %.3g on values up to %.3g\n", md, mx);
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
  (void)run<193>(A, W, C2, M, N, K, 2, dbg);
  stamps("all behind the barrier");
  (void)run<192>(A, W, C2, M, N, K, 2, dbg);
  stamps("waves 4..7 at k-step 1, 32x32x16");
  (void)run<128>(A, W, C2, M, N, K, 2, dbg);
  stamps("the product kernel (16x16x32, waves 4..7 behind the first MFMA groups)");

  std::vector<float> c1(1 << 20), c2(1 << 20);
  (void)hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost);
  size_t diff = 0;
  for (size_t i = 0; i < c1.size(); ++i) diff += c1[i] != c2[i];
  printf("differing elements among the first 2^20: %zu\n", diff);
  return 0;
}
