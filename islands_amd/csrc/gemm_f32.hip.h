// float32 GEMM on the matrix cores, shared by the encoder (Linear layers) and the batched
// distance matrix (query x candidate contraction).
#pragma once

#include "common.hpp"

namespace isl_gemm {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// epilogues beyond the Linear ones (0 bias, 1 bias + GELU(erf), 2 bias + GELU(tanh)): distances
// from the dot products, with `bias` = per-column and `R` = per-row squared norms
constexpr int EPI_COSINE = 3, EPI_DOT = 4, EPI_EUCLIDEAN = 5;

__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x / 1.41421356237309515f)); }
__device__ __forceinline__ float gelu_tanh_f(float x) {
  return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}

// C[M,N] = A[M,K] W[N,K]^T + bias (+ R) with an optional GELU: Linear layers of the encoder.
// 128x128 tile per 256-thread workgroup, 64x64 per wave = 2x2 MFMA 32x32 blocks, K in slabs of
// 32 staged k-major in LDS (the next slab is fetched into registers while the current one feeds
// the matrix cores).  v_mfma_f32_32x32x2_f32: lane l supplies A[l%32][l/32] and B[l/32][l%32],
// accumulator register r of lane l is C[8*(r/4) + 4*(l/32) + r%4][l%32].
constexpr int BM = 128, BN = 128, BK = 32, LDT = BM + 4;
template <int ACT, bool RES>
__global__ __launch_bounds__(256) void gemm_tn_f32(const float* __restrict__ A,
                                                   const float* __restrict__ W,
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ R, float* __restrict__ C,
                                                   uint32_t M, uint32_t N, uint32_t K) {
  __shared__ float As[BK][LDT];
  __shared__ float Bs[BK][LDT];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const uint64_t m0 = (uint64_t)blockIdx.y * BM, n0 = (uint64_t)blockIdx.x * BN;
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  // A slab and W slab are 128 rows x 32 floats = 1024 float4 each: thread t owns float4 number
  // t, t+256, t+512, t+768 -> row = idx / 8, k-octet = idx % 8.  Rows / columns past the matrix
  // are clamped to the last valid one (the loads stay unconditional 16-byte loads) and zeroed
  // after the load on edge tiles only.
  const uint32_t kq = (tid & 7) * 4;
  uint32_t ra[4], rb[4];
  bool za[4], zb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t r = (tid + 256u * i) >> 3;
    za[i] = m0 + r >= M;
    zb[i] = n0 + r >= N;
    ra[i] = (uint32_t)(za[i] ? M - 1 - m0 : r);
    rb[i] = (uint32_t)(zb[i] ? N - 1 - n0 : r);
  }
  const bool edge = m0 + BM > M || n0 + BN > N || (K % BK) != 0;
  const float* Ab = A + m0 * K + kq;
  const float* Wb = W + n0 * K + kq;
  float4 pa[4], pb[4];
  auto fetch = [&](uint32_t k0) {
    if (!edge) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pa[i] = *reinterpret_cast<const float4*>(Ab + (uint64_t)ra[i] * K + k0);
        pb[i] = *reinterpret_cast<const float4*>(Wb + (uint64_t)rb[i] * K + k0);
      }
    } else {
      const bool kout = k0 + kq >= K;  // K is a multiple of 4 (checked on the host)
      const uint32_t kc = kout ? 0u : k0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pa[i] = *reinterpret_cast<const float4*>(Ab + (uint64_t)ra[i] * K + kc);
        pb[i] = *reinterpret_cast<const float4*>(Wb + (uint64_t)rb[i] * K + kc);
        if (kout || za[i]) pa[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kout || zb[i]) pb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t r = (tid + 256u * i) >> 3;
      As[kq + 0][r] = pa[i].x; As[kq + 1][r] = pa[i].y; As[kq + 2][r] = pa[i].z; As[kq + 3][r] = pa[i].w;
      Bs[kq + 0][r] = pb[i].x; Bs[kq + 1][r] = pb[i].y; Bs[kq + 2][r] = pb[i].z; Bs[kq + 3][r] = pb[i].w;
    }
  };
  fetch(0);
  const uint32_t kh = lane >> 5, c32 = lane & 31;
  for (uint32_t k0 = 0; k0 < K; k0 += BK) {
    stage();
    __syncthreads();
    if (k0 + BK < K) fetch(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float a0 = As[2 * kk + kh][wm + c32], a1 = As[2 * kk + kh][wm + 32 + c32];
      const float b0 = Bs[2 * kk + kh][wn + c32], b1 = Bs[2 * kk + kh][wn + 32 + c32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint64_t n = n0 + wn + j * 32 + c32;
      if (n >= N) continue;
      const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint64_t m = m0 + wm + i * 32 + 8 * (r / 4) + 4 * kh + (r % 4);
        if (m >= M) continue;
        float v = acc[i][j][r];
        if (ACT <= 2) {
          v += bv;
          if (ACT == 1) v = gelu_erf_f(v);
          if (ACT == 2) v = gelu_tanh_f(v);
          if (RES) v += R[m * N + n];
        } else if (ACT == EPI_COSINE) {  // bias = |w_n|^2 per column, R = |a_m|^2 per row
          const float norm = sqrtf(R[m] * bv);
          v = norm == 0.0f ? 1.0f : 1.0f - v / norm;
        } else if (ACT == EPI_DOT) {
          v = -v;
        } else if (ACT == EPI_EUCLIDEAN) {  // |a|^2 + |w|^2 - 2 a.w, clamped
          v = R[m] + bv - 2.0f * v;
          v = sqrtf(v > 0.0f ? v : 0.0f);
        }
        C[m * N + n] = v;
      }
    }
}

template <int ACT, bool RES>
void launch_gemm(const float* A, const float* W, const float* bias, const float* R, float* C,
                 uint64_t M, uint64_t N, uint64_t K, hipStream_t st) {
  dim3 grid((uint32_t)((N + BN - 1) / BN), (uint32_t)((M + BM - 1) / BM));
  hipLaunchKernelGGL((gemm_tn_f32<ACT, RES>), grid, dim3(256), 0, st, A, W, bias, R, C, (uint32_t)M,
                     (uint32_t)N, (uint32_t)K);
}


}  // namespace isl_gemm
