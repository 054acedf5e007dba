// bf16-input GEMM on the matrix cores for the encoder's optional reduced-precision mode
// (isl_encoder_set_precision): C[M,N] = A[M,K] W[N,K]^T + bias (+ GELU, + residual) with A in
// float32 (rounded to bf16 while it is staged into LDS), W pre-converted to bf16, float32
// accumulation in v_mfma_f32_32x32x16_bf16.  Same tiling idea as gemm_f32.hip.h: 128x128 tile per
// 256-thread workgroup, 64x64 per wave = 2x2 MFMA blocks, K in slabs of 64, the next slab
// prefetched into registers.  Operand layout of the 32x32x16 MFMA: lane l supplies row (or
// column) l % 32 and the 8 consecutive k values 8 * (l / 32) .. + 7 of each 16-wide k step.
#pragma once

#include <algorithm>
#include <type_traits>

#include "common.hpp"
#include "gemm_f32.hip.h"

namespace isl_gemm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int HBK = 64;          // k per slab
constexpr int HLD = HBK + 8;     // LDS row pitch in bf16 (144 bytes: conflict-free 16-byte reads)

// A16: A is already bf16 (activations kept in bf16 between the kernels); C16: C is written as bf16.
template <int ACT, bool RES, bool A16, bool C16, int MF>
__global__ __launch_bounds__(256) void gemm_tn_bf16(const void* __restrict__ Av,
                                                    const __bf16* __restrict__ W,
                                                    const float* __restrict__ bias,
                                                    const float* __restrict__ R, void* __restrict__ Cv,
                                                    uint32_t M, uint32_t N, uint32_t K) {
  const float* A = reinterpret_cast<const float*>(Av);
  const __bf16* Ah = reinterpret_cast<const __bf16*>(Av);
  float* C = reinterpret_cast<float*>(Cv);
  __bf16* Ch = reinterpret_cast<__bf16*>(Cv);
  constexpr int TM = 64 * MF;  // tile rows: every wave owns MF x 2 blocks of 32 x 32
  __shared__ __attribute__((aligned(16))) __bf16 As[TM * HLD];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[BN * HLD];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t wm = (wave >> 1) * (32 * MF), wn = (wave & 1) * 64;
  const uint64_t m0 = (uint64_t)blockIdx.y * TM, n0 = (uint64_t)blockIdx.x * BN;
  floatx16 acc[MF][2];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  // A slab: 128 rows x 64 floats = 2048 float4 -> 8 per thread (row = idx / 16, 4-float column
  // group = idx % 16); W slab: 128 rows x 64 bf16 = 1024 x 16 bytes -> 4 per thread (row = idx / 8)
  constexpr int NA32 = TM / 16, NA16 = TM / 32;  // loads per thread of an A slab (f32 / bf16)
  float4 pa[A16 ? 1 : NA32];
  bf16x8 pa16[A16 ? NA16 : 1];
  bf16x8 pb[4];
  const bool edge = m0 + TM > M || n0 + BN > N || (K % HBK) != 0;
  auto fetch = [&](uint32_t k0) {
    if constexpr (A16) {
#pragma unroll
      for (int i = 0; i < NA16; ++i) {
        const uint32_t idx = tid + 256u * i, r = idx >> 3, c = (idx & 7) * 8;
        const bool out = edge && (m0 + r >= M || k0 + c >= K);
        const uint64_t rr = (m0 + r < M) ? m0 + r : (uint64_t)M - 1;
        const uint32_t kc = (k0 + c < K) ? k0 + c : 0u;
        pa16[i] = *reinterpret_cast<const bf16x8*>(Ah + rr * K + kc);
        if (out) {
#pragma unroll
          for (int e = 0; e < 8; ++e) pa16[i][e] = (__bf16)0.0f;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA32; ++i) {
        const uint32_t idx = tid + 256u * i, r = idx >> 4, c = (idx & 15) * 4;
        const bool out = edge && (m0 + r >= M || k0 + c >= K);  // K is a multiple of 8 (host check)
        const uint64_t rr = (m0 + r < M) ? m0 + r : (uint64_t)M - 1;
        const uint32_t kc = (k0 + c < K) ? k0 + c : 0u;
        pa[i] = *reinterpret_cast<const float4*>(A + rr * K + kc);
        if (out) pa[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t idx = tid + 256u * i, r = idx >> 3, c = (idx & 7) * 8;
      const bool out = edge && (n0 + r >= N || k0 + c >= K);
      const uint64_t rr = (n0 + r < N) ? n0 + r : (uint64_t)N - 1;
      const uint32_t kc = (k0 + c < K) ? k0 + c : 0u;
      pb[i] = *reinterpret_cast<const bf16x8*>(W + rr * K + kc);
      if (out) {
#pragma unroll
        for (int e = 0; e < 8; ++e) pb[i][e] = (__bf16)0.0f;
      }
    }
  };
  auto stage = [&]() {
    if constexpr (A16) {
#pragma unroll
      for (int i = 0; i < NA16; ++i) {
        const uint32_t idx = tid + 256u * i, r = idx >> 3, c = (idx & 7) * 8;
        *reinterpret_cast<bf16x8*>(&As[r * HLD + c]) = pa16[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA32; ++i) {
        const uint32_t idx = tid + 256u * i, r = idx >> 4, c = (idx & 15) * 4;
        bf16x4 v;
        v[0] = (__bf16)pa[i].x; v[1] = (__bf16)pa[i].y; v[2] = (__bf16)pa[i].z; v[3] = (__bf16)pa[i].w;
        *reinterpret_cast<bf16x4*>(&As[r * HLD + c]) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t idx = tid + 256u * i, r = idx >> 3, c = (idx & 7) * 8;
      *reinterpret_cast<bf16x8*>(&Bs[r * HLD + c]) = pb[i];
    }
  };
  fetch(0);
  const uint32_t kh = lane >> 5, c32 = lane & 31;
  for (uint32_t k0 = 0; k0 < K; k0 += HBK) {
    stage();
    __syncthreads();
    if (k0 + HBK < K) fetch(k0 + HBK);
#pragma unroll
    for (int ks = 0; ks < HBK / 16; ++ks) {
      const uint32_t ko = 16 * ks + 8 * kh;
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&Bs[(wn + c32) * HLD + ko]);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Bs[(wn + 32 + c32) * HLD + ko]);
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(&As[(wm + 32 * i + c32) * HLD + ko]);
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[i][1], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint64_t n = n0 + wn + j * 32 + c32;
      if (n >= N) continue;
      const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint64_t m = m0 + wm + i * 32 + 8 * (r / 4) + 4 * kh + (r % 4);
        if (m >= M) continue;
        float v = acc[i][j][r] + bv;
        if (ACT == 1) v = gelu_erf_f(v);
        if (ACT == 2) v = gelu_tanh_f(v);
        if (RES) v += R[m * N + n];
        if constexpr (C16) Ch[m * N + n] = (__bf16)v;
        else C[m * N + n] = v;
      }
    }
}

// Same contraction with both operands staged by LDS-DMA (global_load_lds_dwordx4: global -> LDS
// with no register stop) into two 32 KiB buffers: the slab k+1 is in flight while slab k feeds
// the matrix cores, one barrier per slab.  The DMA writes lane-linear images (a wave
// instruction fills 8 rows of 128 bytes), so the bank-conflict swizzle is applied to the SOURCE
// address: LDS chunk c of row r holds the row's 16-byte chunk c ^ ((r >> 1) & 7), and the fragment
// reads look their chunk up under the same XOR (16 consecutive rows at one logical chunk then
// touch 16 different 16-byte slots of the 256-byte bank row).  Requirements (host-checked): A and W bf16,
// K % 64 == 0; rows past M / N are clamped (their products are never stored).  Workgroup ids are
// remapped so that the column tiles of one row tile share an XCD (its L2 then serves A once).
// (isl_lds_void / isl_glb_void: gemm_f32.hip.h)

// WM x WN waves, each owning MF x NF blocks of 32 x 32: tile = (32 MF WM) x (32 NF WN).
//   <2,2,2,2>: 128 x 128, 256 threads, 64 KiB LDS, two workgroups per CU;
//   <2,4,4,2>: 256 x 256, 512 threads, 128 KiB LDS, one per CU -- 128 x 64 per wave reads 6 operand
//   fragments per 8 MFMAs instead of 4 per 4, and a slab byte feeds twice the flops.
// EXP (tools/microbench/gemm_bf16_exp.hip only): bit 0 = all waves issue their DMA pieces behind the barrier
// (without it waves 4..7 issue theirs one k-step later than waves 0..3; bits 2-3: another k-step); bit 1 =
// staggered start of the workgroups; bit 7 = s_memtime stamps of waves 0 and 4 of workgroup 300 -> dbg.
template <int ACT, bool RES, bool C16, int WM, int WN, int MF, int NF, int EXP = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_tn_bf16_dma(const __bf16* __restrict__ A,
                                                                 const __bf16* __restrict__ W,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ R,
                                                                 void* __restrict__ Cv, uint32_t M, uint32_t N,
                                                                 uint32_t K, uint32_t ntn, uint64_t ldc,
                                                                 uint64_t* dbg = nullptr) {
  constexpr uint32_t TM = 32 * MF * WM, TN = 32 * NF * WN, NW = WM * WN;
  constexpr uint32_t ABYTES = TM * HBK * 2, WBYTES = TN * HBK * 2, BUF = ABYTES + WBYTES;
  constexpr int NIA = TM * 8 / (64 * NW), NIW = TN * 8 / (64 * NW);  // DMA instructions per wave and slab
  float* C = reinterpret_cast<float*>(Cv);
  __bf16* Ch = reinterpret_cast<__bf16*>(Cv);
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // [2][A slab | W slab]
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware, bijective remap of the linear workgroup id (8 XCDs, round-robin dispatch)
  const uint32_t nwg = gridDim.x, orig = blockIdx.x;
  const uint32_t q8 = nwg / 8, r8 = nwg % 8, xcd = orig % 8;
  const uint32_t wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + orig / 8;
  // grouped order inside an XCD's range: the workgroups resident on an XCD together cover 8 row
  // tiles x a few column tiles, whose current A and W slabs stay in its 4 MiB L2
  const uint32_t ntm = nwg / ntn;
  constexpr uint32_t GM = 8;
  const uint32_t group = wgid / (GM * ntn), in_group = wgid % (GM * ntn);
  const uint32_t gm = ntm - group * GM < GM ? ntm - group * GM : GM;
  const uint64_t m0 = (uint64_t)(group * GM + in_group % gm) * TM, n0 = (uint64_t)(in_group / gm) * TN;
  const uint32_t wm = (wave / WN) * (32 * MF), wn = (wave % WN) * (32 * NF);
  // v_mfma_f32_16x16x32_bf16 on 16 x 16 blocks instead of 32x32x16 on 32 x 32 ones: the same LDS image,
  // fragment bytes and accumulator registers per wave, twice the MFMAs at half the cycles each, the
  // same bits in every output -- and 7-8 % faster on 4096 x 65536 x 4096 (1.21-1.23 against 1.12-1.14
  // PFLOP/s kernel-only; the chip holds a higher clock under the smaller instruction).  EXP bit 6
  // selects the 32x32x16 form for comparison.
  constexpr bool MI16 = (EXP & 64) == 0;
  typedef float floatx4_t __attribute__((ext_vector_type(4)));
  floatx16 acc[MI16 ? 1 : MF][MI16 ? 1 : NF];
  floatx4_t acc4[MI16 ? 2 * MF : 1][MI16 ? 2 * NF : 1];
#pragma unroll
  for (int i = 0; i < (MI16 ? 1 : MF); ++i)
#pragma unroll
    for (int j = 0; j < (MI16 ? 1 : NF); ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
  for (int i = 0; i < (MI16 ? 2 * MF : 1); ++i)
#pragma unroll
    for (int j = 0; j < (MI16 ? 2 * NF : 1); ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc4[i][j][r] = 0.0f;
  // staging: instruction i of wave w fills LDS chunks [(NW i + w) * 64, + 64) of a slab image;
  // chunk = (row, c) with row = chunk / 8
  const __bf16* asrc[NIA];
  const __bf16* wsrc[NIW];
#pragma unroll
  for (int i = 0; i < NIA; ++i) {
    const uint32_t chunk = (NW * i + wave) * 64u + lane, row = chunk >> 3, c = (chunk & 7u) ^ ((row >> 1) & 7u);
    const uint64_t ra = m0 + row < M ? m0 + row : (uint64_t)M - 1;
    asrc[i] = A + ra * K + c * 8u;
  }
#pragma unroll
  for (int i = 0; i < NIW; ++i) {
    const uint32_t chunk = (NW * i + wave) * 64u + lane, row = chunk >> 3, c = (chunk & 7u) ^ ((row >> 1) & 7u);
    const uint64_t rw = n0 + row < N ? n0 + row : (uint64_t)N - 1;
    wsrc[i] = W + rw * K + c * 8u;
  }
  auto issue = [&](uint32_t k0, uint32_t buf) {
    unsigned char* ab = lds + buf * BUF;
    unsigned char* wb = ab + ABYTES;
#pragma unroll
    for (int i = 0; i < NIA; ++i)  // wave-uniform destination, lanes follow linearly
      __builtin_amdgcn_global_load_lds((isl_glb_void*)(asrc[i] + k0), (isl_lds_void*)(ab + (NW * i + wave) * 1024u), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NIW; ++i)
      __builtin_amdgcn_global_load_lds((isl_glb_void*)(wsrc[i] + k0), (isl_lds_void*)(wb + (NW * i + wave) * 1024u), 16, 0, 0);
  };
  const uint32_t kh = lane >> 5, c32 = lane & 31;
  const uint32_t nk = K / HBK;
  if ((EXP & 2) && gridDim.x > 256) {
    const uint32_t naps = (blockIdx.x * 37u) % 12u;
    for (uint32_t z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(127);
  }
  // the two waves of a SIMD (w and w + 4) do not queue on the texture-address unit together: the
  // second issues its DMA pieces one k-step (a quarter of the slab) later (+3 % on 4096 x 65536 x 4096;
  // EXP bit 0 turns it off, bits 2-3 move it)
  const bool late = !(EXP & 1) && NW == 8 && wave >= NW / 2;
  const bool stamp = (EXP & 128) && blockIdx.x == 300 && (wave & 3) == 0 && lane == 0;
  uint64_t tacc[5] = {0, 0, 0, 0, 0};
  issue(0, 0);
  for (uint32_t kt = 0; kt < nk; ++kt) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (EXP & 128) t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's share of slab kt has landed
    if (EXP & 128) t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();  // everyone's share has; everyone is done reading the other buffer
    if (EXP & 128) t2 = __builtin_amdgcn_s_memtime();
    if (kt + 1 < nk && !late) issue((kt + 1) * HBK, (kt + 1) & 1);
    if (EXP & 128) t3 = __builtin_amdgcn_s_memtime();
    const unsigned char* ab = lds + (kt & 1) * BUF;
    const unsigned char* wb = ab + ABYTES;
    if constexpr (MI16) {
#pragma unroll
      for (int s32 = 0; s32 < HBK / 32; ++s32) {
        if (s32 == 0 && late && kt + 1 < nk) {}  // (late issue below, behind the first MFMA groups)
        const uint32_t g = lane >> 4, c16 = lane & 15, cl = 4 * s32 + g;  // lane group g holds k = 8 g .. 8 g + 7 of the step
        bf16x8 b[2 * NF], a[2 * MF];
#pragma unroll
        for (int j = 0; j < 2 * NF; ++j) {
          const uint32_t rb = wn + 16 * j + c16;
          b[j] = *reinterpret_cast<const bf16x8*>(wb + rb * 128u + ((cl ^ ((rb >> 1) & 7u)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 2 * MF; ++i) {
          const uint32_t ra = wm + 16 * i + c16;
          a[i] = *reinterpret_cast<const bf16x8*>(ab + ra * 128u + ((cl ^ ((ra >> 1) & 7u)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 2 * MF; ++i) {
#pragma unroll
          for (int j = 0; j < 2 * NF; ++j) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc4[i][j], 0, 0, 0);
          // (EXP bits 2-3 move the late waves' issue point: after MFMA group 1 / 2 / 6 of 16 instead of 4)
          constexpr int LATE_AT = ((EXP >> 2) & 3) == 1 ? 0 : ((EXP >> 2) & 3) == 2 ? 1 : ((EXP >> 2) & 3) == 3 ? 5 : MF - 1;
          if (s32 == 0 && i == LATE_AT && late && kt + 1 < nk) issue((kt + 1) * HBK, (kt + 1) & 1);
        }
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < HBK / 16; ++ks) {
      if (ks == (((EXP >> 2) & 3) ? ((EXP >> 2) & 3) : 1) && late && kt + 1 < nk) issue((kt + 1) * HBK, (kt + 1) & 1);
      const uint32_t cl = 2 * ks + kh;  // logical 16-byte chunk of the row
      bf16x8 b[NF];
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const uint32_t rb = wn + 32 * j + c32;
        b[j] = *reinterpret_cast<const bf16x8*>(wb + rb * 128u + ((cl ^ ((rb >> 1) & 7u)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const uint32_t ra = wm + 32 * i + c32;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(ab + ra * 128u + ((cl ^ ((ra >> 1) & 7u)) << 4));
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc[i][j], 0, 0, 0);
      }
    }
    }
    if (EXP & 128) {
      const uint64_t t4 = __builtin_amdgcn_s_memtime();
      if (kt >= 8 && kt < 56) { tacc[0] += t1 - t0; tacc[1] += t2 - t1; tacc[2] += t3 - t2; tacc[3] += t4 - t3; tacc[4] += 1; }
    }
  }
  if ((EXP & 128) && stamp && dbg) {
#pragma unroll
    for (int z = 0; z < 5; ++z) dbg[(wave >> 2) * 8 + z] = tacc[z];
  }
  // rm = R[m] for the distance epilogues (|a_m|^2), loaded once per row by the caller
  auto emit = [&](uint64_t m, uint64_t n, float bv, float rm, float v) {
    if (ACT <= 2) {
      v += bv;
      if (ACT == 1) v = gelu_erf_f(v);
      if (ACT == 2) v = gelu_tanh_f(v);
      if (RES) v += R[m * N + n];
    } else if (ACT == EPI_COSINE) {  // bias = |w_n|^2 per column, R = |a_m|^2 per row
      v = epi_cosine(v, rm, bv);
    } else if (ACT == EPI_DOT) {
      v = -v;
    } else if (ACT == EPI_EUCLIDEAN) {  // |a|^2 + |w|^2 - 2 a.w, clamped
      v = epi_euclidean(v, rm, bv);
    }
    if constexpr (C16) Ch[m * N + n] = (__bf16)v;
    else C[(uint64_t)m * ldc + n] = v;
  };
  constexpr bool ROWNORM = ACT == EPI_COSINE || ACT == EPI_EUCLIDEAN;
  if constexpr (MI16) {
    float bvs[2 * NF];
#pragma unroll
    for (int j = 0; j < 2 * NF; ++j) {
      const uint64_t n = n0 + wn + j * 16 + (lane & 15);
      bvs[j] = (bias && n < N) ? bias[n] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 2 * MF; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint64_t m = m0 + wm + i * 16 + 4 * (lane >> 4) + r;
        if (m >= M) continue;
        const float rm = ROWNORM ? R[m] : 0.0f;
#pragma unroll
        for (int j = 0; j < 2 * NF; ++j) {
          const uint64_t n = n0 + wn + j * 16 + (lane & 15);
          if (n < N) emit(m, n, bvs[j], rm, acc4[i][j][r]);
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const uint64_t n = n0 + wn + j * 32 + c32;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint64_t m = m0 + wm + i * 32 + 8 * (r / 4) + 4 * kh + (r % 4);
          if (m < M) emit(m, n, bv, ROWNORM ? R[m] : 0.0f, acc[i][j][r]);
        }
      }
  }
}

// ---- the 256 x 256 tile on the eight-phase schedule of the CDNA4 guide's GEMM template ----
// (cdna_hip_programming.md, "The 256^2 8-phase template": the structure is the guide's, written out
// here from its description.)  Same tile, wave decomposition (2 x 4 waves, 128 x 64 per wave),
// 16x16x32 MFMAs, LDS image and accumulation order per output element as gemm_tn_bf16_dma<..,2,4,4,2>
// above -- every output bit is the same -- but the K loop is cut differently:
//   * a K-tile (64 deep) is FOUR half-tiles of 16 KiB: A0 / A1 = the rows of the first / second
//     64-row half of every wave's 128 rows, B0 / B1 = the first / second 32 columns of every wave's 64;
//     a wave's quadrant (ai, bj) of its output reads half-tiles A<ai> and B<bj> only;
//   * four phases per K-tile, one quadrant each: (0,0) (0,1) (1,1) (1,0).  A phase = [fragment reads of
//     what the quadrant still lacks: B0 + A0 | B1 | A1 | nothing] + [ONE half-tile of prefetch: 2 LDS-DMA
//     instructions per wave] -> s_barrier -> lgkmcnt(0) -> 16 MFMAs -> s_barrier;
//   * the prefetch runs three half-tiles ahead of the K-tile being waited for: phase 1 of K-tile s
//     stages A1(s+1), phases 2, 3, 4 stage B0, A0, B1 of K-tile s+2 into the buffer being read -- each
//     slot two phases after its last fragment read (B0: one phase, its reads are retired by the
//     lgkmcnt(8) in front of phase 1's barrier) -- and the only vmcnt wait of a K-tile sits in phase 4,
//     counted (6 = the three half-tiles staged since), never 0: nothing drains;
//   * waves 4..7 run one barrier behind waves 0..3, so that on every SIMD one wave's MFMA section
//     lies beside its partner's read / DMA section (matrix beside memory).
// Round 2's stamps said the 8-wave kernel above loses a third of a slab's time to the DMA issue of
// all pieces right behind the barrier; here a phase issues two.
// VAR (tools/microbench/gemm_bf16_ph8.hip only): bit 0 = no s_setprio around the MFMA clusters; bit 1 = one
// static s_setprio 1 for waves 4..7 instead.
template <int ACT, bool RES, bool C16, int VAR = 0>
__global__ __launch_bounds__(512) void gemm_tn_bf16_ph8(const __bf16* __restrict__ A, const __bf16* __restrict__ W,
                                                        const float* __restrict__ bias, const float* __restrict__ R,
                                                        void* __restrict__ Cv, uint32_t M, uint32_t N, uint32_t K,
                                                        uint32_t ntn, uint64_t ldc) {
  constexpr uint32_t TM = 256, TN = 256, HT = 16384, KT = 4 * HT;  // half-tile, K-tile image [A0 | A1 | B0 | B1]
  float* C = reinterpret_cast<float*>(Cv);
  __bf16* Ch = reinterpret_cast<__bf16*>(Cv);
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // [2][KT]
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t nwg = gridDim.x, orig = blockIdx.x;
  const uint32_t q8 = nwg / 8, r8 = nwg % 8, xcd = orig % 8;
  const uint32_t wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + orig / 8;
  const uint32_t ntm = nwg / ntn;
  constexpr uint32_t GM = 8;
  const uint32_t group = wgid / (GM * ntn), in_group = wgid % (GM * ntn);
  const uint32_t gm = ntm - group * GM < GM ? ntm - group * GM : GM;
  const uint64_t m0 = (uint64_t)(group * GM + in_group % gm) * TM, n0 = (uint64_t)(in_group / gm) * TN;
  const uint32_t wr = wave >> 2, wc = wave & 3;
  const uint32_t wm = wr * 128, wn = wc * 64;
  typedef float floatx4_t __attribute__((ext_vector_type(4)));
  floatx4_t acc4[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc4[i][j][r] = 0.0f;
  // staging: piece q = 8 i + wave (i = 0, 1) of a half-tile image = its rows 8 q .. 8 q + 7 (1 KiB); LDS
  // chunk `pos` of image row r holds the row's 16-byte chunk pos ^ ((r >> 1) & 7)
  const __bf16* src[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t chunk = (8u * i + wave) * 64u + lane, r = chunk >> 3, c = (chunk & 7u) ^ ((r >> 1) & 7u);
      if (t < 2) {  // A<t>: image row r = 64 wr' + x  <->  tile row 128 wr' + 64 t + x
        const uint64_t row = m0 + (r >> 6) * 128u + t * 64u + (r & 63u);
        src[t][i] = A + (row < M ? row : (uint64_t)M - 1) * K + c * 8u;
      } else {      // B<t-2>: image row r = 32 wc' + x  <->  tile column 64 wc' + 32 (t - 2) + x
        const uint64_t col = n0 + (r >> 5) * 64u + (t - 2) * 32u + (r & 31u);
        src[t][i] = W + (col < N ? col : (uint64_t)N - 1) * K + c * 8u;
      }
    }
  auto stage = [&](auto tt, uint32_t kt) {
    constexpr int t = decltype(tt)::value;
    unsigned char* dst = lds + (kt & 1u) * KT + t * HT + wave * 1024u;
    __builtin_amdgcn_global_load_lds((isl_glb_void*)(src[t][0] + (uint64_t)kt * HBK), (isl_lds_void*)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((isl_glb_void*)(src[t][1] + (uint64_t)kt * HBK), (isl_lds_void*)(dst + 8192u), 16, 0, 0);
  };
  using T_A0 = std::integral_constant<int, 0>; using T_A1 = std::integral_constant<int, 1>;
  using T_B0 = std::integral_constant<int, 2>; using T_B1 = std::integral_constant<int, 3>;
  // fragment reads: lane group g = lane / 16 holds k = 8 g .. 8 g + 7 of a 32-deep step, c16 = its row in
  // the 16-row block; (row >> 1) & 7 == c16 >> 1 for every block (block starts are multiples of 16)
  const uint32_t g = lane >> 4, c16 = lane & 15, sw = c16 >> 1;
  const uint32_t aoff0 = (wr * 64u + c16) * 128u + (((0u + g) ^ sw) << 4), aoff1 = (wr * 64u + c16) * 128u + (((4u + g) ^ sw) << 4);
  const uint32_t boff0 = (wc * 32u + c16) * 128u + (((0u + g) ^ sw) << 4), boff1 = (wc * 32u + c16) * 128u + (((4u + g) ^ sw) << 4);
  bf16x8 fa[4][2];      // the A half of the current quadrants: 4 row blocks x 2 k-steps
  bf16x8 fb0[2][2], fb1[2][2];  // B0 and B1 stay in registers for the whole K-tile
  auto read_a = [&](const unsigned char* img) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i][0] = *reinterpret_cast<const bf16x8*>(img + aoff0 + i * 2048u);
      fa[i][1] = *reinterpret_cast<const bf16x8*>(img + aoff1 + i * 2048u);
    }
  };
  auto read_b = [&](bf16x8 (&fb)[2][2], const unsigned char* img) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fb[j][0] = *reinterpret_cast<const bf16x8*>(img + boff0 + j * 2048u);
      fb[j][1] = *reinterpret_cast<const bf16x8*>(img + boff1 + j * 2048u);
    }
  };
  auto quadrant = [&](auto ai_, auto bj_, const bf16x8 (&fb)[2][2]) {
    constexpr int ai = decltype(ai_)::value, bj = decltype(bj_)::value;
    if constexpr (!(VAR & 1)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s32 = 0; s32 < 2; ++s32)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc4[4 * ai + i][2 * bj + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][s32], fb[j][s32], acc4[4 * ai + i][2 * bj + j], 0, 0, 0);
    if constexpr (!(VAR & 1)) __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  auto barrier = [] {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto reads_done = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  const uint32_t nk = K / HBK;
  // prologue: K-tile 0 whole, then the three half-tiles of K-tile 1 that phases 2-4 of "K-tile -1" would have staged
  stage(T_A0{}, 0); stage(T_B0{}, 0); stage(T_B1{}, 0); stage(T_A1{}, 0);
  if (nk > 1) {
    stage(T_B0{}, 1); stage(T_A0{}, 1); stage(T_B1{}, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  barrier();               // every wave's pieces of K-tile 0 have landed
  if (wr == 1) barrier();  // waves 4..7 run one barrier behind from here on
  if constexpr ((VAR & 2) != 0) {
    if (__builtin_amdgcn_readfirstlane(wr) == 1) __builtin_amdgcn_s_setprio(1);
  }
  for (uint32_t s = 0; s < nk; ++s) {
    const unsigned char* cur = lds + (s & 1u) * KT;
    // phase 1: quadrant (0, 0)
    read_b(fb0, cur + 2 * HT);
    __builtin_amdgcn_sched_barrier(0);
    read_a(cur);
    if (s + 1 < nk) stage(T_A1{}, s + 1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the four B0 reads are back: its slot is restaged next phase
    barrier();
    reads_done();
    quadrant(I0{}, I0{}, fb0);
    barrier();
    // phase 2: quadrant (0, 1)
    read_b(fb1, cur + 3 * HT);
    if (s + 2 < nk) stage(T_B0{}, s + 2);
    barrier();
    reads_done();
    quadrant(I0{}, I1{}, fb1);
    barrier();
    // phase 3: quadrant (1, 1)
    read_a(cur + HT);
    if (s + 2 < nk) stage(T_A0{}, s + 2);
    barrier();
    reads_done();
    quadrant(I1{}, I1{}, fb1);
    barrier();
    // phase 4: quadrant (1, 0); the K-tile's one wait: everything but the last three half-tiles has landed
    if (s + 2 < nk) {
      stage(T_B1{}, s + 2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (s + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    barrier();
    quadrant(I1{}, I0{}, fb0);
    barrier();
  }
  if (wr == 0) barrier();  // (the barrier waves 4..7 took at the start)
  auto emit = [&](uint64_t m, uint64_t n, float bv, float rm, float v) {
    if (ACT <= 2) {
      v += bv;
      if (ACT == 1) v = gelu_erf_f(v);
      if (ACT == 2) v = gelu_tanh_f(v);
      if (RES) v += R[m * N + n];
    } else if (ACT == EPI_COSINE) {
      v = epi_cosine(v, rm, bv);
    } else if (ACT == EPI_DOT) {
      v = -v;
    } else if (ACT == EPI_EUCLIDEAN) {
      v = epi_euclidean(v, rm, bv);
    }
    if constexpr (C16) Ch[m * N + n] = (__bf16)v;
    else C[(uint64_t)m * ldc + n] = v;
  };
  constexpr bool ROWNORM = ACT == EPI_COSINE || ACT == EPI_EUCLIDEAN;
  float bvs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t n = n0 + wn + j * 16 + (lane & 15);
    bvs[j] = (bias && n < N) ? bias[n] : 0.0f;
  }
  if constexpr ((VAR & 4) != 0 && !C16 && ACT >= EPI_COSINE) {
    // Round 4: the distance epilogues go through a wave-private patch of the (now idle) K-tile buffers --
    // 8 rows x 64 columns, pitch 288 bytes: the four row groups of an MFMA block land on different banks --
    // ds_write_b32 of the finished values, then ds_read_b128 and ONE global_store_dwordx4 per four whole
    // 256-byte rows: 32 store instructions per lane instead of 128 of four 64-byte row segments each.
    // Same values, bit for bit (the launcher takes this form when ldc % 4 == 0 and C is 16-byte aligned).
    // Measured with it (tools/microbench/gemm_bf16_ph8.hip, profiles/r04_gemm_bf16_ph8p_variants.log): as part
    // of a persistent workgroup per CU that also stages the next tile's first K-tiles under its epilogue
    // (gemm_tn_bf16_ph8p below) the cosine kernel gains 2.5 % over the per-element epilogue, but the
    // persistent structure itself loses 5 % to this one-tile kernel, and non-temporal stores another 2-4 %.
    constexpr uint32_t EP_PITCH = 72;
    float* const ep = reinterpret_cast<float*>(lds) + wave * (8 * EP_PITCH);
    const uint32_t g4 = lane >> 4, c = lane & 15;
    // VAR & 8 (round 4): the wave's 128 row norms come through LDS, loaded ONCE in front of the stores.  On gfx950
    // loads and stores share vmcnt and the compiler must take a counter with both kinds pending as unordered:
    // every R[m] read between the stores became `global_load_dword; s_waitcnt vmcnt(0)` -- 32 times per lane
    // the wave sat until ALL its earlier result stores were acknowledged (the cosine kernel's 0.19 ms over the
    // dot kernel's 1.75 at 4096 x 65536 x 4096).  And when no (row, column) pair of the wave's block has
    // |a|^2 |w|^2 below FLT_MIN (min over rows x min over columns: float multiplication is monotonic), the
    // block takes epi_cosine's main branch without the per-element test: same bits.
    constexpr bool LNORM = (VAR & 8) != 0 && ROWNORM;
    float* const rnl = reinterpret_cast<float*>(lds) + 8 * (8 * EP_PITCH) + wave * 128;
    int fast_i = 0;
    if constexpr (LNORM) {
      float mr = 3.0e38f, mc = 3.0e38f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint64_t m = m0 + wm + lane + 64 * t;
        const float v = m < M ? R[m] : 0.0f;
        rnl[lane + 64 * t] = v;
        if (m < M) mr = fminf(mr, v);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + wn + j * 16 + c < N) mc = fminf(mc, bvs[j]);
#pragma unroll
      for (int off = 32; off; off >>= 1) {
        mr = fminf(mr, __shfl_xor(mr, off));
        mc = fminf(mc, __shfl_xor(mc, off));
      }
      fast_i = __builtin_amdgcn_readfirstlane((ACT == EPI_COSINE && mr * mc >= 1.17549435e-38f) ? 1 : 0);
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
    }
    auto body = [&](auto fast_) {
      constexpr bool FAST = decltype(fast_)::value;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // rows 4 g4 + 2 h + {0, 1} of the 16-row block -> patch rows 2 g4 + {0, 1}
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * h + rr;
            const uint64_t m = m0 + wm + i * 16 + 4 * g4 + r;
            float rm;
            if constexpr (LNORM) rm = rnl[i * 16 + 4 * g4 + r];
            else rm = (ROWNORM && m < M) ? R[m] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v = acc4[i][j][r];
              if (ACT == EPI_COSINE) v = FAST ? 1.0f - v * __builtin_amdgcn_rsqf(rm * bvs[j]) : epi_cosine(v, rm, bvs[j]);
              else if (ACT == EPI_DOT) v = -v;
              else v = epi_euclidean(v, rm, bvs[j]);
              ep[(g4 * 2 + rr) * EP_PITCH + j * 16 + c] = v;
            }
          }
          __builtin_amdgcn_wave_barrier();
          asm volatile("" ::: "memory");
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const uint32_t lrow = 4 * t + g4;  // patch row: group lrow / 2, row rr = lrow % 2
            const uint64_t m = m0 + wm + i * 16 + 4 * (lrow >> 1) + 2 * h + (lrow & 1u);
            const uint64_t n = n0 + wn + c * 4;
            const floatx4_t v = *reinterpret_cast<const floatx4_t*>(&ep[lrow * EP_PITCH + c * 4]);
            if (m < M) {
              float* out = C + m * ldc + n;
              if (n + 3 < N) {
                *reinterpret_cast<floatx4_t*>(out) = v;
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (n + e < N) out[e] = v[e];
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
          asm volatile("" ::: "memory");
        }
    };
    if (LNORM && fast_i) body(std::true_type{});
    else body(std::false_type{});
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint64_t m = m0 + wm + i * 16 + 4 * (lane >> 4) + r;
      if (m >= M) continue;
      const float rm = ROWNORM ? R[m] : 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint64_t n = n0 + wn + j * 16 + (lane & 15);
        if (n < N) emit(m, n, bvs[j], rm, acc4[i][j][r]);
      }
    }
}

// ---- the eight-phase kernel as ONE persistent workgroup per CU (round 4) ----
// gemm_tn_bf16_ph8 pays, per 256 x 256 tile: a prologue (K-tile 0 fetched with nothing to compute), and an
// epilogue of 128 global_store_dword per lane -- four 64-byte row segments per instruction -- with the
// matrix cores idle: 0.12 ms of a 1.8 ms call at 4096 x 65536 x 4096.  Here a workgroup walks its tiles
// (virtual block ids blockIdx.x, + gridDim.x, ...: the same XCD-aware tile order as the one-tile kernel):
//   * the K loop runs on across the tile boundary: the phases of a tile's last two K-tiles stage the
//     NEXT tile's K-tiles 0 and 1 (the slots they take are the ones K-tiles nk and nk + 1 would have
//     taken; buffer parity counts K-tiles globally), which land during the epilogue -- no prologue after
//     the first tile, and the phase-4 wait stays the counted vmcnt(6) (the epilogue's stores are older
//     than the six loads it leaves outstanding, loads return in order); the one-barrier stagger of waves
//     4..7 is taken up at every tile's start and given back before its epilogue;
//   * the epilogue goes through a wave-private LDS patch (8 rows x 64 columns, pitch 288 bytes: the four
//     row groups of an MFMA block land on different banks): ds_write_b32 of the finished values, then
//     ds_read_b128 and ONE global_store_dwordx4 per four whole 256-byte rows -- 32 store instructions per
//     lane instead of 128.
// MEASURED (profiles/r04_gemm_bf16_ph8p_variants.log, 4096 x 65536 x 4096, rounds interleaved with the one-tile
// kernel on one card): dot epilogue 1199 TFLOP/s against 1255 for gemm_tn_bf16_ph8, cosine 1145 against
// 1138; with this kernel's per-element epilogue 1191 / 1118 -- so the LDS-transposed epilogue is worth 0.7 % /
// 2.5 %, the persistent structure itself costs 5 % (static tile striding instead of the dispatcher's
// first-free-CU order, 251 registers against 229), and non-temporal result stores cost another 2-4 %.  The
// library therefore keeps the one-tile kernel and takes only the epilogue (gemm_tn_bf16_ph8, VAR & 4);
// this kernel stays for the micro-benchmark (ISL_GEMM_PERSIST=1 selects it in the library).
// Same accumulation order, same epilogue arithmetic: every output bit equals gemm_tn_bf16_ph8's
// (tools/microbench/gemm_bf16_ph8.hip compares all of them).  Needs K >= 128, ldc % 4 == 0, C 16-byte
// aligned, float32 output.  VAR (micro-benchmark only): bit 0 = non-temporal result stores,
// bit 1 = the one-tile kernel's per-element epilogue.
template <int ACT, bool RES, int VAR = 0>
__global__ __launch_bounds__(512) void gemm_tn_bf16_ph8p(const __bf16* __restrict__ A, const __bf16* __restrict__ W,
                                                         const float* __restrict__ bias, const float* __restrict__ R,
                                                         float* __restrict__ C, uint32_t M, uint32_t N, uint32_t K,
                                                         uint32_t ntn, uint64_t ldc, uint32_t ntiles) {
  constexpr uint32_t TM = 256, TN = 256, HT = 16384, KT = 4 * HT;
  constexpr uint32_t EP_PITCH = 72;  // floats per patch row (288 bytes: consecutive row pairs are 16 banks apart)
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // [2][KT] | 8 waves x [8][EP_PITCH] floats
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* const ep = reinterpret_cast<float*>(lds + 2 * KT) + wave * (8 * EP_PITCH);
  const uint32_t G = gridDim.x;
  const uint32_t ntm = ntiles / ntn;
  auto tile_origin = [&](uint32_t vb, uint32_t& m0, uint32_t& n0) {
    const uint32_t q8 = ntiles / 8, r8 = ntiles % 8, xcd = vb % 8;
    const uint32_t wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + vb / 8;
    constexpr uint32_t GM = 8;
    const uint32_t group = wgid / (GM * ntn), in_group = wgid % (GM * ntn);
    const uint32_t gm = ntm - group * GM < GM ? ntm - group * GM : GM;
    m0 = (group * GM + in_group % gm) * TM;
    n0 = (in_group / gm) * TN;
  };
  const uint32_t wr = wave >> 2, wc = wave & 3;
  const uint32_t wm = wr * 128, wn = wc * 64;
  typedef float floatx4_t __attribute__((ext_vector_type(4)));
  floatx4_t acc4[8][4];
  // staging source of piece i (= 0, 1) of half-tile t of the tile at (m0, n0): see gemm_tn_bf16_ph8
  // Staging source of piece i (= 0, 1) of half-tile t of the tile at (m0, n0), as an element offset from
  // the tile's operand panel (A + m0 K or W + n0 K: wave-uniform bases, so that the pieces are eight
  // 32-bit offsets instead of eight pointers); rows past M / N are clamped to the last one (their
  // products are never stored).  `tid_` = threadIdx.x: the look-ahead into the next tile passes an
  // opaque copy of it, so that its offsets are computed where they are used -- two K-tiles per tile --
  // instead of being hoisted out of the K loop as live registers.
  auto piece_off = [&](int t, int i, uint32_t m0, uint32_t n0, uint32_t tid_) -> uint32_t {
    const uint32_t chunk = 512u * i + tid_, r = chunk >> 3, c = (chunk & 7u) ^ ((r >> 1) & 7u);
    if (t < 2) {
      const uint32_t row = m0 + (r >> 6) * 128u + t * 64u + (r & 63u);
      return ((row < M ? row : M - 1) - m0) * K + c * 8u;
    }
    const uint32_t col = n0 + (r >> 5) * 64u + (t - 2) * 32u + (r & 31u);
    return ((col < N ? col : N - 1) - n0) * K + c * 8u;
  };
  uint32_t vb = blockIdx.x;
  uint32_t m0, n0, m1 = 0, n1 = 0;
  tile_origin(vb, m0, n0);
  bool has_next = vb + G < ntiles;
  if (has_next) tile_origin(vb + G, m1, n1);
  uint32_t soff[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i) soff[t][i] = piece_off(t, i, m0, n0, tid);
  const uint32_t nk = K / HBK;
  uint32_t par = 0;  // parity of this tile's K-tile 0 in the global count of K-tiles
  // K-tile `sl` of the current tile (sl < nk) or, past its end, K-tile sl - nk of the next one
  auto stage = [&](auto tt, uint32_t sl) {
    constexpr int t = decltype(tt)::value;
    unsigned char* dst = lds + ((par + sl) & 1u) * KT + t * HT + wave * 1024u;
    const __bf16 *p0, *p1;
    if (sl < nk) {
      const __bf16* base = (t < 2 ? A + (uint64_t)m0 * K : W + (uint64_t)n0 * K) + (uint64_t)sl * HBK;
      p0 = base + soff[t][0];
      p1 = base + soff[t][1];
    } else {
      uint32_t tid_ = tid;
      asm volatile("" : "+v"(tid_));
      const __bf16* base = (t < 2 ? A + (uint64_t)m1 * K : W + (uint64_t)n1 * K) + (uint64_t)(sl - nk) * HBK;
      p0 = base + piece_off(t, 0, m1, n1, tid_);
      p1 = base + piece_off(t, 1, m1, n1, tid_);
    }
    __builtin_amdgcn_global_load_lds((isl_glb_void*)p0, (isl_lds_void*)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((isl_glb_void*)p1, (isl_lds_void*)(dst + 8192u), 16, 0, 0);
  };
  using T_A0 = std::integral_constant<int, 0>; using T_A1 = std::integral_constant<int, 1>;
  using T_B0 = std::integral_constant<int, 2>; using T_B1 = std::integral_constant<int, 3>;
  const uint32_t g = lane >> 4, c16 = lane & 15, sw = c16 >> 1;
  const uint32_t aoff0 = (wr * 64u + c16) * 128u + (((0u + g) ^ sw) << 4), aoff1 = (wr * 64u + c16) * 128u + (((4u + g) ^ sw) << 4);
  const uint32_t boff0 = (wc * 32u + c16) * 128u + (((0u + g) ^ sw) << 4), boff1 = (wc * 32u + c16) * 128u + (((4u + g) ^ sw) << 4);
  bf16x8 fa[4][2];
  bf16x8 fb0[2][2], fb1[2][2];
  auto read_a = [&](const unsigned char* img) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i][0] = *reinterpret_cast<const bf16x8*>(img + aoff0 + i * 2048u);
      fa[i][1] = *reinterpret_cast<const bf16x8*>(img + aoff1 + i * 2048u);
    }
  };
  auto read_b = [&](bf16x8 (&fb)[2][2], const unsigned char* img) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fb[j][0] = *reinterpret_cast<const bf16x8*>(img + boff0 + j * 2048u);
      fb[j][1] = *reinterpret_cast<const bf16x8*>(img + boff1 + j * 2048u);
    }
  };
  auto quadrant = [&](auto ai_, auto bj_, const bf16x8 (&fb)[2][2]) {
    constexpr int ai = decltype(ai_)::value, bj = decltype(bj_)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s32 = 0; s32 < 2; ++s32)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc4[4 * ai + i][2 * bj + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][s32], fb[j][s32], acc4[4 * ai + i][2 * bj + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  auto barrier = [] {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto reads_done = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // prologue (first tile only): K-tile 0 whole, then the three half-tiles of K-tile 1 that phases 2-4 stage
  stage(T_A0{}, 0); stage(T_B0{}, 0); stage(T_B1{}, 0); stage(T_A1{}, 0);
  stage(T_B0{}, 1); stage(T_A0{}, 1); stage(T_B1{}, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  barrier();               // every wave's pieces of K-tile 0 have landed
  constexpr bool ROWNORM = ACT == EPI_COSINE || ACT == EPI_EUCLIDEAN;
  auto value = [&](uint64_t m, uint64_t n, float bv, float rm, float v) -> float {
    if (ACT <= 2) {
      v += bv;
      if (ACT == 1) v = gelu_erf_f(v);
      if (ACT == 2) v = gelu_tanh_f(v);
      if (RES) v += R[m * N + n];
    } else if (ACT == EPI_COSINE) {
      v = epi_cosine(v, rm, bv);
    } else if (ACT == EPI_DOT) {
      v = -v;
    } else if (ACT == EPI_EUCLIDEAN) {
      v = epi_euclidean(v, rm, bv);
    }
    return v;
  };
  for (;;) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[i][j][r] = 0.0f;
    if (wr == 1) barrier();  // waves 4..7 run one barrier behind through the tile's K loop
    for (uint32_t s = 0; s < nk; ++s) {
      const unsigned char* cur = lds + ((par + s) & 1u) * KT;
      // phase 1: quadrant (0, 0)
      read_b(fb0, cur + 2 * HT);
      __builtin_amdgcn_sched_barrier(0);
      read_a(cur);
      if (s + 1 < nk || has_next) stage(T_A1{}, s + 1);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the four B0 reads are back: its slot is restaged next phase
      barrier();
      reads_done();
      quadrant(I0{}, I0{}, fb0);
      barrier();
      // phase 2: quadrant (0, 1)
      const bool ahead2 = s + 2 < nk || has_next;
      read_b(fb1, cur + 3 * HT);
      if (ahead2) stage(T_B0{}, s + 2);
      barrier();
      reads_done();
      quadrant(I0{}, I1{}, fb1);
      barrier();
      // phase 3: quadrant (1, 1)
      read_a(cur + HT);
      if (ahead2) stage(T_A0{}, s + 2);
      barrier();
      reads_done();
      quadrant(I1{}, I1{}, fb1);
      barrier();
      // phase 4: quadrant (1, 0); the K-tile's one wait
      if (ahead2) {
        stage(T_B1{}, s + 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else if (s + 1 < nk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      barrier();
      quadrant(I1{}, I0{}, fb0);
      barrier();
    }
    // ---- epilogue of the tile at (m0, n0): all eight waves together (the barrier waves 4..7 took at the
    // tile's start; staggered, each half of the workgroup would sit out the other half's epilogue)
    if (wr == 0) barrier();
    float bvs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t n = (uint64_t)n0 + wn + j * 16 + (lane & 15);
      bvs[j] = (bias && n < N) ? bias[n] : 0.0f;
    }
    if constexpr ((VAR & 2) != 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint64_t m = (uint64_t)m0 + wm + i * 16 + 4 * (lane >> 4) + r;
          if (m >= M) continue;
          const float rm = ROWNORM ? R[m] : 0.0f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint64_t n = (uint64_t)n0 + wn + j * 16 + (lane & 15);
            if (n < N) C[(uint64_t)m * ldc + n] = value(m, n, bvs[j], rm, acc4[i][j][r]);
          }
        }
    } else {
      const uint32_t g4 = lane >> 4, c = lane & 15;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // rows 4 g4 + 2 h + {0, 1} of the 16-row block -> patch rows 2 g4 + {0, 1}
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * h + rr;
            const uint64_t m = (uint64_t)m0 + wm + i * 16 + 4 * g4 + r;
            const float rm = (ROWNORM && m < M) ? R[m] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint64_t n = (uint64_t)n0 + wn + j * 16 + c;
              ep[(g4 * 2 + rr) * EP_PITCH + j * 16 + c] = (m < M && n < N) ? value(m, n, bvs[j], rm, acc4[i][j][r]) : 0.0f;
            }
          }
          __builtin_amdgcn_wave_barrier();
          asm volatile("" ::: "memory");
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const uint32_t lrow = 4 * t + g4;  // patch row: group lrow / 2, row rr = lrow % 2
            const uint64_t m = (uint64_t)m0 + wm + i * 16 + 4 * (lrow >> 1) + 2 * h + (lrow & 1u);
            const uint64_t n = (uint64_t)n0 + wn + c * 4;
            const floatx4_t v = *reinterpret_cast<const floatx4_t*>(&ep[lrow * EP_PITCH + c * 4]);
            if (m < M) {
              float* out = C + m * ldc + n;
              if (n + 3 < N) {
                if constexpr ((VAR & 1) != 0) __builtin_nontemporal_store(v, reinterpret_cast<floatx4_t*>(out));
                else *reinterpret_cast<floatx4_t*>(out) = v;
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (n + e < N) out[e] = v[e];
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
          asm volatile("" ::: "memory");
        }
    }
    if (!has_next) break;
    par = (par + nk) & 1u;
    vb += G;
    m0 = m1;
    n0 = n1;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) soff[t][i] = piece_off(t, i, m0, n0, tid);
    has_next = vb + G < ntiles;
    if (has_next) tile_origin(vb + G, m1, n1);
  }
}

// Picks the tile by the size of the problem: the 256 x 256 tile needs enough tiles to fill the chip.
template <int ACT, bool RES, bool C16>
void launch_gemm_bf16_dma(const __bf16* A, const __bf16* W, const float* bias, const float* R, void* C,
                          uint64_t M, uint64_t N, uint64_t K, uint64_t ldc, hipStream_t st) {
  static const int tile_env = [] { const char* e = getenv("ISL_GEMM_TILE"); return e ? atoi(e) : 0; }();  // 128 / 256: A/B switch
  const uint64_t big = ((M + 255) / 256) * ((N + 255) / 256);
  const bool use_big = tile_env == 256 || (tile_env != 128 && big >= 512);
  // the 256 x 256 tile runs on the eight-phase schedule (gemm_tn_bf16_ph8: same outputs bit for bit, 4-5 %
  // faster on 4096 x 65536 x 4096, tools/microbench/gemm_bf16_ph8.hip); ISL_GEMM_PH8=0 keeps round 2's loop
  static const bool ph8 = [] { const char* e = getenv("ISL_GEMM_PH8"); return !e || atoi(e) != 0; }();
  // ... and as one persistent workgroup per CU with the LDS-transposed epilogue (gemm_tn_bf16_ph8p, round 4):
  // float32 outputs only; ISL_GEMM_PERSIST=0 keeps the one-tile kernel
  // (measured slower than the one-tile kernel, see gemm_tn_bf16_ph8p: off unless ISL_GEMM_PERSIST=1)
  static const bool persist = [] { const char* e = getenv("ISL_GEMM_PERSIST"); return e && atoi(e) != 0; }();
  if constexpr (!C16) {
    if (use_big && ph8 && persist && K >= 2 * HBK && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 && big < 0x7FFFFFFFull) {
      static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n > 0 ? n : 256;
      }();
      auto kern = gemm_tn_bf16_ph8p<ACT, RES>;
      constexpr size_t lds = 2 * (256 + 256) * HBK * 2 + 8 * 8 * 72 * 4;
      static const bool once = [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return true;
      }();
      (void)once;
      const uint32_t grid = (uint32_t)std::min<uint64_t>(big, (uint64_t)ncu);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, A, W, bias, R, (float*)C, (uint32_t)M, (uint32_t)N,
                         (uint32_t)K, (uint32_t)((N + 255) / 256), ldc, (uint32_t)big);
      return;
    }
  }
  if constexpr (!C16 && ACT >= EPI_COSINE) {
    // distance epilogues: rows stored whole through the LDS patch (gemm_tn_bf16_ph8, VAR & 4); ISL_GEMM_EPT=0: A/B
    static const bool ept = [] { const char* e = getenv("ISL_GEMM_EPT"); return !e || atoi(e) != 0; }();
    if (use_big && ph8 && ept && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0) {
      auto kern = gemm_tn_bf16_ph8<ACT, RES, C16, 12>;
      constexpr size_t lds = 2 * (256 + 256) * HBK * 2;
      static const bool once = [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return true;
      }();
      (void)once;
      hipLaunchKernelGGL(kern, dim3((uint32_t)big), dim3(512), lds, st, A, W, bias, R, C, (uint32_t)M, (uint32_t)N,
                         (uint32_t)K, (uint32_t)((N + 255) / 256), ldc);
      return;
    }
  }
  if (use_big && ph8) {
    auto kern = gemm_tn_bf16_ph8<ACT, RES, C16>;
    constexpr size_t lds = 2 * (256 + 256) * HBK * 2;
    static const bool once = [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      return true;
    }();
    (void)once;
    hipLaunchKernelGGL(kern, dim3((uint32_t)big), dim3(512), lds, st, A, W, bias, R, C, (uint32_t)M, (uint32_t)N,
                       (uint32_t)K, (uint32_t)((N + 255) / 256), ldc);
  } else if (use_big) {
    auto kern = gemm_tn_bf16_dma<ACT, RES, C16, 2, 4, 4, 2>;
    constexpr size_t lds = 2 * (256 + 256) * HBK * 2;
    static const bool once = [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      return true;
    }();
    (void)once;
    hipLaunchKernelGGL(kern, dim3((uint32_t)big), dim3(512), lds, st, A, W, bias, R, C, (uint32_t)M, (uint32_t)N,
                       (uint32_t)K, (uint32_t)((N + 255) / 256), ldc, (uint64_t*)nullptr);
  } else {
    const uint64_t ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    auto kern = gemm_tn_bf16_dma<ACT, RES, C16, 2, 2, 2, 2>;
    constexpr size_t lds = 2 * (BM + BN) * HBK * 2;
    hipLaunchKernelGGL(kern, dim3((uint32_t)(ntm * ntn)), dim3(256), lds, st, A, W, bias, R, C, (uint32_t)M,
                       (uint32_t)N, (uint32_t)K, (uint32_t)ntn, ldc, (uint64_t*)nullptr);
  }
}

template <int ACT, bool RES, bool A16, bool C16>
void launch_gemm_bf16(const void* A, const __bf16* W, const float* bias, const float* R, void* C,
                      uint64_t M, uint64_t N, uint64_t K, hipStream_t st) {
  // MF = 4 (256-row tiles, 128 x 64 per wave: 6 LDS operand reads per 8 MFMAs instead of 4 per 4)
  // was measured slower: 272 registers leave one wave per SIMD (27 ms against 18 ms end to end)
  static const bool no_dma = getenv("ISL_GEMM_NO_DMA") != nullptr;  // A/B switch for measurements
  if constexpr (A16) {
    const uint64_t ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    if (!no_dma && K % HBK == 0 && ntm * ntn < 0x7FFFFFFFull && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0) {
      launch_gemm_bf16_dma<ACT, RES, C16>((const __bf16*)A, W, bias, R, C, M, N, K, N, st);
      return;
    }
  }
  dim3 grid((uint32_t)((N + BN - 1) / BN), (uint32_t)((M + BM - 1) / BM));
  hipLaunchKernelGGL((gemm_tn_bf16<ACT, RES, A16, C16, 2>), grid, dim3(256), 0, st, A, W, bias, R, C,
                     (uint32_t)M, (uint32_t)N, (uint32_t)K);
}

// Distance epilogues (EPI_COSINE / EPI_DOT / EPI_EUCLIDEAN) over bf16 queries and rows: out [M][ldc] f32.
template <int EPI>
void launch_gemm_bf16_distance(const __bf16* Q, const __bf16* Rows, const float* row_norm2, const float* q_norm2,
                               float* out, uint64_t M, uint64_t N, uint64_t K, uint64_t ldc, hipStream_t st) {
  launch_gemm_bf16_dma<EPI, false, false>(Q, Rows, row_norm2, q_norm2, out, M, N, K, ldc, st);
}

static __global__ void f32_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (__bf16)src[i];
}

}  // namespace isl_gemm
