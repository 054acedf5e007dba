// float32 GEMM on the matrix cores, shared by the encoder (Linear layers) and the batched
// distance matrix (query x candidate contraction).
#pragma once

#include "common.hpp"

namespace isl_gemm {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// epilogues beyond the Linear ones (0 bias, 1 bias + GELU(erf), 2 bias + GELU(tanh)): distances
// from the dot products, with `bias` = per-column and `R` = per-row squared norms
constexpr int EPI_COSINE = 3, EPI_DOT = 4, EPI_EUCLIDEAN = 5;

// The distance epilogues of the GEMM forms of batch_calculate.  These paths are specified to 1e-5 of
// the reference (the matrix cores sum in another order than distance.rs does anyway), so the
// epilogue uses the hardware's 1-ulp v_rsq_f32 / v_sqrt_f32 instead of the correctly rounded sqrt
// and division this library is otherwise built with: per element ~5 instructions instead of ~50,
// which on the 4096 x 65536 x 4096 bf16 product was 11 % of the kernel.
__device__ __forceinline__ float epi_cosine(float dot, float na, float nb) {
  const float nn = na * nb;
  // v_rsq_f32 flushes a denormal input to zero (-> inf, NaN when dot == 0): below FLT_MIN take the
  // reference's own form, norm = sqrt(na * nb), 1 - dot / norm (distance.rs:82-87), correctly rounded
  if (nn < 1.17549435e-38f) return nn == 0.0f ? 1.0f : 1.0f - dot / sqrtf(nn);
  return 1.0f - dot * __builtin_amdgcn_rsqf(nn);
}
__device__ __forceinline__ float epi_euclidean(float dot, float na, float nb) {
  const float v = na + nb - 2.0f * dot;
  return v > 0.0f ? __builtin_amdgcn_sqrtf(v) : 0.0f;
}

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
          v = epi_cosine(v, R[m], bv);
        } else if (ACT == EPI_DOT) {
          v = -v;
        } else if (ACT == EPI_EUCLIDEAN) {  // |a|^2 + |w|^2 - 2 a.w, clamped
          v = epi_euclidean(v, R[m], bv);
        }
        C[m * N + n] = v;
      }
    }
}

// The same contraction with both operands staged by LDS-DMA (global_load_lds_dwordx4, no register
// stop) into two buffers of 32-deep slabs -- the arrangement of gemm_tn_bf16_dma (gemm_bf16.hip.h) for
// float32 operands.  A float32 MFMA takes 64 cycles, so a 256 x 256 x 32 slab is 128 MFMAs = 8192
// matrix-pipe cycles per wave: one barrier and one DMA round trip per slab disappear behind it, and
// a lane fetches its operands with one ds_read_b128 per four MFMAs instead of one ds_read_b32 per
// MFMA.  LDS image: row r = 128 bytes, 16-byte chunk c stored at c ^ ((r >> 1) & 7).  The k pairing
// of v_mfma_f32_32x32x2_f32 (lanes 0..31 supply one k, lanes 32..63 another) is (8m + e, 8m + 4 + e):
// lane half kh reads chunk 2m + kh and feeds its element e to MFMA e of step m.  The order in
// which an output element accumulates its products depends on K alone -- not on the tile variant,
// the position in the tile or M -- so an embedding does not depend on what it is batched with.
// Requirements (host-checked): K % 32 == 0, 16-byte aligned operands; rows past M / N are clamped.
typedef __attribute__((address_space(3))) void isl_lds_void;
typedef const __attribute__((address_space(1))) void isl_glb_void;
constexpr int FBK = 32;  // floats per slab row

// EXP (tools/microbench/gemm_f32_exp.hip only): bit 0 = per-element epilogue stores, bit 1 = no staggered start.
template <int ACT, bool RES, int WM, int WN, int MF, int NF, int EXP = 0>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((flatten)) void gemm_tn_f32_dma(const float* __restrict__ A,
                                                                const float* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ R,
                                                                float* __restrict__ C, uint32_t M, uint32_t N,
                                                                uint32_t K, uint32_t ntn, uint64_t ldc,
                                                                uint32_t ntiles) {
  constexpr uint32_t TM = 32 * MF * WM, TN = 32 * NF * WN, NW = WM * WN;
  constexpr uint32_t ABYTES = TM * FBK * 4, WBYTES = TN * FBK * 4, BUF = ABYTES + WBYTES;
  constexpr int NIA = TM * 8 / (64 * NW), NIW = TN * 8 / (64 * NW);  // DMA instructions per wave and slab
  constexpr int NP = NIA + NIW, STEPS = FBK / 8;
  constexpr uint32_t SCRATCH = 32 * 32 * NF * 4;  // the epilogue's transposing scratch per wave: 32 rows of the wave's columns
  static_assert(NP % STEPS == 0, "pieces are spread evenly over the steps of a slab");
  static_assert(NW * SCRATCH <= BUF, "the epilogue's scratch fits the buffer of the tile's last slab");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // [2][A slab | W slab]
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t kh = lane >> 5, c32 = lane & 31;
  const uint32_t wm = (wave / WN) * (32 * MF), wn = (wave % WN) * (32 * NF);
  const uint32_t nk = K / FBK;
  // Tile t of the (virtual) grid of ntiles workgroups: an XCD-aware, bijective remap of the linear id
  // (8 XCDs, round-robin dispatch; a persistent workgroup's tiles t, t + gridDim.x, ... stay on its XCD
  // when gridDim.x is a multiple of 8); inside an XCD's range the resident workgroups cover 8 row
  // tiles x a few column tiles.
  auto tile_origin = [&](uint32_t t, uint64_t& m0, uint64_t& n0) {
    const uint32_t q8 = ntiles / 8, r8 = ntiles % 8, xcd = t % 8;
    const uint32_t wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + t / 8;
    const uint32_t ntm = ntiles / ntn;
    constexpr uint32_t GM = 8;
    const uint32_t group = wgid / (GM * ntn), in_group = wgid % (GM * ntn);
    const uint32_t gm = ntm - group * GM < GM ? ntm - group * GM : GM;
    m0 = (uint64_t)(group * GM + in_group % gm) * TM;
    n0 = (uint64_t)(in_group / gm) * TN;
  };
  const float* asrc[NIA];
  const float* wsrc[NIW];
  auto set_sources = [&](uint64_t m0, uint64_t n0) {
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const uint32_t chunk = (NW * i + wave) * 64u + lane, row = chunk >> 3, c = (chunk & 7u) ^ ((row >> 1) & 7u);
      const uint64_t ra = m0 + row < M ? m0 + row : (uint64_t)M - 1;
      asrc[i] = A + ra * K + c * 4u;
    }
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
      const uint32_t chunk = (NW * i + wave) * 64u + lane, row = chunk >> 3, c = (chunk & 7u) ^ ((row >> 1) & 7u);
      const uint64_t rw = n0 + row < N ? n0 + row : (uint64_t)N - 1;
      wsrc[i] = W + rw * K + c * 4u;
    }
  };
  // DMA piece p of a slab (the A pieces first): 1 KiB = 8 rows per wave instruction
  auto piece = [&](int p, uint32_t k0, uint32_t buf) {
    unsigned char* ab = lds + buf * BUF;
    if (p < NIA)
      __builtin_amdgcn_global_load_lds((isl_glb_void*)(asrc[p] + k0), (isl_lds_void*)(ab + (NW * p + wave) * 1024u), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((isl_glb_void*)(wsrc[p - NIA] + k0),
                                       (isl_lds_void*)(ab + ABYTES + (NW * (p - NIA) + wave) * 1024u), 16, 0, 0);
  };
  // the fragments of step ms of a slab image: lane half kh reads 16-byte chunk 2 ms + kh (k = 8 ms + 4 kh + e)
  auto frags = [&](const unsigned char* ab, int ms, float4 (&a)[MF], float4 (&b)[NF]) {
    const unsigned char* wb = ab + ABYTES;
    const uint32_t cl = 2 * ms + kh;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const uint32_t rb = wn + 32 * j + c32;
      b[j] = *reinterpret_cast<const float4*>(wb + rb * 128u + ((cl ^ ((rb >> 1) & 7u)) << 4));
    }
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const uint32_t ra = wm + 32 * i + c32;
      a[i] = *reinterpret_cast<const float4*>(ab + ra * 128u + ((cl ^ ((ra >> 1) & 7u)) << 4));
    }
  };
  // the epilogue of one element; rv = the residual element R[m][n] (RES) or the row's squared norm R[m]
  // (distance epilogues), loaded by the caller: a load between the result stores costs a full
  // `s_waitcnt vmcnt(0)` on gfx950 (loads and stores share the counter, and with both kinds pending the
  // compiler must treat it as unordered), i.e. the wave sits until its earlier stores are acknowledged
  auto finish = [&](float v, float bv, float rv) -> float {
    if (ACT <= 2) {
      v += bv;
      if (ACT == 1) v = gelu_erf_f(v);
      if (ACT == 2) v = gelu_tanh_f(v);
      if (RES) v += rv;
    } else if (ACT == EPI_COSINE) {
      v = epi_cosine(v, rv, bv);
    } else if (ACT == EPI_DOT) {
      v = -v;
    } else if (ACT == EPI_EUCLIDEAN) {
      v = epi_euclidean(v, rv, bv);
    }
    return v;
  };
  constexpr bool ROWNORM = ACT == EPI_COSINE || ACT == EPI_EUCLIDEAN;
  const bool vec_ok = (N & 3u) == 0 && (ldc & 3u) == 0 && ((uintptr_t)C & 15u) == 0 && (!bias || ((uintptr_t)bias & 15u) == 0) &&
                      (!RES || ((uintptr_t)R & 15u) == 0);

  uint32_t t = blockIdx.x, par = 0;
  uint64_t m0, n0;
  tile_origin(t, m0, n0);
  set_sources(m0, n0);
  if (!(EXP & 2) && gridDim.x < ntiles) {
    // Equal tiles keep the CUs in lockstep, and then every CU writes its 256 KiB of results in the same
    // few microseconds: a burst the HBM write path cannot absorb while the matrix cores idle.  The
    // workgroups start up to ~45 us apart so that the epilogues spread over the tile time.
    const uint32_t naps = (blockIdx.x * 37u) % 12u;
    for (uint32_t z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(127);
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) piece(p, 0, 0);
  for (;;) {
    floatx16 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const uint32_t tnext = t + gridDim.x;
    const bool has_next = tnext < ntiles;
    uint64_t nm0 = 0, nn0 = 0;
    for (uint32_t kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's share of the slab has landed
      __syncthreads();  // everyone's share has; everyone is done with the other buffer
      const bool more = kt + 1 < nk;
      if (!more && has_next) {  // the last slab's shadow covers the next tile's first slab
        tile_origin(tnext, nm0, nn0);
        set_sources(nm0, nn0);
      }
      const bool fetch = more || has_next;
      const uint32_t knext = more ? (kt + 1) * FBK : 0u, bnext = (par + kt + 1) & 1u;
      const unsigned char* cur = lds + ((par + kt) & 1u) * BUF;
#pragma unroll
      for (int ms = 0; ms < STEPS; ++ms) {
        float4 a[MF], b[NF];
        frags(cur, ms, a, b);
        // e outermost: consecutive MFMAs go to different accumulators; an element still receives its
        // products in ascending (step, e) order
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(e == 0 ? a[i].x : e == 1 ? a[i].y : e == 2 ? a[i].z : a[i].w,
                                                               e == 0 ? b[j].x : e == 1 ? b[j].y : e == 2 ? b[j].z : b[j].w,
                                                               acc[i][j], 0, 0, 0);
          // the next slab's DMA pieces go out in the shadow of this step's MFMAs, not in a burst behind the barrier
          if (e == 0 && fetch) {
#pragma unroll
            for (int p = ms * (NP / STEPS); p < (ms + 1) * (NP / STEPS); ++p) piece(p, knext, bnext);
          }
        }
      }
    }
    if (EXP & 1) {
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
            if (m >= M) continue;
            C[m * ldc + n] = finish(acc[i][j][r], bv, RES ? R[m * N + n] : ROWNORM ? R[m] : 0.0f);
          }
        }
    } else {
      // Epilogue through LDS: an accumulator register holds one element of 16 different rows, so stored
      // as is a wave instruction writes 2 x 128 bytes; transposed through the wave's own 8 KiB of the
      // buffer the tile's last slab occupied, it writes 4 x 256 contiguous bytes (dwordx4 per lane).
      __syncthreads();  // everyone is done reading the last slab
      float* sc = reinterpret_cast<float*>(lds + ((par + nk - 1) & 1u) * BUF + wave * SCRATCH);
      constexpr uint32_t WCOLS = 32 * NF;  // the wave's columns
      constexpr uint32_t NQ = 32 * WCOLS / 256, RPQ = 64 / (WCOLS / 4);  // store rounds per row block; rows per round
      // a lane's columns are the same in every round (c4 = lane mod WCOLS / 4): its bias values are loaded once
      // per tile, and a row block's residual elements / row norms in one go before the block's stores
      const uint32_t c4 = lane % (WCOLS / 4), row0 = lane / (WCOLS / 4);
      const uint64_t nv = n0 + wn + 4 * c4;
      // (loads are unconditional on clamped addresses and pinned by an empty asm: the compiler otherwise sinks
      // each load back to its use between the stores)
      const uint64_t nvc = nv < N ? nv : 0;
      float4 bv4 = float4{0.0f, 0.0f, 0.0f, 0.0f};
      if (vec_ok && bias) {
        bv4 = *reinterpret_cast<const float4*>(bias + nvc);
        asm volatile("" : "+v"(bv4.x), "+v"(bv4.y), "+v"(bv4.z), "+v"(bv4.w));
      }
#pragma clang loop unroll(full)
      for (int i = 0; i < MF; ++i) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[(8 * (r / 4) + 4 * kh + (r % 4)) * WCOLS + 32 * j + c32] = acc[i][j][r];
        // (LDS operations of one wave complete in order: the reads below see the writes above)
        if (vec_ok) {
          auto store_row = [&](uint32_t q, const float4& rv, float rnq) {
            const uint32_t row = q * RPQ + row0;
            const float4 v = *reinterpret_cast<const float4*>(sc + row * WCOLS + 4 * c4);
            const uint64_t m = m0 + wm + 32 * i + row;
            if (m >= M || nv >= N) return;
            float4 o;
            o.x = finish(v.x, bv4.x, RES ? rv.x : rnq);
            o.y = finish(v.y, bv4.y, RES ? rv.y : rnq);
            o.z = finish(v.z, bv4.z, RES ? rv.z : rnq);
            o.w = finish(v.w, bv4.w, RES ? rv.w : rnq);
            *reinterpret_cast<float4*>(C + m * ldc + nv) = o;
          };
          if constexpr (RES || ROWNORM) {
            // eight rows' loads in one go, pinned in front of their eight stores (a 128 x 128 wave tile has 16 per block)
            constexpr uint32_t QC = NQ < 8 ? NQ : 8;
#pragma unroll
            for (uint32_t q0 = 0; q0 < NQ; q0 += QC) {
              float4 rv4[QC];
              float rn[QC];
#pragma unroll
              for (uint32_t q = 0; q < QC; ++q) {
                const uint64_t m = m0 + wm + 32 * i + (q0 + q) * RPQ + row0, mc = m < M ? m : (uint64_t)M - 1;
                rv4[q] = float4{0.0f, 0.0f, 0.0f, 0.0f};
                rn[q] = 0.0f;
                if (RES) rv4[q] = *reinterpret_cast<const float4*>(R + mc * N + nvc);
                if (ROWNORM) rn[q] = R[mc];
              }
#pragma unroll
              for (uint32_t q = 0; q < QC; ++q) {
                if (RES) asm volatile("" : "+v"(rv4[q].x), "+v"(rv4[q].y), "+v"(rv4[q].z), "+v"(rv4[q].w));
                if (ROWNORM) asm volatile("" : "+v"(rn[q]));
              }
#pragma unroll
              for (uint32_t q = 0; q < QC; ++q) store_row(q0 + q, rv4[q], rn[q]);
            }
          } else {
            const float4 zero = float4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (uint32_t q = 0; q < NQ; ++q) store_row(q, zero, 0.0f);
          }
        } else {
#pragma unroll
          for (uint32_t q = 0; q < NQ; ++q) {
            const uint32_t row = q * RPQ + row0;
            const float4 v = *reinterpret_cast<const float4*>(sc + row * WCOLS + 4 * c4);
            const uint64_t m = m0 + wm + 32 * i + row;
            if (m >= M || nv >= N) continue;
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (nv + u < N)
                C[m * ldc + nv + u] = finish(vv[u], bias ? bias[nv + u] : 0.0f, RES ? R[m * N + nv + u] : ROWNORM ? R[m] : 0.0f);
          }
        }
      }
    }
    if (!has_next) break;
    par = (par + nk) & 1u;
    t = tnext;
    m0 = nm0;
    n0 = nn0;
  }
}

template <int ACT, bool RES>
void launch_gemm(const float* A, const float* W, const float* bias, const float* R, float* C,
                 uint64_t M, uint64_t N, uint64_t K, hipStream_t st) {
  static const bool no_dma = getenv("ISL_GEMM_F32_NO_DMA") != nullptr;  // A/B switch for measurements
  if (!no_dma && K % FBK == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0) {
    // the 256 x 256 tile needs enough tiles to fill the chip; both variants add an element's products
    // in the same order
    const uint64_t big = ((M + 255) / 256) * ((N + 255) / 256);
    static const uint32_t cus = [] {
      int dev = 0, n = 0;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      return (uint32_t)(n > 0 ? n : 256);
    }();
    if (big >= 384) {
      // persistent: one workgroup per CU (128 KiB of LDS each) walks its tiles, the next tile's first
      // slab in flight under the current tile's last
      constexpr size_t lds = 2 * (256 + 256) * FBK * 4;
      const uint32_t grid = (uint32_t)(big < cus ? big : cus);
      // Round 4: the plain and the residual epilogue run the 256 x 256 tile on FOUR waves, 128 x 128 per wave (16
      // accumulator blocks = 256 accumulator registers, one wave per SIMD): 8 fragment reads per 64 MFMAs instead of 6
      // per 32, same k order, same bits.  Measured in one process (tools/microbench/gemm_f32_w4.hip,
      // profiles/r04_gemm_f32_w4.log; M = 524288): bias only 135.7 -> 139.5 TFLOP/s at K = 768 and 137.8 -> 143.6 at
      // K = 3072 (hipBLASLt: 141.5), bias + residual 125.4 -> 128.8 / 132.1 -> 139.7.  The GELU epilogue stays on eight
      // waves (126.9 against 118.8: with one wave per SIMD nothing runs beside a wave's 256 inlined erff), and so do
      // the distance epilogues.  ISL_GEMM_F32_W4=0: eight waves everywhere (A/B switch).
      static const bool w4 = [] { const char* e = getenv("ISL_GEMM_F32_W4"); return !e || atoi(e) != 0; }();
      if constexpr (ACT == 0) {
        if (w4) {
          auto kern = gemm_tn_f32_dma<ACT, RES, 2, 2, 4, 4>;
          static const bool once4 = [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            return true;
          }();
          (void)once4;
          hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, A, W, bias, R, C, (uint32_t)M, (uint32_t)N,
                             (uint32_t)K, (uint32_t)((N + 255) / 256), N, (uint32_t)big);
          return;
        }
      }
      auto kern = gemm_tn_f32_dma<ACT, RES, 2, 4, 4, 2>;
      static const bool once = [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return true;
      }();
      (void)once;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, A, W, bias, R, C, (uint32_t)M, (uint32_t)N,
                         (uint32_t)K, (uint32_t)((N + 255) / 256), N, (uint32_t)big);
    } else {
      const uint64_t ntm = (M + 127) / 128, ntn = (N + 127) / 128;
      auto kern = gemm_tn_f32_dma<ACT, RES, 2, 2, 2, 2>;
      constexpr size_t lds = 2 * (128 + 128) * FBK * 4;
      const uint32_t grid = (uint32_t)(ntm * ntn < 2 * cus ? ntm * ntn : 2 * cus);  // two workgroups fit a CU
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, A, W, bias, R, C, (uint32_t)M,
                         (uint32_t)N, (uint32_t)K, (uint32_t)ntn, N, (uint32_t)(ntm * ntn));
    }
    return;
  }
  dim3 grid((uint32_t)((N + BN - 1) / BN), (uint32_t)((M + BM - 1) / BM));
  hipLaunchKernelGGL((gemm_tn_f32<ACT, RES>), grid, dim3(256), 0, st, A, W, bias, R, C, (uint32_t)M,
                     (uint32_t)N, (uint32_t)K);
}


}  // namespace isl_gemm
