// Device helpers shared by the kernels of libislands_amd.so: wave-level utilities and the
// exact-order distance routine (one lane owns one row and runs the reference's strictly
// sequential f32 chain; rows are staged through an LDS tile with coalesced 16-byte loads).
// Translation units including this file must be built with -ffp-contract=off.
#pragma once

#include "common.hpp"

namespace isl_dev {

constexpr int SLAB = 64;            // floats of each row staged per step (256 B)
constexpr int TILE_LD = SLAB + 4;   // LDS row pitch in floats: 68*r mod 64 = 4r -> conflict-free b128
constexpr int TILE_ROWS = 64;
constexpr int METRIC_SUMSQ = 100;   // internal: sqrt(sum x*x) (normalize_vector, distance.rs:126)
constexpr int METRIC_EUCLID_SQ = 101;  // internal: sum (q-x)^2 without sqrt (pq.rs:295-299)

// ---------------------------------------------------------------- wave helpers
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ uint32_t rl_u(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ float rl_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t uni(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// OrderedFloat<f32> total order mapped to unsigned: -0 == +0, NaN == NaN, NaN greatest.
__device__ __forceinline__ uint32_t ordkey(float d) {
  if (d != d) return 0xFFFFFFFFu;
  uint32_t u = __float_as_uint(d);
  if (u == 0x80000000u) u = 0u;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ uint32_t hslot(uint32_t id, uint32_t bits) {
  return (id * 0x9E3779B1u) >> (32u - bits);
}

// apply_pruning_strategy, leann.rs:991-1016: Global/Local keep a prefix.
// Proportional (thread_rng, leann.rs:1043) takes the deterministic fallback `take(num_to_keep)`.
__device__ __forceinline__ uint32_t prune_keep(float prune_ratio, uint32_t strategy, uint32_t n,
                                               uint32_t results_len, uint32_t ef) {
  if (prune_ratio == 0.0f || n == 0) return n;
  float keepf = ceilf((float)n * (1.0f - prune_ratio));
  uint32_t num_to_keep = (uint32_t)keepf;
  if (num_to_keep < 1) num_to_keep = 1;
  if (strategy == ISL_PRUNE_GLOBAL) {
    float ratio = (float)results_len / (float)ef;
    float adj = ceilf((float)n * (1.0f - ratio * prune_ratio));
    uint32_t adjusted = adj > 0.0f ? (uint32_t)adj : 0u;
    if (adjusted < 1) adjusted = 1;
    return adjusted < n ? adjusted : n;
  }
  return num_to_keep < n ? num_to_keep : n;
}

// ------------------------------------------------------- exact-order distances
// One step of the reference's scalar loops (distance.rs:71-122); q = query element
// (a), x = row element (b).  Separate roundings: this file is built -ffp-contract=off.
template <int METRIC>
__device__ __forceinline__ void dstep(float q, float x, float& a0, float& a1) {
  if (METRIC == ISL_METRIC_COSINE) {
    a0 += q * x;  // dot += x*y
    a1 += x * x;  // norm_b += y*y
  } else if (METRIC == ISL_METRIC_EUCLIDEAN) {
    float diff = q - x;
    a0 += diff * diff;
  } else if (METRIC == ISL_METRIC_DOT) {
    a0 += q * x;
  } else if (METRIC == METRIC_SUMSQ) {
    a0 += x * x;
  } else if (METRIC == METRIC_EUCLID_SQ) {
    float diff = q - x;
    a0 += diff * diff;
  } else {
    a0 += fabsf(q - x);
  }
}

template <int METRIC>
__device__ __forceinline__ float dfinish(float a0, float a1, float q_norm) {
  if (METRIC == ISL_METRIC_COSINE) {
    float norm = sqrtf(q_norm * a1);  // (norm_a * norm_b).sqrt(), distance.rs:82
    if (norm == 0.0f) return 1.0f;
    return 1.0f - (a0 / norm);
  } else if (METRIC == ISL_METRIC_EUCLIDEAN) {
    return sqrtf(a0);
  } else if (METRIC == ISL_METRIC_DOT) {
    return -a0;
  } else if (METRIC == METRIC_SUMSQ) {
    return sqrtf(a0);
  }
  return a0;  // Manhattan, squared Euclidean
}

// Distances of R (<= 64) rows to the query held in LDS; lane r < R owns row `rid`
// and returns its distance.  tile: TILE_ROWS x TILE_LD floats of LDS.
template <int METRIC>
__device__ __forceinline__ float wave_distances(const float* __restrict__ emb, uint64_t stride, uint32_t d,
                                uint32_t rid, uint32_t R, const float* qs, float* tile,
                                float q_norm) {
  const int lane = threadIdx.x;
  const int sub = lane & 15;   // which float4 of the 64-float slab this lane moves
  const int rgrp = lane >> 4;  // which of the 4 rows of a piece
  const uint32_t npieces = (R + 3) >> 2;
  const uint32_t nslab = (d + SLAB - 1) / SLAB;
  // 16 named pieces (hipcc keeps a float4[16] indexed from unrolled loops in scratch)
#define ISL_FOR16(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
#define ISL_DECL(p)                                                        \
  float4 r##p = make_float4(0.f, 0.f, 0.f, 0.f);                           \
  const bool on##p = (uint32_t)(p) < npieces && (uint32_t)(4 * (p) + rgrp) < R; \
  const float* rp##p = emb + (uint64_t)__shfl(rid, (4 * (p) + rgrp) & 63) * stride + sub * 4;
  ISL_FOR16(ISL_DECL)
// piece p = rows 4p..4p+3, 256 B each: one fully coalesced wave-instruction
#define ISL_LOAD(p) if (on##p) r##p = *reinterpret_cast<const float4*>(rp##p + soff);
#define ISL_STORE(p) \
  if (on##p) *reinterpret_cast<float4*>(tile + (4 * (p) + rgrp) * TILE_LD + sub * 4) = r##p;
  float a0 = 0.0f, a1 = 0.0f;
  {
    const size_t soff = 0;
    ISL_FOR16(ISL_LOAD)
  }
  for (uint32_t s = 0; s < nslab; ++s) {
    ISL_FOR16(ISL_STORE)
    __syncthreads();
    if (s + 1 < nslab) {  // in flight while this slab is consumed
      const size_t soff = (size_t)(s + 1) * SLAB;
      ISL_FOR16(ISL_LOAD)
    }
    if ((uint32_t)lane < R) {
      const float* trow = tile + lane * TILE_LD;
      const float* qv = qs + s * SLAB;
      uint32_t cnt = d - s * SLAB;
      if (cnt >= (uint32_t)SLAB) {
#pragma unroll
        for (int j = 0; j < SLAB; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          float4 q = *reinterpret_cast<const float4*>(qv + j);
          dstep<METRIC>(q.x, x.x, a0, a1);
          dstep<METRIC>(q.y, x.y, a0, a1);
          dstep<METRIC>(q.z, x.z, a0, a1);
          dstep<METRIC>(q.w, x.w, a0, a1);
        }
      } else {
        for (uint32_t j = 0; j < cnt; ++j) dstep<METRIC>(qv[j], trow[j], a0, a1);
      }
    }
    __syncthreads();
  }
#undef ISL_FOR16
#undef ISL_DECL
#undef ISL_LOAD
#undef ISL_STORE
  return dfinish<METRIC>(a0, a1, q_norm);
}

// Loads query `qi` into LDS and returns norm_a (cosine) computed in reference order.
template <int METRIC>
__device__ __forceinline__ float load_query(const float* __restrict__ queries, uint32_t qi, uint32_t d, float* qs) {
  const float* q = queries + (uint64_t)qi * d;
  for (uint32_t j = threadIdx.x; j < d; j += 64) qs[j] = q[j];
  __syncthreads();
  float na = 0.0f;
  if (METRIC == ISL_METRIC_COSINE) {
    for (uint32_t j = 0; j < d; ++j) {
      float x = qs[j];
      na += x * x;  // norm_a += x*x, distance.rs:78
    }
  }
  return na;
}


}  // namespace isl_dev
