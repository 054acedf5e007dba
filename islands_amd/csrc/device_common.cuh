// Device helpers shared by the kernels of libislands_amd.so: wave-level utilities and the
// exact-order distance routine (one lane owns one row and runs the reference's strictly
// sequential f32 chain; rows are staged through an LDS tile with coalesced 16-byte loads).
// Translation units including this file must be built with -ffp-contract=off.
#pragma once

#include "common.hpp"

namespace isl_dev {

constexpr int PIECE = 128;           // floats of one row staged per step (512 B; a wave-instruction
                                     // moves the pieces of two rows)
constexpr int TILE_LD = PIECE + 4;   // LDS row pitch in floats: 132*r mod 64 = 4r -> conflict-free b128
constexpr int GROUP = 16;            // rows staged together
constexpr int TILE_ROWS = GROUP;
constexpr int METRIC_SUMSQ = 100;   // internal: sqrt(sum x*x) (normalize_vector, distance.rs:126)
constexpr int METRIC_EUCLID_SQ = 101;  // internal: sum (q-x)^2 without sqrt (pq.rs:295-299)
constexpr int METRIC_SUMSQ_RAW = 102;  // internal: sum x*x (norm_b of distance.rs:79, precomputed per row)
constexpr int METRIC_COSINE_PRE = 103; // internal: cosine with norm_b supplied per row (same value as
                                       // the on-the-fly chain: it is a sum over the row alone)

// ---------------------------------------------------------------- wave helpers
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ uint32_t rl_u(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ float rl_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// lane i receives the value of lane i-1 (lane 0 keeps `v`): DPP wave_shr:1, a plain VALU move
// on gfx9-family parts, instead of the LDS-crossbar round trip of __shfl_up.
__device__ __forceinline__ uint32_t shr1_u(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ float shr1_f(float v) { return __uint_as_float(shr1_u(__float_as_uint(v))); }
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
  } else if (METRIC == METRIC_COSINE_PRE) {
    a0 += q * x;  // dot += x*y; norm_b comes precomputed
  } else if (METRIC == METRIC_SUMSQ || METRIC == METRIC_SUMSQ_RAW) {
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
  if (METRIC == ISL_METRIC_COSINE || METRIC == METRIC_COSINE_PRE) {
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

// One group of Rg <= 2*NI rows (NI = load instructions per piece, compile-time).  Every
// wave-instruction moves one 512-byte piece (PIECE floats) of TWO rows, fully coalesced
// (32 lanes x 16 B per row); three pieces per row are in flight (register ring A/B/C) while
// lanes 0..Rg-1 run the sequential chains of the piece that already sits in the LDS tile
// (GROUP x TILE_LD floats, read back row-per-lane with conflict-free ds_read_b128:
// 132*r mod 64 = 4r).  All loads and LDS stores of a variant are unconditional straight-line
// code -- a predicate per instruction makes hipcc fall back to s_waitcnt vmcnt(0) before every
// store, which serialises the ring; lanes whose row does not exist re-read the group's first
// row (same cache lines) into a tile row that no lane consumes.
template <int METRIC, int NI>
__device__ __forceinline__ float group_distances(const float* __restrict__ emb, uint64_t stride,
                                                 uint32_t d, uint32_t rid, uint32_t g0, uint32_t Rg,
                                                 const float* qs, float* tile, float q_norm,
                                                 float row_aux, uint64_t* prof3) {
  const int lane = threadIdx.x;
  const int half = lane >> 5;        // which of the two rows of a load instruction
  const int col = (lane & 31) * 4;   // this lane's float4 inside the piece
  const uint32_t nT = (d + PIECE - 1) / PIECE;
#define ISL_FOR8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define ISL_DECL(j)                                                                          \
  const uint32_t rw##j = (uint32_t)(2 * (j) + half) < Rg ? (uint32_t)(2 * (j) + half) : 0u;  \
  const float* rp##j = emb + (uint64_t)__shfl(rid, (int)((g0 + rw##j) & 63)) * stride + col; \
  float4 ra##j = make_float4(0.f, 0.f, 0.f, 0.f), rb##j = ra##j, rc##j = ra##j;
  ISL_FOR8(ISL_DECL)
#define ISL_LOAD_A(j) if constexpr ((j) < NI) ra##j = *reinterpret_cast<const float4*>(rp##j + poff);
#define ISL_LOAD_B(j) if constexpr ((j) < NI) rb##j = *reinterpret_cast<const float4*>(rp##j + poff);
#define ISL_LOAD_C(j) if constexpr ((j) < NI) rc##j = *reinterpret_cast<const float4*>(rp##j + poff);
#define ISL_STORE_A(j) \
  if constexpr ((j) < NI) *reinterpret_cast<float4*>(tile + (2 * (j) + half) * TILE_LD + col) = ra##j;
#define ISL_STORE_B(j) \
  if constexpr ((j) < NI) *reinterpret_cast<float4*>(tile + (2 * (j) + half) * TILE_LD + col) = rb##j;
#define ISL_STORE_C(j) \
  if constexpr ((j) < NI) *reinterpret_cast<float4*>(tile + (2 * (j) + half) * TILE_LD + col) = rc##j;
  float a0 = 0.0f, a1 = 0.0f;
  auto consume = [&](uint32_t t) {
    if ((uint32_t)lane < Rg) {
      const float* trow = tile + lane * TILE_LD;
      // unroll 8 keeps 16 ds_read_b128 in flight: measured sweet spot for one wave
      // (16.2 cycles/element; unroll 16 -> 26.5, unroll 4 -> 18.9)
      const float* qv = qs + t * PIECE;
      const uint32_t cnt = d - t * PIECE;
      if (cnt >= (uint32_t)PIECE) {
#pragma unroll 8
        for (int j = 0; j < PIECE; j += 4) {
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
  };
  uint64_t tw0 = prof3 ? __builtin_amdgcn_s_memrealtime() : 0;
  {
    const size_t poff = 0;
    ISL_FOR8(ISL_LOAD_A)
  }
  if (nT > 1) {
    const size_t poff = PIECE;
    ISL_FOR8(ISL_LOAD_B)
  }
  if (nT > 2) {
    const size_t poff = 2 * PIECE;
    ISL_FOR8(ISL_LOAD_C)
  }
  for (uint32_t t = 0; t < nT; t += 3) {
    ISL_FOR8(ISL_STORE_A)
    __syncthreads();
    if (prof3 && t == 0) { uint64_t n_ = __builtin_amdgcn_s_memrealtime(); prof3[0] += n_ - tw0; tw0 = n_; }
    if (t + 3 < nT) {
      const size_t poff = (size_t)(t + 3) * PIECE;
      ISL_FOR8(ISL_LOAD_A)
    }
    consume(t);
    __syncthreads();
    if (prof3 && t == 0) { uint64_t n_ = __builtin_amdgcn_s_memrealtime(); prof3[1] += n_ - tw0; tw0 = n_; }
    if (t + 1 < nT) {
      ISL_FOR8(ISL_STORE_B)
      __syncthreads();
      if (t + 4 < nT) {
        const size_t poff = (size_t)(t + 4) * PIECE;
        ISL_FOR8(ISL_LOAD_B)
      }
      consume(t + 1);
      __syncthreads();
    }
    if (t + 2 < nT) {
      ISL_FOR8(ISL_STORE_C)
      __syncthreads();
      if (t + 5 < nT) {
        const size_t poff = (size_t)(t + 5) * PIECE;
        ISL_FOR8(ISL_LOAD_C)
      }
      consume(t + 2);
      __syncthreads();
    }
  }
#undef ISL_FOR8
#undef ISL_DECL
#undef ISL_LOAD_A
#undef ISL_LOAD_B
#undef ISL_LOAD_C
#undef ISL_STORE_A
#undef ISL_STORE_B
#undef ISL_STORE_C
  if (METRIC == METRIC_COSINE_PRE) a1 = __shfl(row_aux, (int)((g0 + lane) & 63));
  return dfinish<METRIC>(a0, a1, q_norm);
}

// Distances of R (<= 64) rows to the query held in LDS (`qs`); lane r < R owns row `rid` and
// returns its distance.  Rows are handled in groups of GROUP = 16 (see group_distances).
template <int METRIC>
__device__ __forceinline__ float wave_distances(const float* __restrict__ emb, uint64_t stride,
                                                uint32_t d, uint32_t rid, uint32_t R,
                                                const float* qs, float* tile, float q_norm,
                                                float row_aux = 0.0f, uint64_t* prof3 = nullptr) {
  const int lane = threadIdx.x;
  float result = 0.0f;
  for (uint32_t g0 = 0; g0 < R; g0 += GROUP) {
    const uint32_t Rg = R - g0 < (uint32_t)GROUP ? R - g0 : (uint32_t)GROUP;
    float dist;
    if (Rg <= 2) dist = group_distances<METRIC, 1>(emb, stride, d, rid, g0, Rg, qs, tile, q_norm, row_aux, prof3);
    else if (Rg <= 4) dist = group_distances<METRIC, 2>(emb, stride, d, rid, g0, Rg, qs, tile, q_norm, row_aux, prof3);
    else if (Rg <= 8) dist = group_distances<METRIC, 4>(emb, stride, d, rid, g0, Rg, qs, tile, q_norm, row_aux, prof3);
    else dist = group_distances<METRIC, 8>(emb, stride, d, rid, g0, Rg, qs, tile, q_norm, row_aux, prof3);
    // lane j of this group computed row g0 + j: hand the value to lane g0 + j
    float moved = __shfl(dist, (lane - (int)g0) & 63);
    if ((uint32_t)lane >= g0 && (uint32_t)lane < g0 + Rg) result = moved;
  }
  return result;
}

// Copies the query at `q` (global) into LDS and returns norm_a of cosine_distance
// (distance.rs:78) computed in reference order.
template <int METRIC>
__device__ __forceinline__ float load_query(const float* __restrict__ q, uint32_t d, float* qs) {
  for (uint32_t j = threadIdx.x; j < d; j += 64) qs[j] = q[j];
  __syncthreads();
  float na = 0.0f;
  if (METRIC == ISL_METRIC_COSINE || METRIC == METRIC_COSINE_PRE) {
    for (uint32_t j = 0; j < d; ++j) {
      float x = qs[j];
      na += x * x;
    }
  }
  return na;
}

}  // namespace isl_dev
