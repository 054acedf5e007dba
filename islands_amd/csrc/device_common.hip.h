// Device helpers shared by the kernels of libislands_amd.so: wave-level utilities and the
// exact-order distance routine (the four lanes of a quad own one row: each multiplies four of
// every 16 elements, and the reference's strictly sequential f32 sum travels round the quad --
// see "The quad chain" below; rows are staged through an LDS tile with coalesced 16-byte loads,
// or fetched straight into a register ring).
// Translation units including this file must be built with -ffp-contract=off.
#pragma once

#include "common.hpp"

namespace isl_dev {

constexpr int PIECE = 128;           // floats of one row staged per step (512 B; a wave-instruction
                                     // moves the pieces of two rows)
constexpr int TILE_LD = PIECE + 16;  // LDS row pitch in floats: lane (row r, quarter s) reads the float4
                                     // at 144 r + 16 i + 4 s -> bank 16 r + 4 s (mod 64): the 16 lanes
                                     // of four consecutive rows cover the 64 banks exactly once
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

// Ordering point for LDS traffic inside a ONE-WAVE workgroup: the LDS executes a wave's
// instructions in order, so a plain __syncthreads() (s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier)
// would only add a wait for every outstanding global store / load of the wave.  This keeps the
// compiler from moving memory operations across it and emits no wait for vector memory.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
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
// the same for a table of any size: the top of the 64-bit product (cap = 1 << bits gives hslot's value)
__device__ __forceinline__ uint32_t hslot_cap(uint32_t id, uint32_t cap) {
  return __umulhi(id * 0x9E3779B1u, cap);
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

// value of `v` in lane SEL of the caller's quad (DPP quad_perm:[SEL,SEL,SEL,SEL]; folds into the
// consuming v_add_f32 as its DPP source operand)
template <int SEL>
__device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), SEL * 0x55, 0xf, 0xf, true));
}

// value of `v` in the previous lane of the caller's quad (lane 0 takes lane 3's): DPP
// quad_perm:[3,0,1,2], folds into the consuming v_add_f32 as its DPP source operand
__device__ __forceinline__ float quad_prev(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x93, 0xf, 0xf, true));
}

// ------------------------------------------------------- exact-order distances
// The quad chain.  Lane s of a quad owns elements 16i + 4s .. 16i + 4s + 3 of its row (8 per lane
// and 32 per step for bf16 rows).  The partial sum travels round the quad: in turn t of a step lane
// t takes it from lane t - 1 (one v_add_f32 with a DPP source, 12.3 cycles when the chain depends
// on it) and adds its own elements with plain adds (5.25 cycles each) -- (12.3 + 3 x 5.25) / 4 = 7
// cycles per element where a chain that fetches every addend from the owning lane
// (v_add_f32_dpp quad_perm:[s,s,s,s], round 1) paid 14.5.  All four lanes execute every turn; the
// three that do not hold the live sum compute values nobody reads.  After a whole step the sum sits
// in lane 3, which is where the caller broadcasts it from (quad_bcast<3>).  The order of the
// additions is the reference's: element 0, 1, 2, ... of the row.
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

// The addend one element contributes to the main chain (dstep's `a0 += ...` operand).
template <int METRIC>
__device__ __forceinline__ float dterm(float q, float x) {
  if (METRIC == ISL_METRIC_COSINE || METRIC == ISL_METRIC_DOT || METRIC == METRIC_COSINE_PRE) {
    return q * x;
  } else if (METRIC == ISL_METRIC_EUCLIDEAN || METRIC == METRIC_EUCLID_SQ) {
    float diff = q - x;
    return diff * diff;
  } else if (METRIC == METRIC_SUMSQ || METRIC == METRIC_SUMSQ_RAW) {
    return x * x;
  }
  return fabsf(q - x);
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
// quads 0..Rg-1 run the sequential chains of the piece that already sits in the LDS tile
// (GROUP x TILE_LD floats, read back with conflict-free ds_read_b128).  All loads and LDS stores of a variant are unconditional straight-line
// code -- a predicate per instruction makes hipcc fall back to s_waitcnt vmcnt(0) before every
// store, which serialises the ring; lanes whose row does not exist re-read the group's first
// row (same cache lines) into a tile row that no lane consumes.
template <int METRIC, int NI>
__device__ __forceinline__ float group_distances(const float* __restrict__ emb, uint64_t stride,
                                                 uint32_t d, uint32_t rid, uint32_t g0, uint32_t Rg,
                                                 const float* qs, float* tile, float q_norm,
                                                 float row_aux) {
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
  // The chain of row r runs in all four lanes of quad r (same operands, same order -> same
  // bits); lane s of the quad owns elements 16i + 4s .. 16i + 4s + 3 of every 16: one
  // ds_read_b128 of x and of q and four multiplies per 16 elements instead of per 4, the 16
  // dependent adds take their addend straight from the owning lane (v_add_f32_dpp quad_perm).
  // The dependent add (about 14.5 cycles on gfx950) is the critical path either way; what this
  // layout buys is issue slots: 1.25 VALU and 0.125 LDS instructions per element, not 2 and 0.5.
#define ISL_TURN(ss)                                                                          \
  a0 = quad_prev(a0) + p0; a0 += p1; a0 += p2; a0 += p3;                                       \
  if constexpr (METRIC == ISL_METRIC_COSINE) { a1 = quad_prev(a1) + n0; a1 += n1; a1 += n2; a1 += n3; }
#define ISL_ADD_G1(ss, c) if (j + 4 * (ss) + (c) < cnt) { a0 += p##c; if constexpr (METRIC == ISL_METRIC_COSINE) a1 += n##c; }
#define ISL_TURN_G(ss)                                                                        \
  a0 = quad_prev(a0);                                                                          \
  if constexpr (METRIC == ISL_METRIC_COSINE) a1 = quad_prev(a1);                               \
  ISL_ADD_G1(ss, 0) ISL_ADD_G1(ss, 1) ISL_ADD_G1(ss, 2) ISL_ADD_G1(ss, 3)
#define ISL_TERMS                                                                              \
  const float p0 = dterm<METRIC>(q.x, x.x), p1 = dterm<METRIC>(q.y, x.y),                      \
              p2 = dterm<METRIC>(q.z, x.z), p3 = dterm<METRIC>(q.w, x.w);                      \
  const float n0 = x.x * x.x, n1 = x.y * x.y, n2 = x.z * x.z, n3 = x.w * x.w;                  \
  (void)n0; (void)n1; (void)n2; (void)n3;
  auto consume = [&](uint32_t t) {
    if ((uint32_t)(lane >> 2) < Rg) {
      const uint32_t s4 = (uint32_t)(lane & 3) * 4u;
      const float* trow = tile + (lane >> 2) * TILE_LD + s4;
      const float* qv = qs + t * PIECE + s4;
      const uint32_t cnt = d - t * PIECE;
      if (cnt >= (uint32_t)PIECE) {
#pragma unroll 4
        for (int j = 0; j < PIECE; j += 16) {
          const float4 x = *reinterpret_cast<const float4*>(trow + j);
          const float4 q = *reinterpret_cast<const float4*>(qv + j);
          ISL_TERMS
          ISL_TURN(0) ISL_TURN(1) ISL_TURN(2) ISL_TURN(3)
        }
      } else {
        for (uint32_t j = 0; j < cnt; j += 16) {
          float4 x = make_float4(0.f, 0.f, 0.f, 0.f), q = x;
          if (j + s4 < cnt) {  // `qs` holds d rounded up to 4 floats
            x = *reinterpret_cast<const float4*>(trow + j);
            q = *reinterpret_cast<const float4*>(qv + j);
          }
          ISL_TERMS
          // elements at or past `cnt` (stale tile / query words) are multiplied but never added
          ISL_TURN_G(0) ISL_TURN_G(1) ISL_TURN_G(2) ISL_TURN_G(3)
        }
      }
    }
  };
#undef ISL_TURN
#undef ISL_ADD_G1
#undef ISL_TURN_G
#undef ISL_TERMS
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
    if (t + 3 < nT) {
      const size_t poff = (size_t)(t + 3) * PIECE;
      ISL_FOR8(ISL_LOAD_A)
    }
    consume(t);
    __syncthreads();
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
  a0 = quad_bcast<3>(a0);  // the sums end their round in lane 3
  a1 = quad_bcast<3>(a1);
  if (METRIC == METRIC_COSINE_PRE) a1 = __shfl(row_aux, (int)((g0 + (lane >> 2)) & 63));
  return dfinish<METRIC>(a0, a1, q_norm);
}

// Distances of R (<= 64) rows to the query held in LDS (`qs`); lane r < R owns row `rid` and
// returns its distance.  Rows are handled in groups of GROUP = 16 (see group_distances).
template <int METRIC>
__device__ __forceinline__ float wave_distances(const float* __restrict__ emb, uint64_t stride,
                                                uint32_t d, uint32_t rid, uint32_t R,
                                                const float* qs, float* tile, float q_norm,
                                                float row_aux = 0.0f) {
  const int lane = threadIdx.x;
  float result = 0.0f;
  for (uint32_t g0 = 0; g0 < R; g0 += GROUP) {
    const uint32_t Rg = R - g0 < (uint32_t)GROUP ? R - g0 : (uint32_t)GROUP;
    float dist;
    if (Rg <= 2) dist = group_distances<METRIC, 1>(emb, stride, d, rid, g0, Rg, qs, tile, q_norm, row_aux);
    else if (Rg <= 4) dist = group_distances<METRIC, 2>(emb, stride, d, rid, g0, Rg, qs, tile, q_norm, row_aux);
    else if (Rg <= 8) dist = group_distances<METRIC, 4>(emb, stride, d, rid, g0, Rg, qs, tile, q_norm, row_aux);
    else dist = group_distances<METRIC, 8>(emb, stride, d, rid, g0, Rg, qs, tile, q_norm, row_aux);
    // quad j of this group computed row g0 + j: hand the value to lane g0 + j
    float moved = __shfl(dist, (4 * (lane - (int)g0)) & 63);
    if ((uint32_t)lane >= g0 && (uint32_t)lane < g0 + Rg) result = moved;
  }
  return result;
}

// ---------------------------------------------------------- tile-free variant
// Same quad layout as group_distances, but every lane fetches its own 16 bytes of every 64
// straight from global memory into a register ring (RING steps of 16 elements in flight per
// lane): a wave-load touches one 64-byte segment of up to 16 rows, consecutive steps walk each
// row sequentially (the second half of every 128-byte line is an L2/TCP hit).  No LDS tile, no
// ds_write, no barrier: the dependent add chain runs uninterrupted and the kernel's LDS
// footprint shrinks to the visited table plus the query.  Used by the search kernel, where a
// hop evaluates ~9 rows; the streaming kernels keep the tile (fully coalesced 512-byte pieces).
constexpr int RING = 12;
typedef float v4f __attribute__((ext_vector_type(4)));

#define ISL_RING(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11)
template <int METRIC>
__device__ __forceinline__ float direct_group(const float* __restrict__ emb, uint64_t stride,
                                              uint32_t d, uint32_t rid, uint32_t g0, uint32_t Rg,
                                              const float* qs, float q_norm, float row_aux) {
  const int lane = threadIdx.x;
  const uint32_t r = (uint32_t)lane >> 2;
  const uint32_t s4 = ((uint32_t)lane & 3u) * 4u;
  const uint32_t row = (uint32_t)__shfl((int)rid, (int)((g0 + (r < Rg ? r : 0u)) & 63u));
  float a0 = 0.0f, a1 = 0.0f;
  if (r < Rg) {
    const float* rp = emb + (uint64_t)row * stride + s4;
    const float* qp = qs + s4;
    const uint32_t nF = d >> 4;          // steps whose 16 elements all exist
    const uint32_t nS = (d + 15u) >> 4;  // steps in total
#define ISL_TURN(ss)                                                                          \
  a0 = quad_prev(a0) + p0; a0 += p1; a0 += p2; a0 += p3;                                       \
  if constexpr (METRIC == ISL_METRIC_COSINE) { a1 = quad_prev(a1) + n0; a1 += n1; a1 += n2; a1 += n3; }
#define ISL_ADD_G1(ss, c) if (e0 + 4 * (ss) + (c) < d) { a0 += p##c; if constexpr (METRIC == ISL_METRIC_COSINE) a1 += n##c; }
#define ISL_TURN_G(ss)                                                                        \
  a0 = quad_prev(a0);                                                                          \
  if constexpr (METRIC == ISL_METRIC_COSINE) a1 = quad_prev(a1);                               \
  ISL_ADD_G1(ss, 0) ISL_ADD_G1(ss, 1) ISL_ADD_G1(ss, 2) ISL_ADD_G1(ss, 3)
#define ISL_ALL_TURNS ISL_TURN(0) ISL_TURN(1) ISL_TURN(2) ISL_TURN(3)
#define ISL_TERMS                                                                              \
  const float p0 = dterm<METRIC>(q.x, x.x), p1 = dterm<METRIC>(q.y, x.y),                      \
              p2 = dterm<METRIC>(q.z, x.z), p3 = dterm<METRIC>(q.w, x.w);                      \
  const float n0 = x.x * x.x, n1 = x.y * x.y, n2 = x.z * x.z, n3 = x.w * x.w;                  \
  (void)n0; (void)n1; (void)n2; (void)n3;
#define ISL_DECLX(k) v4f x##k;
#define ISL_ISSUE(k) x##k = *reinterpret_cast<const v4f*>(rp + 16u * (base + (k)));
  // The register ring only works if the reload of slot k is issued inside step k (behind the
  // multiplies that free the slot, ahead of the 16 dependent adds that hide its latency): the
  // scheduler would otherwise sink every reload to its first use.  sched_barrier pins VMEM and
  // lets ALU / DS instructions move; the empty asm makes the whole 128-bit register the load's
  // only user, so the load is not split into four dword loads.
#define ISL_PIN __builtin_amdgcn_sched_barrier(0x40F);
#define ISL_TAKE(k)                                                               \
    v4f xv = x##k;                                                                \
    asm volatile("" : "+v"(xv));                                                  \
    const float4 x = make_float4(xv.x, xv.y, xv.z, xv.w);
  // the query operand of step k + 1 is read from LDS during step k's adds as well
#define ISL_QNEXT(k)                                                              \
    {                                                                             \
      const uint32_t sn_ = base + (k) + 1u < nS ? base + (k) + 1u : nS - 1u;      \
      qn = *reinterpret_cast<const float4*>(qp + 16u * sn_);                      \
    }
#define ISL_STEP_RELOAD(k)                                                        \
  {                                                                               \
    ISL_TAKE(k)                                                                   \
    const float4 q = qn;                                                          \
    ISL_TERMS                                                                     \
    ISL_PIN                                                                       \
    x##k = *reinterpret_cast<const v4f*>(rp + 16u * (base + RING + (k)));         \
    ISL_QNEXT(k)                                                                  \
    ISL_PIN                                                                       \
    ISL_ALL_TURNS                                                                 \
  }
#define ISL_STEP(k)                                                               \
  {                                                                               \
    ISL_TAKE(k)                                                                   \
    const float4 q = qn;                                                          \
    ISL_TERMS                                                                     \
    ISL_PIN                                                                       \
    ISL_QNEXT(k)                                                                  \
    ISL_PIN                                                                       \
    ISL_ALL_TURNS                                                                 \
  }
    ISL_RING(ISL_DECLX)
    uint32_t base = 0;
    if (nF >= (uint32_t)RING) {
      ISL_RING(ISL_ISSUE)
      float4 qn = *reinterpret_cast<const float4*>(qp);
      ISL_PIN
      while (base + 2u * RING <= nF) {
        ISL_RING(ISL_STEP_RELOAD)
        base += RING;
      }
      ISL_RING(ISL_STEP)
      base += RING;
    }
    if (base < nS) {
      // fewer than RING + 1 steps left (all of them when d < 16 * RING): loads first, clamped to
      // the last step so that they stay unconditional, then the guarded chains
      const uint32_t rem = nS - base;
      const uint32_t last = nS - 1u;
#define ISL_ISSUE_C(k)                                                              \
  {                                                                                 \
    const uint32_t st_ = base + (k) < last ? base + (k) : last;                     \
    x##k = *reinterpret_cast<const v4f*>(rp + 16u * st_);                           \
  }
#define ISL_STEP_G(k)                                                               \
  if ((uint32_t)(k) < rem) {                                                        \
    const uint32_t e0 = 16u * (base + (k));                                         \
    ISL_TAKE(k)                                                                     \
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);                                     \
    if (e0 + s4 < ((d + 3u) & ~3u)) q = *reinterpret_cast<const float4*>(qp + e0); \
    ISL_TERMS                                                                       \
    ISL_TURN_G(0) ISL_TURN_G(1) ISL_TURN_G(2) ISL_TURN_G(3)                         \
  }
      ISL_RING(ISL_ISSUE_C)
      ISL_RING(ISL_STEP_G)
#undef ISL_ISSUE_C
#undef ISL_STEP_G
    }
#undef ISL_TURN
#undef ISL_ADD_G1
#undef ISL_TURN_G
#undef ISL_ALL_TURNS
#undef ISL_TERMS
#undef ISL_DECLX
#undef ISL_ISSUE
#undef ISL_PIN
#undef ISL_TAKE
#undef ISL_QNEXT
#undef ISL_STEP_RELOAD
#undef ISL_STEP
  }
  a0 = quad_bcast<3>(a0);  // the sums end their round in lane 3
  a1 = quad_bcast<3>(a1);
  if (METRIC == METRIC_COSINE_PRE) a1 = __shfl(row_aux, (int)((g0 + r) & 63u));
  return dfinish<METRIC>(a0, a1, q_norm);
}
#undef ISL_RING

// bf16 rows (ISL_DTYPE_BF16): the same quad layout with 8 elements per 16-byte load, so a step
// covers 32 elements (lane s of the quad owns elements 32i + 8s .. + 7).  Every bf16 value is
// exactly representable in f32 and widened before use: the arithmetic is the f32 chain of the
// reference run on the widened rows.
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
// The query operand of one step (8 elements per lane) as it waits in registers.  QH = the query
// sits in LDS as bf16 bit patterns (every element of it is a bf16 value: checked where it is
// loaded), half the LDS per wave -- at d = 4096 that is what bounds the waves per CU; the values
// are widened exactly like the row's.
template <bool QH> struct QStep;
template <> struct QStep<false> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* qs, uint32_t e) {
    a = *reinterpret_cast<const float4*>(qs + e);
    b = *reinterpret_cast<const float4*>(qs + e + 4u);
  }
  __device__ __forceinline__ void load_guarded(const float* qs, uint32_t e, uint32_t d) {
    const uint32_t dq = (d + 3u) & ~3u;
    a = make_float4(0.f, 0.f, 0.f, 0.f);
    b = a;
    if (e < dq) a = *reinterpret_cast<const float4*>(qs + e);
    if (e + 4u < dq) b = *reinterpret_cast<const float4*>(qs + e + 4u);
  }
  __device__ __forceinline__ float4 lo() const { return a; }
  __device__ __forceinline__ float4 hi() const { return b; }
};
template <> struct QStep<true> {
  v4u w;
  __device__ __forceinline__ void load(const float* qs, uint32_t e) {
    w = *reinterpret_cast<const v4u*>(reinterpret_cast<const uint16_t*>(qs) + e);
  }
  __device__ __forceinline__ void load_guarded(const float* qs, uint32_t e, uint32_t d) {
    w = v4u{0u, 0u, 0u, 0u};
    if (e < ((d + 7u) & ~7u)) w = *reinterpret_cast<const v4u*>(reinterpret_cast<const uint16_t*>(qs) + e);
  }
  __device__ __forceinline__ float4 lo() const {
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u),
                       __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xFFFF0000u));
  }
  __device__ __forceinline__ float4 hi() const {
    return make_float4(__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xFFFF0000u),
                       __uint_as_float(w.w << 16), __uint_as_float(w.w & 0xFFFF0000u));
  }
};
#define ISL_RING(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11)
template <int METRIC, bool QH = false>
__device__ __forceinline__ float direct_group_bf16(const uint16_t* __restrict__ emb, uint64_t stride,
                                                   uint32_t d, uint32_t rid, uint32_t g0, uint32_t Rg,
                                                   const float* qs, float q_norm, float row_aux) {
  const int lane = threadIdx.x;
  const uint32_t r = (uint32_t)lane >> 2;
  const uint32_t s8 = ((uint32_t)lane & 3u) * 8u;
  const uint32_t row = (uint32_t)__shfl((int)rid, (int)((g0 + (r < Rg ? r : 0u)) & 63u));
  float a0 = 0.0f, a1 = 0.0f;
  if (r < Rg) {
    const uint16_t* rp = emb + (uint64_t)row * stride + s8;
    const uint32_t nF = d >> 5;
    const uint32_t nS = (d + 31u) >> 5;
#define ISL_TURN(ss)                                                                          \
  a0 = quad_prev(a0) + p0; a0 += p1; a0 += p2; a0 += p3; a0 += p4; a0 += p5; a0 += p6; a0 += p7; \
  if constexpr (METRIC == ISL_METRIC_COSINE) {                                                 \
    a1 = quad_prev(a1) + n0; a1 += n1; a1 += n2; a1 += n3; a1 += n4; a1 += n5; a1 += n6; a1 += n7; \
  }
#define ISL_ADD_G1(ss, c) if (e0 + 8 * (ss) + (c) < d) { a0 += p##c; if constexpr (METRIC == ISL_METRIC_COSINE) a1 += n##c; }
#define ISL_TURN_G(ss)                                                                        \
  a0 = quad_prev(a0);                                                                          \
  if constexpr (METRIC == ISL_METRIC_COSINE) a1 = quad_prev(a1);                               \
  ISL_ADD_G1(ss, 0) ISL_ADD_G1(ss, 1) ISL_ADD_G1(ss, 2) ISL_ADD_G1(ss, 3)                      \
  ISL_ADD_G1(ss, 4) ISL_ADD_G1(ss, 5) ISL_ADD_G1(ss, 6) ISL_ADD_G1(ss, 7)
#define ISL_WIDEN                                                                               \
  const float w0 = __uint_as_float(xv.x << 16), w1 = __uint_as_float(xv.x & 0xFFFF0000u),      \
              w2 = __uint_as_float(xv.y << 16), w3 = __uint_as_float(xv.y & 0xFFFF0000u),      \
              w4 = __uint_as_float(xv.z << 16), w5 = __uint_as_float(xv.z & 0xFFFF0000u),      \
              w6 = __uint_as_float(xv.w << 16), w7 = __uint_as_float(xv.w & 0xFFFF0000u);
#define ISL_TERMS                                                                               \
  const float p0 = dterm<METRIC>(qa.x, w0), p1 = dterm<METRIC>(qa.y, w1),                      \
              p2 = dterm<METRIC>(qa.z, w2), p3 = dterm<METRIC>(qa.w, w3),                      \
              p4 = dterm<METRIC>(qb.x, w4), p5 = dterm<METRIC>(qb.y, w5),                      \
              p6 = dterm<METRIC>(qb.z, w6), p7 = dterm<METRIC>(qb.w, w7);                      \
  const float n0 = w0 * w0, n1 = w1 * w1, n2 = w2 * w2, n3 = w3 * w3, n4 = w4 * w4,            \
              n5 = w5 * w5, n6 = w6 * w6, n7 = w7 * w7;                                        \
  (void)n0; (void)n1; (void)n2; (void)n3; (void)n4; (void)n5; (void)n6; (void)n7;
#define ISL_DECLX(k) v4u x##k;
#define ISL_ISSUE(k) x##k = *reinterpret_cast<const v4u*>(rp + 32u * (base + (k)));
#define ISL_PIN __builtin_amdgcn_sched_barrier(0x40F);
#define ISL_TAKE(k)                                                               \
    v4u xv = x##k;                                                                \
    asm volatile("" : "+v"(xv));
#define ISL_QNEXT(k)                                                              \
    {                                                                             \
      const uint32_t sn_ = base + (k) + 1u < nS ? base + (k) + 1u : nS - 1u;      \
      qn.load(qs, s8 + 32u * sn_);                                                \
    }
#define ISL_ALL_ADDS ISL_TURN(0) ISL_TURN(1) ISL_TURN(2) ISL_TURN(3)
#define ISL_STEP_RELOAD(k)                                                        \
  {                                                                               \
    ISL_TAKE(k)                                                                   \
    const float4 qa = qn.lo(), qb = qn.hi();                                      \
    ISL_WIDEN                                                                     \
    ISL_TERMS                                                                     \
    ISL_PIN                                                                       \
    x##k = *reinterpret_cast<const v4u*>(rp + 32u * (base + RING + (k)));         \
    ISL_QNEXT(k)                                                                  \
    ISL_PIN                                                                       \
    ISL_ALL_ADDS                                                                  \
  }
#define ISL_STEP(k)                                                               \
  {                                                                               \
    ISL_TAKE(k)                                                                   \
    const float4 qa = qn.lo(), qb = qn.hi();                                      \
    ISL_WIDEN                                                                     \
    ISL_TERMS                                                                     \
    ISL_PIN                                                                       \
    ISL_QNEXT(k)                                                                  \
    ISL_PIN                                                                       \
    ISL_ALL_ADDS                                                                  \
  }
    ISL_RING(ISL_DECLX)
    uint32_t base = 0;
    if (nF >= (uint32_t)RING) {
      ISL_RING(ISL_ISSUE)
      QStep<QH> qn;
      qn.load(qs, s8);
      ISL_PIN
      while (base + 2u * RING <= nF) {
        ISL_RING(ISL_STEP_RELOAD)
        base += RING;
      }
      ISL_RING(ISL_STEP)
      base += RING;
    }
    if (base < nS) {
      const uint32_t rem = nS - base;
      const uint32_t last = nS - 1u;
#define ISL_ISSUE_C(k)                                                              \
  {                                                                                 \
    const uint32_t st_ = base + (k) < last ? base + (k) : last;                     \
    x##k = *reinterpret_cast<const v4u*>(rp + 32u * st_);                           \
  }
#define ISL_STEP_G(k)                                                               \
  if ((uint32_t)(k) < rem) {                                                        \
    const uint32_t e0 = 32u * (base + (k));                                         \
    ISL_TAKE(k)                                                                     \
    QStep<QH> qg_;                                                                  \
    qg_.load_guarded(qs, e0 + s8, d);                                               \
    const float4 qa = qg_.lo(), qb = qg_.hi();                                      \
    ISL_WIDEN                                                                       \
    ISL_TERMS                                                                       \
    ISL_TURN_G(0) ISL_TURN_G(1) ISL_TURN_G(2) ISL_TURN_G(3)                         \
  }
      ISL_RING(ISL_ISSUE_C)
      ISL_RING(ISL_STEP_G)
#undef ISL_ISSUE_C
#undef ISL_STEP_G
    }
#undef ISL_TURN
#undef ISL_ADD_G1
#undef ISL_TURN_G
#undef ISL_WIDEN
#undef ISL_TERMS
#undef ISL_DECLX
#undef ISL_ISSUE
#undef ISL_PIN
#undef ISL_TAKE
#undef ISL_QNEXT
#undef ISL_ALL_ADDS
#undef ISL_STEP_RELOAD
#undef ISL_STEP
  }
  a0 = quad_bcast<3>(a0);  // the sums end their round in lane 3
  a1 = quad_bcast<3>(a1);
  if (METRIC == METRIC_COSINE_PRE) a1 = __shfl(row_aux, (int)((g0 + r) & 63u));
  return dfinish<METRIC>(a0, a1, q_norm);
}
#undef ISL_RING

// Tile-free counterpart of wave_distances: lane j < R receives the distance of row rid(j).
// ROWT = float or uint16_t (bf16 bits).
template <int METRIC, typename ROWT = float, bool QH = false>
__device__ __forceinline__ float direct_distances(const ROWT* __restrict__ emb, uint64_t stride,
                                                  uint32_t d, uint32_t rid, uint32_t R,
                                                  const float* qs, float q_norm, float row_aux = 0.0f) {
  const int lane = threadIdx.x;
  float result = 0.0f;
  for (uint32_t g0 = 0; g0 < R; g0 += GROUP) {
    const uint32_t Rg = R - g0 < (uint32_t)GROUP ? R - g0 : (uint32_t)GROUP;
    float dist;
    if constexpr (sizeof(ROWT) == 2)
      dist = direct_group_bf16<METRIC, QH>(reinterpret_cast<const uint16_t*>(emb), stride, d, rid, g0, Rg, qs, q_norm, row_aux);
    else
      dist = direct_group<METRIC>(reinterpret_cast<const float*>(emb), stride, d, rid, g0, Rg, qs, q_norm, row_aux);
    float moved = __shfl(dist, (4 * (lane - (int)g0)) & 63);
    if ((uint32_t)lane >= g0 && (uint32_t)lane < g0 + Rg) result = moved;
  }
  return result;
}

// Slow generic form for bf16 rows in the heap-exact kernel: lane j < R walks its own row.
template <int METRIC>
__device__ __forceinline__ float lane_distances_bf16(const uint16_t* __restrict__ emb, uint64_t stride,
                                                     uint32_t d, uint32_t rid, uint32_t R,
                                                     const float* qs, float q_norm, float row_aux) {
  float a0 = 0.0f, a1 = 0.0f;
  if ((uint32_t)threadIdx.x < R) {
    const uint16_t* rp = emb + (uint64_t)rid * stride;
    for (uint32_t j = 0; j < d; ++j) dstep<METRIC>(qs[j], __uint_as_float((uint32_t)rp[j] << 16), a0, a1);
  }
  if (METRIC == METRIC_COSINE_PRE) a1 = row_aux;
  return dfinish<METRIC>(a0, a1, q_norm);
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

// The same with the query kept in LDS as bf16 bit patterns (QStep<true>): *representable = false
// when an element is not a bf16 value (the caller then answers the query with the f32-query kernel).
template <int METRIC>
__device__ __forceinline__ float load_query_bf16(const float* __restrict__ q, uint32_t d, float* qs,
                                                 bool* representable) {
  uint16_t* qh = reinterpret_cast<uint16_t*>(qs);
  bool bad = false;
  const uint32_t d8 = (d + 7u) & ~7u;
  for (uint32_t j = threadIdx.x; j < d8; j += 64) {
    const uint32_t u = j < d ? __float_as_uint(q[j]) : 0u;
    bad |= (u & 0xFFFFu) != 0u;
    qh[j] = (uint16_t)(u >> 16);
  }
  __syncthreads();
  *representable = !ballot(bad);
  float na = 0.0f;
  if (METRIC == ISL_METRIC_COSINE || METRIC == METRIC_COSINE_PRE) {
    for (uint32_t j = 0; j < d; ++j) {
      float x = __uint_as_float((uint32_t)qh[j] << 16);
      na += x * x;
    }
  }
  return na;
}

}  // namespace isl_dev
