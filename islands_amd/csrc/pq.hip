// Product-quantisation distance ops of src/core/pq.rs on gfx950:
//   build_distance_tables (pq.rs:307-338), table_distance (:341-348),
//   asymmetric_distance (:275-304), encode / find_nearest (:221-244, :86-106).
// Every sum keeps the reference's left-to-right f32 order (device_common.hip.h).
#include "device_common.hip.h"

#include <algorithm>
#include <cfloat>

namespace {

using namespace isl_dev;

// Block (one wave) = 64 centroids of one subquantizer for one query: table[j][c] =
// sum_i (q_sub[i] - cent[c][i])^2, sequential over i (pq.rs:322-332; powi(2) == x*x).
__global__ __launch_bounds__(64) void pq_tables_kernel(const float* __restrict__ queries,
                                                       uint32_t d, const float* __restrict__ cb,
                                                       uint32_t m, uint32_t K, uint32_t dsub,
                                                       uint32_t cstride, float* __restrict__ tables) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  float* qs = tile + TILE_ROWS * TILE_LD;
  const int lane = threadIdx.x;
  const uint32_t cblocks = (K + 63) / 64;
  const uint32_t q = blockIdx.y;
  const uint32_t j = blockIdx.x / cblocks, c0 = (blockIdx.x % cblocks) * 64;
  (void)load_query<METRIC_EUCLID_SQ>(queries + (uint64_t)q * d + (uint64_t)j * dsub, dsub, qs);
  uint32_t R = K - c0 < 64 ? K - c0 : 64;
  const float* rows = cb + ((uint64_t)j * K + c0) * cstride;
  float v = wave_distances<METRIC_EUCLID_SQ>(rows, cstride, dsub, (uint32_t)lane, R, qs, tile, 0.f);
  if ((uint32_t)lane < R) tables[((uint64_t)q * m + j) * K + c0 + lane] = v;
}

// Short subvectors (dsub <= 64, the usual case: 8 at d = 768 / m = 96, 64 at d = 4096 / m = 64): one
// workgroup per (query, subquantizer), one thread per centroid walking its dsub elements in order
// -- the same left fold, without the wave-tile machinery (a tenth of its time at dsub = 8).
__global__ __launch_bounds__(256) void pq_tables_direct_kernel(const float* __restrict__ queries, uint32_t d,
                                                              const float* __restrict__ cb, uint32_t m,
                                                              uint32_t K, uint32_t dsub, uint32_t cstride,
                                                              float* __restrict__ tables) {
  __shared__ float qs[64];
  const uint32_t j = blockIdx.x, q = blockIdx.y;
  if (threadIdx.x < dsub) qs[threadIdx.x] = queries[(uint64_t)q * d + (uint64_t)j * dsub + threadIdx.x];
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < K; c += 256) {
    const float* row = cb + ((uint64_t)j * K + c) * cstride;
    float s = 0.0f;
    for (uint32_t i = 0; i < dsub; ++i) {
      const float df = qs[i] - row[i];
      s += df * df;  // (a - b).powi(2), summed left to right (pq.rs:326-331)
    }
    tables[((uint64_t)q * m + j) * K + c] = s;
  }
}

// table_distance, pq.rs:341-348: sqrt(sum_j tables[j][code_j]), left fold over j.
__global__ void pq_table_distance_kernel(const float* __restrict__ tables,
                                         const uint16_t* __restrict__ codes, uint64_t n, uint32_t m,
                                         uint32_t K, float* __restrict__ out,
                                         uint32_t* __restrict__ flags) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint16_t* c = codes + i * m;
  float s = 0.0f;
  bool bad = false;
  for (uint32_t j = 0; j < m; ++j) {
    uint32_t code = c[j];
    if (code >= K) { bad = true; code = 0; }
    s += tables[(uint64_t)j * K + code];
  }
  if (bad) atomicOr(flags, 1u);
  out[i] = sqrtf(s);
}

// encode, pq.rs:221-244: per (vector, subquantizer) arg-min over centroids of
// metric.calculate(sub, centroid); strict `<` from f32::MAX, first index wins (pq.rs:94-103).
template <int METRIC>
__global__ __launch_bounds__(64) void pq_encode_kernel(const float* __restrict__ vectors, uint32_t d,
                                                       const float* __restrict__ cb, uint32_t m,
                                                       uint32_t K, uint32_t dsub, uint32_t cstride,
                                                       uint16_t* __restrict__ codes) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  float* qs = tile + TILE_ROWS * TILE_LD;
  const int lane = threadIdx.x;
  const uint32_t j = blockIdx.x;
  const uint64_t v = blockIdx.y;
  const float q_norm = load_query<METRIC>(vectors + v * d + (uint64_t)j * dsub, dsub, qs);
  float best = FLT_MAX;
  uint32_t best_idx = 0;
  bool have = false;
  for (uint32_t c0 = 0; c0 < K; c0 += 64) {
    uint32_t R = K - c0 < 64 ? K - c0 : 64;
    const float* rows = cb + ((uint64_t)j * K + c0) * cstride;
    float dist = wave_distances<METRIC>(rows, cstride, dsub, (uint32_t)lane, R, qs, tile, q_norm);
    bool ok = (uint32_t)lane < R && dist < best;  // can this lane beat the running best?
    // smallest distance among the candidates of this block, lowest index on ties
    float bd = ok ? dist : 0.0f;
    uint32_t bi = c0 + lane;
    bool bv = ok;
    for (int off = 32; off > 0; off >>= 1) {
      float od = __shfl_xor(bd, off);
      uint32_t oi = __shfl_xor(bi, off);
      bool ov = __shfl_xor((int)bv, off) != 0;
      if (ov && (!bv || od < bd || (od == bd && oi < bi))) { bd = od; bi = oi; bv = true; }
    }
    if (bv) { best = bd; best_idx = bi; have = true; }
    __syncthreads();
  }
  (void)have;
  if (lane == 0) codes[v * m + j] = (uint16_t)best_idx;
}

size_t pq_lds(uint32_t dsub) {
  return (size_t)TILE_ROWS * TILE_LD * 4 + (size_t)((dsub + 3) / 4 * 4) * 4 + 64;
}

struct Staged {
  std::vector<void*> owned;
  ~Staged() { for (void* p : owned) (void)hipFree(p); }
  void* alloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return nullptr;
    owned.push_back(p);
    return p;
  }
};

isl_status launch_tables(const isl_pq* pq, const float* d_queries, uint64_t nq, float* d_tables,
                         hipStream_t st) {
  size_t lds = pq_lds((uint32_t)pq->dsub);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pq_tables_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  uint32_t cblocks = (uint32_t)((pq->K + 63) / 64);
  for (uint64_t q0 = 0; q0 < nq; q0 += 65535) {
    uint32_t nb = (uint32_t)std::min<uint64_t>(65535, nq - q0);
    if (pq->dsub <= 64) {
      hipLaunchKernelGGL(pq_tables_direct_kernel, dim3((uint32_t)pq->m, nb), dim3(256), 0, st,
                         d_queries + q0 * pq->dimension, (uint32_t)pq->dimension, pq->d_codebooks, (uint32_t)pq->m,
                         (uint32_t)pq->K, (uint32_t)pq->dsub, (uint32_t)pq->cstride, d_tables + q0 * pq->m * pq->K);
      continue;
    }
    hipLaunchKernelGGL(pq_tables_kernel, dim3((uint32_t)pq->m * cblocks, nb), dim3(64), lds, st,
                       d_queries + q0 * pq->dimension, (uint32_t)pq->dimension, pq->d_codebooks,
                       (uint32_t)pq->m, (uint32_t)pq->K, (uint32_t)pq->dsub, (uint32_t)pq->cstride,
                       d_tables + q0 * pq->m * pq->K);
  }
  ISL_HIP(hipGetLastError());
  return ISL_OK;
}

}  // namespace

namespace isl {
isl_status pq_launch_tables(const isl_pq* pq, const float* d_queries, uint64_t nq, float* d_tables,
                            hipStream_t st) {
  return launch_tables(pq, d_queries, nq, d_tables, st);
}
}  // namespace isl

extern "C" {

isl_status isl_pq_new(uint64_t dimension, uint64_t m, uint64_t K, const float* codebooks,
                      int32_t metric, int32_t device, isl_pq** out) {
  if (!out || !codebooks) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  // PQConfig::validate, pq.rs:37-55
  if (m == 0)
    return isl::fail(ISL_ERR_INVALID_CONFIG, "Invalid configuration: num_subquantizers must be > 0");
  if (dimension % m != 0)
    return isl::fail(ISL_ERR_INVALID_CONFIG,
                     "Invalid configuration: dimension %llu must be divisible by num_subquantizers %llu",
                     (unsigned long long)dimension, (unsigned long long)m);
  if (K == 0 || K > 65536)
    return isl::fail(ISL_ERR_INVALID_CONFIG,
                     "Invalid configuration: num_centroids must be in range [1, 65536]");
  if (metric < 0 || metric > ISL_METRIC_MANHATTAN)
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "unknown metric");
  ISL_TRY(isl::use_device(device));
  isl_pq* pq = new isl_pq();
  pq->dimension = dimension;
  pq->m = m;
  pq->K = K;
  pq->dsub = dimension / m;
  pq->cstride = (pq->dsub + 3) / 4 * 4;
  pq->metric = metric;
  pq->device = device;
  size_t bytes = (size_t)(m * K * pq->cstride + 256) * 4;
  if (hipMalloc(&pq->d_codebooks, bytes) != hipSuccess ||
      hipMemset(pq->d_codebooks, 0, bytes) != hipSuccess ||
      hipMemcpy2D(pq->d_codebooks, pq->cstride * 4, codebooks, pq->dsub * 4, pq->dsub * 4, m * K,
                  hipMemcpyHostToDevice) != hipSuccess) {
    isl_pq_free(pq);
    return isl::fail(ISL_ERR_DEVICE, "codebook upload failed");
  }
  *out = pq;
  return ISL_OK;
}

void isl_pq_free(isl_pq* pq) {
  if (!pq) return;
  if (pq->d_codebooks) (void)hipFree(pq->d_codebooks);
  delete pq;
}

isl_status isl_pq_build_distance_tables(const isl_pq* pq, const float* queries, uint64_t nq,
                                        uint64_t d, float* tables, int32_t mem, void* stream) {
  if (!pq) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "pq is NULL");
  if (d != pq->dimension) return isl::fail_dim(pq->dimension, d);  // pq.rs:308-313
  if (nq == 0) return ISL_OK;
  if (!queries || !tables) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  ISL_TRY(isl::use_device(pq->device));
  hipStream_t st = (hipStream_t)stream;
  Staged sg;
  const float* dq = queries;
  float* dt = tables;
  size_t tbytes = (size_t)nq * pq->m * pq->K * 4;
  if (mem == ISL_MEM_HOST) {
    void* a = sg.alloc((size_t)nq * d * 4);
    void* b = sg.alloc(tbytes);
    if (!a || !b) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
    ISL_HIP(hipMemcpyAsync(a, queries, (size_t)nq * d * 4, hipMemcpyHostToDevice, st));
    dq = (const float*)a;
    dt = (float*)b;
  }
  ISL_TRY(launch_tables(pq, dq, nq, dt, st));
  if (mem == ISL_MEM_HOST) ISL_HIP(hipMemcpyAsync(tables, dt, tbytes, hipMemcpyDeviceToHost, st));
  ISL_HIP(hipStreamSynchronize(st));
  return ISL_OK;
}

static isl_status table_distance_impl(const isl_pq* pq, const float* d_tables,
                                      const uint16_t* codes, uint64_t n, float* out, int32_t mem,
                                      hipStream_t st) {
  Staged sg;
  const uint16_t* dc = codes;
  float* dout = out;
  if (mem == ISL_MEM_HOST) {
    void* a = sg.alloc((size_t)n * pq->m * 2);
    void* b = sg.alloc((size_t)n * 4);
    if (!a || !b) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
    ISL_HIP(hipMemcpyAsync(a, codes, (size_t)n * pq->m * 2, hipMemcpyHostToDevice, st));
    dc = (const uint16_t*)a;
    dout = (float*)b;
  }
  uint32_t* d_flags = (uint32_t*)sg.alloc(4);
  if (!d_flags) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
  ISL_HIP(hipMemsetAsync(d_flags, 0, 4, st));
  uint32_t blocks = (uint32_t)((n + 255) / 256);
  hipLaunchKernelGGL(pq_table_distance_kernel, dim3(blocks), dim3(256), 0, st, d_tables, dc, n,
                     (uint32_t)pq->m, (uint32_t)pq->K, dout, d_flags);
  ISL_HIP(hipGetLastError());
  uint32_t flags = 0;
  ISL_HIP(hipMemcpyAsync(&flags, d_flags, 4, hipMemcpyDeviceToHost, st));
  if (mem == ISL_MEM_HOST) ISL_HIP(hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, st));
  ISL_HIP(hipStreamSynchronize(st));
  if (flags)  // pq.rs:290-292 "Invalid code"
    return isl::fail(ISL_ERR_PQ, "Product quantization error: Invalid code (>= %llu centroids)",
                     (unsigned long long)pq->K);
  return ISL_OK;
}

isl_status isl_pq_table_distance(const isl_pq* pq, const float* tables, const uint16_t* codes,
                                 uint64_t n, float* out, int32_t mem, void* stream) {
  if (!pq) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "pq is NULL");
  if (n == 0) return ISL_OK;
  if (!tables || !codes || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  ISL_TRY(isl::use_device(pq->device));
  hipStream_t st = (hipStream_t)stream;
  Staged sg;
  const float* dt = tables;
  if (mem == ISL_MEM_HOST) {
    void* a = sg.alloc((size_t)pq->m * pq->K * 4);
    if (!a) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
    ISL_HIP(hipMemcpyAsync(a, tables, (size_t)pq->m * pq->K * 4, hipMemcpyHostToDevice, st));
    dt = (const float*)a;
  }
  return table_distance_impl(pq, dt, codes, n, out, mem, st);
}

// asymmetric_distance (pq.rs:275-304) adds the same per-subquantizer sums in the same order
// as table_distance over freshly built tables, so it is computed exactly that way.
isl_status isl_pq_asymmetric_distance(const isl_pq* pq, const float* query, uint64_t d,
                                      const uint16_t* codes, uint64_t n, float* out, int32_t mem,
                                      void* stream) {
  if (!pq) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "pq is NULL");
  if (d != pq->dimension) return isl::fail_dim(pq->dimension, d);  // pq.rs:276-281
  if (n == 0) return ISL_OK;
  if (!query || !codes || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  ISL_TRY(isl::use_device(pq->device));
  hipStream_t st = (hipStream_t)stream;
  Staged sg;
  float* dt = (float*)sg.alloc((size_t)pq->m * pq->K * 4);
  if (!dt) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
  const float* dq = query;
  if (mem == ISL_MEM_HOST) {
    void* a = sg.alloc((size_t)d * 4);
    if (!a) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
    ISL_HIP(hipMemcpyAsync(a, query, (size_t)d * 4, hipMemcpyHostToDevice, st));
    dq = (const float*)a;
  }
  ISL_TRY(launch_tables(pq, dq, 1, dt, st));
  return table_distance_impl(pq, dt, codes, n, out, mem, st);
}

isl_status isl_pq_encode(const isl_pq* pq, const float* vectors, uint64_t n, uint64_t d,
                         uint16_t* codes, int32_t mem, void* stream) {
  if (!pq) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "pq is NULL");
  if (d != pq->dimension) return isl::fail_dim(pq->dimension, d);  // pq.rs:225-230
  if (n == 0) return ISL_OK;
  if (!vectors || !codes) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  ISL_TRY(isl::use_device(pq->device));
  hipStream_t st = (hipStream_t)stream;
  Staged sg;
  const float* dv = vectors;
  uint16_t* dc = codes;
  if (mem == ISL_MEM_HOST) {
    void* a = sg.alloc((size_t)n * d * 4);
    void* b = sg.alloc((size_t)n * pq->m * 2);
    if (!a || !b) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
    ISL_HIP(hipMemcpyAsync(a, vectors, (size_t)n * d * 4, hipMemcpyHostToDevice, st));
    dv = (const float*)a;
    dc = (uint16_t*)b;
  }
  size_t lds = pq_lds((uint32_t)pq->dsub);
  for (uint64_t v0 = 0; v0 < n; v0 += 65535) {
    uint32_t nb = (uint32_t)std::min<uint64_t>(65535, n - v0);
    dim3 grid((uint32_t)pq->m, nb);
#define ISL_ENC(M)                                                                             \
  {                                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pq_encode_kernel<M>),              \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
    hipLaunchKernelGGL(pq_encode_kernel<M>, grid, dim3(64), lds, st, dv + v0 * d, (uint32_t)d, \
                       pq->d_codebooks, (uint32_t)pq->m, (uint32_t)pq->K, (uint32_t)pq->dsub,  \
                       (uint32_t)pq->cstride, dc + v0 * pq->m);                                \
  }
    switch (pq->metric) {
      case ISL_METRIC_COSINE: ISL_ENC(ISL_METRIC_COSINE) break;
      case ISL_METRIC_EUCLIDEAN: ISL_ENC(ISL_METRIC_EUCLIDEAN) break;
      case ISL_METRIC_DOT: ISL_ENC(ISL_METRIC_DOT) break;
      default: ISL_ENC(ISL_METRIC_MANHATTAN) break;
    }
#undef ISL_ENC
  }
  ISL_HIP(hipGetLastError());
  if (mem == ISL_MEM_HOST)
    ISL_HIP(hipMemcpyAsync(codes, dc, (size_t)n * pq->m * 2, hipMemcpyDeviceToHost, st));
  ISL_HIP(hipStreamSynchronize(st));
  return ISL_OK;
}

}  // extern "C"
