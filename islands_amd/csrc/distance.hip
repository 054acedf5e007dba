// Batched distance ops: Distance::{calculate, calculate_squared, batch_calculate}
// (src/core/distance.rs:22-66) and normalize_vector (distance.rs:125-132) on gfx950.
// Arithmetic order is the reference's (see device_common.hip.h): one lane owns one row.
#include "device_common.hip.h"

#include <algorithm>
#include <vector>

namespace {

using namespace isl_dev;

// One wave per 64 consecutive rows; grid-strides over row blocks.
template <int METRIC>
__global__ __launch_bounds__(64) void distance_batch_kernel(const float* __restrict__ query,
                                                            const float* __restrict__ rows,
                                                            uint64_t n, uint32_t d,
                                                            uint64_t row_stride,
                                                            float* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  float* qs = tile + TILE_ROWS * TILE_LD;
  const int lane = threadIdx.x;
  const float q_norm = load_query<METRIC>(query, d, qs);
  for (uint64_t base = (uint64_t)blockIdx.x * 64; base < n; base += (uint64_t)gridDim.x * 64) {
    uint32_t R = (uint32_t)(n - base < 64 ? n - base : 64);
    // row ids relative to `base` keep the 32-bit id type of wave_distances
    float v = wave_distances<METRIC>(rows + base * row_stride, row_stride, d, (uint32_t)lane, R,
                                     qs, tile, q_norm);
    if ((uint32_t)lane < R) out[base + lane] = v;
  }
}

// rows[i][j] /= ||rows[i]|| when the norm is > 0 (distance.rs:125-132).
__global__ __launch_bounds__(64) void normalize_rows_kernel(float* __restrict__ rows, uint64_t n,
                                                            uint32_t d) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  float* norms = tile + TILE_ROWS * TILE_LD;  // 64 floats
  float* qs = norms + 64;                     // never written: SUMSQ ignores the query operand
  const int lane = threadIdx.x;
  for (uint64_t base = (uint64_t)blockIdx.x * 64; base < n; base += (uint64_t)gridDim.x * 64) {
    uint32_t R = (uint32_t)(n - base < 64 ? n - base : 64);
    float nv = wave_distances<METRIC_SUMSQ>(rows + base * d, d, d, (uint32_t)lane, R, qs, tile,
                                            0.0f);
    if ((uint32_t)lane < R) norms[lane] = nv;
    __syncthreads();
    for (uint64_t e = lane; e < (uint64_t)R * d; e += 64) {
      float nr = norms[e / d];
      if (nr > 0.0f) rows[base * d + e] = rows[base * d + e] / nr;
    }
    __syncthreads();
  }
}

size_t dist_lds(uint32_t d) {
  return (size_t)TILE_ROWS * TILE_LD * 4 + (size_t)((d + 3) / 4 * 4) * 4 + 64;
}

template <int METRIC>
void launch_dist(uint32_t grid, hipStream_t st, const float* q, const float* rows, uint64_t n,
                 uint32_t d, uint64_t stride, float* out) {
  auto k = distance_batch_kernel<METRIC>;
  size_t lds = dist_lds(d);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, q, rows, n, d, stride, out);
}

// rows must be readable one PIECE past the last row end -> stage into a padded buffer
isl_status run_distance(int32_t metric, const float* query, uint64_t d, const float* rows,
                        uint64_t n, float* out, int32_t mem, int32_t device, hipStream_t st) {
  ISL_TRY(isl::use_device(device));
  if (n == 0) return ISL_OK;
  if (d == 0 || d > 65536) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "dimension out of range");
  const uint64_t stride = (d + 3) / 4 * 4;
  float *dq = nullptr, *drows = nullptr, *dout = nullptr;
  auto cleanup = [&]() {
    if (dq) (void)hipFree(dq);
    if (drows) (void)hipFree(drows);
    if (dout && mem == ISL_MEM_HOST) (void)hipFree(dout);
  };
  hipMemcpyKind kin = mem == ISL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  // private padded copies keep the kernel's whole-slab reads inside the allocation
  size_t rbytes = (size_t)(n * stride + 256) * 4;
  if (hipMalloc(&dq, (size_t)stride * 4) != hipSuccess ||
      hipMalloc(&drows, rbytes) != hipSuccess) {
    cleanup();
    return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed in isl_distance_batch");
  }
  hipError_t e = hipMemsetAsync(drows, 0, rbytes, st);
  if (e == hipSuccess) e = hipMemcpyAsync(dq, query, d * 4, kin, st);
  if (e == hipSuccess) {
    if (stride == d) e = hipMemcpyAsync(drows, rows, (size_t)n * d * 4, kin, st);
    else e = hipMemcpy2DAsync(drows, stride * 4, rows, d * 4, d * 4, n, kin, st);
  }
  if (e == hipSuccess) {
    if (mem == ISL_MEM_DEVICE) dout = out;
    else e = hipMalloc(&dout, n * 4);
  }
  if (e != hipSuccess) {
    cleanup();
    return isl::fail(ISL_ERR_DEVICE, "staging failed in isl_distance_batch: %s",
                     hipGetErrorString(e));
  }
  uint32_t grid = (uint32_t)std::min<uint64_t>((n + 63) / 64, 4096);
  switch (metric) {
    case ISL_METRIC_COSINE: launch_dist<ISL_METRIC_COSINE>(grid, st, dq, drows, n, (uint32_t)d, stride, dout); break;
    case ISL_METRIC_EUCLIDEAN: launch_dist<ISL_METRIC_EUCLIDEAN>(grid, st, dq, drows, n, (uint32_t)d, stride, dout); break;
    case ISL_METRIC_DOT: launch_dist<ISL_METRIC_DOT>(grid, st, dq, drows, n, (uint32_t)d, stride, dout); break;
    case ISL_METRIC_MANHATTAN: launch_dist<ISL_METRIC_MANHATTAN>(grid, st, dq, drows, n, (uint32_t)d, stride, dout); break;
    case METRIC_EUCLID_SQ: launch_dist<METRIC_EUCLID_SQ>(grid, st, dq, drows, n, (uint32_t)d, stride, dout); break;
    default:
      cleanup();
      return isl::fail(ISL_ERR_INVALID_ARGUMENT, "unknown metric %d", metric);
  }
  e = hipGetLastError();
  if (e == hipSuccess && mem == ISL_MEM_HOST)
    e = hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  cleanup();
  if (e != hipSuccess)
    return isl::fail(ISL_ERR_DEVICE, "isl_distance_batch failed: %s", hipGetErrorString(e));
  return ISL_OK;
}

}  // namespace

extern "C" {

isl_status isl_distance_batch(int32_t metric, const float* query, uint64_t d, const float* rows,
                              uint64_t n, uint64_t row_len, float* out, int32_t mem, int32_t device,
                              void* stream) {
  if (metric < 0 || metric > ISL_METRIC_MANHATTAN)
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "unknown metric %d", metric);
  if (n && row_len != d) return isl::fail_dim(d, row_len);  // distance.rs:39-44
  if (n == 0) return ISL_OK;                                  // empty batch, distance.rs:374-382
  if (!query || !rows || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  return run_distance(metric, query, d, rows, n, out, mem, device, (hipStream_t)stream);
}

isl_status isl_distance(int32_t metric, const float* a, uint64_t na, const float* b, uint64_t nb,
                        float* out) {
  if (!a || !b || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  if (na != nb) return isl::fail_dim(na, nb);
  if (na == 0) {  // empty vectors: every accumulator stays 0
    *out = metric == ISL_METRIC_COSINE ? 1.0f : (metric == ISL_METRIC_DOT ? -0.0f : 0.0f);
    return ISL_OK;
  }
  return isl_distance_batch(metric, a, na, b, 1, nb, out, ISL_MEM_HOST, 0, nullptr);
}

isl_status isl_distance_squared(int32_t metric, const float* a, uint64_t na, const float* b,
                                uint64_t nb, float* out) {
  if (!a || !b || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  if (na != nb) return isl::fail_dim(na, nb);
  if (metric == ISL_METRIC_EUCLIDEAN) {  // distance.rs:63
    if (na == 0) { *out = 0.0f; return ISL_OK; }
    return run_distance(METRIC_EUCLID_SQ, a, na, b, 1, out, ISL_MEM_HOST, 0, nullptr);
  }
  float dist = 0.0f;
  ISL_TRY(isl_distance(metric, a, na, b, nb, &dist));
  *out = dist * dist;  // distance.rs:64
  return ISL_OK;
}

isl_status isl_normalize_rows(float* rows, uint64_t n, uint64_t d, int32_t mem, int32_t device,
                              void* stream) {
  if (n == 0 || d == 0) return ISL_OK;
  if (!rows) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  if (d % 4 != 0)
    return isl::fail(ISL_ERR_UNSUPPORTED, "isl_normalize_rows needs d to be a multiple of 4");
  ISL_TRY(isl::use_device(device));
  hipStream_t st = (hipStream_t)stream;
  float* dr = nullptr;
  size_t bytes = (size_t)(n * d + 256) * 4;
  ISL_HIP(hipMalloc(&dr, bytes));
  hipMemcpyKind kin = mem == ISL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  hipMemcpyKind kout = mem == ISL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  hipError_t e = hipMemsetAsync(dr + n * d, 0, 256 * 4, st);
  if (e == hipSuccess) e = hipMemcpyAsync(dr, rows, (size_t)n * d * 4, kin, st);
  if (e == hipSuccess) {
    size_t lds = (size_t)TILE_ROWS * TILE_LD * 4 + 64 * 4 + (size_t)d * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(normalize_rows_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    uint32_t grid = (uint32_t)std::min<uint64_t>((n + 63) / 64, 4096);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(grid), dim3(64), lds, st, dr, n, (uint32_t)d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(rows, dr, (size_t)n * d * 4, kout, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(dr);
  if (e != hipSuccess)
    return isl::fail(ISL_ERR_DEVICE, "isl_normalize_rows failed: %s", hipGetErrorString(e));
  return ISL_OK;
}

}  // extern "C"
