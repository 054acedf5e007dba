// Batched distances as a query x candidate GEMM on the matrix cores: every (query, row) pair of
// Distance::batch_calculate (src/core/distance.rs:32-34, benches/vector_ops.rs:60-79) at once,
// and exact brute-force top-k on top of it (the ground truth of the recall measurement).
// The dot products come out of v_mfma_f32_32x32x2_f32 in the MFMA's accumulation order, so the
// values agree with the reference's sequential sums to float32 rounding (<= 1e-5 on normalised
// rows), not bit for bit -- the traversal keeps the exact-order kernels.
#include "device_common.hip.h"
#include "gemm_f32.hip.h"
#include "gemm_bf16.hip.h"

#include <algorithm>
#include <vector>

namespace {

using namespace isl_gemm;

// sum of squares of every row (one wave per row, 16 bytes per lane and load when the rows allow it)
__global__ __launch_bounds__(256) void sumsq_rows_kernel(const float* __restrict__ x, uint64_t n, uint32_t d,
                                                         uint64_t stride, float* __restrict__ out) {
  const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* rp = x + row * stride;
  float s = 0.0f;
  if ((d & 3u) == 0 && (stride & 3u) == 0 && ((uintptr_t)x & 15u) == 0) {
    for (uint32_t j = lane * 4; j < d; j += 256) {
      const float4 v = *reinterpret_cast<const float4*>(rp + j);
      s += v.x * v.x;
      s += v.y * v.y;
      s += v.z * v.z;
      s += v.w * v.w;
    }
  } else {
    for (uint32_t j = lane; j < d; j += 64) { float v = rp[j]; s += v * v; }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[row] = s;
}

// the same for bf16 rows (their exact f32 images are squared): one wave per row, 16 bytes per lane
// and load when the row allows it
__global__ __launch_bounds__(256) void sumsq_rows_bf16_kernel(const uint16_t* __restrict__ x, uint64_t n, uint32_t d,
                                                              float* __restrict__ out) {
  const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (row >= n) return;
  const uint16_t* rp = x + row * d;
  float s = 0.0f;
  if ((d & 7u) == 0 && ((uintptr_t)x & 15u) == 0) {
    for (uint32_t j = lane * 8; j < d; j += 512) {
      const uint4 w = *reinterpret_cast<const uint4*>(rp + j);
      const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float lo = __uint_as_float(ws[u] << 16), hi = __uint_as_float(ws[u] & 0xFFFF0000u);
        s += lo * lo;
        s += hi * hi;
      }
    }
  } else {
    for (uint32_t j = lane; j < d; j += 64) {
      const float v = __uint_as_float((uint32_t)rp[j] << 16);
      s += v * v;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[row] = s;
}

// Keeps the k smallest (distance, id) of every query across column chunks: one wave per query,
// the running list sorted ascending in LDS, ties resolved towards the smaller id.  The scan takes
// 16 bytes per lane and step (the next step's load is issued before the current one is examined);
// only values that beat the k-th best so far reach the serial insertion, k ln(n / k) of them over a
// whole scan of n columns.
__device__ __forceinline__ void topk_insert(float* bd, uint64_t* bi, uint32_t& cnt, uint32_t k, float nv, uint64_t ni) {
  if (cnt == k && !(nv < bd[k - 1] || (nv == bd[k - 1] && ni < bi[k - 1]))) return;  // no longer among the k best
  uint32_t pos = cnt < k ? cnt : k - 1;
  while (pos > 0 && (nv < bd[pos - 1] || (nv == bd[pos - 1] && ni < bi[pos - 1]))) {
    bd[pos] = bd[pos - 1];
    bi[pos] = bi[pos - 1];
    --pos;
  }
  bd[pos] = nv;
  bi[pos] = ni;
  if (cnt < k) ++cnt;
}

__global__ __launch_bounds__(64) void topk_chunk_kernel(const float* __restrict__ dist, uint64_t ld,
                                                        uint32_t cols, uint64_t col0, uint32_t k,
                                                        float* __restrict__ best_d, uint64_t* __restrict__ best_i,
                                                        uint32_t* __restrict__ best_n) {
  extern __shared__ unsigned char sm[];
  float* bd = reinterpret_cast<float*>(sm);
  uint64_t* bi = reinterpret_cast<uint64_t*>(sm + ((k * 4 + 7) & ~7u));
  const uint32_t q = blockIdx.x, lane = threadIdx.x;
  uint32_t cnt = best_n[q];
  for (uint32_t i = lane; i < cnt; i += 64) { bd[i] = best_d[(uint64_t)q * k + i]; bi[i] = best_i[(uint64_t)q * k + i]; }
  __syncthreads();
  const float* row = dist + (uint64_t)q * ld;
  const bool vec = (ld & 3u) == 0 && ((uintptr_t)dist & 15u) == 0;
  const float kInf = __builtin_inff();
  auto load4 = [&](uint32_t c) -> float4 {  // columns c .. c + 3 of the row, +inf past the end
    if (vec && c + 4 <= cols) return *reinterpret_cast<const float4*>(row + c);
    float4 v;
    v.x = c < cols ? row[c] : kInf;
    v.y = c + 1 < cols ? row[c + 1] : kInf;
    v.z = c + 2 < cols ? row[c + 2] : kInf;
    v.w = c + 3 < cols ? row[c + 3] : kInf;
    return v;
  };
  float4 nxt = load4(lane * 4);
  for (uint32_t c0 = 0; c0 < cols; c0 += 256) {
    const uint32_t c = c0 + lane * 4;
    const float4 v4 = nxt;
    if (c0 + 256 < cols) nxt = load4(c + 256);
    const float vs[4] = {v4.x, v4.y, v4.z, v4.w};
    const bool full = cnt == k;
    const float kth = full ? bd[k - 1] : 0.0f;
    const uint64_t kid = full ? bi[k - 1] : 0ull;
    uint32_t cm = 0;  // which of this lane's four columns may enter the list
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool in = c + j < cols;
      if (in && (!full || vs[j] < kth || (vs[j] == kth && col0 + c + j < kid))) cm |= 1u << j;
    }
    uint64_t m = __ballot(cm != 0);
    while (m) {
      const int l = __ffsll((long long)m) - 1;
      m &= m - 1;
      const uint32_t lm = (uint32_t)__shfl((int)cm, l);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float nv = __shfl(vs[j], l);
        if (lane == 0 && ((lm >> j) & 1u)) topk_insert(bd, bi, cnt, k, nv, col0 + c0 + (uint32_t)l * 4 + j);
      }
      cnt = (uint32_t)__shfl((int)cnt, 0);
      __syncthreads();
    }
  }
  __syncthreads();
  for (uint32_t i = lane; i < cnt; i += 64) { best_d[(uint64_t)q * k + i] = bd[i]; best_i[(uint64_t)q * k + i] = bi[i]; }
  if (lane == 0) best_n[q] = cnt;
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

// device copy of a [n][d] matrix with rows padded to a multiple of 4 floats (GEMM loads 16 B)
isl_status stage_matrix(Staged& s, const float* src, uint64_t n, uint64_t d, int32_t mem, hipStream_t st,
                        const float** out, uint64_t* stride) {
  const uint64_t ld = (d + 3) / 4 * 4;
  *stride = ld;
  if (mem == ISL_MEM_DEVICE && ld == d) { *out = src; return ISL_OK; }
  float* p = (float*)s.alloc(n * ld * 4);
  if (!p) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
  if (ld != d) ISL_HIP(hipMemsetAsync(p, 0, n * ld * 4, st));
  ISL_HIP(hipMemcpy2DAsync(p, ld * 4, src, d * 4, d * 4, n,
                           mem == ISL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  *out = p;
  return ISL_OK;
}

isl_status launch_distance_gemm(int32_t metric, const float* dq, const float* dr, const float* qn,
                                const float* rn, float* dout, uint64_t nq, uint64_t n, uint64_t ld,
                                hipStream_t st) {
  switch (metric) {
    case ISL_METRIC_COSINE: launch_gemm<EPI_COSINE, false>(dq, dr, rn, qn, dout, nq, n, ld, st); break;
    case ISL_METRIC_DOT: launch_gemm<EPI_DOT, false>(dq, dr, nullptr, nullptr, dout, nq, n, ld, st); break;
    case ISL_METRIC_EUCLIDEAN: launch_gemm<EPI_EUCLIDEAN, false>(dq, dr, rn, qn, dout, nq, n, ld, st); break;
    default:
      return isl::fail(ISL_ERR_UNSUPPORTED, "the Manhattan distance is not a contraction: use isl_distance_batch");
  }
  ISL_HIP(hipGetLastError());
  return ISL_OK;
}

isl_status launch_distance_gemm_bf16(int32_t metric, const uint16_t* dq, const uint16_t* dr, const float* qn,
                                     const float* rn, float* dout, uint64_t nq, uint64_t n, uint64_t d,
                                     uint64_t ldc, hipStream_t st) {
  const __bf16* q = reinterpret_cast<const __bf16*>(dq);
  const __bf16* r = reinterpret_cast<const __bf16*>(dr);
  switch (metric) {
    case ISL_METRIC_COSINE: launch_gemm_bf16_distance<EPI_COSINE>(q, r, rn, qn, dout, nq, n, d, ldc, st); break;
    case ISL_METRIC_DOT: launch_gemm_bf16_distance<EPI_DOT>(q, r, nullptr, nullptr, dout, nq, n, d, ldc, st); break;
    case ISL_METRIC_EUCLIDEAN: launch_gemm_bf16_distance<EPI_EUCLIDEAN>(q, r, rn, qn, dout, nq, n, d, ldc, st); break;
    default:
      return isl::fail(ISL_ERR_UNSUPPORTED, "the Manhattan distance is not a contraction: use isl_distance_batch");
  }
  ISL_HIP(hipGetLastError());
  return ISL_OK;
}

}  // namespace

extern "C" {

// bf16 rows and queries (BASELINE config 5): products of two bf16 values are exact in float32, the
// sums run in the accumulation order of v_mfma_f32_32x32x16_bf16.  d must be a multiple of 64.
isl_status isl_distance_matrix_bf16(int32_t metric, const uint16_t* queries, uint64_t nq, const uint16_t* rows,
                                    uint64_t n, uint64_t d, float* out, int32_t mem, int32_t device,
                                    void* stream) {
  return isl_distance_matrix_bf16_norms(metric, queries, nq, rows, n, d, nullptr, nullptr, out, mem, device, stream);
}

// sum of squares of every row of a bf16 matrix (its exact float32 images), in the order the distance
// epilogues take it: what isl_distance_matrix_bf16 computes per call, for callers that keep the rows
// resident and ask for their distances to many query batches (config 5: 26 batches x 153 row blocks)
isl_status isl_row_sumsq_bf16(const uint16_t* rows, uint64_t n, uint64_t d, float* out, int32_t mem, int32_t device,
                              void* stream) {
  if (n == 0) return ISL_OK;
  if (!rows || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  if (d == 0) return isl::fail(ISL_ERR_EMPTY_COLLECTION, "Empty vector collection");
  ISL_TRY(isl::use_device(device));
  hipStream_t st = (hipStream_t)stream;
  Staged s;
  const uint16_t* dr = rows;
  float* dout = out;
  if (mem == ISL_MEM_HOST) {
    uint16_t* b = (uint16_t*)s.alloc(n * d * 2);
    dout = (float*)s.alloc(n * 4);
    if (!b || !dout) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
    ISL_HIP(hipMemcpyAsync(b, rows, n * d * 2, hipMemcpyHostToDevice, st));
    dr = b;
  }
  for (uint64_t r0 = 0; r0 < n; r0 += 0x7FFFFFFCull)  // grid.x limit (four rows per workgroup)
    hipLaunchKernelGGL(sumsq_rows_bf16_kernel, dim3((uint32_t)((std::min<uint64_t>(n - r0, 0x7FFFFFFCull) + 3) / 4)), dim3(256),
                       0, st, dr + r0 * d, std::min<uint64_t>(n - r0, 0x7FFFFFFCull), (uint32_t)d, dout + r0);
  ISL_HIP(hipGetLastError());
  if (mem == ISL_MEM_HOST) ISL_HIP(hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, st));
  ISL_HIP(hipStreamSynchronize(st));
  return ISL_OK;
}

// q_sumsq / row_sumsq: isl_row_sumsq_bf16 of the queries / rows (same memory space as the matrices), or
// NULL = computed here.  The outputs do not depend on who computed them (same kernel, same order).
isl_status isl_distance_matrix_bf16_norms(int32_t metric, const uint16_t* queries, uint64_t nq, const uint16_t* rows,
                                          uint64_t n, uint64_t d, const float* q_sumsq, const float* row_sumsq,
                                          float* out, int32_t mem, int32_t device, void* stream) {
  if (nq == 0 || n == 0) return ISL_OK;
  if (!queries || !rows || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  if (d == 0) return isl::fail(ISL_ERR_EMPTY_COLLECTION, "Empty vector collection");
  if (d % 64) return isl::fail(ISL_ERR_UNSUPPORTED, "bf16 distance matrix: the dimension must be a multiple of 64");
  if (nq > 0xFFFFFFFFull || n > 0xFFFFFFFFull || ((nq + 127) / 128) * ((n + 127) / 128) >= 0x7FFFFFFFull)
    return isl::fail(ISL_ERR_UNSUPPORTED, "matrix too large for one launch: split the rows");
  ISL_TRY(isl::use_device(device));
  hipStream_t st = (hipStream_t)stream;
  Staged s;
  const uint16_t *dq = queries, *dr = rows;
  float* dout = out;
  if (mem == ISL_MEM_HOST) {
    uint16_t* a = (uint16_t*)s.alloc(nq * d * 2);
    uint16_t* b = (uint16_t*)s.alloc(n * d * 2);
    dout = (float*)s.alloc(nq * n * 4);
    if (!a || !b || !dout) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
    ISL_HIP(hipMemcpyAsync(a, queries, nq * d * 2, hipMemcpyHostToDevice, st));
    ISL_HIP(hipMemcpyAsync(b, rows, n * d * 2, hipMemcpyHostToDevice, st));
    dq = a;
    dr = b;
  }
  if (((uintptr_t)dq & 15) || ((uintptr_t)dr & 15))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "bf16 matrices must be 16-byte aligned");
  const float *qn = nullptr, *rn = nullptr;
  if (metric != ISL_METRIC_DOT) {  // (the dot epilogue reads no norms)
    float* qbuf = (q_sumsq && mem == ISL_MEM_DEVICE) ? nullptr : (float*)s.alloc(nq * 4);
    float* rbuf = (row_sumsq && mem == ISL_MEM_DEVICE) ? nullptr : (float*)s.alloc(n * 4);
    if ((!qbuf && !(q_sumsq && mem == ISL_MEM_DEVICE)) || (!rbuf && !(row_sumsq && mem == ISL_MEM_DEVICE)))
      return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
    if (q_sumsq && mem == ISL_MEM_HOST) ISL_HIP(hipMemcpyAsync(qbuf, q_sumsq, nq * 4, hipMemcpyHostToDevice, st));
    else if (!q_sumsq)
      hipLaunchKernelGGL(sumsq_rows_bf16_kernel, dim3((uint32_t)((nq + 3) / 4)), dim3(256), 0, st, dq, nq, (uint32_t)d, qbuf);
    if (row_sumsq && mem == ISL_MEM_HOST) ISL_HIP(hipMemcpyAsync(rbuf, row_sumsq, n * 4, hipMemcpyHostToDevice, st));
    else if (!row_sumsq)
      hipLaunchKernelGGL(sumsq_rows_bf16_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, st, dr, n, (uint32_t)d, rbuf);
    qn = qbuf ? qbuf : q_sumsq;
    rn = rbuf ? rbuf : row_sumsq;
  }
  ISL_TRY(launch_distance_gemm_bf16(metric, dq, dr, qn, rn, dout, nq, n, d, n, st));
  if (mem == ISL_MEM_HOST) ISL_HIP(hipMemcpyAsync(out, dout, nq * n * 4, hipMemcpyDeviceToHost, st));
  ISL_HIP(hipStreamSynchronize(st));
  return ISL_OK;
}

// The same GEMM, enqueued only: device buffers, the sums of squares handed in, nothing allocated and nothing
// waited for -- the caller's next kernel on `stream` (a top-k over the block, the next block's call) finds the
// distances there.  A call that synchronises leaves the chip idle between two blocks and lets it drop its
// clocks (1.93 ms per 4096 x 65536 x 4096 block alone against 1.70 back to back: DESIGN.md section 3.4).
isl_status isl_distance_matrix_bf16_enqueue(int32_t metric, const uint16_t* queries, uint64_t nq, const uint16_t* rows,
                                            uint64_t n, uint64_t d, const float* q_sumsq, const float* row_sumsq,
                                            float* out, int32_t device, void* stream) {
  if (nq == 0 || n == 0) return ISL_OK;
  if (!queries || !rows || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  if (metric != ISL_METRIC_DOT && (!q_sumsq || !row_sumsq))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "the enqueue-only form takes the sums of squares from the caller (isl_row_sumsq_bf16)");
  if (d == 0) return isl::fail(ISL_ERR_EMPTY_COLLECTION, "Empty vector collection");
  if (d % 64) return isl::fail(ISL_ERR_UNSUPPORTED, "bf16 distance matrix: the dimension must be a multiple of 64");
  if (nq > 0xFFFFFFFFull || n > 0xFFFFFFFFull || ((nq + 127) / 128) * ((n + 127) / 128) >= 0x7FFFFFFFull)
    return isl::fail(ISL_ERR_UNSUPPORTED, "matrix too large for one launch: split the rows");
  if (((uintptr_t)queries & 15) || ((uintptr_t)rows & 15))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "bf16 matrices must be 16-byte aligned");
  ISL_TRY(isl::use_device(device));
  return launch_distance_gemm_bf16(metric, queries, rows, q_sumsq, row_sumsq, out, nq, n, d, n, (hipStream_t)stream);
}

isl_status isl_distance_matrix(int32_t metric, const float* queries, uint64_t nq, const float* rows,
                               uint64_t n, uint64_t d, float* out, int32_t mem, int32_t device,
                               void* stream) {
  if (nq == 0 || n == 0) return ISL_OK;
  if (!queries || !rows || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  if (d == 0) return isl::fail(ISL_ERR_EMPTY_COLLECTION, "Empty vector collection");
  if (nq > 0xFFFFFFFFull || n > 0xFFFFFFFFull) return isl::fail(ISL_ERR_UNSUPPORTED, "matrix side above 2^32");
  ISL_TRY(isl::use_device(device));
  hipStream_t st = (hipStream_t)stream;
  Staged s;
  const float *dq, *dr;
  uint64_t ldq, ldr;
  ISL_TRY(stage_matrix(s, queries, nq, d, mem, st, &dq, &ldq));
  ISL_TRY(stage_matrix(s, rows, n, d, mem, st, &dr, &ldr));
  float* qn = (float*)s.alloc(nq * 4);
  float* rn = (float*)s.alloc(n * 4);
  float* dout = mem == ISL_MEM_DEVICE ? out : (float*)s.alloc(nq * n * 4);
  if (!qn || !rn || !dout) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
  hipLaunchKernelGGL(sumsq_rows_kernel, dim3((uint32_t)((nq + 3) / 4)), dim3(256), 0, st, dq, nq, (uint32_t)d, ldq, qn);
  hipLaunchKernelGGL(sumsq_rows_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, st, dr, n, (uint32_t)d, ldr, rn);
  ISL_TRY(launch_distance_gemm(metric, dq, dr, qn, rn, dout, nq, n, ldq, st));
  if (mem == ISL_MEM_HOST) ISL_HIP(hipMemcpyAsync(out, dout, nq * n * 4, hipMemcpyDeviceToHost, st));
  ISL_HIP(hipStreamSynchronize(st));
  return ISL_OK;
}

isl_status isl_bruteforce_topk(int32_t metric, const float* queries, uint64_t nq, const float* rows,
                               uint64_t n, uint64_t d, uint64_t k, uint64_t* out_ids, float* out_dist,
                               uint32_t* out_count, int32_t mem, int32_t device, void* stream) {
  if (nq == 0) return ISL_OK;
  if (!queries || !out_count || (k && (!out_ids || !out_dist)) || (!rows && n))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  if (k > 1024) return isl::fail(ISL_ERR_UNSUPPORTED, "k <= 1024");
  if (nq > 0xFFFFFFFFull) return isl::fail(ISL_ERR_UNSUPPORTED, "too many queries");
  ISL_TRY(isl::use_device(device));
  hipStream_t st = (hipStream_t)stream;
  Staged s;
  const uint64_t kk = k ? k : 1;
  float* bd = (float*)s.alloc(nq * kk * 4);
  uint64_t* bi = (uint64_t*)s.alloc(nq * kk * 8);
  uint32_t* bn = (uint32_t*)s.alloc(nq * 4);
  if (!bd || !bi || !bn) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
  ISL_HIP(hipMemsetAsync(bn, 0, nq * 4, st));
  if (n && k && d) {
    const float *dq, *dr;
    uint64_t ldq, ldr;
    ISL_TRY(stage_matrix(s, queries, nq, d, mem, st, &dq, &ldq));
    ISL_TRY(stage_matrix(s, rows, n, d, mem, st, &dr, &ldr));
    // column chunks sized so that the distance block stays near 1 GiB
    const uint64_t chunk = std::max<uint64_t>(4096, std::min<uint64_t>(n, (1ull << 28) / std::max<uint64_t>(nq, 1)));
    float* qn = (float*)s.alloc(nq * 4);
    float* rn = (float*)s.alloc(n * 4);
    float* blk = (float*)s.alloc(nq * chunk * 4);
    if (!qn || !rn || !blk) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
    hipLaunchKernelGGL(sumsq_rows_kernel, dim3((uint32_t)((nq + 3) / 4)), dim3(256), 0, st, dq, nq, (uint32_t)d, ldq, qn);
    for (uint64_t r0 = 0; r0 < n; r0 += 0x7FFFFFFCull)  // grid.x limit (four rows per workgroup)
      hipLaunchKernelGGL(sumsq_rows_kernel, dim3((uint32_t)((std::min<uint64_t>(n - r0, 0x7FFFFFFCull) + 3) / 4)), dim3(256), 0,
                         st, dr + r0 * ldr, std::min<uint64_t>(n - r0, 0x7FFFFFFCull), (uint32_t)d, ldr, rn + r0);
    const size_t lds = ((kk * 4 + 7) & ~7ull) + kk * 8;
    for (uint64_t c0 = 0; c0 < n; c0 += chunk) {
      const uint64_t cols = std::min(chunk, n - c0);
      ISL_TRY(launch_distance_gemm(metric, dq, dr + c0 * ldr, qn, rn + c0, blk, nq, cols, ldq, st));
      hipLaunchKernelGGL(topk_chunk_kernel, dim3((uint32_t)nq), dim3(64), lds, st, blk, cols, (uint32_t)cols, c0,
                         (uint32_t)k, bd, bi, bn);
    }
    ISL_HIP(hipGetLastError());
  }
  hipMemcpyKind kind = mem == ISL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (k) {
    ISL_HIP(hipMemcpyAsync(out_ids, bi, nq * k * 8, kind, st));
    ISL_HIP(hipMemcpyAsync(out_dist, bd, nq * k * 4, kind, st));
  }
  ISL_HIP(hipMemcpyAsync(out_count, bn, nq * 4, kind, st));
  ISL_HIP(hipStreamSynchronize(st));
  return ISL_OK;
}

// The same over bf16 rows and queries: distance blocks from the bf16 matrix cores
// (isl_distance_matrix_bf16's kernel), exact top-k under THOSE distances.  What makes an exact kNN
// graph or a ground truth over 10M rows a matter of minutes (1.5e17 flops at d = 768).
isl_status isl_bruteforce_topk_bf16(int32_t metric, const uint16_t* queries, uint64_t nq, const uint16_t* rows,
                                    uint64_t n, uint64_t d, uint64_t k, uint64_t* out_ids, float* out_dist,
                                    uint32_t* out_count, int32_t mem, int32_t device, void* stream) {
  if (nq == 0) return ISL_OK;
  if (!queries || !out_count || (k && (!out_ids || !out_dist)) || (!rows && n))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  if (k > 1024) return isl::fail(ISL_ERR_UNSUPPORTED, "k <= 1024");
  if (nq > 0xFFFFFFFFull) return isl::fail(ISL_ERR_UNSUPPORTED, "too many queries");
  if (d == 0 && n) return isl::fail(ISL_ERR_EMPTY_COLLECTION, "Empty vector collection");
  if (d % 64) return isl::fail(ISL_ERR_UNSUPPORTED, "bf16 distance matrix: the dimension must be a multiple of 64");
  ISL_TRY(isl::use_device(device));
  hipStream_t st = (hipStream_t)stream;
  Staged s;
  const uint64_t kk = k ? k : 1;
  float* bd = (float*)s.alloc(nq * kk * 4);
  uint64_t* bi = (uint64_t*)s.alloc(nq * kk * 8);
  uint32_t* bn = (uint32_t*)s.alloc(nq * 4);
  if (!bd || !bi || !bn) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
  ISL_HIP(hipMemsetAsync(bn, 0, nq * 4, st));
  if (n && k) {
    const uint16_t *dq = queries, *dr = rows;
    if (mem == ISL_MEM_HOST) {
      uint16_t* a = (uint16_t*)s.alloc(nq * d * 2);
      uint16_t* b = (uint16_t*)s.alloc(n * d * 2);
      if (!a || !b) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
      ISL_HIP(hipMemcpyAsync(a, queries, nq * d * 2, hipMemcpyHostToDevice, st));
      ISL_HIP(hipMemcpyAsync(b, rows, n * d * 2, hipMemcpyHostToDevice, st));
      dq = a;
      dr = b;
    }
    if (((uintptr_t)dq & 15) || ((uintptr_t)dr & 15))
      return isl::fail(ISL_ERR_INVALID_ARGUMENT, "bf16 matrices must be 16-byte aligned");
    // column chunks sized so that the distance block stays near 1 GiB (a multiple of 256 columns: the
    // GEMM's tile, and the top-k scan's 16-byte loads stay aligned in every row)
    const uint64_t chunk = std::max<uint64_t>(4096, std::min<uint64_t>((n + 255) / 256 * 256,
                                                                      ((1ull << 28) / std::max<uint64_t>(nq, 1)) / 256 * 256));
    float* qn = (float*)s.alloc(nq * 4);
    float* rn = (float*)s.alloc(n * 4);
    float* blk = (float*)s.alloc(nq * chunk * 4);
    if (!qn || !rn || !blk) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
    if (metric != ISL_METRIC_DOT) {
      hipLaunchKernelGGL(sumsq_rows_bf16_kernel, dim3((uint32_t)((nq + 3) / 4)), dim3(256), 0, st, dq, nq, (uint32_t)d, qn);
      for (uint64_t r0 = 0; r0 < n; r0 += 0x7FFFFFFCull)  // grid.x limit (four rows per workgroup)
        hipLaunchKernelGGL(sumsq_rows_bf16_kernel, dim3((uint32_t)((std::min<uint64_t>(n - r0, 0x7FFFFFFCull) + 3) / 4)),
                           dim3(256), 0, st, dr + r0 * d, std::min<uint64_t>(n - r0, 0x7FFFFFFCull), (uint32_t)d, rn + r0);
    }
    const size_t lds = ((kk * 4 + 7) & ~7ull) + kk * 8;
    for (uint64_t c0 = 0; c0 < n; c0 += chunk) {
      const uint64_t cols = std::min(chunk, n - c0);
      ISL_TRY(launch_distance_gemm_bf16(metric, dq, dr + c0 * d, qn, rn + c0, blk, nq, cols, d, chunk, st));
      hipLaunchKernelGGL(topk_chunk_kernel, dim3((uint32_t)nq), dim3(64), lds, st, blk, chunk, (uint32_t)cols, c0,
                         (uint32_t)k, bd, bi, bn);
    }
    ISL_HIP(hipGetLastError());
  }
  hipMemcpyKind kind = mem == ISL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (k) {
    ISL_HIP(hipMemcpyAsync(out_ids, bi, nq * k * 8, kind, st));
    ISL_HIP(hipMemcpyAsync(out_dist, bd, nq * k * 4, kind, st));
  }
  ISL_HIP(hipMemcpyAsync(out_count, bn, nq * 4, kind, st));
  ISL_HIP(hipStreamSynchronize(st));
  return ISL_OK;
}

}  // extern "C"
