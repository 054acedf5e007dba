// Masked mean-pool + L2 normalise: the part of the recompute encoder that is islands' own
// code (src/core/embedding/candle_provider.rs:434-488).  The BERT forward itself is
// third-party (candle-transformers 0.9.1, not in the reference tree) and is NOT built in
// round 1; summation order here is sequential over the sequence / hidden axis, matching the
// oracle's restatement (the reference's tensor-library reduction order is unpinned).
#include "common.hpp"

namespace {

// one thread per (batch row, hidden unit): sum_t hidden[b][t][h] * mask[b][t] / clamp(sum_t mask, 1e-9)
__global__ void mean_pool_kernel(const float* __restrict__ hidden, const float* __restrict__ mask,
                                 uint32_t B, uint32_t L, uint32_t H, float* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)B * H) return;
  uint32_t b = (uint32_t)(i / H), h = (uint32_t)(i % H);
  float sum_mask = 0.0f;
  for (uint32_t t = 0; t < L; ++t) sum_mask += mask[(uint64_t)b * L + t];
  if (sum_mask < 1e-9f) sum_mask = 1e-9f;  // clamp(1e-9, MAX), :455-459
  float s = 0.0f;
  for (uint32_t t = 0; t < L; ++t) s += hidden[((uint64_t)b * L + t) * H + h] * mask[(uint64_t)b * L + t];
  out[i] = s / sum_mask;  // :462-463
}

// one thread per batch row: x / clamp(sqrt(sum x^2), 1e-12), :466-481
__global__ void l2_normalize_kernel(float* __restrict__ out, uint32_t B, uint32_t H) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float* row = out + (uint64_t)b * H;
  float ss = 0.0f;
  for (uint32_t h = 0; h < H; ++h) ss += row[h] * row[h];
  float norm = sqrtf(ss);
  if (norm < 1e-12f) norm = 1e-12f;
  for (uint32_t h = 0; h < H; ++h) row[h] = row[h] / norm;
}

}  // namespace

extern "C" isl_status isl_mean_pool_normalize(const float* hidden, const float* mask, uint64_t B,
                                              uint64_t L, uint64_t H, int32_t normalize, float* out,
                                              int32_t mem, int32_t device, void* stream) {
  if (B == 0 || H == 0) return ISL_OK;
  if (!hidden || !mask || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  ISL_TRY(isl::use_device(device));
  hipStream_t st = (hipStream_t)stream;
  const float *dh = hidden, *dm = mask;
  float* dout = out;
  float *th = nullptr, *tm = nullptr, *to = nullptr;
  auto cleanup = [&]() {
    if (th) (void)hipFree(th);
    if (tm) (void)hipFree(tm);
    if (to) (void)hipFree(to);
  };
  hipError_t e = hipSuccess;
  if (mem == ISL_MEM_HOST) {
    if (hipMalloc(&th, B * L * H * 4 + 4) != hipSuccess || hipMalloc(&tm, B * L * 4 + 4) != hipSuccess ||
        hipMalloc(&to, B * H * 4) != hipSuccess) {
      cleanup();
      return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed in isl_mean_pool_normalize");
    }
    e = hipMemcpyAsync(th, hidden, B * L * H * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(tm, mask, B * L * 4, hipMemcpyHostToDevice, st);
    dh = th; dm = tm; dout = to;
  }
  if (e == hipSuccess) {
    uint64_t n = B * H;
    hipLaunchKernelGGL(mean_pool_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, dh, dm,
                       (uint32_t)B, (uint32_t)L, (uint32_t)H, dout);
    if (normalize)
      hipLaunchKernelGGL(l2_normalize_kernel, dim3((uint32_t)((B + 63) / 64)), dim3(64), 0, st, dout,
                         (uint32_t)B, (uint32_t)H);
    e = hipGetLastError();
  }
  if (e == hipSuccess && mem == ISL_MEM_HOST)
    e = hipMemcpyAsync(out, dout, B * H * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  cleanup();
  if (e != hipSuccess)
    return isl::fail(ISL_ERR_DEVICE, "isl_mean_pool_normalize failed: %s", hipGetErrorString(e));
  return ISL_OK;
}
