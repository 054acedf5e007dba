#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, const float* __restrict__ qg, unsigned long long* t, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 4352 + 1024; i += 64) lds[i] = (float)(i % 97) * 1e-3f;
  __syncthreads();
  float acc = out[threadIdx.x];
  const float* trow = lds + (threadIdx.x & 15) * 260;
  const float* qv = lds + 4352;
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x < 16) {
    for (int i = 0; i < iters; ++i) {
      if (MODE == 0) {  // x and q from LDS
#pragma unroll 16
        for (int j = 0; j < 256; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          float4 q = *reinterpret_cast<const float4*>(qv + j);
          acc += q.x * x.x; acc += q.y * x.y; acc += q.z * x.z; acc += q.w * x.w;
        }
      } else if (MODE == 1) {  // x from LDS, q constant register
        float q = out[64];
#pragma unroll 16
        for (int j = 0; j < 256; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          acc += q * x.x; acc += q * x.y; acc += q * x.z; acc += q * x.w;
        }
      } else if (MODE == 2) {  // x from LDS, q from global via uniform (scalar) loads
#pragma unroll 16
        for (int j = 0; j < 256; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          acc += qg[j] * x.x; acc += qg[j + 1] * x.y; acc += qg[j + 2] * x.z; acc += qg[j + 3] * x.w;
        }
      } else if (MODE == 3) {  // no LDS at all
        float q = out[64], x = out[65];
#pragma unroll 16
        for (int j = 0; j < 256; j += 4) {
          acc += q * x; acc += q * x; acc += q * x; acc += q * x;
          asm volatile("" : "+v"(x));
        }
      } else if (MODE == 4) {  // x via ds_read_b32 ... scalar floats
#pragma unroll 16
        for (int j = 0; j < 256; j += 4) {
          float x0 = trow[j], x1 = trow[j+1], x2 = trow[j+2], x3 = trow[j+3];
          float q = out[64];
          acc += q * x0; acc += q * x1; acc += q * x2; acc += q * x3;
        }
      }
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x + blockIdx.x * 64] = acc;
  if (threadIdx.x == 0) atomicAdd(&t[0], c1 - c0);
}
template <int MODE> void run(float* out, float* qg, unsigned long long* t, const char* name) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 500, grid = 1024;
  hipMemset(t, 0, 64);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 37120, 0, out, qg, t, iters);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("%-40s %.2f cyc/elem\n", name, (double)h / grid / (iters * 256.0));
}
int main() {
  float *out, *qg; unsigned long long* t; hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20); hipMalloc(&t, 256); hipMalloc(&qg, 4096); hipMemset(qg, 0, 4096);
  run<0>(out, qg, t, "x LDS b128 + q LDS b128");
  run<1>(out, qg, t, "x LDS b128, q register");
  run<2>(out, qg, t, "x LDS b128, q scalar global");
  run<3>(out, qg, t, "no LDS (registers only)");
  run<4>(out, qg, t, "x LDS scalar reads, q register");
  return 0;
}
