#include <hip/hip_runtime.h>
#include <cstdio>
// faithful copy of the kernel's consume loop (16 lanes, pitch 260, aligned dynamic LDS)
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, const float* __restrict__ qg, unsigned long long* t, int iters) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* lds = reinterpret_cast<float*>(smem);
  for (int i = threadIdx.x; i < 16 * 260 + 1024; i += 64) lds[i] = (float)(i % 97) * 1e-3f;
  __syncthreads();
  float a0 = out[threadIdx.x];
  const int lane = threadIdx.x;
  const float* trow = lds + lane * 260;
  const float* qv = lds + 16 * 260;
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  if (lane < 16) {
    for (int i = 0; i < iters; ++i) {
      if (MODE == 0) {
#pragma unroll 16
        for (int j = 0; j < 256; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          float4 q = *reinterpret_cast<const float4*>(qv + j);
          a0 += q.x * x.x; a0 += q.y * x.y; a0 += q.z * x.z; a0 += q.w * x.w;
        }
      } else if (MODE == 1) {  // q through scalar loads
#pragma unroll 16
        for (int j = 0; j < 256; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          a0 += qg[j] * x.x; a0 += qg[j + 1] * x.y; a0 += qg[j + 2] * x.z; a0 += qg[j + 3] * x.w;
        }
      } else if (MODE == 2) {  // unroll 8
#pragma unroll 8
        for (int j = 0; j < 256; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          float4 q = *reinterpret_cast<const float4*>(qv + j);
          a0 += q.x * x.x; a0 += q.y * x.y; a0 += q.z * x.z; a0 += q.w * x.w;
        }
      } else if (MODE == 4) {
#pragma unroll 4
        for (int j = 0; j < 256; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          float4 q = *reinterpret_cast<const float4*>(qv + j);
          a0 += q.x * x.x; a0 += q.y * x.y; a0 += q.z * x.z; a0 += q.w * x.w;
        }
      } else if (MODE == 5) {
#pragma unroll 2
        for (int j = 0; j < 256; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          float4 q = *reinterpret_cast<const float4*>(qv + j);
          a0 += q.x * x.x; a0 += q.y * x.y; a0 += q.z * x.z; a0 += q.w * x.w;
        }
      } else if (MODE == 6) {  // q scalar, unroll 8
#pragma unroll 8
        for (int j = 0; j < 256; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          a0 += qg[j] * x.x; a0 += qg[j + 1] * x.y; a0 += qg[j + 2] * x.z; a0 += qg[j + 3] * x.w;
        }
      } else if (MODE == 7) {  // q scalar, unroll 4
#pragma unroll 4
        for (int j = 0; j < 256; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          a0 += qg[j] * x.x; a0 += qg[j + 1] * x.y; a0 += qg[j + 2] * x.z; a0 += qg[j + 3] * x.w;
        }
      } else if (MODE == 3) {  // products first (independent), then the chain of adds
#pragma unroll 4
        for (int j = 0; j < 256; j += 16) {
          float p[16];
#pragma unroll
          for (int u = 0; u < 16; u += 4) {
            float4 x = *reinterpret_cast<const float4*>(trow + j + u);
            float4 q = *reinterpret_cast<const float4*>(qv + j + u);
            p[u] = q.x * x.x; p[u + 1] = q.y * x.y; p[u + 2] = q.z * x.z; p[u + 3] = q.w * x.w;
          }
#pragma unroll
          for (int u = 0; u < 16; ++u) a0 += p[u];
        }
      }
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x + blockIdx.x * 64] = a0;
  if (threadIdx.x == 0) atomicAdd(&t[0], c1 - c0);
}
template <int MODE> void run(float* out, float* qg, unsigned long long* t, const char* name) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 500, grid = 1024;
  hipMemset(t, 0, 64);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 37120, 0, out, qg, t, iters);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("%-44s %.2f cyc/elem  (%.0f cycles per 768-chain)\n", name, (double)h / grid / (iters * 256.0), (double)h / grid / iters * 3);
}
int main() {
  float *out, *qg; unsigned long long* t; hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20); hipMalloc(&t, 256); hipMalloc(&qg, 4096); hipMemset(qg, 0, 4096);
  run<0>(out, qg, t, "x,q LDS b128, unroll 16 (kernel today)");
  run<1>(out, qg, t, "x LDS b128, q scalar loads");
  run<2>(out, qg, t, "x,q LDS b128, unroll 8");
  run<3>(out, qg, t, "products first, then add chain");
  run<4>(out, qg, t, "x,q LDS b128, unroll 4");
  run<5>(out, qg, t, "x,q LDS b128, unroll 2");
  run<6>(out, qg, t, "x LDS, q scalar, unroll 8");
  run<7>(out, qg, t, "x LDS, q scalar, unroll 4");
  return 0;
}
