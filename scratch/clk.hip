#include <hip/hip_runtime.h>
#include <cstdio>
// measures shader clock, dependent v_add chain cost, mul+add chain with LDS reads, for 1 wave/SIMD
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* t, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 4352; i += 64) lds[i] = (float)i * 1e-3f;
  __syncthreads();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  float a = out[threadIdx.x];
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) a += 1.0f;   // dependent adds (compiler can't fold: fp)
  }
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  // chain with LDS reads like consume(): lane-per-row
  float acc = 0.f;
  const float* trow = lds + (threadIdx.x & 15) * 260;
  const float* qv = lds + 4160 - 4160;  // same array
  if (threadIdx.x < 16) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll 16
      for (int j = 0; j < 256; j += 4) {
        float4 x = *reinterpret_cast<const float4*>(trow + j);
        float4 q = *reinterpret_cast<const float4*>(qv + j);
        acc += q.x * x.x; acc += q.y * x.y; acc += q.z * x.z; acc += q.w * x.w;
      }
    }
  }
  unsigned long long r2 = __builtin_amdgcn_s_memrealtime(), c2 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x + blockIdx.x * 64] = a + acc;
  if (threadIdx.x == 0) { atomicAdd(&t[0], r1 - r0); atomicAdd(&t[1], c1 - c0); atomicAdd(&t[2], r2 - r1); atomicAdd(&t[3], c2 - c1); }
  unsigned simd = 0; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(simd));
  if (threadIdx.x == 0) { unsigned cu = (simd >> 8) & 0xf, sh = (simd>>12)&1, se = (simd>>13)&7, sid = (simd >> 4) & 3; atomicAdd(&t[8 + sid], 1ull); }
}
int main() {
  float* out; unsigned long long* t; hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20); hipMalloc(&t, 256);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int lds : {17408, 37120, 80000}) for (int grid : {256, 1024}) for (int bs : {64}) {
    int iters = 500;
    hipMemset(t, 0, 128);
    hipLaunchKernelGGL(k, dim3(grid), dim3(bs), lds, 0, out, t, iters);
    hipDeviceSynchronize();
    unsigned long long h[16]; hipMemcpy(h, t, 128, hipMemcpyDeviceToHost);
    double us2 = h[2] / 100.0 / grid;
    printf("lds %d grid %d: dep add %.2f cyc | mul+add+lds chain: %.2f cyc/elem, %.3f us per 256 elems | waves by SIMD id: %llu %llu %llu %llu\n",
           lds, grid, (double)h[1] / grid / (iters * 64.0), (double)h[3] / grid / (iters * 256.0), us2 / iters, h[8], h[9], h[10], h[11]);
  }
  return 0;
}
