#include <hip/hip_runtime.h>
#include <cstdio>
// pure LDS read throughput of ONE wave: 16 independent reads in flight, then lgkmcnt(0)
template <int WIDTH>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* t, int iters, int nactive) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 64 * 260; i += 64) lds[i] = (float)(i % 97) * 1e-3f;
  __syncthreads();
  int lane = threadIdx.x;
  unsigned addr = (unsigned)(lane * 260 * 4);
  float4 r[16];
  float acc = 0.f;
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  if (lane < nactive) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (WIDTH == 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[j]) : "v"(addr), "n"(j * 16));
        else if (WIDTH == 8) { float2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(j * 8)); r[j].x = v.x; r[j].y = v.y; }
        else { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(j * 4)); r[j].x = v; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += r[j].x;
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x + blockIdx.x * 64] = acc;
  if (threadIdx.x == 0) atomicAdd(&t[0], c1 - c0);
}
template <int W> void run(float* out, unsigned long long* t, int grid, int nactive) {
  hipFuncSetAttribute((const void*)k<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 2000;
  hipMemset(t, 0, 64);
  hipLaunchKernelGGL(k<W>, dim3(grid), dim3(64), 70000, 0, out, t, iters, nactive);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("grid %4d width %2d active %2d: %.1f cycles per read (incl. 1 dependent add each)\n", grid, W, nactive, (double)h / grid / (iters * 16.0));
}
int main() {
  float *out; unsigned long long* t; hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20); hipMalloc(&t, 256);
  for (int grid : {256, 512}) for (int na : {64, 16}) { run<16>(out, t, grid, na); run<8>(out, t, grid, na); run<4>(out, t, grid, na); }
  return 0;
}
