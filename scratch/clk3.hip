#include <hip/hip_runtime.h>
#include <cstdio>
// cost of ds_read_b128 per wave-instruction for different active-lane patterns (1 wave per SIMD)
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* t, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 64 * 260; i += 64) lds[i] = (float)(i % 97) * 1e-3f;
  __syncthreads();
  float acc = out[threadIdx.x];
  int lane = threadIdx.x;
  bool active; int row;
  if (MODE == 0) { active = true; row = lane; }                 // 64 lanes, 64 rows
  else if (MODE == 1) { active = lane < 16; row = lane; }        // lanes 0..15
  else if (MODE == 2) { active = (lane & 3) == 0; row = lane >> 2; }  // every 4th lane
  else if (MODE == 3) { active = lane < 32; row = lane; }        // lanes 0..31
  else { active = lane < 16; row = 0; }                          // 16 lanes, same row (broadcast)
  const float* trow = lds + row * 260;
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  if (active) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll 16
      for (int j = 0; j < 256; j += 4) {
        float4 x = *reinterpret_cast<const float4*>(trow + j);
        acc += x.x;   // 1 VALU per read: the read dominates
        asm volatile("" : "+v"(acc));
      }
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x + blockIdx.x * 64] = acc;
  if (threadIdx.x == 0) atomicAdd(&t[0], c1 - c0);
}
template <int MODE> void run(float* out, unsigned long long* t, const char* name, int grid) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 500;
  hipMemset(t, 0, 64);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 70000, 0, out, t, iters);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("grid %4d %-36s %.1f cycles per ds_read_b128\n", grid, name, (double)h / grid / (iters * 64.0));
}
int main() {
  float *out; unsigned long long* t; hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20); hipMalloc(&t, 256);
  for (int grid : {256, 512}) {
    run<0>(out, t, "64 lanes, own rows", grid);
    run<1>(out, t, "lanes 0-15, own rows", grid);
    run<2>(out, t, "every 4th lane, own rows", grid);
    run<3>(out, t, "lanes 0-31, own rows", grid);
    run<4>(out, t, "lanes 0-15, same row", grid);
  }
  return 0;
}
