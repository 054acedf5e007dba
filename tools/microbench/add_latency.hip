#include <hip/hip_runtime.h>
#include <cstdio>
// dependent f32 add latency on gfx950: plain VOP2 vs DPP source, with/without an interleaved multiply
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* t, int iters, float seed) {
  float a = seed, p0 = seed * 0.5f, p1 = seed * 0.25f, p2 = seed * 0.125f, p3 = seed * 0.0625f;
  float x = seed * 1.5f, q = seed * 1.25f;
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) { a += p0; a += p1; a += p2; a += p3; }
      if (MODE == 1) {
        a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p0), 0x00, 0xf, 0xf, true));
        a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p1), 0x55, 0xf, 0xf, true));
        a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p2), 0xAA, 0xf, 0xf, true));
        a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p3), 0xFF, 0xf, 0xf, true));
      }
      if (MODE == 2) {  // mul + add per element (lane mode), operands in registers
        float m0 = x * q; a += m0; x += 1.0f;
        float m1 = x * q; a += m1; q += 1.0f;
        float m2 = x * q; a += m2; x += 1.0f;
        float m3 = x * q; a += m3; q += 1.0f;
      }
      asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x + blockIdx.x * 64] = a + x + q;
  if (threadIdx.x == 0) atomicAdd(&t[0], c1 - c0);
}
template <int MODE> void run(float* out, unsigned long long* t, const char* name, int grid) {
  int iters = 2000;
  (void)hipMemset(t, 0, 64);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0f);
  hipError_t e = hipDeviceSynchronize();
  unsigned long long h; (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("%-34s grid %5d  %.2f cycles per dependent add (err=%d)\n", name, grid, (double)h / grid / (iters * 64.0), (int)e);
}
int main() {
  float* out; unsigned long long* t; (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&t, 256);
  for (int grid : {256, 1024, 2048, 4096}) {
    run<0>(out, t, "plain v_add_f32 chain", grid);
    run<1>(out, t, "v_add_f32_dpp quad_perm chain", grid);
    run<2>(out, t, "mul + add per element", grid);
  }
  return 0;
}
