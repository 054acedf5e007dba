#include <hip/hip_runtime.h>
#include <cstdio>
// chain with q held in VGPRs (16 values per register across the 16-lane row) and fetched by the
// multiply itself through DPP row_newbcast; x from LDS (ds_read_b128).
#define MULB(K) asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "=v"(p) : "v"(qr), "v"(xv));
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* t, int iters, float* chk) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* lds = reinterpret_cast<float*>(smem);
  for (int i = threadIdx.x; i < 16 * 132 + 1024; i += 64) lds[i] = (float)((i * 7) % 97) * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x;
  const float* trow = lds + lane * 132;
  const float* qv = lds + 16 * 132;
  float a0 = 0.f;
  // q piece (128 floats) in 8 registers: register r, lane l (l<16) holds q[16 r + l]
  float qreg[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) qreg[r] = qv[16 * r + (lane & 15)];
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  if (lane < 16) {
    for (int i = 0; i < iters; ++i) {
      if (MODE == 0) {
#pragma unroll 8
        for (int j = 0; j < 128; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          float4 q = *reinterpret_cast<const float4*>(qv + j);
          a0 += q.x * x.x; a0 += q.y * x.y; a0 += q.z * x.z; a0 += q.w * x.w;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float qr = qreg[r];
          float4 x0 = *reinterpret_cast<const float4*>(trow + 16 * r);
          float4 x1 = *reinterpret_cast<const float4*>(trow + 16 * r + 4);
          float4 x2 = *reinterpret_cast<const float4*>(trow + 16 * r + 8);
          float4 x3 = *reinterpret_cast<const float4*>(trow + 16 * r + 12);
          float p, xv;
          xv = x0.x; MULB(0) a0 += p; xv = x0.y; MULB(1) a0 += p; xv = x0.z; MULB(2) a0 += p; xv = x0.w; MULB(3) a0 += p;
          xv = x1.x; MULB(4) a0 += p; xv = x1.y; MULB(5) a0 += p; xv = x1.z; MULB(6) a0 += p; xv = x1.w; MULB(7) a0 += p;
          xv = x2.x; MULB(8) a0 += p; xv = x2.y; MULB(9) a0 += p; xv = x2.z; MULB(10) a0 += p; xv = x2.w; MULB(11) a0 += p;
          xv = x3.x; MULB(12) a0 += p; xv = x3.y; MULB(13) a0 += p; xv = x3.z; MULB(14) a0 += p; xv = x3.w; MULB(15) a0 += p;
        }
      }
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x + blockIdx.x * 64] = a0;
  if (blockIdx.x == 0 && lane < 16) chk[lane] = a0;
  if (threadIdx.x == 0) atomicAdd(&t[0], c1 - c0);
}
template <int MODE> void run(float* out, unsigned long long* t, float* chk, const char* name) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 500, grid = 1024;
  hipMemset(t, 0, 64);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 20000, 0, out, t, iters, chk);
  hipError_t e = hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  float c[16]; hipMemcpy(c, chk, 64, hipMemcpyDeviceToHost);
  printf("%-34s %.2f cyc/elem  (err=%d) chk %g %g %g\n", name, (double)h / grid / (iters * 128.0), (int)e, c[0], c[1], c[15]);
}
int main() {
  float *out, *chk; unsigned long long* t; hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20); hipMalloc(&t, 256); hipMalloc(&chk, 64);
  run<0>(out, t, chk, "x,q LDS b128 unroll 8");
  run<1>(out, t, chk, "x LDS, q VGPR via row_newbcast");
  return 0;
}
