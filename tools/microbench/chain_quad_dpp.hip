#include <hip/hip_runtime.h>
#include <cstdio>
// MODE 0: current consume loop: lane r<16 owns row r, x and q from LDS b128, mul+add per element.
// MODE 1: quad mode: lanes 4r..4r+3 own row r; lane (r,s) multiplies elements 16i+4s..+3, the add
//         chain runs redundantly in all 4 lanes and fetches the products with DPP quad_perm.
// MODE 2: pair mode: lanes 2r,2r+1 own row r (32 rows).
__device__ __forceinline__ float qb(float v, int sel) {  // compile-time sel after unrolling
  switch (sel) {
    case 0: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x00, 0xf, 0xf, true));
    case 1: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x55, 0xf, 0xf, true));
    case 2: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xAA, 0xf, 0xf, true));
    default: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xFF, 0xf, 0xf, true));
  }
}
__device__ __forceinline__ float pb(float v, int sel) {  // pairs: [0,0,2,2] / [1,1,3,3]
  if (sel == 0) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xA0, 0xf, 0xf, true));
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xF5, 0xf, 0xf, true));
}
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* t, int iters, float* chk) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* lds = reinterpret_cast<float*>(smem);
  constexpr int LD = MODE == 0 ? 132 : 144;
  for (int i = threadIdx.x; i < 32 * LD + 1024; i += 64) lds[i] = (float)((i * 7) % 97) * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x;
  const float* qv = lds + 32 * LD;
  float a0 = 0.f;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(); unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 0) {
    const float* trow = lds + lane * LD;
    if (lane < 16) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll 8
        for (int j = 0; j < 128; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          float4 q = *reinterpret_cast<const float4*>(qv + j);
          a0 += q.x * x.x; a0 += q.y * x.y; a0 += q.z * x.z; a0 += q.w * x.w;
        }
      }
    }
  } else if (MODE == 1) {
    const int r = lane >> 2, s = lane & 3;
    const float* trow = lds + r * LD + 4 * s;
    const float* qq = qv + 4 * s;
    for (int i = 0; i < iters; ++i) {
#pragma unroll 4
      for (int j = 0; j < 128; j += 16) {
        float4 x = *reinterpret_cast<const float4*>(trow + j);
        float4 q = *reinterpret_cast<const float4*>(qq + j);
        float p0 = q.x * x.x, p1 = q.y * x.y, p2 = q.z * x.z, p3 = q.w * x.w;
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
          a0 += qb(p0, ss); a0 += qb(p1, ss); a0 += qb(p2, ss); a0 += qb(p3, ss);
        }
      }
    }
  } else if (MODE == 3) {
    const int r = lane >> 2, s = lane & 3;
    const float* trow = lds + r * LD + 4 * s;
    const float* qq = qv + 4 * s;
    for (int i = 0; i < iters; ++i) {
#pragma unroll 4
      for (int j = 0; j < 128; j += 16) {
        float4 x = *reinterpret_cast<const float4*>(trow + j);
        float4 q = *reinterpret_cast<const float4*>(qq + j);
        float p0 = q.x * x.x, p1 = q.y * x.y, p2 = q.z * x.z, p3 = q.w * x.w;
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
          // the cross-lane move is taken off the dependent chain: v_mov_b32_dpp into a plain
          // register (kept from folding into the add by the empty asm), then a plain v_add_f32
          float t0 = qb(p0, ss), t1 = qb(p1, ss), t2 = qb(p2, ss), t3 = qb(p3, ss);
          asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
          a0 += t0; a0 += t1; a0 += t2; a0 += t3;
        }
      }
    }
  } else {
    const int r = lane >> 1, s = lane & 1;
    const float* trow = lds + r * LD + 4 * s;
    const float* qq = qv + 4 * s;
    for (int i = 0; i < iters; ++i) {
#pragma unroll 8
      for (int j = 0; j < 128; j += 8) {
        float4 x = *reinterpret_cast<const float4*>(trow + j);
        float4 q = *reinterpret_cast<const float4*>(qq + j);
        float p0 = q.x * x.x, p1 = q.y * x.y, p2 = q.z * x.z, p3 = q.w * x.w;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          a0 += pb(p0, ss); a0 += pb(p1, ss); a0 += pb(p2, ss); a0 += pb(p3, ss);
        }
      }
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(); unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[threadIdx.x + blockIdx.x * 64] = a0;
  if (blockIdx.x == 0) chk[lane] = a0;
  if (threadIdx.x == 0) { atomicAdd(&t[0], c1 - c0); atomicAdd(&t[1], r1 - r0); }
}
template <int MODE> void run(float* out, unsigned long long* t, float* chk, const char* name, int grid) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 500;
  hipMemset(t, 0, 64);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 24000, 0, out, t, iters, chk);
  hipError_t e = hipDeviceSynchronize();
  unsigned long long hh[2]; hipMemcpy(hh, t, 16, hipMemcpyDeviceToHost); unsigned long long h = hh[0];
  float c[64]; hipMemcpy(c, chk, 256, hipMemcpyDeviceToHost);
  int stride = MODE == 0 ? 1 : ((MODE == 1 || MODE == 3) ? 4 : 2);
  printf("%-28s grid %5d  %.2f cyc/elem %.3f ns/elem (err=%d) chk row0 %.9g row1 %.9g row15 %.9g\n", name, grid,
         (double)h / grid / (iters * 128.0), (double)hh[1] * 10.0 / grid / (iters * 128.0), (int)e, c[0], c[stride], c[15 * stride]);
}
int main() {
  float *out, *chk; unsigned long long* t; hipMalloc(&out, 1 << 22); hipMemset(out, 0, 1 << 22); hipMalloc(&t, 256); hipMalloc(&chk, 256);
  for (int grid : {256, 1024, 2048}) {
    run<0>(out, t, chk, "lane-per-row (16 rows)", grid);
    run<1>(out, t, chk, "quad-per-row (16 rows)", grid);
    run<2>(out, t, chk, "pair-per-row (32 rows)", grid);
    run<3>(out, t, chk, "quad, mov_dpp + plain add", grid);
  }
  return 0;
}
