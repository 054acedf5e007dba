// Recompute encoder of the LEANN path: CandleEmbedder::embed_texts_raw,
// src/core/embedding/candle_provider.rs:353-507 -- BERT forward (:429-432), masked mean pooling
// (:434-463), optional L2 normalisation (:466-488).
//
// The model is third-party code that is not in the reference tree (candle-transformers 0.9.1
// `models::bert::BertModel`, Cargo.lock:1113-1114; a port of HuggingFace modeling_bert.py).  Its
// published algorithm is what runs here, in float32 like the reference's CPU path:
//   embeddings = word + token_type + position -> LayerNorm
//   per layer:  QKV = x W^T + b (one fused GEMM) -> softmax(Q K^T / sqrt(dh) + (1-mask)*f32::MIN) V
//               -> dense + residual -> LayerNorm -> dense + GELU -> dense + residual -> LayerNorm
// The linear layers (>98 % of the flops: 24 h^2 L of 24 h^2 L + 4 L^2 h per layer and sequence)
// run on the matrix cores with v_mfma_f32_32x32x2_f32; attention, LayerNorm and the embedding
// gather are VALU kernels.  Every output row depends on its own sequence only and is reduced in a
// fixed order, so a node's embedding does not depend on what it is batched with.
#include "common.hpp"
#include "encoder.hpp"
#include "gemm_f32.hip.h"
#include "gemm_bf16.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

using namespace isl_gemm;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// LayerNorm of one token row held in LDS (`row`, h floats): biased variance of the centred
// values, eps inside the square root (candle_nn::LayerNorm / torch.nn.LayerNorm).
__device__ __forceinline__ void ln_row(const float* row, uint32_t h, const float* __restrict__ w,
                                       const float* __restrict__ b, float eps, float* __restrict__ out,
                                       __bf16* __restrict__ out16 = nullptr) {
  const uint32_t lane = threadIdx.x;
  float s = 0.0f;
  for (uint32_t j = lane; j < h; j += 64) s += row[j];
  const float mean = wave_sum(s) / (float)h;
  float v = 0.0f;
  for (uint32_t j = lane; j < h; j += 64) { float c = row[j] - mean; v += c * c; }
  const float var = wave_sum(v) / (float)h;
  const float inv = 1.0f / sqrtf(var + eps);
  for (uint32_t j = lane; j < h; j += 64) {
    const float y = (row[j] - mean) * inv * w[j] + b[j];
    out[j] = y;
    if (out16) out16[j] = (__bf16)y;  // bf16 mode: the copy the next GEMM reads
  }
}

// BertEmbeddings: word + token_type + position -> LayerNorm.  One wave per token.
__global__ __launch_bounds__(64) void embed_ln_kernel(const int64_t* __restrict__ ids,
                                                      const int64_t* __restrict__ tt, uint32_t L,
                                                      uint32_t h, uint32_t vocab, uint32_t ntypes,
                                                      const float* __restrict__ word,
                                                      const float* __restrict__ pos,
                                                      const float* __restrict__ type,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ b, float eps,
                                                      float* __restrict__ out, uint32_t* flag,
                                                      __bf16* __restrict__ out16) {
  extern __shared__ float row[];
  const uint64_t tok = blockIdx.x;
  const uint32_t p = (uint32_t)(tok % L);
  int64_t id = ids[tok], ty = tt ? tt[tok] : 0;
  if (id < 0 || id >= (int64_t)vocab || ty < 0 || ty >= (int64_t)ntypes) {
    if (threadIdx.x == 0) atomicOr(flag, 1u);
    id = 0;
    ty = 0;
  }
  for (uint32_t j = threadIdx.x; j < h; j += 64)
    row[j] = word[(uint64_t)id * h + j] + type[(uint64_t)ty * h + j] + pos[(uint64_t)p * h + j];
  __syncthreads();
  ln_row(row, h, w, b, eps, out + tok * h, out16 ? out16 + tok * h : nullptr);
}

// y = LayerNorm(a) (the residual is already added by the GEMM epilogue).  One wave per token.
__global__ __launch_bounds__(64) void ln_kernel(const float* __restrict__ a, uint32_t h,
                                                const float* __restrict__ w,
                                                const float* __restrict__ b, float eps,
                                                float* __restrict__ out, __bf16* __restrict__ out16) {
  extern __shared__ float row[];
  const uint64_t tok = blockIdx.x;
  for (uint32_t j = threadIdx.x; j < h; j += 64) row[j] = a[tok * h + j];
  __syncthreads();
  ln_row(row, h, w, b, eps, out + tok * h, out16 ? out16 + tok * h : nullptr);
}

// BertSelfAttention for one (sequence, head) and 64 query rows per wave: lane i owns query row
// i.  Keys / values of the head are staged through LDS 64 at a time; scores of the chunk go to
// an LDS row per lane, the softmax is carried online across chunks (running max / sum).
// scores = q.k / sqrt(dh) + (1 - mask[key]) * f32::MIN, exactly candle's broadcast_add of the
// extended mask; padded query rows are computed like any other (the reference does, too).
template <int DH>
__global__ __launch_bounds__(64) void attention_kernel(const float* __restrict__ qkv,
                                                       const float* __restrict__ mask, uint32_t L,
                                                       uint32_t heads, float* __restrict__ ctx) {
  __shared__ float Ks[64][DH + 4];
  __shared__ float Vs[64][DH + 4];
  __shared__ float Ss[64][65];
  const uint32_t lane = threadIdx.x;
  const uint32_t b = blockIdx.x / heads, hd = blockIdx.x % heads;
  const uint32_t h = heads * DH, ld = 3 * h;
  const uint32_t qi = blockIdx.y * 64 + lane;
  const bool qok = qi < L;
  const float scale = sqrtf((float)DH);
  float q[DH], acc[DH];
  {
    const float* qp = qkv + ((uint64_t)b * L + (qok ? qi : 0)) * ld + hd * DH;
#pragma unroll
    for (int c = 0; c < DH; c += 4) {
      float4 v = *reinterpret_cast<const float4*>(qp + c);
      q[c] = v.x; q[c + 1] = v.y; q[c + 2] = v.z; q[c + 3] = v.w;
    }
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] = 0.0f;
  }
  float mrun = -INFINITY, lrun = 0.0f;
  for (uint32_t j0 = 0; j0 < L; j0 += 64) {
    const uint32_t nk = L - j0 < 64 ? L - j0 : 64;
    __syncthreads();
    // stage K and V rows j0 .. j0+nk: 64 lanes x float4 cover 4 rows of DH=64 (or 8 of DH=32)
    constexpr int LPR = DH / 4;  // lanes per row
    for (uint32_t rr = lane / LPR; rr < nk; rr += 64 / LPR) {
      const uint32_t c = (lane % LPR) * 4;
      const float* base = qkv + ((uint64_t)b * L + j0 + rr) * ld + hd * DH + c;
      float4 kv = *reinterpret_cast<const float4*>(base + h);
      float4 vv = *reinterpret_cast<const float4*>(base + 2 * h);
      Ks[rr][c] = kv.x; Ks[rr][c + 1] = kv.y; Ks[rr][c + 2] = kv.z; Ks[rr][c + 3] = kv.w;
      Vs[rr][c] = vv.x; Vs[rr][c + 1] = vv.y; Vs[rr][c + 2] = vv.z; Vs[rr][c + 3] = vv.w;
    }
    __syncthreads();
    float cmax = -INFINITY;
    for (uint32_t j = 0; j < nk; ++j) {
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        float4 kv = *reinterpret_cast<const float4*>(&Ks[j][c]);
        s += q[c] * kv.x; s += q[c + 1] * kv.y; s += q[c + 2] * kv.z; s += q[c + 3] * kv.w;
      }
      s = s / scale + (1.0f - mask[(uint64_t)b * L + j0 + j]) * -3.40282347e+38f;
      Ss[lane][j] = s;
      cmax = fmaxf(cmax, s);
    }
    const float mnew = fmaxf(mrun, cmax);
    const float corr = expf(mrun - mnew);  // first chunk: exp(-inf) = 0
    lrun *= corr;
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] *= corr;
    for (uint32_t j = 0; j < nk; ++j) {
      const float p = expf(Ss[lane][j] - mnew);
      lrun += p;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        float4 vv = *reinterpret_cast<const float4*>(&Vs[j][c]);
        acc[c] += p * vv.x; acc[c + 1] += p * vv.y; acc[c + 2] += p * vv.z; acc[c + 3] += p * vv.w;
      }
    }
    mrun = mnew;
  }
  if (qok) {
    float* op = ctx + ((uint64_t)b * L + qi) * h + hd * DH;
    const float inv = 1.0f / lrun;
#pragma unroll
    for (int c = 0; c < DH; c += 4)
      *reinterpret_cast<float4*>(op + c) = make_float4(acc[c] * inv, acc[c + 1] * inv, acc[c + 2] * inv, acc[c + 3] * inv);
  }
}

// BertSelfAttention on the matrix cores for head sizes 32 and 64: one wave per (sequence, head)
// and block of 64 queries.  The transposed problem is computed so that the probabilities never
// leave the registers: S^T = K Q^T (rows = keys, columns = queries) with v_mfma_f32_32x32x2_f32
// puts a query's 64 scores into one lane pair (lane l and l^32 hold column l%32), so the
// softmax is lane-local plus one exchange, and the accumulator registers of S^T are exactly the
// B operand of O^T = V^T P^T once the contraction index is enumerated in the order the
// accumulator holds it (register 4a+b of lane l holds key 8a + 4(l/32) + b).  Keys are walked
// in blocks of 64 with an online softmax (running max / sum per query).  Same arithmetic as
// the VALU kernel up to summation order: q.k / sqrt(dh) + (1 - mask[key]) * f32::MIN.
template <int DH>
__global__ __launch_bounds__(64) void attention_mfma_kernel(const float* __restrict__ qkv,
                                                            const float* __restrict__ mask, uint32_t L,
                                                            uint32_t heads, float* __restrict__ ctx,
                                                            __bf16* __restrict__ ctx16) {
  constexpr int LD = DH + 1;   // tile pitch: conflict-free column reads
  constexpr int CF = DH / 32;  // 32-wide blocks of the head dimension
  __shared__ float Ta[64 * LD];  // Q block, later the V block
  __shared__ float Tb[64 * LD];  // K block
  const uint32_t lane = threadIdx.x, c32 = lane & 31, kh = lane >> 5;
  const uint32_t b = blockIdx.x / heads, hd = blockIdx.x % heads;
  const uint32_t h = heads * DH, ld = 3 * h;
  const uint32_t q0 = blockIdx.y * 64;
  const float inv_scale = sqrtf((float)DH);
  constexpr int LPR = DH / 4;  // lanes per row when staging with float4 loads
  // Staging of 64 rows of Q (0) / K (1) / V (2) in two halves: all loads of the block are issued
  // together (one memory round trip per stage instead of one per four rows), the LDS stores follow
  // when the tile is free -- the V block is fetched while the scores are still being computed.
  constexpr int RPI = 64 / LPR;  // rows per iteration
  const uint32_t sc = (lane % LPR) * 4;
  auto stage_load = [&](float4 (&v)[LPR], uint32_t row0, uint32_t which) {
#pragma unroll
    for (int it = 0; it < LPR; ++it) {
      const uint32_t rr = lane / LPR + it * RPI;
      const uint32_t row = row0 + rr < L ? row0 + rr : L - 1;  // clamped: the load stays unconditional
      v[it] = *reinterpret_cast<const float4*>(qkv + ((uint64_t)b * L + row) * ld + which * h + hd * DH + sc);
      if (row0 + rr >= L) v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage_store = [&](float* T, const float4 (&v)[LPR]) {
#pragma unroll
    for (int it = 0; it < LPR; ++it) {
      const uint32_t rr = lane / LPR + it * RPI;
      T[rr * LD + sc] = v[it].x; T[rr * LD + sc + 1] = v[it].y; T[rr * LD + sc + 2] = v[it].z; T[rr * LD + sc + 3] = v[it].w;
    }
  };
  floatx16 o[CF][2];
#pragma unroll
  for (int cf = 0; cf < CF; ++cf)
#pragma unroll
    for (int qf = 0; qf < 2; ++qf)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[cf][qf][r] = 0.0f;
  float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.0f, 0.0f};
  for (uint32_t j0 = 0; j0 < L; j0 += 64) {
    __syncthreads();
    {
      float4 vq[LPR], vk[LPR];
      stage_load(vq, q0, 0);
      stage_load(vk, j0, 1);
      stage_store(Ta, vq);
      stage_store(Tb, vk);
    }
    float4 vv[LPR];
    stage_load(vv, j0, 2);  // in flight during the score MFMAs
    __syncthreads();
    // S^T[j][q] = sum_c K[j][c] Q[q][c]
    floatx16 sT[2][2];
#pragma unroll
    for (int jf = 0; jf < 2; ++jf)
#pragma unroll
      for (int qf = 0; qf < 2; ++qf)
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[jf][qf][r] = 0.0f;
#pragma unroll 4
    for (int kk = 0; kk < DH / 2; ++kk) {
      const float a0 = Tb[c32 * LD + 2 * kk + kh], a1 = Tb[(32 + c32) * LD + 2 * kk + kh];
      const float b0 = Ta[c32 * LD + 2 * kk + kh], b1 = Ta[(32 + c32) * LD + 2 * kk + kh];
      sT[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, sT[0][0], 0, 0, 0);
      sT[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, sT[0][1], 0, 0, 0);
      sT[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, sT[1][0], 0, 0, 0);
      sT[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, sT[1][1], 0, 0, 0);
    }
    __syncthreads();
    stage_store(Ta, vv);  // V block replaces the Q block
    // scores: scale, mask bias per key (row), keys past L excluded
    float cmax[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int jf = 0; jf < 2; ++jf)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t j = j0 + 32 * jf + 8 * (r / 4) + 4 * kh + (r % 4);
        const bool live = j < L;
        const float bias = live ? (1.0f - mask[(uint64_t)b * L + j]) * -3.40282347e+38f : 0.0f;
#pragma unroll
        for (int qf = 0; qf < 2; ++qf) {
          const float sv = live ? sT[jf][qf][r] / inv_scale + bias : -INFINITY;
          sT[jf][qf][r] = sv;
          cmax[qf] = fmaxf(cmax[qf], sv);
        }
      }
    float corr[2];
#pragma unroll
    for (int qf = 0; qf < 2; ++qf) {
      cmax[qf] = fmaxf(cmax[qf], __shfl_xor(cmax[qf], 32));
      const float mnew = fmaxf(mrun[qf], cmax[qf]);
      corr[qf] = expf(mrun[qf] - mnew);
      mrun[qf] = mnew;
      float ps = 0.0f;
#pragma unroll
      for (int jf = 0; jf < 2; ++jf)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = expf(sT[jf][qf][r] - mnew);  // exp(-inf) = 0 for keys past L
          sT[jf][qf][r] = pv;
          ps += pv;
        }
      ps += __shfl_xor(ps, 32);
      lrun[qf] = lrun[qf] * corr[qf] + ps;
#pragma unroll
      for (int cf = 0; cf < CF; ++cf)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cf][qf][r] *= corr[qf];
    }
    __syncthreads();  // V staged
    // O^T[c][q] += sum_j V[j][c] P^T[j][q]; contraction slot (t, half) <-> key 32 jf + 8a + 4 half + b
#pragma unroll
    for (int jf = 0; jf < 2; ++jf)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const uint32_t j = 32 * jf + 8 * (t / 4) + 4 * kh + (t % 4);
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) {
          const float av = Ta[j * LD + 32 * cf + c32];
          o[cf][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, sT[jf][0][t], o[cf][0], 0, 0, 0);
          o[cf][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, sT[jf][1][t], o[cf][1], 0, 0, 0);
        }
      }
  }
  // ctx[q][c] = O^T[c][q] / l_q
#pragma unroll
  for (int qf = 0; qf < 2; ++qf) {
    const uint32_t q = q0 + 32 * qf + c32;
    if (q >= L) continue;
    const float inv = 1.0f / lrun[qf];
    const uint64_t obase = ((uint64_t)b * L + q) * h + hd * DH;
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t c = 32 * cf + 8 * (r / 4) + 4 * kh + (r % 4);
        if (ctx16) ctx16[obase + c] = (__bf16)(o[cf][qf][r] * inv);  // bf16 mode: only the next GEMM reads it
        else ctx[obase + c] = o[cf][qf][r] * inv;
      }
  }
}

// masked mean pooling + optional L2 normalisation, candle_provider.rs:434-488; one wave per
// sequence, lane j owns hidden units j, j+64, ...; sequential over the tokens.
__global__ __launch_bounds__(64) void pool_kernel(const float* __restrict__ hid,
                                                  const float* __restrict__ mask, uint32_t L,
                                                  uint32_t h, int normalize, float* __restrict__ out_base,
                                                  const uint32_t* __restrict__ out_rows, uint64_t out_stride) {
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  float* out = out_base + (uint64_t)(out_rows ? out_rows[b] : b) * out_stride;
  float sm = 0.0f;
  for (uint32_t t = 0; t < L; ++t) sm += mask[(uint64_t)b * L + t];
  if (sm < 1e-9f) sm = 1e-9f;
  float ss = 0.0f;
  for (uint32_t j = lane; j < h; j += 64) {
    float s = 0.0f;
    for (uint32_t t = 0; t < L; ++t) s += hid[((uint64_t)b * L + t) * h + j] * mask[(uint64_t)b * L + t];
    s = s / sm;
    out[j] = s;
    ss += s * s;
  }
  if (normalize) {
    float norm = sqrtf(wave_sum(ss));
    if (norm < 1e-12f) norm = 1e-12f;
    for (uint32_t j = lane; j < h; j += 64) out[j] = out[j] / norm;
  }
}

// the buffers one model pass works in: the encoder's own, or its side set
struct EncWs {
  float *x, *x1, *t, *qkv, *ctx, *inter;
  void *x16, *x1_16;
  float* d_mask;
  int64_t *d_ids, *d_tt;
  uint32_t* d_flag;
};
EncWs main_ws(const isl_encoder* e) {
  return EncWs{e->x, e->x1, e->t, e->qkv, e->ctx, e->inter, e->x16, e->x1_16, e->d_mask, e->d_ids, e->d_tt, e->d_flag};
}
EncWs side_ws(const isl_encoder* e) {
  const auto& s = e->side;
  return EncWs{s.x, s.x1, s.t, s.qkv, s.ctx, s.inter, s.x16, s.x1_16, s.d_mask, s.d_ids, s.d_tt, s.d_flag};
}

void free_side(isl_encoder* e) {
  auto& s = e->side;
  void* olds[] = {s.x, s.x1, s.t, s.qkv, s.ctx, s.inter, s.d_mask, s.d_ids, s.d_tt, s.x16, s.x1_16};
  for (void* p : olds)
    if (p) (void)hipFree(p);
  s.x = s.x1 = s.t = s.qkv = s.ctx = s.inter = s.d_mask = nullptr;
  s.x16 = s.x1_16 = nullptr;
  s.d_ids = s.d_tt = nullptr;
  s.ws_tokens = 0;
}

// the side workspace for B sequences of padded length L, its stream and its two events
isl_status ensure_side(isl_encoder* e, uint64_t B, uint64_t L) {
  auto& s = e->side;
  if (!s.stream) {
    hipStream_t st = nullptr;
    ISL_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    s.stream = st;
  }
  if (!s.ev_in) { hipEvent_t ev = nullptr; ISL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); s.ev_in = ev; }
  if (!s.ev_out) { hipEvent_t ev = nullptr; ISL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); s.ev_out = ev; }
  if (!s.d_flag && hipMalloc(&s.d_flag, 4) != hipSuccess) return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
  const uint64_t tokens = B * L;
  if (tokens <= s.ws_tokens) return ISL_OK;
  free_side(e);
  const uint64_t h = e->cfg.hidden, I = e->cfg.intermediate;
  if (hipMalloc(&s.x, tokens * h * 4) != hipSuccess || hipMalloc(&s.x1, tokens * h * 4) != hipSuccess ||
      hipMalloc(&s.t, tokens * h * 4) != hipSuccess || hipMalloc(&s.qkv, tokens * 3 * h * 4) != hipSuccess ||
      hipMalloc(&s.ctx, tokens * h * 4) != hipSuccess || hipMalloc(&s.inter, tokens * I * 4) != hipSuccess ||
      hipMalloc(&s.d_mask, tokens * 4) != hipSuccess || hipMalloc(&s.d_ids, tokens * 8) != hipSuccess ||
      hipMalloc(&s.d_tt, tokens * 8) != hipSuccess || hipMalloc(&s.x16, tokens * h * 2) != hipSuccess ||
      hipMalloc(&s.x1_16, tokens * h * 2) != hipSuccess) {
    free_side(e);
    return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed for the encoder's side workspace (%llu tokens)",
                     (unsigned long long)tokens);
  }
  s.ws_tokens = tokens;
  return ISL_OK;
}

isl_status ensure_ws(isl_encoder* e, uint64_t B, uint64_t L) {
  const uint64_t tokens = B * L;
  if (tokens <= e->ws_tokens) return ISL_OK;
  void* olds[] = {e->x, e->x1, e->t, e->qkv, e->ctx, e->inter, e->d_mask, e->d_ids, e->d_tt, e->x16, e->x1_16};
  for (void* p : olds)
    if (p) (void)hipFree(p);
  e->x = e->x1 = e->t = e->qkv = e->ctx = e->inter = e->d_mask = nullptr;
  e->x16 = e->x1_16 = nullptr;
  e->d_ids = e->d_tt = nullptr;
  e->ws_tokens = 0;
  const uint64_t h = e->cfg.hidden, I = e->cfg.intermediate;
  if (hipMalloc(&e->x, tokens * h * 4) != hipSuccess || hipMalloc(&e->x1, tokens * h * 4) != hipSuccess ||
      hipMalloc(&e->t, tokens * h * 4) != hipSuccess || hipMalloc(&e->qkv, tokens * 3 * h * 4) != hipSuccess ||
      hipMalloc(&e->ctx, tokens * h * 4) != hipSuccess || hipMalloc(&e->inter, tokens * I * 4) != hipSuccess ||
      hipMalloc(&e->d_mask, tokens * 4) != hipSuccess || hipMalloc(&e->d_ids, tokens * 8) != hipSuccess ||
      hipMalloc(&e->d_tt, tokens * 8) != hipSuccess || hipMalloc(&e->x16, tokens * h * 2) != hipSuccess ||
      hipMalloc(&e->x1_16, tokens * h * 2) != hipSuccess)
    return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed for the encoder workspace (%llu tokens)",
                     (unsigned long long)tokens);
  e->ws_tokens = tokens;
  return ISL_OK;
}

__global__ void fill_f32(float* p, uint64_t n, float v) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// The model on B sequences of padded length L whose ids / types / mask already sit in the
// workspace (w.d_ids, w.d_tt when has_tt, w.d_mask); the last hidden state is left in w.x.
isl_status compute_forward(isl_encoder* enc, const EncWs& w, bool has_tt, uint64_t B, uint64_t L, hipStream_t st) {
  const EncWs* e = &w;  // (the body below reads its buffers through `e`)
  const isl_bert_config& c = enc->cfg;
  const uint64_t M = B * L, h = c.hidden, I = c.intermediate;
  ISL_HIP(hipMemsetAsync(e->d_flag, 0, 4, st));
  const uint32_t dh = c.hidden / c.heads;
  static const bool valu_attention = getenv("ISL_ATTENTION_VALU") != nullptr;
  // bf16 mode: every GEMM input is kept as a bf16 copy written by its producer (LayerNorm,
  // attention, the GELU epilogue); ctx and inter only exist in bf16 then (in their f32 buffers)
  const bool half = enc->precision == ISL_DTYPE_BF16 && !enc->layers16.empty() && h % 8 == 0 && I % 8 == 0 &&
                    (dh == 64 || dh == 32) && !valu_attention;
  __bf16* x16 = half ? reinterpret_cast<__bf16*>(e->x16) : nullptr;
  __bf16* x1_16 = half ? reinterpret_cast<__bf16*>(e->x1_16) : nullptr;
  __bf16* ctx16 = half ? reinterpret_cast<__bf16*>(e->ctx) : nullptr;
  hipLaunchKernelGGL(embed_ln_kernel, dim3((uint32_t)M), dim3(64), h * 4, st, e->d_ids,
                     has_tt ? e->d_tt : nullptr, (uint32_t)L, (uint32_t)h, c.vocab_size, c.type_vocab,
                     enc->word, enc->pos, enc->type, enc->eln_w, enc->eln_b, c.layer_norm_eps, e->x, e->d_flag, x16);
  for (size_t li = 0; li < enc->layers.size(); ++li) {
    const auto& ly = enc->layers[li];
    if (half) launch_gemm_bf16<0, false, true, false>(x16, (const __bf16*)enc->layers16[li].wqkv, ly.bqkv, nullptr, e->qkv, M, 3 * h, h, st);
    else launch_gemm<0, false>(e->x, ly.wqkv, ly.bqkv, nullptr, e->qkv, M, 3 * h, h, st);
    dim3 ag((uint32_t)(B * c.heads), (uint32_t)((L + 63) / 64));
    if (dh == 64 && !valu_attention) hipLaunchKernelGGL(attention_mfma_kernel<64>, ag, dim3(64), 0, st, e->qkv, e->d_mask, (uint32_t)L, c.heads, e->ctx, ctx16);
    else if (dh == 32 && !valu_attention) hipLaunchKernelGGL(attention_mfma_kernel<32>, ag, dim3(64), 0, st, e->qkv, e->d_mask, (uint32_t)L, c.heads, e->ctx, ctx16);
    else if (dh == 64) hipLaunchKernelGGL(attention_kernel<64>, ag, dim3(64), 0, st, e->qkv, e->d_mask, (uint32_t)L, c.heads, e->ctx);
    else if (dh == 32) hipLaunchKernelGGL(attention_kernel<32>, ag, dim3(64), 0, st, e->qkv, e->d_mask, (uint32_t)L, c.heads, e->ctx);
    else hipLaunchKernelGGL(attention_kernel<16>, ag, dim3(64), 0, st, e->qkv, e->d_mask, (uint32_t)L, c.heads, e->ctx);
    if (half) launch_gemm_bf16<0, true, true, false>(ctx16, (const __bf16*)enc->layers16[li].wo, ly.bo, e->x, e->t, M, h, h, st);
    else launch_gemm<0, true>(e->ctx, ly.wo, ly.bo, e->x, e->t, M, h, h, st);
    hipLaunchKernelGGL(ln_kernel, dim3((uint32_t)M), dim3(64), h * 4, st, e->t, (uint32_t)h, ly.ln1w, ly.ln1b, c.layer_norm_eps, e->x1, x1_16);
    if (half) {
      if (c.gelu_tanh) launch_gemm_bf16<2, false, true, true>(x1_16, (const __bf16*)enc->layers16[li].wi, ly.bi, nullptr, e->inter, M, I, h, st);
      else launch_gemm_bf16<1, false, true, true>(x1_16, (const __bf16*)enc->layers16[li].wi, ly.bi, nullptr, e->inter, M, I, h, st);
      launch_gemm_bf16<0, true, true, false>(e->inter, (const __bf16*)enc->layers16[li].wo2, ly.bo2, e->x1, e->t, M, h, I, st);
    } else {
      if (c.gelu_tanh) launch_gemm<2, false>(e->x1, ly.wi, ly.bi, nullptr, e->inter, M, I, h, st);
      else launch_gemm<1, false>(e->x1, ly.wi, ly.bi, nullptr, e->inter, M, I, h, st);
      launch_gemm<0, true>(e->inter, ly.wo2, ly.bo2, e->x1, e->t, M, h, I, st);
    }
    hipLaunchKernelGGL(ln_kernel, dim3((uint32_t)M), dim3(64), h * 4, st, e->t, (uint32_t)h, ly.ln2w, ly.ln2b, c.layer_norm_eps, e->x, x16);
  }
  ISL_HIP(hipGetLastError());
  return ISL_OK;
}

// Uploads the inputs of B sequences of padded length L and runs the model.
isl_status run_forward(isl_encoder* e, const int64_t* ids, const int64_t* tt, const float* mask,
                       uint64_t B, uint64_t L, int32_t mem, hipStream_t st) {
  if (L > e->cfg.max_position)
    return isl::fail(ISL_ERR_EMBEDDING, "Embedding error: sequence length %llu exceeds max_position %u",
                     (unsigned long long)L, e->cfg.max_position);
  ISL_TRY(ensure_ws(e, B, L));
  const uint64_t M = B * L;
  hipMemcpyKind kind = mem == ISL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  ISL_HIP(hipMemcpyAsync(e->d_ids, ids, M * 8, kind, st));
  if (tt) ISL_HIP(hipMemcpyAsync(e->d_tt, tt, M * 8, kind, st));
  if (mask) ISL_HIP(hipMemcpyAsync(e->d_mask, mask, M * 4, kind, st));
  else hipLaunchKernelGGL(fill_f32, dim3((uint32_t)((M + 255) / 256)), dim3(256), 0, st, e->d_mask, M, 1.0f);
  return compute_forward(e, main_ws(e), tt != nullptr, B, L, st);
}

// token table row -> the int64 ids / f32 mask the model kernels read
__global__ void gather_tokens_kernel(const uint16_t* __restrict__ tokens, const uint16_t* __restrict__ lens,
                                     uint32_t L, const uint32_t* __restrict__ node_ids, uint64_t n,
                                     int64_t* __restrict__ ids, float* __restrict__ mask) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * L) return;
  const uint64_t b = i / L;
  const uint32_t t = (uint32_t)(i % L);
  const uint32_t node = node_ids[b];
  const uint32_t len = lens ? lens[node] : L;
  const bool live = t < len;
  ids[i] = live ? (int64_t)tokens[(uint64_t)node * L + t] : 0;  // padding id 0, candle_provider.rs:398
  mask[i] = live ? 1.0f : 0.0f;
}

isl_status check_ids_flag(isl_encoder* e, hipStream_t st, bool side_too = false) {
  uint32_t flag = 0, flag2 = 0;
  ISL_HIP(hipMemcpyAsync(&flag, e->d_flag, 4, hipMemcpyDeviceToHost, st));
  if (side_too) ISL_HIP(hipMemcpyAsync(&flag2, e->side.d_flag, 4, hipMemcpyDeviceToHost, st));
  ISL_HIP(hipStreamSynchronize(st));
  if (flag | flag2) return isl::fail(ISL_ERR_EMBEDDING, "Embedding error: token or token-type id out of range");
  return ISL_OK;
}

}  // namespace

namespace isl {
// Batch sizes the model passes of encoder_embed_nodes run at full waves of GEMM tiles: *chunk = the
// sequences of one pass, *quantum = the largest batch whose narrowest Linear (hidden -> hidden, 256 x 256
// tiles, one workgroup per CU) still fits one wave of tiles over the chip.  (h = 768, L = 64, 256 CUs:
// 85 row tiles x 3 column tiles = 255 tiles -> 340 sequences; two such batches fill two waves, and so on.)
void encoder_batch_quantum(const isl_encoder* e, uint32_t L, uint32_t* quantum, uint32_t* chunk) {
  *quantum = 0;
  *chunk = 2048;
  if (!e || !L) return;
  const uint32_t ncu = (uint32_t)device_cu_count(e->device);
  const uint32_t col_tiles = (e->cfg.hidden + 255) / 256;
  const uint32_t row_tiles = ncu / std::max<uint32_t>(col_tiles, 1);
  const uint64_t q = (uint64_t)row_tiles * 256 / L;
  if (q >= 16 && q <= *chunk) *quantum = (uint32_t)q;
}

isl_status encoder_embed_nodes(isl_encoder* e, const uint16_t* d_tokens, const uint16_t* d_lens,
                               uint32_t L, const uint32_t* d_node_ids, uint64_t n, int normalize,
                               float* d_rows, uint64_t stride, hipStream_t st, const uint32_t* d_out_rows) {
  if (n == 0) return ISL_OK;
  if (!d_out_rows) d_out_rows = d_node_ids;
  if (L == 0 || L > e->cfg.max_position)
    return isl::fail(ISL_ERR_EMBEDDING, "Embedding error: token rows of %u slots do not fit max_position %u",
                     L, e->cfg.max_position);
  std::lock_guard<std::mutex> lock(e->mu);
  const uint64_t chunk = 2048;  // sequences per model pass: (2048 x 64 tokens) x 3072 floats = 1.6 GB
  // Two halves side by side (round 4, opt-in: ISL_ENCODER_SPLIT=<sequences from which a pass is split>).  One pass
  // over B sequences runs its GEMMs as waves of 256 x 256 tiles, one workgroup per CU: 860 sequences x 64 tokens are
  // 645 tiles of a hidden x hidden Linear = 2.52 waves, and the third wave leaves half the chip idle; LayerNorm,
  // attention and the embedding gather (8 % of the time) leave the matrix cores idle altogether.  Here the halves
  // of the batch go through the model on two streams with a workspace each: a CU that one half's GEMM has no tile
  // left for takes the other half's workgroups, one half's memory-bound kernels lie beside the other's GEMMs.
  // Same embeddings, bit for bit (every row is reduced in a fixed order that does not depend on the batch).
  // MEASURED (config 3 at 1M nodes, 256 queries, ~500 misses per round; profiles/r04_recompute_1m_split_ab.jsonl,
  // all four settings interleaved in one process): whole passes 34.4 queries/s, halves side by side 35.4, whole
  // passes over batches cut to whole tile waves (assign_slots_kernel's quantum) 35.6, both 35.4 -- the two remedies
  // recover the same idle wave and do not add up; the quantum needs no second workspace, so it is the default and
  // this stays a switch (read per call).
  const char* se = getenv("ISL_ENCODER_SPLIT");
  const uint64_t split_min = se ? (uint64_t)std::max(0, atoi(se)) : 0;  // 0 = never
  auto half = [&](const EncWs& w, uint64_t o, uint64_t B, hipStream_t s) -> isl_status {
    hipLaunchKernelGGL(gather_tokens_kernel, dim3((uint32_t)((B * L + 255) / 256)), dim3(256), 0, s, d_tokens,
                       d_lens, L, d_node_ids + o, B, w.d_ids, w.d_mask);
    ISL_TRY(compute_forward(e, w, false, B, L, s));
    hipLaunchKernelGGL(pool_kernel, dim3((uint32_t)B), dim3(64), 0, s, w.x, w.d_mask, L,
                       (uint32_t)e->cfg.hidden, normalize, d_rows, d_out_rows + o, stride);
    ISL_HIP(hipGetLastError());
    return ISL_OK;
  };
  for (uint64_t o = 0; o < n; o += chunk) {
    const uint64_t B = std::min(chunk, n - o);
    if (split_min && B >= split_min && B >= 8) {
      const uint64_t B0 = ((B / 2 + 3) / 4) * 4, B1 = B - B0;  // (whole 256-row tiles at 64 tokens per sequence)
      ISL_TRY(ensure_ws(e, std::min(n, chunk), L));
      ISL_TRY(ensure_side(e, (std::min(n, chunk) + 1) / 2, L));
      hipStream_t side = (hipStream_t)e->side.stream;
      ISL_HIP(hipEventRecord((hipEvent_t)e->side.ev_in, st));     // what the caller enqueued so far (the node lists)
      ISL_HIP(hipStreamWaitEvent(side, (hipEvent_t)e->side.ev_in, 0));
      isl_status rc = half(side_ws(e), o + B0, B1, side);
      if (rc == ISL_OK) rc = half(main_ws(e), o, B0, st);
      (void)hipEventRecord((hipEvent_t)e->side.ev_out, side);     // joined whatever happened: nothing stays behind on the side stream
      (void)hipStreamWaitEvent(st, (hipEvent_t)e->side.ev_out, 0);
      if (rc != ISL_OK) { (void)hipStreamSynchronize(st); return rc; }
      ISL_TRY(check_ids_flag(e, st, true));
    } else {
      ISL_TRY(ensure_ws(e, std::min(n, chunk), L));
      ISL_TRY(half(main_ws(e), o, B, st));
      ISL_TRY(check_ids_flag(e, st));
    }
  }
  return ISL_OK;
}
}  // namespace isl

extern "C" {

void isl_encoder_free(isl_encoder* e) {
  if (!e) return;
  if (e->device >= 0) {
    (void)hipSetDevice(e->device);
    for (void* p : e->owned) (void)hipFree(p);
    void* ws[] = {e->x, e->x1, e->t, e->qkv, e->ctx, e->inter, e->d_mask, e->d_ids, e->d_tt, e->d_flag, e->x16, e->x1_16};
    for (void* p : ws)
      if (p) (void)hipFree(p);
    free_side(e);
    if (e->side.d_flag) (void)hipFree(e->side.d_flag);
    if (e->side.ev_in) (void)hipEventDestroy((hipEvent_t)e->side.ev_in);
    if (e->side.ev_out) (void)hipEventDestroy((hipEvent_t)e->side.ev_out);
    if (e->side.stream) (void)hipStreamDestroy((hipStream_t)e->side.stream);
  }
  delete e;
}

isl_status isl_encoder_new(const isl_bert_config* cfg, int32_t device, isl_encoder** out) {
  if (!cfg || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  const isl_bert_config& c = *cfg;
  if (!c.vocab_size || !c.hidden || !c.layers || !c.heads || !c.intermediate || !c.max_position || !c.type_vocab)
    return isl::fail(ISL_ERR_INVALID_CONFIG, "Invalid configuration: zero-sized encoder dimension");
  if (c.hidden % c.heads)
    return isl::fail(ISL_ERR_INVALID_CONFIG, "Invalid configuration: hidden must be a multiple of heads");
  const uint32_t dh = c.hidden / c.heads;
  if (dh != 16 && dh != 32 && dh != 64)
    return isl::fail(ISL_ERR_UNSUPPORTED, "head size %u is not built (16, 32 and 64 are)", dh);
  if (c.hidden % 4 || c.intermediate % 4)
    return isl::fail(ISL_ERR_UNSUPPORTED, "hidden and intermediate sizes must be multiples of 4");
  ISL_TRY(isl::use_device(device));
  isl_encoder* e = new isl_encoder();
  e->cfg = c;
  e->device = device;
  bool ok = true;
  auto alloc = [&](uint64_t n) -> float* {
    void* p = nullptr;
    if (hipMalloc(&p, n * 4) != hipSuccess || hipMemset(p, 0, n * 4) != hipSuccess) { ok = false; return nullptr; }
    e->owned.push_back(p);
    return (float*)p;
  };
  const uint64_t h = c.hidden, I = c.intermediate;
  e->word = alloc((uint64_t)c.vocab_size * h);
  e->pos = alloc((uint64_t)c.max_position * h);
  e->type = alloc((uint64_t)c.type_vocab * h);
  e->eln_w = alloc(h);
  e->eln_b = alloc(h);
  e->layers.resize(c.layers);
  for (auto& ly : e->layers) {
    ly.wqkv = alloc(3 * h * h); ly.bqkv = alloc(3 * h);
    ly.wo = alloc(h * h); ly.bo = alloc(h);
    ly.ln1w = alloc(h); ly.ln1b = alloc(h);
    ly.wi = alloc(I * h); ly.bi = alloc(I);
    ly.wo2 = alloc(h * I); ly.bo2 = alloc(h);
    ly.ln2w = alloc(h); ly.ln2b = alloc(h);
  }
  if (ok && hipMalloc(&e->d_flag, 4) != hipSuccess) ok = false;
  if (!ok) {
    isl_encoder_free(e);
    return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed for the encoder weights");
  }
  *out = e;
  return ISL_OK;
}

// Tensor names as in the HuggingFace checkpoint candle's VarBuilder reads
// (candle_provider.rs:267-284), with or without the "bert." prefix.
isl_status isl_encoder_set_weight(isl_encoder* e, const char* name, const float* data, uint64_t count,
                                  int32_t mem) {
  if (!e || !name || !data) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  ISL_TRY(isl::use_device(e->device));
  std::string n(name);
  if (n.rfind("bert.", 0) == 0) n = n.substr(5);
  const uint64_t h = e->cfg.hidden, I = e->cfg.intermediate;
  float* dst = nullptr;
  uint64_t want = 0;
  if (n == "embeddings.word_embeddings.weight") { dst = e->word; want = (uint64_t)e->cfg.vocab_size * h; }
  else if (n == "embeddings.position_embeddings.weight") { dst = e->pos; want = (uint64_t)e->cfg.max_position * h; }
  else if (n == "embeddings.token_type_embeddings.weight") { dst = e->type; want = (uint64_t)e->cfg.type_vocab * h; }
  else if (n == "embeddings.LayerNorm.weight") { dst = e->eln_w; want = h; }
  else if (n == "embeddings.LayerNorm.bias") { dst = e->eln_b; want = h; }
  else if (n.rfind("encoder.layer.", 0) == 0) {
    size_t dot = n.find('.', 14);
    if (dot == std::string::npos) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "unknown tensor %s", name);
    uint64_t li = strtoull(n.substr(14, dot - 14).c_str(), nullptr, 10);
    if (li >= e->layers.size()) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "layer index out of range in %s", name);
    auto& ly = e->layers[li];
    std::string r = n.substr(dot + 1);
    if (r == "attention.self.query.weight") { dst = ly.wqkv; want = h * h; }
    else if (r == "attention.self.key.weight") { dst = ly.wqkv + h * h; want = h * h; }
    else if (r == "attention.self.value.weight") { dst = ly.wqkv + 2 * h * h; want = h * h; }
    else if (r == "attention.self.query.bias") { dst = ly.bqkv; want = h; }
    else if (r == "attention.self.key.bias") { dst = ly.bqkv + h; want = h; }
    else if (r == "attention.self.value.bias") { dst = ly.bqkv + 2 * h; want = h; }
    else if (r == "attention.output.dense.weight") { dst = ly.wo; want = h * h; }
    else if (r == "attention.output.dense.bias") { dst = ly.bo; want = h; }
    else if (r == "attention.output.LayerNorm.weight") { dst = ly.ln1w; want = h; }
    else if (r == "attention.output.LayerNorm.bias") { dst = ly.ln1b; want = h; }
    else if (r == "intermediate.dense.weight") { dst = ly.wi; want = I * h; }
    else if (r == "intermediate.dense.bias") { dst = ly.bi; want = I; }
    else if (r == "output.dense.weight") { dst = ly.wo2; want = h * I; }
    else if (r == "output.dense.bias") { dst = ly.bo2; want = h; }
    else if (r == "output.LayerNorm.weight") { dst = ly.ln2w; want = h; }
    else if (r == "output.LayerNorm.bias") { dst = ly.ln2b; want = h; }
  }
  if (!dst) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "unknown tensor %s", name);
  if (count != want) {
    return isl::fail_dim(want, count);
  }
  ISL_HIP(hipMemcpy(dst, data, count * 4, mem == ISL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice));
  return ISL_OK;
}

// Optional reduced-precision mode of the Linear layers: ISL_DTYPE_BF16 rounds the weights (once,
// here) and the activations (per GEMM) to bf16 and accumulates in float32 on the bf16 matrix
// cores; ISL_DTYPE_F32 (default) is the reference's arithmetic.  Call after the weights are set.
isl_status isl_encoder_set_precision(isl_encoder* e, int32_t dtype) {
  if (!e) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "encoder is NULL");
  if (dtype != ISL_DTYPE_F32 && dtype != ISL_DTYPE_BF16) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "unknown dtype");
  ISL_TRY(isl::use_device(e->device));
  std::lock_guard<std::mutex> lock(e->mu);
  const uint64_t h = e->cfg.hidden, I = e->cfg.intermediate;
  if (dtype == ISL_DTYPE_BF16) {
    if (h % 8 || I % 8) return isl::fail(ISL_ERR_UNSUPPORTED, "bf16 mode needs hidden and intermediate sizes that are multiples of 8");
    if (e->layers16.empty()) e->layers16.resize(e->layers.size());
    for (size_t li = 0; li < e->layers.size(); ++li) {
      const float* src[4] = {e->layers[li].wqkv, e->layers[li].wo, e->layers[li].wi, e->layers[li].wo2};
      const uint64_t cnt[4] = {3 * h * h, h * h, I * h, h * I};
      void** dst[4] = {&e->layers16[li].wqkv, &e->layers16[li].wo, &e->layers16[li].wi, &e->layers16[li].wo2};
      for (int t = 0; t < 4; ++t) {
        if (!*dst[t]) {
          ISL_HIP(hipMalloc(dst[t], cnt[t] * 2));
          e->owned.push_back(*dst[t]);
        }
        hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((uint32_t)((cnt[t] + 255) / 256)), dim3(256), 0, 0, src[t],
                           (__bf16*)*dst[t], cnt[t]);
      }
    }
    ISL_HIP(hipGetLastError());
    ISL_HIP(hipDeviceSynchronize());
  }
  e->precision = dtype;
  return ISL_OK;
}

isl_status isl_encoder_forward(isl_encoder* e, const int64_t* input_ids, const int64_t* token_type_ids,
                               const float* attention_mask, uint64_t B, uint64_t L, float* out_hidden,
                               int32_t mem, void* stream) {
  if (!e || !input_ids || !out_hidden) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  if (B == 0 || L == 0) return ISL_OK;
  ISL_TRY(isl::use_device(e->device));
  std::lock_guard<std::mutex> lock(e->mu);
  hipStream_t st = (hipStream_t)stream;
  ISL_TRY(run_forward(e, input_ids, token_type_ids, attention_mask, B, L, mem, st));
  ISL_HIP(hipMemcpyAsync(out_hidden, e->x, B * L * e->cfg.hidden * 4,
                         mem == ISL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
  return check_ids_flag(e, st);
}

isl_status isl_encoder_embed(isl_encoder* e, const int64_t* input_ids, const int64_t* token_type_ids,
                             const float* attention_mask, uint64_t B, uint64_t L, int32_t normalize,
                             float* out, int32_t mem, void* stream) {
  if (!e || !input_ids || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  if (B == 0) return ISL_OK;  // embed_texts_raw(&[]) -> Ok(vec![]), candle_provider.rs:354-356
  if (L == 0) return isl::fail(ISL_ERR_EMBEDDING, "Embedding error: empty sequences");
  ISL_TRY(isl::use_device(e->device));
  std::lock_guard<std::mutex> lock(e->mu);
  hipStream_t st = (hipStream_t)stream;
  ISL_TRY(run_forward(e, input_ids, token_type_ids, attention_mask, B, L, mem, st));
  const uint64_t h = e->cfg.hidden;
  float* d_out = out;
  if (mem == ISL_MEM_HOST) d_out = e->t;  // B*h <= tokens*h
  hipLaunchKernelGGL(pool_kernel, dim3((uint32_t)B), dim3(64), 0, st, e->x, e->d_mask, (uint32_t)L, (uint32_t)h,
                     normalize, d_out, (const uint32_t*)nullptr, h);
  ISL_HIP(hipGetLastError());
  if (mem == ISL_MEM_HOST) ISL_HIP(hipMemcpyAsync(out, d_out, B * h * 4, hipMemcpyDeviceToHost, st));
  return check_ids_flag(e, st);
}

}  // extern "C"
