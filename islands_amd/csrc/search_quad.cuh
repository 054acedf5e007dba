// leann_search_quad: FOUR queries per 64-lane wave, 16 lanes ("quarter") each.
//
// Why: the exact-order distance chain of a row is strictly sequential, so one lane owns one
// row.  An expansion yields ~9 new rows on average, so a wave that serves one query keeps
// <= 16 of its 64 lanes busy while every chain instruction is issued for the whole wave; the
// single-query kernel is bound by VALU/LDS instruction issue (rocprofv3: ~316k VALU + 63k LDS
// instructions per query), not by HBM.  Packing four queries into one wave lets one chain
// instruction advance four queries' rows, which divides the issue cost per query by ~3-4.
//
// The four queries advance in lockstep through the same phases (select, adjacency, visited,
// row fetch, chains, insertion); every per-query quantity lives in the 16 lanes of its quarter
// (a DPP row), quarter-uniform values are replicated in VGPRs.  Semantics are exactly those of
// leann_search_fast (see search.hip and DESIGN.md section 3.2-3.3).
#pragma once

namespace {

constexpr int QW = 16;              // lanes per query
constexpr int QPIECE = 64;          // floats of a row staged per step (256 B, 4 rows per load)
constexpr int QTILE_LD = QPIECE + 4;
constexpr int QTILE_ROWS = 64;
constexpr uint32_t QNONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t qballot(bool p, int qbase) {
  return (uint32_t)(__ballot(p) >> qbase) & 0xFFFFu;
}
template <typename T>
__device__ __forceinline__ T qshfl(T v, uint32_t k, int qbase) {
  return __shfl(v, (int)(qbase | (k & 15)));
}
// lane l receives lane l-1 of its quarter (lane 0 keeps `v`): DPP row_shr:1
__device__ __forceinline__ uint32_t qshr1_u(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
}
// lane 0 of the quarter receives lane 15 (rotate right by one inside the row): DPP row_ror:1
__device__ __forceinline__ uint32_t qror1_u(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121 /* row_ror:1 */, 0xf, 0xf, false);
}

// Sorted result set of one quarter: entry e lives in slot e >> 4 of lane e & 15.
template <int NS>
struct QSet {
  float d[NS];
  uint32_t id[NS];
  uint32_t len;  // quarter-uniform (VGPR)

  __device__ void init() {
#pragma unroll
    for (int s = 0; s < NS; ++s) { d[s] = 0.0f; id[s] = 0u; }
    len = 0;
  }
  __device__ float dist_at(uint32_t e, int qbase) const {
    float v = d[0];
#pragma unroll
    for (int s = 1; s < NS; ++s) v = (e >> 4) == (uint32_t)s ? d[s] : v;
    return qshfl(v, e, qbase);
  }
  __device__ uint32_t id_at(uint32_t e, int qbase) const {
    uint32_t v = id[0];
#pragma unroll
    for (int s = 1; s < NS; ++s) v = (e >> 4) == (uint32_t)s ? id[s] : v;
    return qshfl(v, e, qbase);
  }
  __device__ uint32_t first_unexpanded(int l, int qbase) const {
    uint32_t e = QNONE;
#pragma unroll
    for (int s = NS - 1; s >= 0; --s) {
      uint32_t m = qballot((uint32_t)(s * QW + l) < len && !(id[s] & FLAG_EXP), qbase);
      if (m) e = s * QW + (uint32_t)__ffs((int)m) - 1;
    }
    return e;
  }
  __device__ void mark_expanded(uint32_t e, int l) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if ((e >> 4) == (uint32_t)s && (uint32_t)l == (e & 15)) id[s] |= FLAG_EXP;
  }
  // `doit` is quarter-uniform: quarters that do not insert keep their state
  __device__ void insert(bool doit, float nd, uint32_t nid, int l, int qbase) {
    const uint32_t nk = ordkey(nd);
    uint32_t pos = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      uint32_t ek = ordkey(d[s]);
      bool less = (uint32_t)(s * QW + l) < len && (ek < nk || (ek == nk && (id[s] & ID_MASK) < nid));
      pos += (uint32_t)__popc(qballot(less, qbase));
    }
#pragma unroll
    for (int s = NS - 1; s >= 0; --s) {
      uint32_t ud = qshr1_u(__float_as_uint(d[s]));
      uint32_t ui = qshr1_u(id[s]);
      if (s > 0) {
        uint32_t cd = qror1_u(__float_as_uint(d[s - 1]));
        uint32_t ci = qror1_u(id[s - 1]);
        if (l == 0) { ud = cd; ui = ci; }
      }
      uint32_t e = s * QW + l;
      if (doit) {
        if (e > pos) { d[s] = __uint_as_float(ud); id[s] = ui; }
        else if (e == pos) { d[s] = nd; id[s] = nid; }
      }
    }
    if (doit) len += 1;
  }
};

// Distances of up to 64 rows, 16 per quarter: lane (qt, l) owns row l of its quarter when
// l < cnt (quarter-uniform).  NIQ = ceil(max cnt / 4) is compile-time so that all loads and
// LDS stores are straight-line code (counted s_waitcnt vmcnt).  Load instruction j moves the
// 256-byte pieces of rows 4j..4j+3 of the wave-wide row list (row r = 16*quarter + index).
template <int METRIC, int NIQ>
__device__ __forceinline__ float quad_distances(const float* __restrict__ emb, uint64_t stride,
                                                uint32_t d, const uint32_t* rowids /* LDS [64] */,
                                                uint32_t cnt, const float* qrow /* LDS, per quarter */,
                                                float* tile, float q_norm, float row_aux) {
  const int lane = threadIdx.x;
  const int l = lane & 15;
  const int rsub = lane >> 4;       // which of the 4 rows of a load instruction
  const int col = (lane & 15) * 4;  // this lane's float4 inside the piece
  const uint32_t nT = (d + QPIECE - 1) / QPIECE;
#define ISL_FOR16(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
  // instruction j is needed iff (j & 3) < NIQ; rows that do not exist carry a valid dummy id
#define ISL_QDECL(j)                                                                  \
  const float* rp##j = emb + (uint64_t)rowids[4 * (j) + rsub] * stride + col;         \
  float4 ra##j = make_float4(0.f, 0.f, 0.f, 0.f), rb##j = ra##j, rc##j = ra##j;
  ISL_FOR16(ISL_QDECL)
#define ISL_QLOAD_A(j) if constexpr (((j) & 3) < NIQ) ra##j = *reinterpret_cast<const float4*>(rp##j + poff);
#define ISL_QLOAD_B(j) if constexpr (((j) & 3) < NIQ) rb##j = *reinterpret_cast<const float4*>(rp##j + poff);
#define ISL_QLOAD_C(j) if constexpr (((j) & 3) < NIQ) rc##j = *reinterpret_cast<const float4*>(rp##j + poff);
#define ISL_QSTORE_A(j) \
  if constexpr (((j) & 3) < NIQ) *reinterpret_cast<float4*>(tile + (4 * (j) + rsub) * QTILE_LD + col) = ra##j;
#define ISL_QSTORE_B(j) \
  if constexpr (((j) & 3) < NIQ) *reinterpret_cast<float4*>(tile + (4 * (j) + rsub) * QTILE_LD + col) = rb##j;
#define ISL_QSTORE_C(j) \
  if constexpr (((j) & 3) < NIQ) *reinterpret_cast<float4*>(tile + (4 * (j) + rsub) * QTILE_LD + col) = rc##j;
  float a0 = 0.0f, a1 = 0.0f;
  auto consume = [&](uint32_t t) {
    if ((uint32_t)l < cnt) {
      const float* trow = tile + lane * QTILE_LD;  // row index == lane index
      const float* qv = qrow + t * QPIECE;
      const uint32_t n = d - t * QPIECE;
      if (n >= (uint32_t)QPIECE) {
#pragma unroll 8
        for (int j = 0; j < QPIECE; j += 4) {
          float4 x = *reinterpret_cast<const float4*>(trow + j);
          float4 q = *reinterpret_cast<const float4*>(qv + j);
          dstep<METRIC>(q.x, x.x, a0, a1);
          dstep<METRIC>(q.y, x.y, a0, a1);
          dstep<METRIC>(q.z, x.z, a0, a1);
          dstep<METRIC>(q.w, x.w, a0, a1);
        }
      } else {
        for (uint32_t j = 0; j < n; ++j) dstep<METRIC>(qv[j], trow[j], a0, a1);
      }
    }
  };
  {
    const size_t poff = 0;
    ISL_FOR16(ISL_QLOAD_A)
  }
  if (nT > 1) {
    const size_t poff = QPIECE;
    ISL_FOR16(ISL_QLOAD_B)
  }
  if (nT > 2) {
    const size_t poff = 2 * QPIECE;
    ISL_FOR16(ISL_QLOAD_C)
  }
  for (uint32_t t = 0; t < nT; t += 3) {
    ISL_FOR16(ISL_QSTORE_A)
    __syncthreads();
    if (t + 3 < nT) {
      const size_t poff = (size_t)(t + 3) * QPIECE;
      ISL_FOR16(ISL_QLOAD_A)
    }
    consume(t);
    __syncthreads();
    if (t + 1 < nT) {
      ISL_FOR16(ISL_QSTORE_B)
      __syncthreads();
      if (t + 4 < nT) {
        const size_t poff = (size_t)(t + 4) * QPIECE;
        ISL_FOR16(ISL_QLOAD_B)
      }
      consume(t + 1);
      __syncthreads();
    }
    if (t + 2 < nT) {
      ISL_FOR16(ISL_QSTORE_C)
      __syncthreads();
      if (t + 5 < nT) {
        const size_t poff = (size_t)(t + 5) * QPIECE;
        ISL_FOR16(ISL_QLOAD_C)
      }
      consume(t + 2);
      __syncthreads();
    }
  }
#undef ISL_FOR16
#undef ISL_QDECL
#undef ISL_QLOAD_A
#undef ISL_QLOAD_B
#undef ISL_QLOAD_C
#undef ISL_QSTORE_A
#undef ISL_QSTORE_B
#undef ISL_QSTORE_C
  if (METRIC == METRIC_COSINE_PRE) a1 = row_aux;
  return dfinish<METRIC>(a0, a1, q_norm);
}

template <int NS, int METRIC_API>
__global__ __launch_bounds__(64) void leann_search_quad(SearchParams p) {
  constexpr int METRIC = METRIC_API == ISL_METRIC_COSINE ? METRIC_COSINE_PRE : METRIC_API;
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int l = lane & 15;
  const int qt = lane >> 4;
  const int qbase = lane & 48;
  const uint32_t hcap = 1u << p.hbits;
  const uint32_t hmask = hcap - 1;
  const uint32_t hlimit = hcap - hcap / 8;
  const uint32_t qstride = ((p.d + 3) / 4 * 4) + 16;  // +16 floats: quarters read different banks
  uint32_t* htab = reinterpret_cast<uint32_t*>(smem) + (size_t)qt * hcap;
  float* tile = reinterpret_cast<float*>(smem + (size_t)4 * hcap * 4);
  uint32_t* rowids = reinterpret_cast<uint32_t*>(tile + QTILE_ROWS * QTILE_LD);  // [64]
  uint32_t* ulist = rowids + 64 + (size_t)qt * 64;                               // [4][64]
  float* qs = reinterpret_cast<float*>(rowids + 64 + 256) + (size_t)qt * qstride;
  const uint32_t ocap = 1u << p.obits;
  const uint32_t omask = ocap - 1;
  const uint32_t olimit = ocap - ocap / 4;
  uint32_t* otab = p.otab + ((size_t)blockIdx.x * 4 + qt) * ocap;
  const uint32_t ef = p.ef;

  // ---- per-quarter state (quarter-uniform values replicated in the quarter's lanes)
  QSet<NS> rs;
  rs.init();
  uint32_t qi = QNONE;       // query of this quarter
  bool exhausted = false;    // the work queue has no more queries
  bool virt = false;         // first step of a query: the "row list" is the entry point alone
  uint32_t hcount = 0, ocount = 0;
  bool ovf = false;
  uint32_t status = QS_OK;
  uint64_t payload = 0;
  uint32_t cH = 0, cE = 0, cV = 0, cP = 0;
  uint32_t t_id = 0, tcount = 0;
  float q_norm = 0.0f;
  uint64_t t_start = 0;

  for (;;) {
    // ------------------------------------------------------------ (1) new queries
    bool need = qi == QNONE && !exhausted;
    if (__ballot(need)) {
      uint32_t t = 0;
      if (need && l == 0) t = atomicAdd(&p.ticket[0], 1u);
      t = qshfl(t, 0, qbase);
      if (need) {
        if (t >= p.nq) {
          exhausted = true;
        } else {
          qi = t;
          rs.init();
          hcount = ocount = 0;
          ovf = false;
          status = QS_OK;
          payload = 0;
          cH = cE = cV = cP = 0;
          tcount = 0;
          virt = true;
          t_start = __builtin_amdgcn_s_memrealtime();
        }
      }
      const bool fresh = need && qi != QNONE;
      // clear the quarter's visited table, stage its query, norm_a in reference order
      for (uint32_t i = l; i < hcap; i += QW)
        if (fresh) htab[i] = EMPTY;
      const float* qg = p.queries + (uint64_t)(fresh ? qi : 0) * p.d;
      for (uint32_t j = l; j < p.d; j += QW)
        if (fresh) qs[j] = qg[j];
      __syncthreads();
      if (METRIC == ISL_METRIC_COSINE || METRIC == METRIC_COSINE_PRE) {
        float na = 0.0f;
        for (uint32_t j = 0; j < p.d; ++j) {
          float x = qs[j];
          na += x * x;  // distance.rs:78
        }
        if (fresh) q_norm = na;
      }
      if (fresh) {
        if ((uint64_t)p.entry >= p.nvec) {  // provider.compute_embedding(entry), leann.rs:911
          status = QS_NODE_NOT_FOUND;
          payload = p.entry;
        } else if (l == 0) {
          htab[hslot(p.entry, p.hbits)] = p.entry;  // visited.insert(entry), leann.rs:914
        }
        hcount = 1;
      }
      __syncthreads();
    }
    if (!__ballot(qi != QNONE)) break;  // every quarter idle and the queue is empty

    // ------------------------------------------------------------ (2) candidates.pop()
    const bool running = qi != QNONE && status == QS_OK;
    uint32_t cid = 0;
    bool expand = false;  // this quarter expands `cid` in this step
    bool finish = qi != QNONE && status != QS_OK;
    if (running && !virt) {
      uint32_t e = rs.first_unexpanded(l, qbase);
      if (e != QNONE) {
        cid = rs.id_at(e, qbase) & ID_MASK;
        rs.mark_expanded(e, l);
        expand = true;
      } else if (tcount > 0) {
        // tie-evicted candidates (distance == worst): smallest id first, see search.hip
        uint32_t best = QNONE, bl = 0;
        for (uint32_t i = 0; i < 16; ++i) {
          uint32_t v = qshfl(t_id, i, qbase);
          if (i < tcount && v < best) { best = v; bl = i; }
        }
        uint32_t last = qshfl(t_id, tcount - 1, qbase);
        if ((uint32_t)l == bl) t_id = last;
        tcount -= 1;
        cid = best;
        expand = true;
      } else {
        finish = true;  // every remaining candidate is farther than the worst result
      }
    } else if (running && virt) {
      // other quarters run rs ops above; keep the wave-wide intrinsics balanced
      (void)rs.first_unexpanded(l, qbase);
    }

    // ------------------------------------------------------------ (3) adjacency + visited
    uint32_t nu = 0;  // unvisited ids of this step, in CSR order, in ulist[0..nu)
    uint32_t deg = 0;
    uint64_t o0 = 0;
    if (expand && (uint64_t)cid < p.num_nodes) {  // get_neighbors -> None otherwise
      o0 = p.off[cid];
      deg = (uint32_t)(p.off[cid + 1] - o0);
      cH += 1;
      cE += deg;
      if (deg > 64) { status = QS_REDO; payload = 1; deg = 0; expand = false; finish = true; }
    }
    if (virt && running) {
      if (l == 0) ulist[0] = p.entry;
      nu = 1;
    }
    if (!ovf && hcount + deg > hlimit) ovf = true;
    if (__ballot(deg > 0)) {
      // all (<= 64) neighbour ids of the quarter's row are requested at once
      uint32_t nbr[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) nbr[k] = (uint32_t)(k * QW + l) < deg ? p.adj[o0 + k * QW + l] : EMPTY;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t base = k * QW;
        if (!__ballot(base < deg)) break;
        const bool act = base + l < deg;
        const uint32_t nid = nbr[k];
        bool is_new = false;
        if (act) {
          uint32_t h = hslot(nid, p.hbits);
          if (!ovf) {
            for (;;) {
              uint32_t old = atomicCAS(&htab[h], EMPTY, nid);
              if (old == EMPTY) { is_new = true; break; }
              if (old == nid) break;
              h = (h + 1) & hmask;
            }
          } else {
            bool found = false;
            for (;;) {
              uint32_t cur = htab[h];
              if (cur == nid) { found = true; break; }
              if (cur == EMPTY) break;
              h = (h + 1) & hmask;
            }
            if (!found) {
              uint32_t g = hslot(nid, p.obits);
              for (;;) {
                uint32_t old = atomicCAS(&otab[g], EMPTY, nid);
                if (old == EMPTY) { is_new = true; break; }
                if (old == nid) break;
                g = (g + 1) & omask;
              }
            }
          }
        }
        const uint32_t nm = qballot(is_new, qbase);
        const uint32_t rank = (uint32_t)__popc(nm & ((1u << l) - 1u));
        if (is_new) ulist[nu + rank] = nid;
        const uint32_t add = (uint32_t)__popc(nm);
        nu += add;
        if (!ovf) hcount += add;
        else ocount += add;
      }
      if (ovf && ocount > olimit && status == QS_OK) { status = QS_REDO; payload = 2; nu = 0; finish = true; }
    }
    __syncthreads();

    // ------------------------------------------------------------ (4) pruning + provider
    uint32_t keep = 0;
    if (nu > 0 && status == QS_OK)
      keep = virt ? 1u : prune_keep(p.prune_ratio, p.prune_strategy, nu, rs.len, ef);
    {
      // compute_embeddings_batch: the first id without a row fails the query (leann.rs:947)
      uint32_t firstbad = QNONE;
#pragma unroll 1
      for (uint32_t base = 0; base < 64; base += QW) {
        if (!__ballot(base < keep)) break;
        const uint32_t v = base + l < keep ? ulist[base + l] : 0u;
        const uint32_t bm = qballot(base + l < keep && (uint64_t)v >= p.nvec, qbase);
        if (bm && firstbad == QNONE) firstbad = qshfl(v, (uint32_t)__ffs((int)bm) - 1, qbase);
      }
      if (firstbad != QNONE && !virt) {
        status = QS_NODE_NOT_FOUND;
        payload = firstbad;
        keep = 0;
        finish = true;
      }
    }
    cV += keep;

    // ------------------------------------------------------------ (5) rows, chains, insertion
#pragma unroll 1
    for (uint32_t g0 = 0; g0 < 64; g0 += QW) {
      const uint32_t cnt = keep > g0 ? (keep - g0 < (uint32_t)QW ? keep - g0 : (uint32_t)QW) : 0u;
      const uint64_t anym = __ballot(cnt > 0);
      if (!anym) break;
      const uint32_t uid = (uint32_t)l < cnt ? ulist[g0 + l] : 0u;
      rowids[lane] = uid;  // rows that do not exist re-read row 0 of the matrix
      float r_aux = 0.0f;
      if (METRIC == METRIC_COSINE_PRE && (uint32_t)l < cnt) r_aux = p.norm2[uid];
      __syncthreads();
      // widest quarter decides how many load instructions a piece needs
      uint32_t mx = cnt;
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, 16));
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, 32));
      mx = uni(mx);
      float nd;
      if (mx <= 4) nd = quad_distances<METRIC, 1>(p.emb, p.stride, p.d, rowids, cnt, qs, tile, q_norm, r_aux);
      else if (mx <= 8) nd = quad_distances<METRIC, 2>(p.emb, p.stride, p.d, rowids, cnt, qs, tile, q_norm, r_aux);
      else if (mx <= 12) nd = quad_distances<METRIC, 3>(p.emb, p.stride, p.d, rowids, cnt, qs, tile, q_norm, r_aux);
      else nd = quad_distances<METRIC, 4>(p.emb, p.stride, p.d, rowids, cnt, qs, tile, q_norm, r_aux);

      // leann.rs:953-970 in CSR order, one insertion per quarter and iteration
      uint32_t pending = cnt >= 16 ? 0xFFFFu : ((1u << cnt) - 1u);
      uint2* plog = p.plog + (size_t)(qi == QNONE ? 0 : qi) * p.plog_cap;
      while (__ballot(pending != 0)) {
        const bool full = rs.len >= ef;
        const float worst = rs.len ? rs.dist_at(rs.len - 1, qbase) : 0.0f;
        const bool pass = !full || rs.len == 0 || nd < worst;  // raw f32 `<`, leann.rs:959
        const uint32_t pm = qballot(pass, qbase) & pending;
        const bool has = pm != 0;
        const uint32_t r = has ? (uint32_t)__ffs((int)pm) - 1 : 0u;
        const float id_d = qshfl(nd, r, qbase);
        const uint32_t id_i = qshfl(uid, r, qbase);
        if (has && cP < p.plog_cap && l == 0) plog[cP] = make_uint2(__float_as_uint(id_d), id_i);
        const uint32_t old_raw = rs.id_at(ef - 1, qbase);  // the entry that leaves when full
        rs.insert(has, id_d, id_i, l, qbase);
        if (has && full) rs.len = ef;
        const float new_worst = rs.dist_at(ef - 1, qbase);
        if (has && full) {
          if (ordkey(worst) != ordkey(new_worst)) {
            tcount = 0;
          } else if (!(old_raw & FLAG_EXP)) {
            if (tcount >= 16) { status = QS_REDO; payload = 3; }
            else {
              if ((uint32_t)l == tcount) t_id = old_raw & ID_MASK;
              tcount += 1;
            }
          }
        }
        if (has) cP += 1;
        pending = has ? (pending & ~((2u << r) - 1u)) : 0u;
      }
      if (status != QS_OK && qi != QNONE) finish = true;
      __syncthreads();
    }
    virt = false;

    // ------------------------------------------------------------ (6) finished queries
    if (__ballot(finish)) {
      if (finish) {
        uint32_t outn = rs.len < p.k ? rs.len : p.k;
        if (status == QS_OK) {
          // equal distances inside the returned prefix: BinaryHeap array order decides
          const uint32_t chk = rs.len < p.k + 1 ? rs.len : p.k + 1;
          bool tie = false;
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            if ((uint32_t)(s * QW) < chk) {
              const uint32_t e = s * QW + l;
              float nxt = qshfl(rs.d[s], (uint32_t)(l + 1), qbase);
              if (s + 1 < NS) {
                float n0 = qshfl(rs.d[s + 1 < NS ? s + 1 : s], 0u, qbase);
                if (l == 15) nxt = n0;
              }
              if (e + 1 < chk && ordkey(rs.d[s]) == ordkey(nxt)) tie = true;
            }
          }
          if (qballot(tie, qbase)) {
            if (cP <= p.plog_cap) status = QS_REPLAY;
            else { status = QS_REDO; payload = 4; }
          }
        } else {
          // keep the wave-wide shuffles of the branch above balanced
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            (void)qshfl(rs.d[s], (uint32_t)(l + 1), qbase);
            if (s + 1 < NS) (void)qshfl(rs.d[s + 1 < NS ? s + 1 : s], 0u, qbase);
          }
          (void)qballot(false, qbase);
        }
        if (status == QS_OK) {
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            const uint32_t e = s * QW + l;
            if (e < outn) {
              p.out_ids[(uint64_t)qi * p.k + e] = (uint64_t)(rs.id[s] & ID_MASK);
              p.out_dist[(uint64_t)qi * p.k + e] = rs.d[s];
            }
          }
        }
        if (l == 0) {
          p.status[qi] = status;
          p.payload[qi] = status == QS_OK ? (__builtin_amdgcn_s_memrealtime() - t_start) : payload;
          p.out_count[qi] = status == QS_OK ? outn : 0u;
          p.ctr[qi * 4 + 0] = cH;
          p.ctr[qi * 4 + 1] = cE;
          p.ctr[qi * 4 + 2] = cV;
          p.ctr[qi * 4 + 3] = cP;
          if (status == QS_REDO) {
            p.redo[atomicAdd(&p.ticket[1], 1u)] = qi;
            atomicAdd(&p.ticket[8 + ((uint32_t)payload & 3u)], 1u);
          } else if (status == QS_REPLAY) {
            p.replay[atomicAdd(&p.ticket[3], 1u)] = qi;
          }
        }
        if (ovf) {
          for (uint32_t i = l; i < ocap; i += QW) otab[i] = EMPTY;
        }
        qi = QNONE;
      }
    }
  }
}

// Re-orders the tied prefixes flagged by leann_search_quad (one wave per query, lane 0 sifts).
__global__ __launch_bounds__(64) void leann_replay_order(SearchParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const uint32_t ef = p.ef;
  float* res_d = reinterpret_cast<float*>(smem);
  uint32_t* res_i = reinterpret_cast<uint32_t*>(res_d + (ef + 1));
  uint2* stage = reinterpret_cast<uint2*>(((uintptr_t)(res_i + (ef + 1)) + 15) & ~(uintptr_t)15);
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&p.ticket[4], 1u);
    t = uni(t);
    uint32_t nrep = *((volatile uint32_t*)&p.ticket[3]);
    if (t >= nrep) break;
    const uint32_t qi = p.replay[t];
    replay_result_order(p.plog + (size_t)qi * p.plog_cap, p.ctr[qi * 4 + 3], ef, p.k, qi, res_d, res_i,
                        stage, p.out_ids, p.out_dist, p.out_count);
    if (lane == 0) p.status[qi] = QS_OK;
    __syncthreads();
  }
}

}  // namespace
