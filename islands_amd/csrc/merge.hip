// Cross-index (= cross-shard) top-k merge: MultiIndexSearcher::search, src/core/search.rs:211-237.
// Per query: the per-list results are concatenated in list order, stable-sorted by score
// ascending (ties keep list order, i.e. the lower shard rank first) and truncated to top_k.
#include "common.hpp"

#include <algorithm>

namespace {

// One thread per query.  Every list is already ascending (each comes from a search), so a
// k-way merge that prefers the lowest list on equal scores IS the stable sort of the
// concatenation.  NaN scores make the reference panic (partial_cmp().unwrap(), search.rs:231):
// reported through *nan_flag.
// list l's ids / scores / counts start `lstride` BYTES after list l - 1's (0 = the dense
// [list][query][k] layout of isl_merge_topk; the packed shard records otherwise)
__global__ void merge_topk_kernel(uint32_t nlists, uint32_t nq, uint32_t k,
                                  const uint64_t* __restrict__ ids0,
                                  const float* __restrict__ scores0,
                                  const uint32_t* __restrict__ counts0, uint64_t lstride,
                                  const uint64_t* __restrict__ id_base, uint32_t top_k,
                                  uint64_t* __restrict__ out_ids, float* __restrict__ out_scores,
                                  uint32_t* __restrict__ out_src, uint32_t* __restrict__ out_count,
                                  uint32_t* __restrict__ flags, uint32_t service) {
  // service != 0: the product-level merge of src/indexer/service.rs:787-801 -- results whose id
  // has no file entry are dropped (`id_base` then holds files.len() per list), score =
  // 1.0 - distance, stable sort by score DESCENDING, truncate.
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const uint64_t istr = lstride ? lstride : (uint64_t)nq * k * 8, sstr = lstride ? lstride : (uint64_t)nq * k * 4,
                 cstr = lstride ? lstride : (uint64_t)nq * 4;
  auto ids_of = [&](uint32_t l) { return (const uint64_t*)((const char*)ids0 + l * istr) + (uint64_t)q * k; };
  auto sc_of = [&](uint32_t l) { return (const float*)((const char*)scores0 + l * sstr) + (uint64_t)q * k; };
  // a count of ISL_SHARD_POISON_COUNT marks a list whose producer failed the query (multi-GPU exchange:
  // a rank whose shard search failed still takes part in the collective, shard.hip): the list counts
  // as empty here and bit 2 of *flags tells every rank that the batch is not a complete answer
  auto raw_cnt_of = [&](uint32_t l) { return *((const uint32_t*)((const char*)counts0 + l * cstr) + q); };
  auto cnt_of = [&](uint32_t l) { const uint32_t c = raw_cnt_of(l); return c == ISL_SHARD_POISON_COUNT ? 0u : c; };
  constexpr uint32_t MAXL = 64;
  uint32_t pos[MAXL];
  uint32_t total = 0;
  bool sorted = true, has_nan = false, poisoned = false;
  for (uint32_t l = 0; l < nlists; ++l) {
    pos[l] = 0;
    if (raw_cnt_of(l) == ISL_SHARD_POISON_COUNT) poisoned = true;
    uint32_t c = cnt_of(l);
    if (c > k) c = k;
    total += c;
    const float* sc = sc_of(l);
    for (uint32_t i = 0; i < c; ++i) {
      if (sc[i] != sc[i]) has_nan = true;
      if (i && sc[i] < sc[i - 1]) sorted = false;
    }
  }
  if (has_nan && total > 1) atomicOr(flags, 1u);
  if (!sorted) atomicOr(flags, 2u);
  if (poisoned) atomicOr(flags, 4u);
  uint32_t n = 0;
  while (n < top_k) {
    int best = -1;
    float bs = 0.0f;
    for (uint32_t l = 0; l < nlists; ++l) {
      uint32_t c = cnt_of(l);
      if (c > k) c = k;
      if (service && id_base)  // stored.files.get(id) == None -> skipped, service.rs:788
        while (pos[l] < c && ids_of(l)[pos[l]] >= id_base[l]) pos[l]++;
      if (pos[l] >= c) continue;
      float s = sc_of(l)[pos[l]];
      if (service) s = 1.0f - s;  // service.rs:791
      if (best < 0 || (service ? s > bs : s < bs)) {  // strict: equal scores keep the earlier list
        best = (int)l;
        bs = s;
      }
    }
    if (best < 0) break;
    out_ids[(uint64_t)q * top_k + n] = ids_of((uint32_t)best)[pos[best]] + ((id_base && !service) ? id_base[best] : 0ull);
    out_scores[(uint64_t)q * top_k + n] = bs;
    if (out_src) out_src[(uint64_t)q * top_k + n] = (uint32_t)best;
    pos[best]++;
    n++;
  }
  out_count[q] = n;
}

}  // namespace

static isl_status merge_lists(uint32_t service, uint64_t nlists, uint64_t nq, uint64_t k,
                              const uint64_t* ids, const float* scores, const uint32_t* counts,
                              const uint64_t* id_base, uint64_t top_k, uint64_t* out_ids,
                              float* out_scores, uint32_t* out_src, uint32_t* out_count,
                              int32_t mem, int32_t device, void* stream) {
  if (nq == 0) return ISL_OK;
  if (nlists == 0 || nlists > 64)
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "nlists must be in [1, 64]");
  if (!ids || !scores || !counts || !out_count || (top_k && (!out_ids || !out_scores)))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  ISL_TRY(isl::use_device(device));
  hipStream_t st = (hipStream_t)stream;
  const uint64_t nin = nlists * nq * k, nout = nq * top_k;
  const uint64_t *d_ids = ids, *d_base = id_base;
  const float* d_sc = scores;
  const uint32_t* d_cnt = counts;
  uint64_t* d_oi = out_ids;
  float* d_os = out_scores;
  uint32_t *d_osrc = out_src, *d_oc = out_count, *d_flags = nullptr;
  std::vector<void*> owned;
  auto cleanup = [&]() { for (void* p : owned) (void)hipFree(p); };
  auto dalloc = [&](size_t bytes) -> void* {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return nullptr;
    owned.push_back(p);
    return p;
  };
  hipError_t e = hipSuccess;
  d_flags = (uint32_t*)dalloc(4);
  if (!d_flags) { cleanup(); return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed"); }
  e = hipMemsetAsync(d_flags, 0, 4, st);
  void* tmp_base = nullptr;
  if (id_base) {  // id_base is always a host array (one entry per list)
    tmp_base = dalloc(nlists * 8);
    if (tmp_base && e == hipSuccess)
      e = hipMemcpyAsync(tmp_base, id_base, nlists * 8, hipMemcpyHostToDevice, st);
    d_base = (const uint64_t*)tmp_base;
  }
  if (mem == ISL_MEM_HOST) {
    void* a = dalloc(nin * 8); void* b = dalloc(nin * 4); void* c = dalloc(nlists * nq * 4);
    void* oi = dalloc(nout * 8); void* os = dalloc(nout * 4); void* osrc = dalloc(nout * 4);
    void* oc = dalloc(nq * 4);
    if (!a || !b || !c || !oi || !os || !osrc || !oc) {
      cleanup();
      return isl::fail(ISL_ERR_DEVICE, "hipMalloc failed");
    }
    if (e == hipSuccess) e = hipMemcpyAsync(a, ids, nin * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(b, scores, nin * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(c, counts, nlists * nq * 4, hipMemcpyHostToDevice, st);
    d_ids = (const uint64_t*)a; d_sc = (const float*)b; d_cnt = (const uint32_t*)c;
    d_oi = (uint64_t*)oi; d_os = (float*)os; d_osrc = (uint32_t*)osrc; d_oc = (uint32_t*)oc;
  }
  if (e == hipSuccess) {
    uint32_t blocks = (uint32_t)((nq + 63) / 64);
    hipLaunchKernelGGL(merge_topk_kernel, dim3(blocks), dim3(64), 0, st, (uint32_t)nlists,
                       (uint32_t)nq, (uint32_t)k, d_ids, d_sc, d_cnt, (uint64_t)0, d_base, (uint32_t)top_k, d_oi,
                       d_os, d_osrc, d_oc, d_flags, service);
    e = hipGetLastError();
  }
  uint32_t flags = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&flags, d_flags, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && mem == ISL_MEM_HOST) {
    if (top_k) {
      e = hipMemcpyAsync(out_ids, d_oi, nout * 8, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(out_scores, d_os, nout * 4, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess && out_src)
        e = hipMemcpyAsync(out_src, d_osrc, nout * 4, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_count, d_oc, nq * 4, hipMemcpyDeviceToHost, st);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  cleanup();
  if (e != hipSuccess)
    return isl::fail(ISL_ERR_DEVICE, "merge failed: %s", hipGetErrorString(e));
  if (flags & 1u)
    return isl::fail(ISL_ERR_SEARCH, service
                         ? "Search error: NaN distance in the service merge (order undefined in the reference)"
                         : "Search error: NaN score in merge (the reference panics here)");
  if (flags & 2u)
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "per-list scores must be ascending");
  if (flags & 4u)
    return isl::fail(ISL_ERR_SEARCH, "Search error: a list carries the failed-producer mark (count 0xFFFFFFFF)");
  return ISL_OK;
}

extern "C" isl_status isl_merge_topk(uint64_t nlists, uint64_t nq, uint64_t k, const uint64_t* ids,
                                     const float* scores, const uint32_t* counts,
                                     const uint64_t* id_base, uint64_t top_k, uint64_t* out_ids,
                                     float* out_scores, uint32_t* out_src, uint32_t* out_count,
                                     int32_t mem, int32_t device, void* stream) {
  return merge_lists(0, nlists, nq, k, ids, scores, counts, id_base, top_k, out_ids, out_scores,
                     out_src, out_count, mem, device, stream);
}

extern "C" isl_status isl_merge_service(uint64_t nlists, uint64_t nq, uint64_t k, const uint64_t* ids,
                                        const float* distances, const uint32_t* counts,
                                        const uint64_t* files_len, uint64_t top_k, uint64_t* out_ids,
                                        float* out_scores, uint32_t* out_src, uint32_t* out_count,
                                        int32_t mem, int32_t device, void* stream) {
  return merge_lists(1, nlists, nq, k, ids, distances, counts, files_len, top_k, out_ids, out_scores,
                     out_src, out_count, mem, device, stream);
}

extern "C" uint64_t isl_shard_record_bytes(uint64_t nq, uint64_t k) {
  return (nq * k * 12 + nq * 4 + 15) / 16 * 16;  // ids | distances | counts, padded to 16 bytes
}

extern "C" isl_status isl_merge_topk_packed_async(uint64_t nlists, uint64_t nq, uint64_t k, const void* d_records,
                                                  uint64_t list_stride, const uint64_t* d_id_base,
                                                  uint64_t top_k, uint64_t* d_out_ids, float* d_out_scores,
                                                  uint32_t* d_out_src, uint32_t* d_out_count, uint32_t* d_flags,
                                                  int32_t device, void* stream) {
  if (nq == 0) return ISL_OK;
  if (nlists == 0 || nlists > 64) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "nlists must be in [1, 64]");
  if (!d_records || !d_out_count || !d_flags || (top_k && (!d_out_ids || !d_out_scores)))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  if (list_stride < nq * k * 12 + nq * 4 || list_stride % 8)
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "list_stride must cover a record and be a multiple of 8");
  ISL_TRY(isl::use_device(device));
  const char* base = (const char*)d_records;
  hipLaunchKernelGGL(merge_topk_kernel, dim3((uint32_t)((nq + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                     (uint32_t)nlists, (uint32_t)nq, (uint32_t)k, (const uint64_t*)base,
                     (const float*)(base + nq * k * 8), (const uint32_t*)(base + nq * k * 12), list_stride, d_id_base,
                     (uint32_t)top_k, d_out_ids, d_out_scores, d_out_src, d_out_count, d_flags, 0u);
  ISL_HIP(hipGetLastError());
  return ISL_OK;
}
