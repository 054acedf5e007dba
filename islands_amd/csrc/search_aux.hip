// Heap-exact kernel, two-level search and HnswGraph descent (see search_kernels.hip.h).
#include "search_kernels.hip.h"

namespace {
template <bool HNSW>
void launch_exact_t(int metric, uint32_t grid, size_t lds, hipStream_t st, const SearchParams& p) {
  switch (metric) {
    case ISL_METRIC_COSINE: launch_one(leann_search_exact<ISL_METRIC_COSINE, HNSW>, grid, lds, st, p); break;
    case ISL_METRIC_EUCLIDEAN: launch_one(leann_search_exact<ISL_METRIC_EUCLIDEAN, HNSW>, grid, lds, st, p); break;
    case ISL_METRIC_DOT: launch_one(leann_search_exact<ISL_METRIC_DOT, HNSW>, grid, lds, st, p); break;
    default: launch_one(leann_search_exact<ISL_METRIC_MANHATTAN, HNSW>, grid, lds, st, p); break;
  }
}
template <typename ROWT, bool RESUME, bool QH>
void launch_two_level_t(int metric, uint32_t grid, size_t lds, hipStream_t st, const SearchParams& p) {
  switch (metric) {
    case ISL_METRIC_COSINE: launch_one(leann_search_two_level<ISL_METRIC_COSINE, ROWT, RESUME, QH>, grid, lds, st, p); break;
    case ISL_METRIC_EUCLIDEAN: launch_one(leann_search_two_level<ISL_METRIC_EUCLIDEAN, ROWT, RESUME, QH>, grid, lds, st, p); break;
    case ISL_METRIC_DOT: launch_one(leann_search_two_level<ISL_METRIC_DOT, ROWT, RESUME, QH>, grid, lds, st, p); break;
    default: launch_one(leann_search_two_level<ISL_METRIC_MANHATTAN, ROWT, RESUME, QH>, grid, lds, st, p); break;
  }
}
}  // namespace

void isl_launch::launch_exact(int metric, bool hnsw, uint32_t grid, size_t lds, hipStream_t st, const void* params) {
  const SearchParams& p = *static_cast<const SearchParams*>(params);
  if (hnsw) launch_exact_t<true>(metric, grid, lds, st, p);
  else launch_exact_t<false>(metric, grid, lds, st, p);
}

void isl_launch::launch_two_level(int metric, bool bf16, bool resume, bool qh, uint32_t grid, size_t lds, hipStream_t st,
                                  const void* params) {
  const SearchParams& p = *static_cast<const SearchParams*>(params);
  if (bf16 && qh) launch_two_level_t<uint16_t, false, true>(metric, grid, lds, st, p);
  else if (bf16) launch_two_level_t<uint16_t, false, false>(metric, grid, lds, st, p);
  else if (resume) launch_two_level_t<float, true, false>(metric, grid, lds, st, p);
  else launch_two_level_t<float, false, false>(metric, grid, lds, st, p);
}

void isl_launch::launch_descent(int metric, uint32_t grid, size_t lds, hipStream_t st, const void* params) {
  const SearchParams& p = *static_cast<const SearchParams*>(params);
  switch (metric) {
    case ISL_METRIC_COSINE: launch_one(hnsw_descent_kernel<ISL_METRIC_COSINE>, grid, lds, st, p); break;
    case ISL_METRIC_EUCLIDEAN: launch_one(hnsw_descent_kernel<ISL_METRIC_EUCLIDEAN>, grid, lds, st, p); break;
    case ISL_METRIC_DOT: launch_one(hnsw_descent_kernel<ISL_METRIC_DOT>, grid, lds, st, p); break;
    default: launch_one(hnsw_descent_kernel<ISL_METRIC_MANHATTAN>, grid, lds, st, p); break;
  }
}

void isl_launch::launch_classify(uint32_t grid, hipStream_t st, const void* params) {
  const SearchParams& p = *static_cast<const SearchParams*>(params);
  hipLaunchKernelGGL(classify_queries_kernel, dim3(grid), dim3(64), 0, st, p);
}
