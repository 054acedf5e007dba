// leann_search_fast<S = 8> for every metric, row type and row width (see search_kernels.hip.h).
#include "search_kernels.hip.h"

namespace {
template <typename ROWT, bool WIDE, bool RESUME = false, bool QH = false>
void launch_t(int metric, uint32_t grid, size_t lds, hipStream_t st, const SearchParams& p) {
  switch (metric) {
    case ISL_METRIC_COSINE: launch_one(leann_search_fast<8, ISL_METRIC_COSINE, ROWT, WIDE, RESUME, QH>, grid, lds, st, p); break;
    case ISL_METRIC_EUCLIDEAN: launch_one(leann_search_fast<8, ISL_METRIC_EUCLIDEAN, ROWT, WIDE, RESUME, QH>, grid, lds, st, p); break;
    case ISL_METRIC_DOT: launch_one(leann_search_fast<8, ISL_METRIC_DOT, ROWT, WIDE, RESUME, QH>, grid, lds, st, p); break;
    default: launch_one(leann_search_fast<8, ISL_METRIC_MANHATTAN, ROWT, WIDE, RESUME, QH>, grid, lds, st, p); break;
  }
}
}  // namespace

// wide = the index has adjacency rows of 65..128 ids (its own instantiation: the common case keeps
// its register budget)
void isl_launch::launch_fast_s8(int metric, bool wide, bool bf16, bool resume, bool qh, uint32_t grid, size_t lds,
                                 hipStream_t st, const void* params) {
  const SearchParams& p = *static_cast<const SearchParams*>(params);
  if (qh) {  // bf16 rows, bf16-valued queries: the query operand stays bf16 in LDS
    launch_t<uint16_t, false, false, true>(metric, grid, lds, st, p);
  } else if (resume) {  // searches over the recompute provider (f32 rows) that park and resume
    if (wide) launch_t<float, true, true>(metric, grid, lds, st, p);
    else launch_t<float, false, true>(metric, grid, lds, st, p);
  } else if (bf16) {
    if (wide) launch_t<uint16_t, true>(metric, grid, lds, st, p);
    else launch_t<uint16_t, false>(metric, grid, lds, st, p);
  } else {
    if (wide) launch_t<float, true>(metric, grid, lds, st, p);
    else launch_t<float, false>(metric, grid, lds, st, p);
  }
}
