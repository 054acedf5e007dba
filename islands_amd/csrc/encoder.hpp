// Internal view of the recompute encoder shared by encoder.hip (kernels, C ABI), api_index.hip
// (isl_set_recompute_provider) and search.hip (the rounds that re-encode the nodes a search misses).
#pragma once

#include "common.hpp"

struct isl_encoder {
  isl_bert_config cfg{};
  int device = -1;
  float *word = nullptr, *pos = nullptr, *type = nullptr, *eln_w = nullptr, *eln_b = nullptr;
  struct Layer {
    float *wqkv = nullptr, *bqkv = nullptr, *wo = nullptr, *bo = nullptr, *ln1w = nullptr,
          *ln1b = nullptr, *wi = nullptr, *bi = nullptr, *wo2 = nullptr, *bo2 = nullptr,
          *ln2w = nullptr, *ln2b = nullptr;
  };
  std::vector<Layer> layers;
  // optional reduced-precision mode: bf16 copies of the Linear weights (same order as `layers`)
  struct Layer16 { void *wqkv = nullptr, *wo = nullptr, *wi = nullptr, *wo2 = nullptr; };
  std::vector<Layer16> layers16;
  int32_t precision = 0;  // ISL_DTYPE_F32 / ISL_DTYPE_BF16
  std::vector<void*> owned;
  // workspace, grown on demand (tokens = sequences * padded length)
  uint64_t ws_tokens = 0;
  float *x = nullptr, *x1 = nullptr, *t = nullptr, *qkv = nullptr, *ctx = nullptr, *inter = nullptr;
  void *x16 = nullptr, *x1_16 = nullptr;  // bf16 copies of x / x1 (bf16 mode)
  float* d_mask = nullptr;
  int64_t *d_ids = nullptr, *d_tt = nullptr;
  uint32_t* d_flag = nullptr;
  // second workspace + stream (round 4): encoder_embed_nodes runs the two halves of a batch side by side,
  // so that one half's last, partly filled wave of GEMM tiles and its LayerNorm / attention kernels lie
  // beside the other half's GEMMs (a sequence's embedding does not depend on what it is batched with)
  struct Side {
    uint64_t ws_tokens = 0;
    float *x = nullptr, *x1 = nullptr, *t = nullptr, *qkv = nullptr, *ctx = nullptr, *inter = nullptr;
    void *x16 = nullptr, *x1_16 = nullptr;
    float* d_mask = nullptr;
    int64_t *d_ids = nullptr, *d_tt = nullptr;
    uint32_t* d_flag = nullptr;
    void* stream = nullptr;   // hipStream_t (non-blocking)
    void* ev_in = nullptr;    // hipEvent_t: the caller's stream up to the call
    void* ev_out = nullptr;   // hipEvent_t: the side stream's half is done
  } side;
  std::mutex mu;
};


namespace isl {
// Encodes the nodes listed in d_node_ids (device) from the resident token table (u16 ids, row i
// = node i, `L` slots per node, d_lens[i] of them used; NULL = all) and writes embedding b to
// d_rows + d_node_ids[b] * stride.  Sequences are padded to L: masked keys contribute exact
// zeros, so the result does not depend on the padded length.
// d_out_rows (device, may be NULL = the node ids): embedding b goes to d_rows + d_out_rows[b] * stride.
// batch sizes at which the passes of encoder_embed_nodes run whole waves of GEMM tiles (0 = unknown)
void encoder_batch_quantum(const isl_encoder* e, uint32_t L, uint32_t* quantum, uint32_t* chunk);
isl_status encoder_embed_nodes(isl_encoder* e, const uint16_t* d_tokens, const uint16_t* d_lens,
                               uint32_t L, const uint32_t* d_node_ids, uint64_t n, int normalize,
                               float* d_rows, uint64_t stride, hipStream_t st,
                               const uint32_t* d_out_rows = nullptr);
}  // namespace isl
