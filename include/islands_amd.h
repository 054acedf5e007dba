/*
 * islands_amd.h -- C ABI of the MI355X-native LEANN search path.
 *
 * This is the drop-in boundary for the `islands::core` search hot path of
 * panbanda/islands v1.5.0.  The reference has no FFI of its own (it is a
 * single Rust crate); each entry point below replaces one inherent method /
 * trait method of `src/core`, cited as file:line, and is what a Rust
 * `unsafe extern "C"` block would bind (see INTEGRATION.md for that stub).
 *
 * Conventions
 *   - every function returns isl_status (0 = Ok, otherwise the CoreError
 *     variant of src/core/error.rs:9-62, same order); nothing aborts or throws
 *     across the ABI.  Details of the last error on the calling thread are read
 *     with the isl_last_error_* getters.
 *   - inputs are borrowed for the duration of the call; outputs are written
 *     into caller-allocated buffers, except *_to_bytes (freed with
 *     isl_free_bytes).  An index handle owns its host and device copies.
 *   - plain pointers and sizes only; no C++/torch types.  `stream` arguments
 *     are a hipStream_t passed as void* (NULL = the library's own stream).
 *   - the compute path is HIP on gfx950.  There is NO CPU fallback: without a
 *     usable device every compute entry point returns ISL_ERR_DEVICE.
 */
#ifndef ISLANDS_AMD_H
#define ISLANDS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISL_ABI_VERSION 3

typedef int32_t isl_status;

/* CoreError, src/core/error.rs:9-62 (declaration order), plus ABI-level codes >= 100. */
enum {
  ISL_OK = 0,
  ISL_ERR_DIMENSION_MISMATCH = 1, /* payload: expected, actual */
  ISL_ERR_EMPTY_COLLECTION = 2,
  ISL_ERR_INVALID_CONFIG = 3,
  ISL_ERR_INDEX_NOT_BUILT = 4,
  ISL_ERR_NODE_NOT_FOUND = 5, /* payload: node id */
  ISL_ERR_SERIALIZATION = 6,
  ISL_ERR_DESERIALIZATION = 7,
  ISL_ERR_IO = 8,
  ISL_ERR_HNSW = 9,
  ISL_ERR_PQ = 10,
  ISL_ERR_SEARCH = 11,
  ISL_ERR_EMBEDDING = 12,
  ISL_ERR_DEVICE = 100,           /* no gfx950 device / HIP runtime error */
  ISL_ERR_INVALID_ARGUMENT = 101, /* NULL pointer, bad enum value */
  ISL_ERR_UNSUPPORTED = 102
};

/* DistanceMetric, src/core/distance.rs:9-19 (bincode variant index). */
enum { ISL_METRIC_COSINE = 0, ISL_METRIC_EUCLIDEAN = 1, ISL_METRIC_DOT = 2, ISL_METRIC_MANHATTAN = 3 };
/* PruningStrategy, src/core/leann.rs:168-178. */
enum { ISL_PRUNE_GLOBAL = 0, ISL_PRUNE_LOCAL = 1, ISL_PRUNE_PROPORTIONAL = 2 };
/* Row element types accepted by isl_set_embeddings. */
enum { ISL_DTYPE_F32 = 0, ISL_DTYPE_BF16 = 1 };
/* Where a caller's buffer lives. */
enum { ISL_MEM_HOST = 0, ISL_MEM_DEVICE = 1 };

/* ---- error details (thread-local) ---- */
const char* isl_last_error_message(void);
uint64_t isl_last_error_expected(void); /* DimensionMismatch.expected */
uint64_t isl_last_error_actual(void);   /* DimensionMismatch.actual   */
uint64_t isl_last_error_node(void);     /* NodeNotFound(id)           */
const char* isl_status_name(isl_status s);
uint32_t isl_abi_version(void);
/* Number of usable gfx950 devices (0 when none; never an error). */
int32_t isl_device_count(void);

/* ---- LeannConfig, src/core/leann.rs:322-371 (same fields, same order) ---- */
typedef struct isl_leann_config {
  uint64_t m;
  uint64_t m0;
  uint64_t ef_construction;
  double ml;
  uint64_t max_layers;
  uint32_t metric; /* ISL_METRIC_* */
  uint64_t ef_search;
  uint64_t beam_width;
  float prune_ratio;
  uint32_t pruning_strategy; /* ISL_PRUNE_* */
  uint8_t high_degree_pruning;
  float hub_percentile;
  uint8_t is_compact;
  uint8_t is_recompute;
} isl_leann_config;

void isl_leann_config_paper_default(isl_leann_config* c); /* leann.rs:386-403 */
void isl_leann_config_fast(isl_leann_config* c);          /* leann.rs:406-416 */
void isl_leann_config_accurate(isl_leann_config* c);      /* leann.rs:419-429 */
isl_status isl_leann_config_validate(const isl_leann_config* c); /* leann.rs:432-460 */

/* ---- LeannIndex, src/core/leann.rs:492-546 ---- */
typedef struct isl_index isl_index;

/* LeannIndex::new, leann.rs:504-511 (validates the config). */
isl_status isl_index_new(const isl_leann_config* cfg, isl_index** out);
/* Build an index handle from CsrGraph's public fields (leann.rs:193-208).
 * levels/degree_counts may be NULL (zeros / row lengths).  dimension:
 * has_dimension = 0 encodes None. */
isl_status isl_index_from_csr(const isl_leann_config* cfg, uint64_t num_nodes,
                              const uint64_t* node_offsets, const uint64_t* neighbors,
                              const uint64_t* levels, const uint64_t* degree_counts,
                              int32_t has_entry, uint64_t entry_point, uint64_t max_level,
                              int32_t has_dimension, uint64_t dimension, isl_index** out);
/* Same, from arrays already resident on `device` (u64 offsets, u32 neighbour
 * ids); the handle takes a private device copy, the host CSR is materialised
 * lazily (to_bytes / get_neighbors). */
isl_status isl_index_from_device_csr(const isl_leann_config* cfg, int32_t device,
                                     uint64_t num_nodes, const uint64_t* d_node_offsets,
                                     const uint32_t* d_neighbors, int32_t has_entry,
                                     uint64_t entry_point, int32_t has_dimension,
                                     uint64_t dimension, isl_index** out);
/* LeannIndex::build, leann.rs:560-630, on the device: insertion in id order, construction
 * search with ef_construction (:692-749), high-degree-preserving selection (:761-833) or
 * truncation (:685), bidirectional links with a distance re-sort past m0 (:592-607, :634-658).
 * `levels` replaces random_level (thread_rng, :549-554; NULL = all 0).  `batch` = nodes inserted
 * per step: 1 reproduces the reference's sequential construction (CsrGraph identical field by
 * field), larger values trade that for throughput.  The vectors become the index's in-memory
 * provider.  Limits: m0 <= 128 (LeannConfig::accurate() has 96), ef_construction <= 512. */
isl_status isl_index_build(const isl_leann_config* cfg, const float* vectors, uint64_t n, uint64_t d,
                           const uint64_t* levels, uint64_t batch, int32_t mem, int32_t device,
                           isl_index** out);
/* LeannIndex::from_bytes / to_bytes, leann.rs:1059-1066 (bincode 1.x default layout). */
isl_status isl_index_from_bytes(const uint8_t* bytes, size_t len, isl_index** out);
isl_status isl_index_to_bytes(const isl_index* idx, uint8_t** out, size_t* len);
void isl_free_bytes(uint8_t* p);
void isl_index_free(isl_index* idx);

/* ---- storage.rs: IndexMetadata (:16-47) and the chunk framing of IndexWriter / IndexReader ----
 * A chunk is tag[4] || u64 LE length || payload (:127-135, :159-173); the META chunk carries
 * serde_json::to_vec(&IndexMetadata) (:119-124).  `now` replaces chrono::Utc::now() (:37). */
typedef struct isl_index_metadata {
  uint32_t version;        /* IndexMetadata::CURRENT_VERSION = 1 */
  uint64_t num_vectors;
  uint64_t dimension;
  int64_t created_at;
  int64_t updated_at;
  int32_t has_description; /* Option<String> */
  char description[256];   /* NUL-terminated UTF-8 */
} isl_index_metadata;
void isl_index_metadata_new(uint64_t num_vectors, uint64_t dimension, int64_t now, isl_index_metadata* out);
/* IndexWriter::write_metadata on a buffer: the META chunk (free with isl_free_bytes). */
isl_status isl_storage_write_metadata(const isl_index_metadata* meta, uint8_t** out, size_t* len);
/* IndexReader::read_metadata on a buffer; *consumed = bytes of the chunk.  A different tag is
 * Deserialization("expected META chunk") (:151-153), a short buffer is Io. */
isl_status isl_storage_read_metadata(const uint8_t* bytes, size_t len, isl_index_metadata* meta,
                                     size_t* consumed);
/* One-file persistence: parents created like FileSystemStorage::save (:68-74), chunk META then
 * chunk "LIDX" = LeannIndex::to_bytes().  meta NULL = IndexMetadata::new(len, dimension) now. */
isl_status isl_index_save(const isl_index* idx, const char* path, const isl_index_metadata* meta);
isl_status isl_index_load(const char* path, isl_index** out, isl_index_metadata* meta);

uint64_t isl_index_len(const isl_index* idx);                 /* leann.rs:519-521 */
int32_t isl_index_is_empty(const isl_index* idx);             /* leann.rs:524-526 */
int32_t isl_index_dimension(const isl_index* idx, uint64_t* dim); /* 1 = Some, leann.rs:529 */
uint64_t isl_index_storage_bytes(const isl_index* idx);       /* leann.rs:534, 296-301 */
int32_t isl_index_is_recompute(const isl_index* idx);         /* leann.rs:539 */
int32_t isl_index_is_compact(const isl_index* idx);           /* leann.rs:544 */
isl_status isl_index_config(const isl_index* idx, isl_leann_config* out);
int32_t isl_index_entry_point(const isl_index* idx, uint64_t* entry); /* 1 = Some */
uint64_t isl_index_max_level(const isl_index* idx);
/* CsrGraph::get_neighbors, leann.rs:225-233: returns 1 (Some) / 0 (None).  The
 * slice stays valid until the index is freed. */
int32_t isl_index_get_neighbors(const isl_index* idx, uint64_t node, const uint64_t** ptr,
                                size_t* len);

/* Copy the CSR graph into HBM of `device` (u64 offsets, u32 neighbour ids). */
isl_status isl_index_upload(isl_index* idx, int32_t device);

/* InMemoryEmbeddingProvider::new, leann.rs:111-120: attach n rows of d
 * elements as the provider for this index.  `rows` is row-major, on the host
 * or already on the index's device (mem = ISL_MEM_*).  The handle keeps its
 * own HBM copy (rows padded to 16-byte multiples).  n == 0 -> EmptyCollection.
 * dtype ISL_DTYPE_BF16: `rows` holds bf16 bit patterns (u16) and is stored as such -- half the
 * HBM bytes per visited node; the provider's vectors are their exact f32 images and the
 * arithmetic stays the reference's f32 chain, so results equal the reference's on those f32
 * rows.  (The HnswGraph facade, the builder and the distance / PQ entry points take f32.) */
isl_status isl_set_embeddings(isl_index* idx, const void* rows, uint64_t n, uint64_t d,
                              int32_t dtype, int32_t mem);

/* ---- search ---- */
/* Work counters of one search call (summed over the batch): the inputs of the roofline formula
 * (SURVEY.md section 8d). */
typedef struct isl_search_stats {
  uint64_t queries;
  uint64_t expansions;     /* H: candidates expanded */
  uint64_t edges;          /* E: neighbour ids read */
  uint64_t evals;          /* V: embeddings fetched / distances evaluated */
  uint64_t pushes;         /* heap insertions */
  uint64_t exact_path;     /* queries answered by the heap-exact kernel */
  uint64_t replayed;       /* queries whose tied prefix was re-ordered by the replay kernel */
  double kernel_ms;        /* HIP-event time of the search kernels of that call */
  uint64_t encoded_nodes;    /* recompute provider: embeddings computed by the encoder */
  uint64_t recompute_rounds; /* recompute provider: search rounds of that call (0 otherwise) */
  uint64_t allocations;    /* device / pinned-host allocations and stream / event creations the
                              call had to make (0 for every call that fits isl_index_prepare) */
} isl_search_stats;

/* Sets up everything a search needs so that no later call allocates, creates a stream or
 * synchronises for set-up: `lanes` (1..32) search lanes sized for batches of up to max_nq queries
 * with ef <= max_ef and k <= max_k (streams, events, per-query arrays, overflow tables, push
 * logs, pinned staging buffers for the host-pointer entry points), the padded adjacency, the
 * shared scratch pool of the heap-exact kernel, and one empty launch of the search kernels on
 * every lane's stream (code objects loaded, hardware queues created).  Call it after
 * isl_index_upload + a provider setter.  Lanes run on a per-device pool of at most 16 HIP streams
 * (ISL_MAX_STREAMS; a card serves about that many hardware queues side by side and the rate
 * collapses beyond): lanes past the pool's size share streams, their calls queue in stream order.
 * The reference has no such step (its search allocates
 * per call, leann.rs:903-908); without it the first calls on each lane do this work lazily and
 * report it in isl_search_stats::allocations. */
isl_status isl_index_prepare(isl_index* idx, uint64_t max_nq, uint64_t max_ef, uint64_t max_k,
                             int32_t lanes);

/* LeannIndex::search_with_params over a batch of queries (leann.rs:868-896;
 * batch = Searcher::search_batch semantics, search.rs:179-181: each query is
 * answered independently, results in query order).
 *   queries: nq rows of d floats; out_ids/out_dist: nq*k slots, row i holds
 *   out_count[i] <= k valid entries, ascending distance.
 * Errors that the reference raises per query (NodeNotFound) fail the whole
 * call with the first failing query's error, as the sequential map would.
 * Host buffers in, host buffers out (the caller contract of search.rs:150-181,
 * indexer/service.rs:781-785); re-entrant: concurrent calls run on different lanes. */
isl_status isl_search_batch(const isl_index* idx, const float* queries, uint64_t nq, uint64_t d,
                            uint64_t k, uint64_t ef, uint64_t* out_ids, float* out_dist,
                            uint32_t* out_count);
/* Pipelined form of isl_search_batch: copies `queries` into the lane's pinned staging buffer
 * (they may be reused as soon as the call returns), enqueues H2D copy, search and D2H copies on
 * the lane's stream and returns.  out_ids / out_dist / out_count must stay valid until
 * isl_search_wait[_stats](*token) has returned; they are written there.  Up to 32 calls may be
 * in flight.  token 0 = answered immediately (empty index, nq == 0). */
isl_status isl_search_batch_async(const isl_index* idx, const float* queries, uint64_t nq, uint64_t d,
                                  uint64_t k, uint64_t ef, uint64_t* out_ids, float* out_dist,
                                  uint32_t* out_count, uint64_t* token);
/* Same with every buffer already on the index's device; enqueued on `stream`
 * (NULL = the legacy default stream, i.e. ordered after the caller's earlier
 * default-stream work) and synchronised before returning (status needs the result). */
isl_status isl_search_batch_device(const isl_index* idx, const float* d_queries, uint64_t nq,
                                   uint64_t d, uint64_t k, uint64_t ef, uint64_t* d_out_ids,
                                   float* d_out_dist, uint32_t* d_out_count, void* stream);
/* Asynchronous form: enqueues the search on one of the index's private streams (ordered after
 * the work already enqueued on `stream`) and returns at once; up to 32 searches may be in
 * flight, so consecutive batches overlap and the slowest queries of one batch no longer idle
 * the chip.  Outputs are valid and the per-query status is reported once isl_search_wait
 * returns for *token (token 0 = the call was answered immediately).  An index with the recompute
 * provider is accepted too (this form and isl_search_batch_async): the provider works through the
 * batch in rounds -- search, encode what was missed, resume -- on a host thread of the library's,
 * one set of rounds at a time per index; isl_search_stream_wait then waits on the host.  Calls of
 * this form (and of isl_search_batch_async and isl_search_two_level_batch_device_async) that are waiting for their turn are
 * answered TOGETHER by the call that gets it, when they agree in d, k, ef and search kind: one set of
 * rounds over the union of their queries, so a node several of them need is encoded once and the
 * encoder gets fuller passes (Searcher::search_batch, search.rs:179-181, hands batches over one by one;
 * how they are batched on the device is the library's business).  Each call's answers, status and
 * counters are its own (recompute_rounds / encoded_nodes of its statistics are the shared rounds'); if
 * the union fails, every member is run by itself and gets its own error. */
isl_status isl_search_batch_device_async(const isl_index* idx, const float* d_queries, uint64_t nq,
                                         uint64_t d, uint64_t k, uint64_t ef, uint64_t* d_out_ids,
                                         float* d_out_dist, uint32_t* d_out_count, void* stream,
                                         uint64_t* token);
/* Completes the call `token` names: waits for its stream, copies host-pointer results out, turns
 * per-query failures into the CoreError of the first failing query.  `stats` (may be NULL)
 * receives the counters of exactly that call. */
isl_status isl_search_wait_stats(const isl_index* idx, uint64_t token, isl_search_stats* stats);
isl_status isl_search_wait(const isl_index* idx, uint64_t token);
/* Makes `stream` wait (on the device, no host synchronisation) for the search kernels of the call
 * `token` names: what a consumer of the device-resident answers enqueues on its own stream -- the
 * shard exchange of a multi-GPU search issues its all-gather behind this -- while the host keeps
 * submitting.  The call still has to be completed with isl_search_wait[_stats]. */
isl_status isl_search_stream_wait(const isl_index* idx, uint64_t token, void* stream);
/* LeannIndex::search, leann.rs:858-865: one query, ef = config.ef_search. */
isl_status isl_search(const isl_index* idx, const float* query, uint64_t d, uint64_t k,
                      uint64_t* out_ids, float* out_dist, uint32_t* out_count);

/* Counters of the most recent search call THIS THREAD completed on this index (synchronous entry
 * points and isl_search_wait alike); zeros when there is none.  Calls made by other threads never
 * show up here -- use isl_search_wait_stats for a per-call record. */
isl_status isl_search_last_stats(const isl_index* idx, isl_search_stats* out);

/* ---- distance.rs ---- */
/* Distance::calculate, distance.rs:38-52. */
isl_status isl_distance(int32_t metric, const float* a, uint64_t na, const float* b, uint64_t nb,
                        float* out);
/* Distance::calculate_squared, distance.rs:54-66. */
isl_status isl_distance_squared(int32_t metric, const float* a, uint64_t na, const float* b,
                                uint64_t nb, float* out);
/* Distance::batch_calculate, distance.rs:32-34: query against n contiguous
 * rows of `row_len` floats (row_len != d -> DimensionMismatch). mem applies
 * to query, rows and out alike. */
isl_status isl_distance_batch(int32_t metric, const float* query, uint64_t d, const float* rows,
                              uint64_t n, uint64_t row_len, float* out, int32_t mem, int32_t device,
                              void* stream);
/* normalize_vector, distance.rs:125-132, over n rows in place. */
isl_status isl_normalize_rows(float* rows, uint64_t n, uint64_t d, int32_t mem, int32_t device,
                              void* stream);

/* Every (query, row) distance at once as one GEMM on the matrix cores (float32 MFMA): the
 * batch_calculate of distance.rs:32-34 / benches/vector_ops.rs:60-79 for a whole query batch.
 * out [nq][n].  Cosine, Euclidean (sqrt(|q|^2 + |r|^2 - 2 q.r), clamped at 0) and DotProduct;
 * Manhattan is not a contraction (Unsupported).  The MFMA accumulates in its own order: values
 * agree with the sequential reference sums to float32 rounding, not bit for bit -- the search
 * itself keeps the exact-order kernels. */
isl_status isl_distance_matrix(int32_t metric, const float* queries, uint64_t nq, const float* rows,
                               uint64_t n, uint64_t d, float* out, int32_t mem, int32_t device,
                               void* stream);
/* Exact k nearest rows of every query by brute force over isl_distance_matrix blocks (ties
 * towards the smaller id): the ground truth of recall measurements. */
isl_status isl_bruteforce_topk(int32_t metric, const float* queries, uint64_t nq, const float* rows,
                               uint64_t n, uint64_t d, uint64_t k, uint64_t* out_ids, float* out_dist,
                               uint32_t* out_count, int32_t mem, int32_t device, void* stream);

/* ---- search.rs / indexer merge (multi-index = multi-shard) ---- */
/* MultiIndexSearcher::search merge, search.rs:211-237: per query, nlists
 * candidate lists (list-major: [list][query][k]) concatenated in list order,
 * stable-sorted by score ascending, truncated to top_k.  id_base[l] is added to
 * list l's ids (global id = shard base + local id).  counts: [list][query]. */
isl_status isl_merge_topk(uint64_t nlists, uint64_t nq, uint64_t k, const uint64_t* ids,
                          const float* scores, const uint32_t* counts, const uint64_t* id_base,
                          uint64_t top_k, uint64_t* out_ids, float* out_scores,
                          uint32_t* out_src, uint32_t* out_count, int32_t mem, int32_t device,
                          void* stream);
/* The same merge for the multi-GPU exchange, asynchronous and allocation-free: every
 * shard's answers arrive as one packed record of `list_stride` bytes -- ids u64[nq][k], then
 * distances f32[nq][k], then counts u32[nq] (isl_shard_record_bytes) -- which is what one
 * all-gather delivers in rank order.  Everything lives on the device (d_id_base: u64[nlists]);
 * the kernel is enqueued on `stream` and nothing is waited for.  *d_flags (u32, zeroed by the
 * caller once) collects bit 0 = a NaN score met (the reference panics, search.rs:231), bit 1 =
 * a list was not ascending, bit 2 = a list carried ISL_SHARD_POISON_COUNT as a query's count (its
 * producer failed that query: the list counts as empty); read it when the results are consumed. */
#define ISL_SHARD_POISON_COUNT 0xFFFFFFFFu
uint64_t isl_shard_record_bytes(uint64_t nq, uint64_t k);
isl_status isl_merge_topk_packed_async(uint64_t nlists, uint64_t nq, uint64_t k, const void* d_records,
                                       uint64_t list_stride, const uint64_t* d_id_base, uint64_t top_k,
                                       uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_src,
                                       uint32_t* d_out_count, uint32_t* d_flags, int32_t device, void* stream);
/* ---- multi-GPU: the index sharded by node-id range, one process (rank) per GPU ----
 * Semantics = MultiIndexSearcher::search (src/core/search.rs:211-237) and the product's cross-index
 * merge (src/indexer/service.rs:775-801) with one sub-index per rank: every rank answers the whole
 * query batch in its own sub-graph (local ids), the per-rank top-k lists are concatenated in rank
 * order, stable-sorted by distance (ties -> the lower rank) and truncated to k; global id = the
 * rank's id base + local id.  The exchange is ONE collective per batch -- an all-gather of every
 * rank's packed answer record (isl_shard_record_bytes) -- issued on a side stream behind a device
 * event of the search, followed by the merge kernel on that stream: the host only submits, and the
 * exchange of batch i overlaps the traversals of batches i+1.. on the index's lanes.
 *
 * isl_shard_group = the communicator of the R ranks.  Two transports:
 *   - RCCL (xGMI): rank 0 calls isl_shard_unique_id, hands the 128 bytes to every rank by whatever
 *     means the host has (the reference product would use its own RPC; the tests use a store), and
 *     every rank calls isl_shard_group_create -- ncclCommInitRank, collective over the ranks.
 *     librccl.so is loaded on first use (the copy already in the process, if any).
 *   - host: `allgather(user, send, recv, bytes)` is a blocking all-gather of `bytes` from every rank
 *     into recv (rank-major) over host memory, supplied by the caller (MPI, gloo, a socket ...): for
 *     boxes without xGMI between the ranks' cards and for tests where ranks share one card.  The
 *     exchange is then synchronous inside isl_sharded_submit.
 * world == 1 needs no group (pass NULL to isl_sharded_searcher_new).
 *
 * Failure across ranks: MultiIndexSearcher::search propagates the first index's error (`?`,
 * search.rs:215), so one failing shard fails the whole search -- on EVERY rank, and without leaving
 * the others inside the collective.  A rank whose shard search cannot be enqueued or fails (wrong
 * dimension, no lane, a device error, a query that ends in NodeNotFound ...) still takes part in the
 * batch's all-gather with a record whose counts are ISL_SHARD_POISON_COUNT (the whole batch, or the
 * failed queries); the merge raises flag bit 2 on every rank and isl_sharded_result returns
 * ISL_ERR_SEARCH for that batch everywhere (the failing rank reports its own error).  Waits on an
 * exchange are bounded (ISL_SHARD_TIMEOUT_MS, default 60000): a peer that never arrives is
 * ISL_ERR_DEVICE naming the batch, not a hang. */
#define ISL_SHARD_UNIQUE_ID_BYTES 128
typedef struct isl_shard_group isl_shard_group;
typedef int32_t (*isl_shard_allgather_fn)(void* user, const void* send, void* recv, uint64_t bytes);
isl_status isl_shard_unique_id(uint8_t id[ISL_SHARD_UNIQUE_ID_BYTES]);
isl_status isl_shard_group_create(int32_t device, int32_t world, int32_t rank,
                                  const uint8_t id[ISL_SHARD_UNIQUE_ID_BYTES], isl_shard_group** out);
isl_status isl_shard_group_create_host(int32_t device, int32_t world, int32_t rank,
                                       isl_shard_allgather_fn allgather, void* user, isl_shard_group** out);
/* world / rank as created; *comm_ranks = what the communicator itself reports (ncclCommCount; the
 * host transport reports `world`); *is_rccl = 1 for the RCCL transport. */
isl_status isl_shard_group_info(const isl_shard_group* grp, int32_t* world, int32_t* rank, int32_t* comm_ranks,
                                int32_t* is_rccl);
/* (a group must outlive the searchers created over it) */
void isl_shard_group_free(isl_shard_group* grp);

/* The searcher of one rank: `shard` is this rank's LeannIndex over its id range (resident, provider
 * attached; borrowed, must outlive the searcher).  id_base[r] = first global id of rank r; NULL =
 * even ranges of n_total, rank r owns [r*n_total/R, (r+1)*n_total/R).  Up to `depth` (1..32)
 * batches in flight. */
typedef struct isl_sharded_searcher isl_sharded_searcher;
isl_status isl_sharded_searcher_new(const isl_index* shard, isl_shard_group* grp, uint64_t n_total,
                                    const uint64_t* id_base, int32_t depth, isl_sharded_searcher** out);
void isl_sharded_searcher_free(isl_sharded_searcher* s);
/* Buffers for `depth` batches of up to max_nq queries / max_k results, and isl_index_prepare on the
 * shard: nothing allocates on the submit path afterwards.  Collective over the ranks whenever the
 * searcher has a group (the RCCL communicator's first all-gather, which sets up its channels, is made
 * here): the 16 bytes exchanged carry every rank's local outcome, so a rank whose set-up failed still
 * enters the collective and ALL ranks return an error (the failing one its own, the others
 * ISL_ERR_SEARCH). */
isl_status isl_sharded_prepare(isl_sharded_searcher* s, uint64_t max_nq, uint64_t max_k, uint64_t max_ef);
/* Enqueues one batch (queries on the device, ordered after the work already on `stream`): shard
 * search -> all-gather of the records -> merge.  Every rank must submit the same batches (nq, k) in
 * the same order.  Returns a handle; nothing is waited for (RCCL transport).  Any free slot is taken
 * (the one released longest ago).  When this rank's search cannot be enqueued the exchange is made
 * all the same, with a poisoned record, the call returns the search's error and *handle stays 0 (the
 * slot is reclaimed behind the exchange); the other ranks learn of it from isl_sharded_result. */
isl_status isl_sharded_submit(isl_sharded_searcher* s, const float* d_queries, uint64_t nq, uint64_t d,
                              uint64_t k, uint64_t ef, void* stream, uint64_t* handle);
/* Completes a submitted batch: per-query failures of this rank's shard surface here (the first
 * failing query's CoreError), a failure of any other rank's shard as ISL_ERR_SEARCH, a NaN score in
 * this batch's merge as ISL_ERR_SEARCH (the reference panics, search.rs:231; flags are per batch);
 * the merged answers are on the device -- global ids u64[nq][k], distances f32[nq][k], source rank
 * u32[nq][k], counts u32[nq] -- and stay valid until `depth` further batches have been submitted
 * (slots are reused least-recently-released first).  `stats` (may be NULL) = this rank's search
 * counters.  The wait for the exchange is bounded (ISL_SHARD_TIMEOUT_MS -> ISL_ERR_DEVICE). */
isl_status isl_sharded_result(isl_sharded_searcher* s, uint64_t handle, const uint64_t** d_ids,
                              const float** d_dist, const uint32_t** d_src, const uint32_t** d_count,
                              isl_search_stats* stats);
/* MultiIndexSearcher::search over the shards with host buffers in and out (submit + result +
 * copies); out_src may be NULL.  A NaN score in the merge is ISL_ERR_SEARCH (the reference panics,
 * search.rs:231). */
isl_status isl_sharded_search_batch(isl_sharded_searcher* s, const float* queries, uint64_t nq, uint64_t d,
                                    uint64_t k, uint64_t ef, uint64_t* out_ids, float* out_dist,
                                    uint32_t* out_src, uint32_t* out_count);
/* Merge flags of all completed batches so far, OR-ed (bit 0: NaN score met, bit 1: a list was not
 * ascending, bit 2: a rank failed a batch); synchronises the exchange stream. */
isl_status isl_sharded_flags(isl_sharded_searcher* s, uint32_t* flags);

/* Product-level merge, src/indexer/service.rs:775-801 (IndexerService::search_with_embeddings):
 * per list the (id, distance) results of one index (searched with ef = max(top_k, 100), :781);
 * results whose id has no file entry are dropped (`files_len[l]` = stored.files.len(), host
 * array, NULL = keep all, :788), score = 1.0 - distance (:791), stable sort by score descending
 * (:800), truncate to top_k.  Layouts as in isl_merge_topk. */
isl_status isl_merge_service(uint64_t nlists, uint64_t nq, uint64_t k, const uint64_t* ids,
                             const float* distances, const uint32_t* counts,
                             const uint64_t* files_len, uint64_t top_k, uint64_t* out_ids,
                             float* out_scores, uint32_t* out_src, uint32_t* out_count,
                             int32_t mem, int32_t device, void* stream);

/* ---- pq.rs ---- */
typedef struct isl_pq isl_pq;
/* ProductQuantizer with trained codebooks: m x K x dsub floats (pq.rs:116-129). */
isl_status isl_pq_new(uint64_t dimension, uint64_t m, uint64_t K, const float* codebooks,
                      int32_t metric, int32_t device, isl_pq** out);
void isl_pq_free(isl_pq* pq);
/* build_distance_tables, pq.rs:307-338, for nq queries: tables [nq][m][K]. */
isl_status isl_pq_build_distance_tables(const isl_pq* pq, const float* queries, uint64_t nq,
                                        uint64_t d, float* tables, int32_t mem, void* stream);
/* table_distance, pq.rs:341-348: codes [n][m] u16 against ONE query's tables [m][K]. */
isl_status isl_pq_table_distance(const isl_pq* pq, const float* tables, const uint16_t* codes,
                                 uint64_t n, float* out, int32_t mem, void* stream);
/* asymmetric_distance, pq.rs:275-304: one query against n code rows. */
isl_status isl_pq_asymmetric_distance(const isl_pq* pq, const float* query, uint64_t d,
                                      const uint16_t* codes, uint64_t n, float* out, int32_t mem,
                                      void* stream);
/* encode, pq.rs:221-244: n vectors -> [n][m] u16 codes. */
isl_status isl_pq_encode(const isl_pq* pq, const float* vectors, uint64_t n, uint64_t d,
                         uint16_t* codes, int32_t mem, void* stream);

/* ---- embedding/candle_provider.rs:353-507: the recompute encoder (CandleEmbedder) ----
 * The model is candle-transformers 0.9.1 `BertModel` (third party, not in the reference tree;
 * call sites candle_provider.rs:284, :429-432); its published algorithm runs here in float32,
 * the linear layers on the matrix cores.  isl_bert_config mirrors the fields of the
 * checkpoint's config.json that the forward pass uses. */
typedef struct isl_bert_config {
  uint32_t vocab_size;
  uint32_t hidden;        /* hidden_size */
  uint32_t layers;        /* num_hidden_layers */
  uint32_t heads;         /* num_attention_heads; hidden / heads in {16, 32, 64} */
  uint32_t intermediate;  /* intermediate_size */
  uint32_t max_position;  /* max_position_embeddings */
  uint32_t type_vocab;    /* type_vocab_size */
  float layer_norm_eps;
  uint32_t gelu_tanh;     /* 0: hidden_act "gelu" (erf), 1: "gelu_new"/approximate (tanh) */
} isl_bert_config;
typedef struct isl_encoder isl_encoder;
isl_status isl_encoder_new(const isl_bert_config* cfg, int32_t device, isl_encoder** out);
void isl_encoder_free(isl_encoder* enc);
/* One tensor of the checkpoint by its HuggingFace name ("embeddings.word_embeddings.weight",
 * "encoder.layer.0.attention.self.query.weight", ...; an optional "bert." prefix is ignored), the
 * names candle's VarBuilder resolves (candle_provider.rs:267-284).  A wrong element count is
 * DimensionMismatch{expected, actual}. */
isl_status isl_encoder_set_weight(isl_encoder* enc, const char* name, const float* data,
                                  uint64_t count, int32_t mem);
/* Optional precision of the Linear layers: ISL_DTYPE_F32 (default) is the reference's float32
 * model; ISL_DTYPE_BF16 rounds weights and GEMM inputs to bf16 (float32 accumulation on the bf16
 * matrix cores) -- several times faster, embeddings differ from the float32 ones by about 1e-2
 * relative.  Call after the weights are set (changing a weight afterwards needs a new call). */
isl_status isl_encoder_set_precision(isl_encoder* enc, int32_t dtype);
/* BertModel::forward(input_ids, token_type_ids, Some(attention_mask)), candle_provider.rs:429-432:
 * ids [B, L] (already padded per :385-402), token_type_ids NULL = zeros, attention_mask [B, L]
 * of 0.0 / 1.0, NULL = ones -> last hidden state [B, L, hidden]. */
isl_status isl_encoder_forward(isl_encoder* enc, const int64_t* input_ids,
                               const int64_t* token_type_ids, const float* attention_mask,
                               uint64_t B, uint64_t L, float* out_hidden, int32_t mem, void* stream);
/* embed_texts_raw after tokenisation (candle_provider.rs:404-507): forward, masked mean pooling,
 * L2 normalisation when `normalize` -> [B, hidden]. */
isl_status isl_encoder_embed(isl_encoder* enc, const int64_t* input_ids, const int64_t* token_type_ids,
                             const float* attention_mask, uint64_t B, uint64_t L, int32_t normalize,
                             float* out, int32_t mem, void* stream);

/* Recompute provider: EmbeddingProvider (leann.rs:82-99) backed by the encoder, the mode the
 * paper and `LeannConfig::is_recompute` (leann.rs:366-371) describe -- no stored embeddings, the
 * vectors of the nodes a search visits are computed on the fly from their text.  `tokens` holds
 * node i's tokenised text in row i (n rows of L slots, `lengths[i]` of them used, NULL = all L;
 * tokenisation itself is third-party code outside this library).  After the call searches on
 * `idx` re-encode what they visit: the rows a batch misses are encoded once each, in batches
 * across all its queries.  keep_rows = 0 forgets every row when a call returns (pure recompute),
 * 1 keeps them as an embedding cache.  The encoder is borrowed and must outlive the index;
 * its hidden size becomes the index dimension; `normalize` as in isl_encoder_embed.
 * Results equal those of the in-memory provider holding the same embeddings.  Every search kernel
 * parks a query that meets an absent row and resumes it there once the row is encoded, so the row
 * cache may be far smaller than a traversal (floor: 256 rows). */
isl_status isl_set_recompute_provider(isl_index* idx, isl_encoder* enc, const uint16_t* tokens,
                                      const uint16_t* lengths, uint64_t n, uint64_t L,
                                      int32_t normalize, int32_t keep_rows, int32_t mem);
/* The recompute provider does not store embeddings (leann.rs:366-371); what it keeps on the device
 * is the token table and a bounded row cache: a slab of `rows` embedding rows (default 2^20, or
 * every node of a smaller index; at least 256) plus 4 bytes of slot map per node.  Slots are handed
 * out round-robin, the oldest rows make room.  isl_index_recompute_cache_bytes = what that costs. */
isl_status isl_index_set_recompute_cache_rows(isl_index* idx, uint64_t rows);
uint64_t isl_index_recompute_cache_bytes(const isl_index* idx);

/* ---- embedding/candle_provider.rs:434-488: masked mean-pool + optional L2 normalise ----
 * hidden [B][L][H] f32, mask [B][L] (0/1 as f32), out [B][H].  The BERT forward that
 * produces `hidden` is third-party code (candle-transformers) and out of this round. */
isl_status isl_mean_pool_normalize(const float* hidden, const float* mask, uint64_t B, uint64_t L,
                                   uint64_t H, int32_t normalize, float* out, int32_t mem,
                                   int32_t device, void* stream);

/* ---- HnswGraph, src/core/hnsw.rs:149-515 (search side) ---- */
typedef struct isl_hnsw isl_hnsw;
/* Builds a device-resident HnswGraph from its parts: `levels[i]` = node i's top layer;
 * layer L adjacency in CSR form over ALL nodes (rows of nodes below layer L are empty):
 * layer_offsets[L] has num_nodes + 1 entries, layer_neighbors[L] the ids.  vectors: num_nodes
 * rows of d floats (HnswNode::vector).  m/m0/ef_construction are carried for completeness. */
isl_status isl_hnsw_from_layers(uint64_t m, uint64_t m0, uint64_t ef_construction, int32_t metric,
                                uint64_t num_nodes, uint64_t d, uint64_t num_layers,
                                const uint64_t* const* layer_offsets,
                                const uint64_t* const* layer_neighbors, const uint64_t* levels,
                                int32_t has_entry, uint64_t entry_point, uint64_t max_level,
                                const float* vectors, int32_t device, isl_hnsw** out);
/* HnswGraph::from_bytes, hnsw.rs:511-514: reads the bincode image of a HnswGraph (1.x default
 * layout; HashMap entries in any order, node ids 0..n-1) and makes it resident on `device`.
 * Truncated or inconsistent input -> ISL_ERR_DESERIALIZATION.  No writer: HashMap order makes the
 * reference's own bytes non-deterministic. */
isl_status isl_hnsw_from_bytes(const uint8_t* bytes, size_t len, int32_t device, isl_hnsw** out);
void isl_hnsw_free(isl_hnsw* h);
uint64_t isl_hnsw_len(const isl_hnsw* h);
/* HnswGraph::search (hnsw.rs:458-504) for a batch of queries: greedy descent through layers
 * max_level..1, then search_layer on layer 0 with ef = max(ef, k); heap order on distance only
 * (hnsw.rs:136-141), reproduced with an exact BinaryHeap emulation. */
isl_status isl_hnsw_search_batch(const isl_hnsw* h, const float* queries, uint64_t nq, uint64_t d,
                                 uint64_t k, uint64_t ef, uint64_t* out_ids, float* out_dist,
                                 uint32_t* out_count);
/* Work counters of the most recent search on this graph (see isl_search_last_stats);
 * exact_path = queries in which two equal distances met and the heap-exact kernel decided. */
isl_status isl_hnsw_last_stats(const isl_hnsw* h, isl_search_stats* out);

/* isl_distance_matrix for bf16 queries and rows (bit patterns; BASELINE config 5: d = 4096 bf16): the
 * same batch_calculate (distance.rs:32-34) over the exact float32 images of the stored values, on
 * the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, float32 accumulation).  out [nq][n] f32.
 * d must be a multiple of 64; Manhattan -> Unsupported.  Agreement with the sequential sums:
 * float32 rounding of the accumulation order (<= 1e-5 on normalised rows), not bit for bit. */
isl_status isl_distance_matrix_bf16(int32_t metric, const uint16_t* queries, uint64_t nq, const uint16_t* rows,
                                    uint64_t n, uint64_t d, float* out, int32_t mem, int32_t device,
                                    void* stream);
/* Sum of squares of every row of a bf16 matrix (of the exact float32 images), as the cosine / Euclidean
 * epilogues of isl_distance_matrix_bf16 take it; out [n] f32. */
isl_status isl_row_sumsq_bf16(const uint16_t* rows, uint64_t n, uint64_t d, float* out, int32_t mem, int32_t device,
                              void* stream);
/* isl_distance_matrix_bf16 for callers that keep the rows (or the queries) resident across calls:
 * q_sumsq [nq] / row_sumsq [n] = isl_row_sumsq_bf16 of them, in the matrices' memory space, computed
 * once instead of per call (NULL = computed here).  Same outputs, bit for bit. */
isl_status isl_distance_matrix_bf16_norms(int32_t metric, const uint16_t* queries, uint64_t nq, const uint16_t* rows,
                                          uint64_t n, uint64_t d, const float* q_sumsq, const float* row_sumsq,
                                          float* out, int32_t mem, int32_t device, void* stream);
/* The same GEMM, enqueued on `stream` and nothing else: device buffers only, q_sumsq / row_sumsq required
 * (except for ISL_METRIC_DOT), no allocation, no synchronisation -- the caller's next work on `stream`
 * finds the distances in `out`.  For callers that walk resident rows block by block (config 5: 153 blocks
 * of 65536 rows per query batch) and must not let the chip idle between two blocks.  Same outputs, bit
 * for bit.  Replaces Distance::batch_calculate (distance.rs:32-34) over a block of rows. */
isl_status isl_distance_matrix_bf16_enqueue(int32_t metric, const uint16_t* queries, uint64_t nq, const uint16_t* rows,
                                            uint64_t n, uint64_t d, const float* q_sumsq, const float* row_sumsq,
                                            float* out, int32_t device, void* stream);
/* isl_bruteforce_topk over bf16 rows and queries: blocks of isl_distance_matrix_bf16 + the same running
 * top-k (ties towards the smaller id), exact under the bf16 GEMM's distances.  Ground truth and exact
 * kNN lists at sizes where the float32 GEMM would take tens of minutes (10M x 10M x 768). */
isl_status isl_bruteforce_topk_bf16(int32_t metric, const uint16_t* queries, uint64_t nq, const uint16_t* rows,
                                    uint64_t n, uint64_t d, uint64_t k, uint64_t* out_ids, float* out_dist,
                                    uint32_t* out_count, int32_t mem, int32_t device, void* stream);

/* ---- EXTENSION: two-level search with a PQ filter ----
 * "Algorithm 2: Two-Level Search with Hybrid Distance", docs/leann-specification.md:223-275; the
 * reference promises it (leann.rs:54-56, :855-857: "a two-level search with PQ filtering should
 * be used") and ships no implementation, so there is no reference result to be identical to.
 * The rules the pseudo-code leaves open are fixed in oracle/islands_oracle.c
 * (orc_two_level_search), which the device path matches bit for bit: every new neighbour gets
 * ProductQuantizer::table_distance (pq.rs:341-348) on the tables of build_distance_tables
 * (pq.rs:307-338); after each expansion the not yet promoted members of the first
 * ceil(rerank_ratio * |AQ|) entries of the approximate queue get their exact distance from the
 * index's embedding provider (in-memory or recompute) and enter the result set.
 *
 * isl_index_set_pq_codes: codes [n][pq->m] u16 as ProductQuantizer::encode writes them
 * (pq.rs:221-244), copied to the device; `pq` is borrowed and must outlive the index.  A code
 * >= num_centroids -> ISL_ERR_PQ (the reference's tables[sq][code] would panic). */
isl_status isl_index_set_pq_codes(isl_index* idx, const isl_pq* pq, const uint16_t* codes, uint64_t n,
                                  int32_t mem);
/* Same outputs and error behaviour as isl_search_batch / isl_search_batch_device.
 * isl_search_last_stats afterwards: evals = exact distance evaluations, pushes = approximate
 * (table) evaluations.  ISL_ERR_SEARCH when ceil(rerank_ratio * |AQ|) outgrows the queue window a
 * wave keeps in LDS (up to 16384 entries) -- never a different answer. */
isl_status isl_search_two_level_batch(const isl_index* idx, const float* queries, uint64_t nq, uint64_t d,
                                      uint64_t k, uint64_t ef, float rerank_ratio, uint64_t* out_ids,
                                      float* out_dist, uint32_t* out_count);
isl_status isl_search_two_level_batch_device(const isl_index* idx, const float* d_queries, uint64_t nq,
                                             uint64_t d, uint64_t k, uint64_t ef, float rerank_ratio,
                                             uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count,
                                             void* stream);
/* Asynchronous form (see isl_search_batch_device_async): up to 32 calls in flight on the index's
 * lanes, completed with isl_search_wait[_stats](*token).  The call runs on a host thread of the
 * library's (a query whose queue window was too small is re-run with a larger one before the call
 * completes), so isl_search_stream_wait for such a token waits on the host. */
isl_status isl_search_two_level_batch_device_async(const isl_index* idx, const float* d_queries, uint64_t nq,
                                                   uint64_t d, uint64_t k, uint64_t ef, float rerank_ratio,
                                                   uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count,
                                                   void* stream, uint64_t* token);

#ifdef __cplusplus
}
#endif
#endif /* ISLANDS_AMD_H */
