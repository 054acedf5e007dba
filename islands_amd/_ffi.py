"""ctypes loader for libislands_amd.so (the C ABI of include/islands_amd.h).

The library is the product; this module only binds it.  There is no Python or
CPU fallback: a missing library raises ImportError, a missing GPU surfaces as
CoreError(Device) from the first compute call.
"""
from __future__ import annotations

import ctypes as C
import os

# One hardware queue per search in flight: up to 16 searches may overlap on their own streams
# (isl_search_batch_device_async), the HIP default of 4 queues per process would serialise them.
# Must be in the environment before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

_HERE = os.path.dirname(os.path.abspath(__file__))
# ISL_AMD_LIB: another build of the same library (A/B measurements of two builds on one card)
LIB_PATH = os.environ.get("ISL_AMD_LIB") or os.path.join(_HERE, "lib", "libislands_amd.so")

u64, u32, i32, f32 = C.c_uint64, C.c_uint32, C.c_int32, C.c_float
P = C.POINTER


class LeannConfigC(C.Structure):
    """isl_leann_config == LeannConfig, src/core/leann.rs:322-371."""
    _fields_ = [
        ("m", u64), ("m0", u64), ("ef_construction", u64), ("ml", C.c_double),
        ("max_layers", u64), ("metric", u32), ("ef_search", u64), ("beam_width", u64),
        ("prune_ratio", f32), ("pruning_strategy", u32), ("high_degree_pruning", C.c_uint8),
        ("hub_percentile", f32), ("is_compact", C.c_uint8), ("is_recompute", C.c_uint8),
    ]


class BertConfigC(C.Structure):
    """isl_bert_config (config.json fields used by the forward pass)."""
    _fields_ = [("vocab_size", u32), ("hidden", u32), ("layers", u32), ("heads", u32),
                ("intermediate", u32), ("max_position", u32), ("type_vocab", u32),
                ("layer_norm_eps", f32), ("gelu_tanh", u32)]


class IndexMetadataC(C.Structure):
    """isl_index_metadata == IndexMetadata, src/core/storage.rs:16-29."""
    _fields_ = [("version", u32), ("num_vectors", u64), ("dimension", u64), ("created_at", C.c_int64),
                ("updated_at", C.c_int64), ("has_description", i32), ("description", C.c_char * 256)]


class SearchStatsC(C.Structure):
    _fields_ = [("queries", u64), ("expansions", u64), ("edges", u64), ("evals", u64),
                ("pushes", u64), ("exact_path", u64), ("replayed", u64), ("kernel_ms", C.c_double),
                ("encoded_nodes", u64), ("recompute_rounds", u64), ("allocations", u64)]


# name -> (restype, argtypes); every symbol declared in include/islands_amd.h
SIGNATURES = {
    "isl_last_error_message": (C.c_char_p, []),
    "isl_last_error_expected": (u64, []),
    "isl_last_error_actual": (u64, []),
    "isl_last_error_node": (u64, []),
    "isl_status_name": (C.c_char_p, [i32]),
    "isl_abi_version": (u32, []),
    "isl_device_count": (i32, []),
    "isl_leann_config_paper_default": (None, [P(LeannConfigC)]),
    "isl_leann_config_fast": (None, [P(LeannConfigC)]),
    "isl_leann_config_accurate": (None, [P(LeannConfigC)]),
    "isl_leann_config_validate": (i32, [P(LeannConfigC)]),
    "isl_index_new": (i32, [P(LeannConfigC), P(C.c_void_p)]),
    "isl_index_from_csr": (i32, [P(LeannConfigC), u64, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, i32, u64, u64, i32, u64, P(C.c_void_p)]),
    "isl_index_from_device_csr": (i32, [P(LeannConfigC), i32, u64, C.c_void_p, C.c_void_p, i32,
                                        u64, i32, u64, P(C.c_void_p)]),
    "isl_index_from_bytes": (i32, [C.c_void_p, C.c_size_t, P(C.c_void_p)]),
    "isl_index_to_bytes": (i32, [C.c_void_p, P(C.c_void_p), P(C.c_size_t)]),
    "isl_free_bytes": (None, [C.c_void_p]),
    "isl_index_free": (None, [C.c_void_p]),
    "isl_index_len": (u64, [C.c_void_p]),
    "isl_index_is_empty": (i32, [C.c_void_p]),
    "isl_index_dimension": (i32, [C.c_void_p, P(u64)]),
    "isl_index_storage_bytes": (u64, [C.c_void_p]),
    "isl_index_is_recompute": (i32, [C.c_void_p]),
    "isl_index_is_compact": (i32, [C.c_void_p]),
    "isl_index_config": (i32, [C.c_void_p, P(LeannConfigC)]),
    "isl_index_entry_point": (i32, [C.c_void_p, P(u64)]),
    "isl_index_max_level": (u64, [C.c_void_p]),
    "isl_index_get_neighbors": (i32, [C.c_void_p, u64, P(P(u64)), P(C.c_size_t)]),
    "isl_index_upload": (i32, [C.c_void_p, i32]),
    "isl_set_embeddings": (i32, [C.c_void_p, C.c_void_p, u64, u64, i32, i32]),
    "isl_search_batch": (i32, [C.c_void_p, C.c_void_p, u64, u64, u64, u64, C.c_void_p,
                               C.c_void_p, C.c_void_p]),
    "isl_hnsw_from_bytes": (i32, [C.c_void_p, C.c_size_t, i32, P(C.c_void_p)]),
    "isl_distance_matrix_bf16": (i32, [i32, C.c_void_p, u64, C.c_void_p, u64, u64, C.c_void_p, i32, i32,
                                       C.c_void_p]),
    "isl_index_set_pq_codes": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, u64, i32]),
    "isl_search_two_level_batch": (i32, [C.c_void_p, C.c_void_p, u64, u64, u64, u64, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "isl_search_two_level_batch_device": (i32, [C.c_void_p, C.c_void_p, u64, u64, u64, u64,
                                                C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p]),
    "isl_search_two_level_batch_device_async": (i32, [C.c_void_p, C.c_void_p, u64, u64, u64, u64,
                                                      C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, P(u64)]),
    "isl_search_batch_device": (i32, [C.c_void_p, C.c_void_p, u64, u64, u64, u64, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "isl_search_batch_device_async": (i32, [C.c_void_p, C.c_void_p, u64, u64, u64, u64, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, P(u64)]),
    "isl_search_wait": (i32, [C.c_void_p, u64]),
    "isl_search_stream_wait": (i32, [C.c_void_p, u64, C.c_void_p]),
    "isl_shard_record_bytes": (u64, [u64, u64]),
    "isl_merge_topk_packed_async": (i32, [u64, u64, u64, C.c_void_p, u64, C.c_void_p, u64, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, i32, C.c_void_p]),
    "isl_shard_unique_id": (i32, [C.c_void_p]),
    "isl_shard_group_create": (i32, [i32, i32, i32, C.c_void_p, P(C.c_void_p)]),
    "isl_shard_group_create_host": (i32, [i32, i32, i32, C.c_void_p, C.c_void_p, P(C.c_void_p)]),
    "isl_shard_group_info": (i32, [C.c_void_p, P(i32), P(i32), P(i32), P(i32)]),
    "isl_shard_group_free": (None, [C.c_void_p]),
    "isl_sharded_searcher_new": (i32, [C.c_void_p, C.c_void_p, u64, C.c_void_p, i32, P(C.c_void_p)]),
    "isl_sharded_searcher_free": (None, [C.c_void_p]),
    "isl_sharded_prepare": (i32, [C.c_void_p, u64, u64, u64]),
    "isl_sharded_submit": (i32, [C.c_void_p, C.c_void_p, u64, u64, u64, u64, C.c_void_p, P(u64)]),
    "isl_sharded_result": (i32, [C.c_void_p, u64, P(C.c_void_p), P(C.c_void_p), P(C.c_void_p), P(C.c_void_p),
                                 P(SearchStatsC)]),
    "isl_sharded_search_batch": (i32, [C.c_void_p, C.c_void_p, u64, u64, u64, u64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "isl_sharded_flags": (i32, [C.c_void_p, P(u32)]),
    "isl_search_wait_stats": (i32, [C.c_void_p, u64, P(SearchStatsC)]),
    "isl_search_batch_async": (i32, [C.c_void_p, C.c_void_p, u64, u64, u64, u64, C.c_void_p,
                                     C.c_void_p, C.c_void_p, P(u64)]),
    "isl_index_prepare": (i32, [C.c_void_p, u64, u64, u64, i32]),
    "isl_search": (i32, [C.c_void_p, C.c_void_p, u64, u64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "isl_search_last_stats": (i32, [C.c_void_p, P(SearchStatsC)]),
    "isl_distance": (i32, [i32, C.c_void_p, u64, C.c_void_p, u64, P(f32)]),
    "isl_distance_squared": (i32, [i32, C.c_void_p, u64, C.c_void_p, u64, P(f32)]),
    "isl_distance_batch": (i32, [i32, C.c_void_p, u64, C.c_void_p, u64, u64, C.c_void_p, i32, i32,
                                 C.c_void_p]),
    "isl_normalize_rows": (i32, [C.c_void_p, u64, u64, i32, i32, C.c_void_p]),
    "isl_distance_matrix": (i32, [i32, C.c_void_p, u64, C.c_void_p, u64, u64, C.c_void_p, i32, i32,
                                  C.c_void_p]),
    "isl_bruteforce_topk": (i32, [i32, C.c_void_p, u64, C.c_void_p, u64, u64, u64, C.c_void_p,
                                  C.c_void_p, C.c_void_p, i32, i32, C.c_void_p]),
    "isl_row_sumsq_bf16": (i32, [C.c_void_p, u64, u64, C.c_void_p, i32, i32, C.c_void_p]),
    "isl_distance_matrix_bf16_norms": (i32, [i32, C.c_void_p, u64, C.c_void_p, u64, u64, C.c_void_p, C.c_void_p,
                                             C.c_void_p, i32, i32, C.c_void_p]),
    "isl_distance_matrix_bf16_enqueue": (i32, [i32, C.c_void_p, u64, C.c_void_p, u64, u64, C.c_void_p, C.c_void_p,
                                               C.c_void_p, i32, C.c_void_p]),
    "isl_bruteforce_topk_bf16": (i32, [i32, C.c_void_p, u64, C.c_void_p, u64, u64, u64, C.c_void_p,
                                       C.c_void_p, C.c_void_p, i32, i32, C.c_void_p]),
    "isl_merge_topk": (i32, [u64, u64, u64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, u64,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, i32, i32,
                             C.c_void_p]),
    "isl_merge_service": (i32, [u64, u64, u64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, u64,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, i32, i32,
                             C.c_void_p]),
    "isl_index_build": (i32, [P(LeannConfigC), C.c_void_p, u64, u64, C.c_void_p, u64, i32, i32,
                              P(C.c_void_p)]),
    "isl_index_metadata_new": (None, [u64, u64, C.c_int64, P(IndexMetadataC)]),
    "isl_storage_write_metadata": (i32, [P(IndexMetadataC), P(C.c_void_p), P(C.c_size_t)]),
    "isl_storage_read_metadata": (i32, [C.c_void_p, C.c_size_t, P(IndexMetadataC), P(C.c_size_t)]),
    "isl_index_save": (i32, [C.c_void_p, C.c_char_p, P(IndexMetadataC)]),
    "isl_index_load": (i32, [C.c_char_p, P(C.c_void_p), P(IndexMetadataC)]),
    "isl_encoder_new": (i32, [P(BertConfigC), i32, P(C.c_void_p)]),
    "isl_encoder_free": (None, [C.c_void_p]),
    "isl_encoder_set_weight": (i32, [C.c_void_p, C.c_char_p, C.c_void_p, u64, i32]),
    "isl_encoder_set_precision": (i32, [C.c_void_p, i32]),
    "isl_encoder_forward": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, u64, u64,
                                  C.c_void_p, i32, C.c_void_p]),
    "isl_encoder_embed": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, u64, u64, i32,
                                C.c_void_p, i32, C.c_void_p]),
    "isl_set_recompute_provider": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, u64, u64,
                                         i32, i32, i32]),
    "isl_index_set_recompute_cache_rows": (i32, [C.c_void_p, u64]),
    "isl_index_recompute_cache_bytes": (u64, [C.c_void_p]),
    "isl_mean_pool_normalize": (i32, [C.c_void_p, C.c_void_p, u64, u64, u64, i32, C.c_void_p, i32,
                                      i32, C.c_void_p]),
    "isl_hnsw_from_layers": (i32, [u64, u64, u64, i32, u64, u64, u64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, i32, u64, u64, C.c_void_p, i32, P(C.c_void_p)]),
    "isl_hnsw_free": (None, [C.c_void_p]),
    "isl_hnsw_len": (u64, [C.c_void_p]),
    "isl_hnsw_last_stats": (i32, [C.c_void_p, P(SearchStatsC)]),
    "isl_hnsw_search_batch": (i32, [C.c_void_p, C.c_void_p, u64, u64, u64, u64, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "isl_pq_new": (i32, [u64, u64, u64, C.c_void_p, i32, i32, P(C.c_void_p)]),
    "isl_pq_free": (None, [C.c_void_p]),
    "isl_pq_build_distance_tables": (i32, [C.c_void_p, C.c_void_p, u64, u64, C.c_void_p, i32,
                                           C.c_void_p]),
    "isl_pq_table_distance": (i32, [C.c_void_p, C.c_void_p, C.c_void_p, u64, C.c_void_p, i32,
                                    C.c_void_p]),
    "isl_pq_asymmetric_distance": (i32, [C.c_void_p, C.c_void_p, u64, C.c_void_p, u64,
                                         C.c_void_p, i32, C.c_void_p]),
    "isl_pq_encode": (i32, [C.c_void_p, C.c_void_p, u64, u64, C.c_void_p, i32, C.c_void_p]),
}

# host all-gather callback of isl_shard_group_create_host: (user, send, recv, bytes) -> 0 on success
SHARD_ALLGATHER_FN = C.CFUNCTYPE(i32, C.c_void_p, C.c_void_p, C.c_void_p, u64)

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C islands_amd/csrc` (there is no CPU fallback)")
        # PyTorch wheels bundle their own HIP runtime.  When torch is used in the same process
        # (device tensors handed to the *_device entry points, torch.distributed) it must be the
        # first to load it, otherwise torch later finds "No HIP GPUs".  torch stays optional.
        if os.environ.get("ISL_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib
