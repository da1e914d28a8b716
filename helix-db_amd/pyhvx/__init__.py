"""ctypes binding of libhelix_vec_gfx950.so (include/helix_vec.h) for tests and bench.py.

Names mirror the reference's Rust interface for this path (crates/db/src/search/vector/):
SearchParams (mod.rs:410-516), RestrictedVectorCandidates (restricted.rs:345-371),
ValidatedVectorReadIndex.search / .search_restricted (read_index.rs:83-102), SearchResult
(result.rs:20-41), HelixDbError.  There is NO CPU fallback here: if the HIP library is missing or a
call fails, HelixDbError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HVX_LIB_PATH: measurement scripts load the tuning build (make TUNING=1 -> libhelix_vec_gfx950_tuning.so) through this binding;
# the library itself reads no environment in its release build.
LIB_PATH = os.environ.get("HVX_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libhelix_vec_gfx950.so")

COSINE, EUCLIDEAN, MANHATTAN = 0, 1, 2
F32, BF16, FP8_E4M3 = 0, 1, 2
KERNEL_SCALAR, KERNEL_SSE, KERNEL_AVX, KERNEL_AVX_FMA, KERNEL_NEON = 0, 1, 2, 3, 4  # hvx_float_kernel == FloatSimd (spaces/simple.rs:45-62)
OK, ERR_DIMENSION, ERR_NONFINITE, ERR_ZERO_NORM, ERR_MAGNITUDE, ERR_K_RANGE, ERR_CANDIDATE_LIMIT, \
    ERR_DEVICE, ERR_INVARIANT, ERR_UNSUPPORTED, PENDING, ERR_BUSY, ERR_TIMEOUT = range(13)
DIR_OUT, DIR_IN, DIR_BOTH = 0, 1, 2

_STATUS_NAMES = {
    1: "InvalidDimension", 2: "InvalidVectorComponent", 3: "ZeroNormCosineVector",
    4: "ComponentMagnitudeExceeded", 5: "Query(result count / beam width out of range)",
    6: "Query(too many restricted candidates)", 7: "Device", 8: "InvariantViolation", 9: "Unsupported",
}


class HelixDbError(RuntimeError):
    def __init__(self, status: int, message: str = ""):
        self.status = status
        super().__init__(f"{_STATUS_NAMES.get(status, status)}: {message}")

    def is_invalid_vector_input(self) -> bool:  # crates/server/src/http.rs:143-145
        return self.status in (ERR_DIMENSION, ERR_NONFINITE, ERR_ZERO_NORM, ERR_MAGNITUDE)


class _Desc(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("metric", C.c_uint32), ("dtype", C.c_uint32),
                ("float_kernel", C.c_uint32), ("n", C.c_uint64), ("m", C.c_uint32), ("m0", C.c_uint32),
                ("has_entry", C.c_uint32), ("max_layer", C.c_uint32), ("entry_point", C.c_uint64),
                ("shard_id_lo", C.c_uint64), ("shard_id_hi", C.c_uint64), ("device", C.c_int32),
                ("max_batch", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("expansion_steps", C.c_uint64),
                ("neighbors_examined", C.c_uint64), ("vectors_loaded", C.c_uint64),
                ("distance_computations", C.c_uint64), ("tie_overflow_queries", C.c_uint64),
                ("device_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class QueryStats(C.Structure):
    _fields_ = [("expansion_steps", C.c_uint32), ("neighbors_examined", C.c_uint32),
                ("vectors_loaded", C.c_uint32), ("distance_computations", C.c_uint32)]


SIMHASH_ALWAYS, SIMHASH_ADAPTIVE, SIMHASH_OFF = 0, 1, 2


class _Params(C.Structure):  # hvx_search_params
    _fields_ = [("k", C.c_uint32), ("ef", C.c_uint32), ("simhash_mode", C.c_uint32),
                ("pre_simhash_sampling_ratio_override", C.c_float),
                ("bypass_min_frontier", C.c_uint32), ("bypass_window_expansions", C.c_uint32),
                ("bypass_min_filter_rate", C.c_float), ("read_budget_multiplier", C.c_uint32),
                ("simhash_sampling_ratio_override", C.c_float), ("simhash_failure_prob_override", C.c_float)]


class SimHashConfig(C.Structure):  # hvx_simhash_config: the index-level VectorIndexConfig knobs
    _fields_ = [("seed", C.c_uint64), ("simhash_threshold", C.c_uint32), ("sampling_ratio", C.c_float),
                ("adaptive_enabled", C.c_uint32), ("adaptive_failure_prob", C.c_float), ("resident_snapshot", C.c_uint32)]

    @classmethod
    def default(cls, **over):
        c = cls()
        lib().hvx_simhash_config_default(C.byref(c))
        for key, v in over.items():
            assert hasattr(c, key), key
            setattr(c, key, v)
        return c


class IndexMetadata(C.Structure):  # hvx_index_metadata: VectorIndexMetadata (values/vectors/metadata.rs:22-62)
    _fields_ = [("index_name", C.c_char * 256), ("property_name", C.c_char * 256), ("dimension", C.c_uint64), ("m", C.c_uint64),
                ("m0", C.c_uint64), ("ef_construction", C.c_uint64), ("simhash_threshold", C.c_uint64), ("ml", C.c_float),
                ("sampling_ratio", C.c_float), ("adaptive_failure_prob", C.c_float), ("adaptive_enabled", C.c_uint32),
                ("has_entry_point", C.c_uint32), ("max_layer", C.c_uint32), ("entry_point", C.c_uint64), ("count", C.c_uint64)]


def decode_index_metadata(value: bytes) -> dict:
    """hvx_decode_index_metadata: the rkyv-archived VectorIndexMetadata row (layout restated, parity unpinned)."""
    md = IndexMetadata()
    _check(lib().hvx_decode_index_metadata(value, len(value), C.byref(md)))
    out = {k: getattr(md, k) for k, _ in md._fields_}
    out["index_name"], out["property_name"] = md.index_name.decode(), md.property_name.decode()
    out["entry_point"] = md.entry_point if md.has_entry_point else None
    return out


class BuildParams(C.Structure):  # hvx_build_params
    _fields_ = [("ef_construction", C.c_uint32), ("max_batch", C.c_uint32), ("batch_divisor", C.c_uint32), ("sequential", C.c_uint32),
                ("link_mode", C.c_uint32), ("scatter", C.c_uint32), ("reserve_rows", C.c_uint64), ("reserve_upper_rows", C.c_uint64)]


class BuildStats(C.Structure):  # hvx_build_stats
    _fields_ = [("nodes", C.c_uint64), ("batches", C.c_uint64), ("single_node_batches", C.c_uint64)]


class DeleteStats(C.Structure):  # hvx_delete_stats
    _fields_ = [("requested", C.c_uint32), ("deleted", C.c_uint32), ("missing", C.c_uint32), ("entry_moves", C.c_uint32),
                ("relinked_rows", C.c_uint32), ("reserved", C.c_uint32), ("seconds", C.c_double)]


class BatcherTimes(C.Structure):
    """hvx_batcher_times (include/helix_vec.h)."""
    _fields_ = [("idle_ns", C.c_uint64), ("collect_ns", C.c_uint64), ("drain_ns", C.c_uint64), ("fill_ns", C.c_uint64), ("device_ns", C.c_uint64),
                ("wake_ns", C.c_uint64), ("lanes", C.c_uint32), ("reserved", C.c_uint32)]


class GraphAudit(C.Structure):  # hvx_graph_audit
    _fields_ = [(n, C.c_uint64) for n in ("nodes", "up_rows", "edges_l0", "edges_up", "asymmetric_edges_l0", "asymmetric_edges_up",
                                          "unsorted_entries", "self_loops", "out_of_range_ids", "holes", "level_violations",
                                          "degree_overflow_rows", "unreachable_l0")] + \
               [(n, C.c_uint32) for n in ("max_degree_l0", "max_degree_up", "bfs_levels_l0", "max_layer", "has_entry", "reserved")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "reserved"}


class AdaptiveStats(C.Structure):  # hvx_adaptive_stats
    _fields_ = [(n, C.c_uint32) for n in (
        "simhash_filtered", "simhash_examined", "simhash_passed_before_sampling", "simhash_passed_after_sampling",
        "pre_simhash_sample_kept", "pre_simhash_sample_dropped", "simhash_bypass_expansions",
        "simhash_skipped_candidates", "simhash_bypass_trigger_budget", "simhash_bypass_trigger_low_yield",
        "active_simhash_threshold_sum", "active_simhash_threshold_samples", "effective_beam_len_sum",
        "effective_beam_len_samples", "active_sampling_ratio_samples", "rng_words", "txn_get_simhash_filter")] + \
        [("active_sampling_ratio_sum", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# hvx_option: execution-path selectors of a handle (same results on every setting) / hvx_scan_path flags
OPT_HNSW_GENERAL_KERNEL, OPT_WAVE_LOG2CAP, OPT_FLAT_FORCE_VALU, OPT_FLAT_FIRST_CHUNK, OPT_FLAT_NO_TILE, OPT_FLAT_NO_FILTER, \
    OPT_FLAT_NO_FAST, OPT_FLAT_TILE_BUILD, OPT_FLAT_NO_SMALLB, OPT_HNSW_PAIR, OPT_DELETE_SEQUENTIAL, OPT_RESTRICTED_DIRECT, \
    OPT_RESTRICTED_EXACT_MIB, OPT_FLAT_NO_TAIL = range(14)
PATH_VALU, PATH_MFMA_128, PATH_TILE_256, PATH_FILTERED, PATH_FULL_SPLIT, PATH_VALU_FALLBACK_QUERIES, PATH_WIDENED, \
    PATH_PAIR_OVERFLOW_REPEAT, PATH_SMALL_BATCH, PATH_DIRECT, PATH_EXACT_TAIL = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024
# hvx_restricted_strategy: AUTO = the DEVICE's plan (exact scan while the candidate rows take <= 1 GiB, round 6), REFERENCE_PLAN = the
# reference's (restricted.rs:426-453: exact <= 256 ids / 4 MiB, the filter-aware walk above)
RESTRICTED_AUTO, RESTRICTED_EXACT, RESTRICTED_FILTERED, RESTRICTED_REFERENCE_PLAN = 0, 1, 2, 3
TERM_NONE, TERM_EXHAUSTED, TERM_BEAM_COMPLETE, TERM_ROUTING_BUDGET, TERM_BRIDGE_BUDGET, TERM_VECTOR_BUDGET = range(6)


class RestrictedParams(C.Structure):  # hvx_restricted_params
    _fields_ = [(n, C.c_uint32) for n in ("k", "ef", "strategy", "beam_percent", "directory_enabled", "explicit_budgets", "ef_filtered",
                                          "routing_rows", "bridge_rows", "vector_payloads", "sampled_seeds", "directory_seeds")]

    @classmethod
    def new(cls, k, ef, *, strategy=RESTRICTED_REFERENCE_PLAN, beam_percent=150, directory=True, **budgets):
        """search_restricted under the REFERENCE's plan (restricted.rs:426-453: what the reference runs, what the parity tests compare);
        `budgets` = explicit FilteredGraphBudgets as the reference's tests pass them.  `RestrictedParams.auto` = the device's plan."""
        p = cls(k=k, ef=ef, strategy=strategy, beam_percent=beam_percent, directory_enabled=1 if directory else 0)
        if budgets:
            p.explicit_budgets = 1
            p.strategy = RESTRICTED_FILTERED
            for name in ("ef_filtered", "routing_rows", "bridge_rows", "vector_payloads", "sampled_seeds", "directory_seeds"):
                setattr(p, name, int(budgets.pop(name)))
            assert not budgets, budgets
        return p


    @classmethod
    def auto(cls, k, ef, *, beam_percent=150, directory=True):
        """the library's default (hvx_restricted_params_default): HVX_RESTRICTED_AUTO, the device's cost-based plan"""
        return cls.new(k, ef, strategy=RESTRICTED_AUTO, beam_percent=beam_percent, directory=directory)


class RestrictedStats(C.Structure):  # hvx_restricted_stats: RestrictedSearchStats (restricted.rs:147-166)
    _fields_ = [("strategy", C.c_uint32), ("termination", C.c_uint32)] + [(n, C.c_uint64) for n in (
        "ef_filtered", "directory_scan_calls", "directory_rows", "directory_decoded_bytes", "directory_hits", "simhash_row_requests",
        "companion_row_requests", "routing_rows", "bridge_rows", "bridge_frontier_pushes", "neighbor_multi_get_calls",
        "vector_payload_requests", "vector_bytes", "distance_computations")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class BatcherTicket(C.Structure):  # hvx_batcher_ticket
    _fields_ = [("sequence", C.c_uint64), ("slot", C.c_uint32), ("reserved", C.c_uint32)]


_lib = None
_vp = C.c_void_p


def lib():
    """Load the product library; raises HelixDbError(ERR_DEVICE) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HelixDbError(ERR_DEVICE, f"{LIB_PATH} is missing: run __graft_entry__.build() / make -C helix-db_amd/csrc")
    L = C.CDLL(LIB_PATH)
    # hardware queues for execution lanes: the host's job since round 5 (the library no longer writes the environment when it is
    # loaded); this binding is the host, and calls it before anything can have initialised HIP through this library
    L.hvx_runtime_prepare.restype = C.c_int
    L.hvx_runtime_prepare.argtypes = [C.c_uint32]
    if not os.environ.get("HVX_NO_RUNTIME_PREPARE"):
        L.hvx_runtime_prepare(0)
    L.hvx_device_stream_read_gbs.restype = C.c_int
    L.hvx_device_stream_read_gbs.argtypes = [C.c_int32, C.c_uint64, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.hvx_shard_rccl_version.restype = C.c_int
    L.hvx_shard_rccl_version.argtypes = []
    L.hvx_last_error.restype = C.c_char_p
    L.hvx_version.restype = C.c_char_p
    L.hvx_index_import.restype = C.c_int
    L.hvx_index_import.argtypes = [C.POINTER(_Desc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)]
    L.hvx_index_import_reserve.restype = C.c_int
    L.hvx_index_import_reserve.argtypes = [C.POINTER(_Desc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, C.c_uint64, C.POINTER(_vp)]
    L.hvx_index_free.argtypes = [_vp]
    L.hvx_index_sync.restype = C.c_int
    L.hvx_index_sync.argtypes = [_vp]
    L.hvx_index_stream.restype = _vp
    L.hvx_index_stream.argtypes = [_vp]
    L.hvx_index_set_stream.restype = C.c_int
    L.hvx_index_set_stream.argtypes = [_vp, _vp]
    L.hvx_build_params_default.restype = None
    L.hvx_build_params_default.argtypes = [C.POINTER(BuildParams)]
    L.hvx_index_build.restype = C.c_int
    L.hvx_index_build.argtypes = [C.POINTER(_Desc), _vp, _vp, _vp, C.POINTER(BuildParams), C.POINTER(_vp), C.POINTER(BuildStats)]
    L.hvx_index_graph_sizes.restype = C.c_int
    L.hvx_index_graph_sizes.argtypes = [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.hvx_index_export_graph.restype = C.c_int
    L.hvx_index_export_graph.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp]
    L.hvx_index_link_rows.restype = C.c_int
    L.hvx_index_link_rows.argtypes = [_vp, _vp, _vp, C.c_uint32, C.c_uint32]
    L.hvx_index_audit_graph.restype = C.c_int
    L.hvx_index_audit_graph.argtypes = [_vp, C.POINTER(GraphAudit)]
    L.hvx_shard_group_unique_id.restype = C.c_int
    L.hvx_shard_group_unique_id.argtypes = [_vp]
    L.hvx_shard_group_init.restype = C.c_int
    L.hvx_shard_group_init.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_vp)]
    L.hvx_shard_group_search_batch_device.restype = C.c_int
    L.hvx_shard_group_search_batch_device.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp]
    L.hvx_shard_group_free.argtypes = [_vp]
    L.hvx_index_fork.restype = C.c_int
    L.hvx_index_fork.argtypes = [_vp, C.POINTER(_vp)]
    L.hvx_index_set_occupancy.restype = C.c_int
    L.hvx_index_set_occupancy.argtypes = [_vp, C.c_uint32]
    L.hvx_index_timing_begin.restype = C.c_int
    L.hvx_index_timing_begin.argtypes = [_vp, C.c_uint32]
    L.hvx_index_wave_clocks.restype = C.c_int
    L.hvx_index_wave_clocks.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.hvx_index_timing_collect.restype = C.c_int
    L.hvx_index_timing_collect.argtypes = [_vp, _vp, C.c_uint32, C.POINTER(C.c_uint32)]
    L.hvx_search_batch.restype = C.c_int
    L.hvx_search_batch.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, C.POINTER(Stats)]
    L.hvx_search_batch_device.restype = C.c_int
    L.hvx_search_batch_device.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, _vp, C.POINTER(Stats)]
    L.hvx_search_params_default.restype = None
    L.hvx_search_params_default.argtypes = [C.POINTER(_Params), C.c_uint32]
    L.hvx_simhash_config_default.restype = None
    L.hvx_simhash_config_default.argtypes = [C.POINTER(SimHashConfig)]
    L.hvx_index_set_simhash.restype = C.c_int
    L.hvx_index_set_simhash.argtypes = [_vp, C.POINTER(SimHashConfig), _vp]
    L.hvx_index_get_simhash.restype = C.c_int
    L.hvx_index_get_simhash.argtypes = [_vp, _vp]
    L.hvx_search_batch_params.restype = C.c_int
    L.hvx_search_batch_params.argtypes = [_vp, _vp, C.c_uint32, C.POINTER(_Params), _vp, _vp, _vp, _vp, C.POINTER(Stats), _vp, _vp]
    L.hvx_search_batch_params_device.restype = C.c_int
    L.hvx_search_batch_params_device.argtypes = [_vp, _vp, C.c_uint32, C.POINTER(_Params), _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(Stats)]
    L.hvx_adaptive_threshold_table.restype = C.c_int
    L.hvx_adaptive_threshold_table.argtypes = [C.c_uint32, C.c_float, _vp]
    L.hvx_flat_search_batch.restype = C.c_int
    L.hvx_flat_search_batch.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, C.POINTER(Stats)]
    L.hvx_flat_search_batch_device.restype = C.c_int
    L.hvx_flat_search_batch_device.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, C.POINTER(Stats)]
    L.hvx_search_restricted_batch.restype = C.c_int
    L.hvx_search_restricted_batch.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp, C.POINTER(Stats)]
    L.hvx_merge_topk_device.restype = C.c_int
    L.hvx_merge_topk_device.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]
    L.hvx_topk_payload_bytes.restype = C.c_size_t
    L.hvx_topk_payload_bytes.argtypes = [C.c_uint32, C.c_uint32]
    L.hvx_merge_topk_packed_device.restype = C.c_int
    L.hvx_merge_topk_packed_device.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]
    L.hvx_simhasher_new.restype = C.c_int
    L.hvx_simhasher_new.argtypes = [C.c_uint32, C.c_uint64, C.c_int32, C.POINTER(_vp)]
    L.hvx_simhasher_free.argtypes = [_vp]
    L.hvx_simhash_batch.restype = C.c_int
    L.hvx_simhash_batch.argtypes = [_vp, _vp, C.c_uint64, _vp]
    L.hvx_order_code_from_simhash_bits.restype = C.c_uint64
    L.hvx_order_code_from_simhash_bits.argtypes = [C.c_uint64]
    L.hvx_decode_layer0_row.restype = C.c_int
    L.hvx_decode_layer0_row.argtypes = [C.c_char_p, C.c_size_t, _vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.hvx_component_limit.restype = C.c_float
    L.hvx_component_limit.argtypes = [C.c_uint32, C.c_uint32]
    L.hvx_decode_simhash_row.restype = C.c_int
    L.hvx_decode_simhash_row.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.hvx_decode_entry_candidate_layer.restype = C.c_int
    L.hvx_decode_entry_candidate_layer.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]
    L.hvx_decode_upper_row.restype = C.c_int
    L.hvx_decode_upper_row.argtypes = [C.c_char_p, C.c_size_t, _vp, C.c_uint32, C.POINTER(C.c_uint32)]
    L.hvx_parse_vector_key.restype = C.c_uint32
    L.hvx_parse_vector_key.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.hvx_strip_tenant_envelope.restype = C.c_uint32
    L.hvx_strip_tenant_envelope.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.hvx_decode_index_metadata.restype = C.c_int
    L.hvx_decode_index_metadata.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(IndexMetadata)]
    L.hvx_hydrator_set_metadata.restype = C.c_int
    L.hvx_hydrator_set_metadata.argtypes = [_vp, C.c_char_p, C.c_size_t]
    L.hvx_hydrator_new.restype = C.c_int
    L.hvx_hydrator_new.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(_vp)]
    L.hvx_hydrator_free.argtypes = [_vp]
    L.hvx_hydrator_add_item.restype = C.c_int
    L.hvx_hydrator_add_item.argtypes = [_vp, C.c_uint64, C.c_char_p, C.c_size_t]
    L.hvx_hydrator_add_layer0_row.restype = C.c_int
    L.hvx_hydrator_add_layer0_row.argtypes = [_vp, C.c_uint64, C.c_char_p, C.c_size_t]
    L.hvx_hydrator_add_upper_row.restype = C.c_int
    L.hvx_hydrator_add_upper_row.argtypes = [_vp, C.c_uint64, C.c_uint32, C.c_char_p, C.c_size_t]
    L.hvx_hydrator_set_entry.restype = C.c_int
    L.hvx_hydrator_set_entry.argtypes = [_vp, C.c_uint64, C.c_uint32]
    L.hvx_hydrator_finish.restype = C.c_int
    L.hvx_hydrator_finish.argtypes = [_vp, C.POINTER(_Desc), C.POINTER(_vp)]
    L.hvx_prefilter_search_batch.restype = C.c_int
    L.hvx_prefilter_search_batch.argtypes = [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint32,
                                             C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp,
                                             C.POINTER(C.c_uint64), C.POINTER(Stats)]
    L.hvx_search_restricted_batch_params.restype = C.c_int
    L.hvx_search_restricted_batch_params.argtypes = [_vp, _vp, C.c_uint32, C.POINTER(RestrictedParams), _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp,
                                                     _vp, C.POINTER(Stats)]
    L.hvx_prefilter_search_batch_params.restype = C.c_int
    L.hvx_prefilter_search_batch_params.argtypes = [_vp, _vp, _vp, C.c_uint32, C.POINTER(RestrictedParams), C.c_uint32, _vp, C.c_uint32,
                                                    C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp,
                                                    C.POINTER(C.c_uint64), _vp, C.POINTER(Stats)]
    L.hvx_batcher_new.restype = C.c_int
    L.hvx_batcher_new.argtypes = [_vp, C.POINTER(_Params), C.c_uint32, C.c_uint32, C.POINTER(_vp)]
    L.hvx_batcher_free.argtypes = [_vp]
    L.hvx_batcher_new_restricted.restype = C.c_int
    L.hvx_batcher_new_restricted.argtypes = [_vp, C.POINTER(RestrictedParams), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_vp)]
    L.hvx_batcher_search_restricted.restype = C.c_int
    L.hvx_batcher_search_restricted.argtypes = [_vp, _vp, _vp, C.c_uint32, _vp, _vp, C.POINTER(C.c_uint32)]
    L.hvx_batcher_submit_restricted.restype = C.c_int
    L.hvx_batcher_submit_restricted.argtypes = [_vp, _vp, _vp, C.c_uint32, C.POINTER(BatcherTicket)]
    L.hvx_search_restricted_lists_device.restype = C.c_int
    L.hvx_search_restricted_lists_device.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32, _vp, C.c_uint32, _vp, _vp, _vp, _vp]
    L.hvx_batcher_search.restype = C.c_int
    L.hvx_batcher_search.argtypes = [_vp, _vp, _vp, _vp, C.POINTER(C.c_uint32)]
    L.hvx_batcher_submit.restype = C.c_int
    L.hvx_batcher_submit.argtypes = [_vp, _vp, C.POINTER(BatcherTicket)]
    L.hvx_batcher_poll.restype = C.c_int
    L.hvx_batcher_poll.argtypes = [_vp, C.POINTER(BatcherTicket), _vp, _vp, C.POINTER(C.c_uint32)]
    L.hvx_batcher_wait.restype = C.c_int
    L.hvx_batcher_wait.argtypes = [_vp, C.POINTER(BatcherTicket), C.c_uint32, _vp, _vp, C.POINTER(C.c_uint32)]
    L.hvx_batcher_eventfd.restype = C.c_int
    L.hvx_batcher_eventfd.argtypes = [_vp]
    L.hvx_batcher_stats.restype = C.c_int
    L.hvx_batcher_stats.argtypes = [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.hvx_batcher_lane_times.restype = C.c_int
    L.hvx_batcher_lane_times.argtypes = [_vp, C.POINTER(BatcherTimes)]
    L.hvx_csr_import.restype = C.c_int
    L.hvx_csr_import.argtypes = [C.c_uint64, C.c_uint64, _vp, _vp, _vp, C.c_int32, C.POINTER(_vp)]
    L.hvx_csr_free.argtypes = [_vp]
    L.hvx_traverse_filter.restype = C.c_int
    L.hvx_traverse_filter.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp]
    L.hvx_traverse_ordered.restype = C.c_int
    L.hvx_traverse_ordered.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.c_uint64,
                                       _vp, _vp, _vp, _vp, _vp, _vp]
    L.hvx_traverse_dfs.restype = C.c_int
    L.hvx_traverse_dfs.argtypes = L.hvx_traverse_ordered.argtypes
    L.hvx_traverse_host.restype = C.c_int
    L.hvx_traverse_host.argtypes = [C.c_uint64, C.c_uint64, _vp, _vp, _vp, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint32,
                                    C.c_uint32, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]
    L.hvx_expand_filter.restype = C.c_int
    L.hvx_expand_filter.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32, _vp]
    _lib = L
    return L


def device_stream_read_gbs(device=0, nbytes=3 << 30, iters=5):
    """(best, mean) GB/s of the library's read-only streaming kernel over an `nbytes` HBM buffer: the measured roofline denominator"""
    best, mean = C.c_float(0), C.c_float(0)
    _check(lib().hvx_device_stream_read_gbs(int(device), int(nbytes), int(iters), C.byref(best), C.byref(mean)))
    return float(best.value), float(mean.value)


def rccl_version():
    """NCCL_VERSION_CODE of the RCCL the library bound (0 = none / refused)"""
    return int(lib().hvx_shard_rccl_version())


def _sync_producer(t):
    """A device tensor handed to the library was produced on the caller's (torch) stream; the library reads it on the index's
    own stream.  Wait for the producer first: without this an import / build that starts right after the tensor's kernels were
    ENQUEUED copies rows that have not been written yet (whatever the allocation held before)."""
    try:
        import torch
        if isinstance(t, torch.Tensor) and t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
    except ImportError:
        pass


def _check(rc):
    if rc != OK:
        raise HelixDbError(rc, lib().hvx_last_error().decode())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


@dataclass(frozen=True)
class SearchResult:  # result.rs:20-41
    entity_id: int
    score: np.float32


class SearchParams:
    """mod.rs:410-621 SearchParams.

    `SearchParams.new(k)` is the reference's `SearchParams::new(k)`: ef = max(k, 100), SimHashMode::Adaptive, bypass
    tuning (24, 4, 0.12, 3) -- what the query path runs (access/search/storage.rs:140-141).  `SearchParams(k)` is the
    strict-exhaustive baseline the reference's golden tests and benches use:
    `SearchParams::new(k).with_simhash_mode(Off).with_pre_simhash_sampling_ratio(1.0)` (index.rs:2335-2342).
    """

    def __init__(self, k: int, *, _strict: bool = True):
        if k <= 0:
            raise HelixDbError(ERR_K_RANGE, "result count must be non-zero")
        self.k = int(k)
        self.ef = max(self.k, 100)
        self.simhash_mode = SIMHASH_OFF if _strict else SIMHASH_ADAPTIVE
        self.pre_simhash_sampling_ratio_override = 1.0 if _strict else None
        self.simhash_bypass_min_frontier, self.simhash_bypass_window_expansions = 24, 4
        self.simhash_bypass_min_filter_rate, self.simhash_read_budget_multiplier = 0.12, 3
        self.simhash_sampling_ratio_override = None
        self.simhash_failure_prob_override = None

    @classmethod
    def new(cls, k: int) -> "SearchParams":
        return cls(k, _strict=False)

    @classmethod
    def throughput_profile_floor_92(cls, k: int) -> "SearchParams":  # mod.rs:614-620
        return cls.new(k).with_ef(max(k, 48)).with_simhash_mode(SIMHASH_ADAPTIVE) \
            .with_pre_simhash_sampling_ratio(0.20).with_simhash_bypass_tuning(24, 4, 0.12, 3)

    def with_ef(self, ef: int) -> "SearchParams":
        if ef <= 0 or ef < self.k:
            raise HelixDbError(ERR_K_RANGE, f"search beam width {ef} is below the result count {self.k}")
        self.ef = int(ef)
        return self

    def with_simhash_mode(self, mode: int) -> "SearchParams":
        self.simhash_mode = int(mode)
        return self

    @staticmethod
    def _unit(value, what):  # parameters.rs:170-184
        v = float(value)
        if not np.isfinite(v) or not 0.0 <= v <= 1.0:
            raise HelixDbError(ERR_K_RANGE, f"{what} outside the closed unit interval")
        return v

    def with_pre_simhash_sampling_ratio(self, ratio: float) -> "SearchParams":
        self.pre_simhash_sampling_ratio_override = self._unit(ratio, "pre-sampling ratio")
        return self

    def clear_pre_simhash_sampling_ratio_override(self) -> "SearchParams":
        self.pre_simhash_sampling_ratio_override = None
        return self

    def with_simhash_bypass_tuning(self, min_frontier, window_expansions, min_filter_rate, read_budget_multiplier):
        if min(min_frontier, window_expansions, read_budget_multiplier) <= 0:
            raise HelixDbError(ERR_K_RANGE, "SimHash bypass tuning parameters must be non-zero")
        self.simhash_bypass_min_frontier, self.simhash_bypass_window_expansions = int(min_frontier), int(window_expansions)
        self.simhash_bypass_min_filter_rate = self._unit(min_filter_rate, "bypass filter rate")
        self.simhash_read_budget_multiplier = int(read_budget_multiplier)
        return self

    def with_simhash_sampling_ratio(self, ratio: float) -> "SearchParams":
        self.simhash_sampling_ratio_override = self._unit(ratio, "sampling ratio")
        return self

    def clear_simhash_sampling_ratio_override(self) -> "SearchParams":
        self.simhash_sampling_ratio_override = None
        return self

    def clear_simhash_failure_prob_override(self) -> "SearchParams":
        self.simhash_failure_prob_override = None
        return self

    def with_simhash_failure_prob(self, p: float) -> "SearchParams":
        p = float(p)
        if not np.isfinite(p) or not 0.0 < p < 1.0:  # parameters.rs:196-210
            raise HelixDbError(ERR_K_RANGE, "failure probability outside the open unit interval")
        self.simhash_failure_prob_override = p
        return self

    def requires_query_simhash(self) -> bool:  # mod.rs:546-552
        pre = self.pre_simhash_sampling_ratio_override
        return self.simhash_mode != SIMHASH_OFF or (pre is not None and pre < 1.0)

    def _c(self) -> "_Params":
        none = lambda v: -1.0 if v is None else float(v)
        return _Params(self.k, self.ef, self.simhash_mode, none(self.pre_simhash_sampling_ratio_override),
                       self.simhash_bypass_min_frontier, self.simhash_bypass_window_expansions,
                       self.simhash_bypass_min_filter_rate, self.simhash_read_budget_multiplier,
                       none(self.simhash_sampling_ratio_override), none(self.simhash_failure_prob_override))


class RestrictedVectorCandidates:
    """restricted.rs:345-371: dedupes ids, at most 1,000,000 unique candidates."""

    def __init__(self, ids: np.ndarray):
        self.ids = ids

    @classmethod
    def from_ids(cls, ids) -> "RestrictedVectorCandidates":
        a = np.unique(np.asarray(list(ids) if not isinstance(ids, np.ndarray) else ids, dtype=np.uint64))
        if a.size > 1_000_000:
            raise HelixDbError(ERR_CANDIDATE_LIMIT, "restricted vector search accepts at most 1000000 unique candidates")
        return cls(a)

    @classmethod
    def from_bitmap_words(cls, words: np.ndarray) -> "RestrictedVectorCandidates":
        bits = np.unpackbits(np.ascontiguousarray(words, dtype="<u8").view(np.uint8), bitorder="little")
        return cls.from_ids(np.nonzero(bits)[0].astype(np.uint64))

    def __len__(self):
        return int(self.ids.size)


class ValidatedVectorReadIndex:
    """Device-resident read index (read_index.rs:43-102 over VectorIndex<D>, index.rs:108-140)."""

    def __init__(self, handle, dim, metric, n):
        self._h = handle
        self.dim, self.metric, self.n = dim, metric, n

    @classmethod
    def managed(cls, *, dim, metric, node_ids, vectors, l0_offsets, l0_neighbors, level=None,
                up_offsets=None, up_neighbors=None, entry_point=None, max_layer=0, m=16, m0=32,
                float_kernel=KERNEL_AVX_FMA, device=-1, max_batch=1024, dtype=F32, reserve_rows=0, reserve_upper_rows=0):
        """hvx_index_import; with reserve_rows / reserve_upper_rows: hvx_index_import_reserve (a hydrated image that can grow)"""
        ids = np.ascontiguousarray(node_ids, dtype=np.uint64)
        dev_rows = hasattr(vectors, "data_ptr")  # a torch tensor already resident on the device
        if dev_rows:
            _sync_producer(vectors)
        if dev_rows:
            assert vectors.is_contiguous() and vectors.dtype.is_floating_point and vectors.element_size() == 4 and vectors.numel() == ids.size * dim
            vec = None
        else:
            vec = np.ascontiguousarray(vectors, dtype=np.float32).reshape(ids.size, dim) if ids.size else np.zeros((0, dim), np.float32)
        o0 = np.ascontiguousarray(l0_offsets, dtype=np.uint64)
        n0 = np.ascontiguousarray(l0_neighbors, dtype=np.uint64)
        lv = None if level is None else np.ascontiguousarray(level, dtype=np.uint16)
        uo = None if up_offsets is None else np.ascontiguousarray(up_offsets, dtype=np.uint64)
        un = None if up_neighbors is None else np.ascontiguousarray(up_neighbors, dtype=np.uint64)
        d = _Desc(dim=dim, metric=metric, dtype=dtype, float_kernel=float_kernel, n=ids.size, m=m, m0=m0,
                  has_entry=0 if entry_point is None else 1, max_layer=max_layer,
                  entry_point=0 if entry_point is None else int(entry_point),
                  shard_id_lo=int(ids[0]) if ids.size else 0, shard_id_hi=int(ids[-1]) if ids.size else 0,
                  device=device, max_batch=max_batch)
        h = _vp()
        vp = _vp(vectors.data_ptr()) if dev_rows else _ptr(vec)
        if reserve_rows or reserve_upper_rows:
            _check(lib().hvx_index_import_reserve(C.byref(d), _ptr(ids), vp, _ptr(o0), _ptr(n0), _ptr(lv), _ptr(uo), _ptr(un),
                                                  C.c_uint64(int(reserve_rows)), C.c_uint64(int(reserve_upper_rows)), C.byref(h)))
        else:
            _check(lib().hvx_index_import(C.byref(d), _ptr(ids), vp, _ptr(o0), _ptr(n0), _ptr(lv), _ptr(uo), _ptr(un), C.byref(h)))
        return cls(h, dim, metric, int(ids.size))

    @classmethod
    def build(cls, *, dim, metric, node_ids, vectors, levels=None, m=16, m0=32, ef_construction=200, max_batch=2048,
              batch_divisor=32, sequential=False, device=-1, search_max_batch=None, float_kernel=KERNEL_AVX_FMA, link_mode=0, scatter=False,
              reserve_rows=0, reserve_upper_rows=None):
        """GPU-assisted HNSW build (hvx_index_build): the reference's insert_hnsw for batches of nodes on the device.
        Returns (index, stats dict).  `vectors` may be a host array or a torch tensor resident on the device."""
        ids = np.ascontiguousarray(node_ids, dtype=np.uint64)
        dev_rows = hasattr(vectors, "data_ptr")
        if dev_rows:
            _sync_producer(vectors)
        vec = None if dev_rows else (np.ascontiguousarray(vectors, dtype=np.float32).reshape(ids.size, dim) if ids.size else np.zeros((0, dim), np.float32))
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.uint16)
        d = _Desc(dim=dim, metric=metric, dtype=F32, float_kernel=float_kernel, n=ids.size, m=m, m0=m0, has_entry=0, max_layer=0,
                  entry_point=0, shard_id_lo=int(ids[0]) if ids.size else 0, shard_id_hi=int(ids[-1]) if ids.size else 0,
                  device=device, max_batch=max(max_batch, search_max_batch or 1024))
        if reserve_upper_rows is None:  # the expected number of upper rows of `reserve_rows` nodes under the layer rule, with slack
            reserve_upper_rows = int(reserve_rows / max(m - 1, 1) * 1.5) + (64 if reserve_rows else 0)
        bp = BuildParams(ef_construction=ef_construction, max_batch=max_batch, batch_divisor=batch_divisor, sequential=1 if sequential else 0,
                         link_mode=link_mode, scatter=1 if scatter else 0, reserve_rows=int(reserve_rows), reserve_upper_rows=int(reserve_upper_rows))
        st = BuildStats()
        h = _vp()
        _check(lib().hvx_index_build(C.byref(d), _ptr(ids), _vp(vectors.data_ptr()) if dev_rows else _ptr(vec), _ptr(lv), C.byref(bp),
                                     C.byref(h), C.byref(st)))
        return cls(h, dim, metric, int(ids.size)), {"nodes": st.nodes, "batches": st.batches, "single_node_batches": st.single_node_batches}

    def insert_batch(self, node_ids, vectors, levels=None, *, ef_construction=200, max_batch=2048, batch_divisor=32, sequential=False, link_mode=0):
        """hvx_index_insert_batch: append rows to a growable image (build(..., reserve_rows=...)) and link them into the graph; the new
        generation is visible on this handle when the call returns (forks: refresh()).  Returns the build-stats dict of the batch."""
        ids = np.ascontiguousarray(node_ids, dtype=np.uint64)
        dev_rows = hasattr(vectors, "data_ptr")
        if dev_rows:
            _sync_producer(vectors)
        vec = None if dev_rows else np.ascontiguousarray(vectors, dtype=np.float32).reshape(ids.size, self.dim)
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.uint16)
        bp = BuildParams(ef_construction=ef_construction, max_batch=max_batch, batch_divisor=batch_divisor, sequential=1 if sequential else 0,
                         link_mode=link_mode)
        st = BuildStats()
        L = lib()
        L.hvx_index_insert_batch.restype = C.c_int
        L.hvx_index_insert_batch.argtypes = [_vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(BuildParams), C.POINTER(BuildStats)]
        _check(L.hvx_index_insert_batch(self._h, _ptr(ids), _vp(vectors.data_ptr()) if dev_rows else _ptr(vec), _ptr(lv), int(ids.size), C.byref(bp), C.byref(st)))
        self.n = self.rows()
        return {"nodes": st.nodes, "batches": st.batches, "single_node_batches": st.single_node_batches}

    def upsert_batch(self, node_ids, vectors, levels=None, *, ef_construction=200):
        """hvx_index_upsert_batch: VectorInsertContract::Upsert -- a live id is deleted first, then the vector is linked in under the same
        id (its own row slot, or appended when the id is above the image's); sequential semantics"""
        ids = np.ascontiguousarray(node_ids, dtype=np.uint64).reshape(-1)
        vec = np.ascontiguousarray(vectors, dtype=np.float32).reshape(ids.size, self.dim)
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.uint16).reshape(-1)
        bp = BuildParams()
        lib().hvx_build_params_default(C.byref(bp))
        bp.ef_construction = int(ef_construction)
        st = BuildStats()
        L = lib()
        L.hvx_index_upsert_batch.restype = C.c_int
        L.hvx_index_upsert_batch.argtypes = [_vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(BuildParams), C.POINTER(BuildStats)]
        _check(L.hvx_index_upsert_batch(self._h, _ptr(ids), _ptr(vec), _ptr(lv), int(ids.size), C.byref(bp), C.byref(st)))
        self.n = self.rows()
        return {"nodes": st.nodes, "batches": st.batches, "single_node_batches": st.single_node_batches}

    def delete_batch(self, node_ids) -> dict:
        """hvx_index_delete_batch: VectorIndex::delete (mutation.rs:1606-2055) for every id in order -- unlink, relink the affected
        sources, repair the entry point; ids that are not in the image succeed and count as `missing` (index.rs:2263)"""
        ids = np.ascontiguousarray(node_ids, dtype=np.uint64).reshape(-1)
        st = DeleteStats()
        L = lib()
        L.hvx_index_delete_batch.restype = C.c_int
        L.hvx_index_delete_batch.argtypes = [_vp, _vp, C.c_uint32, C.POINTER(DeleteStats)]
        _check(L.hvx_index_delete_batch(self._h, _ptr(ids), int(ids.size), C.byref(st)))
        return {"requested": st.requested, "deleted": st.deleted, "missing": st.missing, "entry_moves": st.entry_moves,
                "relinked_rows": st.relinked_rows, "seconds": st.seconds}

    def live_rows(self) -> int:
        L = lib()
        L.hvx_index_live_rows.restype = C.c_uint64
        L.hvx_index_live_rows.argtypes = [_vp]
        return int(L.hvx_index_live_rows(self._h))

    def contains(self, node_id) -> bool:
        L = lib()
        L.hvx_index_contains.restype = C.c_int
        L.hvx_index_contains.argtypes = [_vp, C.c_uint64]
        return bool(L.hvx_index_contains(self._h, int(node_id)))

    def refresh(self):
        """hvx_index_refresh: a fork adopts the image's visible generation (rows, entry point, top layer)"""
        L = lib()
        L.hvx_index_refresh.restype = C.c_int
        L.hvx_index_refresh.argtypes = [_vp]
        _check(L.hvx_index_refresh(self._h))
        self.n = self.rows()
        return self

    def rows(self) -> int:
        L = lib()
        L.hvx_index_rows.restype = C.c_uint64
        L.hvx_index_rows.argtypes = [_vp]
        return int(L.hvx_index_rows(self._h))

    def row_capacity(self) -> int:
        L = lib()
        L.hvx_index_row_capacity.restype = C.c_uint64
        L.hvx_index_row_capacity.argtypes = [_vp]
        return int(L.hvx_index_row_capacity(self._h))

    def visible_seq(self) -> int:
        L = lib()
        L.hvx_index_visible_seq.restype = C.c_uint64
        L.hvx_index_visible_seq.argtypes = [_vp]
        return int(L.hvx_index_visible_seq(self._h))

    def export_graph(self) -> dict:
        """The index's graph in hvx_index_import's CSR layout (external ids): what the host persists / tests compare."""
        e0, ur, eu, ep = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        ml, he = C.c_uint32(0), C.c_uint32(0)
        _check(lib().hvx_index_graph_sizes(self._h, C.byref(e0), C.byref(ur), C.byref(eu), C.byref(ep), C.byref(ml), C.byref(he)))
        l0o = np.zeros(self.n + 1, np.uint64); l0n = np.zeros(max(e0.value, 1), np.uint64)
        lv = np.zeros(max(self.n, 1), np.uint16)
        uo = np.zeros(ur.value + 1, np.uint64); un = np.zeros(max(eu.value, 1), np.uint64)
        _check(lib().hvx_index_export_graph(self._h, _ptr(l0o), _ptr(l0n), _ptr(lv), _ptr(uo), _ptr(un)))
        return {"l0_offsets": l0o, "l0_neighbors": l0n[: e0.value], "level": lv[: self.n], "up_offsets": uo, "up_neighbors": un[: eu.value],
                "entry_point": int(ep.value) if he.value else None, "max_layer": int(ml.value)}

    def link_rows(self, from_ids, to_ids, concurrent=False):
        """add_bidirectional_link(from -> to) on layer 0 through the batched build's link kernel (hvx_index_link_rows)."""
        f = np.ascontiguousarray(from_ids, dtype=np.uint64)
        t = np.ascontiguousarray(to_ids, dtype=np.uint64)
        assert f.size == t.size
        _check(lib().hvx_index_link_rows(self._h, _ptr(f), _ptr(t), f.size, 1 if concurrent else 0))

    def audit_graph(self) -> dict:
        """Row invariants, symmetry on every layer and layer-0 reachability of the graph image, counted on the device."""
        a = GraphAudit()
        _check(lib().hvx_index_audit_graph(self._h, C.byref(a)))
        return a.as_dict()

    @classmethod
    def from_export(cls, ex: dict, *, dim, metric, **kw):
        """Build from the dict produced by the oracle's Index.export() (tests) or the GPU builder."""
        return cls.managed(dim=dim, metric=metric, node_ids=ex["node_ids"], vectors=ex["vectors"],
                           l0_offsets=ex["l0_offsets"], l0_neighbors=ex["l0_neighbors"], level=ex["level"],
                           up_offsets=ex["up_offsets"], up_neighbors=ex["up_neighbors"],
                           entry_point=ex["entry_point"], max_layer=ex["max_layer"], **kw)

    def close(self):
        if self._h:
            lib().hvx_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- single-query surface, as the reference's operator calls it ----
    def search(self, query, params: SearchParams):
        ids, sc, cnt, _ = self.search_batch(np.asarray(query, np.float32).reshape(1, -1), params)
        return [SearchResult(int(i), s) for i, s in zip(ids[0, :cnt[0]], sc[0, :cnt[0]])]

    def search_restricted(self, query, params: SearchParams, candidates: RestrictedVectorCandidates):
        ids, sc, cnt = self.search_restricted_batch(np.asarray(query, np.float32).reshape(1, -1), params, candidates)
        return [SearchResult(int(i), s) for i, s in zip(ids[0, :cnt[0]], sc[0, :cnt[0]])]

    # ---- batched surface ----
    def _q(self, queries):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise HelixDbError(ERR_DIMENSION, f"expected dimension {self.dim}, got {q.shape[-1] if q.ndim else 0}")
        return q

    def search_batch(self, queries, params: SearchParams, per_query_status=False):
        q = self._q(queries)
        b, k = q.shape[0], params.k
        ids = np.zeros((b, k), np.uint64); sc = np.zeros((b, k), np.float32)
        cnt = np.zeros(b, np.uint32); st = np.zeros(b, np.uint32)
        stats = Stats()
        if params.requires_query_simhash():
            _check(lib().hvx_search_batch_params(self._h, _ptr(q), b, C.byref(params._c()), _ptr(ids), _ptr(sc), _ptr(cnt),
                                                 _ptr(st) if per_query_status else None, C.byref(stats), None, None))
        else:
            _check(lib().hvx_search_batch(self._h, _ptr(q), b, k, params.ef, _ptr(ids), _ptr(sc), _ptr(cnt),
                                          _ptr(st) if per_query_status else None, C.byref(stats)))
        return (ids, sc, cnt, stats.as_dict()) if not per_query_status else (ids, sc, cnt, stats.as_dict(), st)

    def search_batch_with_stats(self, queries, params: SearchParams):
        """search_with_stats (index.rs:1589-1611) for a batch: results + per-query SearchStats counters of both arms."""
        q = self._q(queries)
        b, k = q.shape[0], params.k
        ids = np.zeros((b, k), np.uint64); sc = np.zeros((b, k), np.float32)
        cnt = np.zeros(b, np.uint32); st = np.zeros(b, np.uint32)
        qs = (QueryStats * b)(); ad = (AdaptiveStats * b)()
        stats = Stats()
        _check(lib().hvx_search_batch_params(self._h, _ptr(q), b, C.byref(params._c()), _ptr(ids), _ptr(sc), _ptr(cnt), _ptr(st),
                                             C.byref(stats), C.cast(qs, _vp), C.cast(ad, _vp)))
        per_query = []
        for i in range(b):
            d = {f: int(getattr(qs[i], f)) for f, _ in QueryStats._fields_}
            d.update(ad[i].as_dict())
            per_query.append(d)
        return ids, sc, cnt, st, per_query, stats.as_dict()

    def set_simhash(self, config: "SimHashConfig" = None, node_hashes=None):
        """Attach the per-node SimHash rows (computed on the device from the f32 rows when node_hashes is None)."""
        cfg = config if config is not None else SimHashConfig.default()
        h = None if node_hashes is None else np.ascontiguousarray(node_hashes, dtype=np.uint64)
        if h is not None and h.size != self.n:
            raise HelixDbError(ERR_INVARIANT, "one SimHash row per node expected")
        _check(lib().hvx_index_set_simhash(self._h, C.byref(cfg), _ptr(h)))
        return self

    def get_simhash(self) -> np.ndarray:
        out = np.zeros(self.n, np.uint64)
        _check(lib().hvx_index_get_simhash(self._h, _ptr(out)))
        return out

    def search_batch_params_device(self, d_queries, params: SearchParams, d_ids, d_scores, d_counts, d_status, d_qstats=None,
                                   d_astats=None, want_stats=False):
        stats = Stats()
        _check(lib().hvx_search_batch_params_device(
            self._h, d_queries.data_ptr(), d_queries.shape[0], C.byref(params._c()), d_ids.data_ptr(), d_scores.data_ptr(),
            d_counts.data_ptr(), d_status.data_ptr(), None if d_qstats is None else d_qstats.data_ptr(),
            None if d_astats is None else d_astats.data_ptr(), C.byref(stats) if want_stats else None))
        return stats.as_dict() if want_stats else None

    def flat_search_batch(self, queries, k, per_query_status=False):
        q = self._q(queries)
        b = q.shape[0]
        ids = np.zeros((b, k), np.uint64); sc = np.zeros((b, k), np.float32)
        cnt = np.zeros(b, np.uint32); st = np.zeros(b, np.uint32)
        stats = Stats()
        _check(lib().hvx_flat_search_batch(self._h, _ptr(q), b, k, _ptr(ids), _ptr(sc), _ptr(cnt),
                                           _ptr(st) if per_query_status else None, C.byref(stats)))
        return (ids, sc, cnt, stats.as_dict()) if not per_query_status else (ids, sc, cnt, stats.as_dict(), st)

    def search_restricted_batch(self, queries, params: SearchParams, candidates, offsets=None):
        q = self._q(queries)
        b, k = q.shape[0], params.k
        al = candidates.ids if isinstance(candidates, RestrictedVectorCandidates) else np.ascontiguousarray(candidates, dtype=np.uint64)
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
        ids = np.zeros((b, k), np.uint64); sc = np.zeros((b, k), np.float32); cnt = np.zeros(b, np.uint32)
        _check(lib().hvx_search_restricted_batch(self._h, _ptr(q), b, k, params.ef, _ptr(al), _ptr(off), al.size,
                                                 _ptr(ids), _ptr(sc), _ptr(cnt), None, None))
        return ids, sc, cnt

    def search_restricted_batch_params(self, queries, rparams: "RestrictedParams", candidates, offsets=None, want_stats=False):
        """search_restricted under the reference's execution plan (hvx_search_restricted_batch_params): exact scan for small
        candidate sets, the filter-aware walk otherwise.  Returns (ids, scores, counts, status, [RestrictedSearchStats dicts])
        (+ the hvx_stats dict when want_stats)."""
        q = self._q(queries)
        b, k = q.shape[0], int(rparams.k)
        al = candidates.ids if isinstance(candidates, RestrictedVectorCandidates) else np.ascontiguousarray(candidates, dtype=np.uint64)
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
        ids = np.zeros((b, k), np.uint64); sc = np.zeros((b, k), np.float32); cnt = np.zeros(b, np.uint32); st = np.zeros(b, np.uint32)
        rs = (RestrictedStats * b)()
        stats = Stats()
        _check(lib().hvx_search_restricted_batch_params(self._h, _ptr(q), b, C.byref(rparams), _ptr(al), _ptr(off), al.size, _ptr(ids), _ptr(sc),
                                                        _ptr(cnt), _ptr(st), C.cast(rs, _vp), C.byref(stats) if want_stats else None))
        out = (ids, sc, cnt, st, [r.as_dict() for r in rs])
        return out + (stats.as_dict(),) if want_stats else out

    def prefilter_search_batch_params(self, graph: "Graph", queries, rparams: "RestrictedParams", seeds, *, traverse=False, max_depth=1,
                                      direction=DIR_OUT, allowed_labels=(), hub_degree=0, include_seeds=True):
        """the fused hop + restricted kNN under the reference's execution plan; returns (ids, scores, counts, n_candidates,
        [RestrictedSearchStats dicts], hvx_stats dict)"""
        q = self._q(queries)
        b, k = q.shape[0], int(rparams.k)
        s = np.ascontiguousarray(seeds, dtype=np.uint64)
        lab = np.ascontiguousarray(allowed_labels, dtype=np.uint32)
        ids = np.zeros((b, k), np.uint64); sc = np.zeros((b, k), np.float32); cnt = np.zeros(b, np.uint32)
        ncand = C.c_uint64(0)
        rs = (RestrictedStats * b)()
        stats = Stats()
        _check(lib().hvx_prefilter_search_batch_params(self._h, graph._h, _ptr(q), b, C.byref(rparams), 1 if traverse else 0, _ptr(s), s.size,
                                                       max_depth, direction, _ptr(lab) if lab.size else None, lab.size, hub_degree,
                                                       1 if include_seeds else 0, _ptr(ids), _ptr(sc), _ptr(cnt), None, C.byref(ncand),
                                                       C.cast(rs, _vp), C.byref(stats)))
        return ids, sc, cnt, int(ncand.value), [r.as_dict() for r in rs], stats.as_dict()

    def prefilter_search_batch(self, graph: "Graph", queries, params: SearchParams, seeds, *, traverse=False, max_depth=1,
                               direction=DIR_OUT, allowed_labels=(), hub_degree=0, include_seeds=True):
        """`where_()` / traversal filter + vector_search in ONE call: the hop's candidate bitmap stays on the device
        (hvx_prefilter_search_batch).  Returns (ids, scores, counts, n_candidates, stats)."""
        q = self._q(queries)
        b, k = q.shape[0], params.k
        s = np.ascontiguousarray(seeds, dtype=np.uint64)
        lab = np.ascontiguousarray(allowed_labels, dtype=np.uint32)
        ids = np.zeros((b, k), np.uint64); sc = np.zeros((b, k), np.float32); cnt = np.zeros(b, np.uint32)
        ncand = C.c_uint64(0)
        stats = Stats()
        _check(lib().hvx_prefilter_search_batch(self._h, graph._h, _ptr(q), b, k, params.ef, 1 if traverse else 0, _ptr(s), s.size,
                                                max_depth, direction, _ptr(lab) if lab.size else None, lab.size, hub_degree,
                                                1 if include_seeds else 0, _ptr(ids), _ptr(sc), _ptr(cnt), None,
                                                C.byref(ncand), C.byref(stats)))
        return ids, sc, cnt, int(ncand.value), stats.as_dict()

    # ---- device-resident surface (torch tensors on the index's GPU) ----
    def search_batch_device(self, d_queries, k, ef, d_ids, d_scores, d_counts, d_status, d_qstats=None, want_stats=False):
        stats = Stats()
        _check(lib().hvx_search_batch_device(
            self._h, d_queries.data_ptr(), d_queries.shape[0], k, ef, d_ids.data_ptr(), d_scores.data_ptr(),
            d_counts.data_ptr(), d_status.data_ptr(), None if d_qstats is None else d_qstats.data_ptr(),
            C.byref(stats) if want_stats else None))
        return stats.as_dict() if want_stats else None

    def flat_search_batch_device(self, d_queries, k, d_ids, d_scores, d_counts, d_status, want_stats=False):
        stats = Stats()
        _check(lib().hvx_flat_search_batch_device(
            self._h, d_queries.data_ptr(), d_queries.shape[0], k, d_ids.data_ptr(), d_scores.data_ptr(),
            d_counts.data_ptr(), d_status.data_ptr(), C.byref(stats) if want_stats else None))
        return stats.as_dict() if want_stats else None

    def merge_topk_device(self, g, b, k, d_ids, d_scores, d_counts, d_out_ids, d_out_scores, d_out_counts):
        _check(lib().hvx_merge_topk_device(self._h, g, b, k, d_ids.data_ptr(), d_scores.data_ptr(), d_counts.data_ptr(),
                                           d_out_ids.data_ptr(), d_out_scores.data_ptr(), d_out_counts.data_ptr()))

    def merge_topk_packed_device(self, g, b, k, d_packed, d_out_ids, d_out_scores, d_out_counts):
        _check(lib().hvx_merge_topk_packed_device(self._h, g, b, k, d_packed.data_ptr(), d_out_ids.data_ptr(),
                                                  d_out_scores.data_ptr(), d_out_counts.data_ptr()))

    def timing_begin(self, capacity: int):
        _check(lib().hvx_index_timing_begin(self._h, capacity))

    def wave_clocks(self, capacity: int, max_batch: int) -> np.ndarray:
        """[launch][query][2] start / end of every query's wavefront (100 MHz device clock) since timing_begin"""
        out = np.zeros((capacity, max_batch, 2), np.uint64)
        n = C.c_uint32(0)
        _check(lib().hvx_index_wave_clocks(self._h, _ptr(out), capacity, max_batch, C.byref(n)))
        return out[: n.value].copy()

    def timing_collect(self, capacity: int) -> np.ndarray:
        """kernel durations (ms) of the un-synchronised search calls since timing_begin, in call order"""
        ms = np.zeros(capacity, np.float32)
        n = C.c_uint32(0)
        _check(lib().hvx_index_timing_collect(self._h, _ptr(ms), capacity, C.byref(n)))
        return ms[: n.value].copy()

    def sync(self):
        _check(lib().hvx_index_sync(self._h))

    def fork(self) -> "ValidatedVectorReadIndex":
        """Another execution lane on the same device image (own stream + scratch); see hvx_index_fork."""
        h = _vp()
        _check(lib().hvx_index_fork(self._h, C.byref(h)))
        return type(self)(h, self.dim, self.metric, self.n)

    def read_rows_device(self, row0: int, n: int, d_out):
        """hvx_index_read_rows_device: the stored (rounded / dequantised) f32 values of rows [row0, row0 + n) into a device tensor"""
        lib().hvx_index_read_rows_device.restype = C.c_int
        lib().hvx_index_read_rows_device.argtypes = [_vp, C.c_uint64, C.c_uint64, _vp]
        _check(lib().hvx_index_read_rows_device(self._h, int(row0), int(n), _vp(d_out.data_ptr())))

    def set_option(self, option: int, value: int):
        """hvx_index_set_option: pin an execution path of this handle (tests, A/B measurements); results do not change"""
        lib().hvx_index_set_option.restype = C.c_int
        lib().hvx_index_set_option.argtypes = [_vp, C.c_uint32, C.c_uint32]
        _check(lib().hvx_index_set_option(self._h, option, value))

    def last_scan_path(self) -> int:
        """hvx_scan_path flags of this handle's last exact scan"""
        lib().hvx_index_last_scan_path.restype = C.c_uint32
        lib().hvx_index_last_scan_path.argtypes = [_vp]
        return int(lib().hvx_index_last_scan_path(self._h))

    def set_occupancy(self, queries_per_simd: int):
        _check(lib().hvx_index_set_occupancy(self._h, queries_per_simd))

    def stream(self) -> int:
        return int(lib().hvx_index_stream(self._h) or 0)

    def set_stream(self, hip_stream):
        """Enqueue on a caller-owned hipStream_t (int handle, e.g. torch.cuda.current_stream().cuda_stream)."""
        _check(lib().hvx_index_set_stream(self._h, _vp(hip_stream) if hip_stream else None))


class ShardGroup:
    """hvx_shard_group: local search -> RCCL all-gather of the packed per-shard top-k -> merge, enqueued on the shard's stream
    by ONE C-ABI call.  `unique_id` = 128 bytes from ShardGroup.unique_id() on rank 0, handed to every rank by
    the host (bench.py uses a torch.distributed broadcast); world == 1 with an id still runs the collective (tests)."""

    def __init__(self, index: "ValidatedVectorReadIndex", unique_id, rank: int, world: int, max_batch: int, max_k: int, _attach_to=None):
        self._ix = index  # keeps the shard handle alive
        self._g = _vp()
        if _attach_to is not None:
            L = lib()
            L.hvx_shard_group_attach.restype = C.c_int
            L.hvx_shard_group_attach.argtypes = [_vp, _vp, C.POINTER(_vp)]
            _check(L.hvx_shard_group_attach(_attach_to._g, index._h, C.byref(self._g)))
            return
        uid = None if unique_id is None else np.frombuffer(bytes(unique_id), np.uint8).copy()
        _check(lib().hvx_shard_group_init(index._h, _ptr(uid), rank, world, max_batch, max_k, C.byref(self._g)))

    def attach(self, lane: "ValidatedVectorReadIndex") -> "ShardGroup":
        """hvx_shard_group_attach: another execution lane of this rank on the SAME communicator (one exchange stream per rank)"""
        return ShardGroup(lane, None, 0, 1, 0, 0, _attach_to=self)

    @staticmethod
    def unique_id() -> bytes:
        out = np.zeros(128, np.uint8)
        _check(lib().hvx_shard_group_unique_id(_ptr(out)))
        return out.tobytes()

    @staticmethod
    def _st(d_status):
        return None if d_status is None else _vp(d_status.data_ptr())

    def wait(self, timeout_ms: int) -> bool:
        """hvx_shard_group_wait: True once the group's last step has drained, False after timeout_ms (a peer that never joined the collective)"""
        L = lib()
        L.hvx_shard_group_wait.restype = C.c_int
        L.hvx_shard_group_wait.argtypes = [_vp, C.c_uint32]
        rc = L.hvx_shard_group_wait(self._g, int(timeout_ms))
        if rc == ERR_TIMEOUT:
            return False
        _check(rc)
        return True

    def abort(self):
        """hvx_shard_group_abort: ncclCommAbort -- the pending collective is cancelled; later steps fail until the group is re-formed"""
        L = lib()
        L.hvx_shard_group_abort.restype = C.c_int
        L.hvx_shard_group_abort.argtypes = [_vp]
        _check(L.hvx_shard_group_abort(self._g))

    def search_batch_device(self, d_queries, k, ef, d_ids, d_scores, d_counts, d_status=None):
        L = lib()
        L.hvx_shard_group_search_batch_device.restype = C.c_int
        L.hvx_shard_group_search_batch_device.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]
        _check(L.hvx_shard_group_search_batch_device(self._g, d_queries.data_ptr(), d_queries.shape[0], k, ef, d_ids.data_ptr(),
                                                     d_scores.data_ptr(), d_counts.data_ptr(), self._st(d_status)))

    def search_batch_params_device(self, d_queries, params: "SearchParams", d_ids, d_scores, d_counts, d_status=None):
        L = lib()
        L.hvx_shard_group_search_batch_params_device.restype = C.c_int
        L.hvx_shard_group_search_batch_params_device.argtypes = [_vp, _vp, C.c_uint32, C.POINTER(_Params), _vp, _vp, _vp, _vp]
        cp = params._c()
        _check(L.hvx_shard_group_search_batch_params_device(self._g, d_queries.data_ptr(), d_queries.shape[0], C.byref(cp), d_ids.data_ptr(),
                                                            d_scores.data_ptr(), d_counts.data_ptr(), self._st(d_status)))

    def flat_search_batch_device(self, d_queries, k, d_ids, d_scores, d_counts, d_status=None):
        L = lib()
        L.hvx_shard_group_flat_search_batch_device.restype = C.c_int
        L.hvx_shard_group_flat_search_batch_device.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]
        _check(L.hvx_shard_group_flat_search_batch_device(self._g, d_queries.data_ptr(), d_queries.shape[0], k, d_ids.data_ptr(),
                                                          d_scores.data_ptr(), d_counts.data_ptr(), self._st(d_status)))

    def search_restricted_batch(self, queries, rparams: "RestrictedParams", candidates):
        """every rank passes the same candidate ids; returns (ids, scores, counts, status) of the merged answer"""
        L = lib()
        L.hvx_shard_group_search_restricted_batch.restype = C.c_int
        L.hvx_shard_group_search_restricted_batch.argtypes = [_vp, _vp, C.c_uint32, C.POINTER(RestrictedParams), _vp, C.c_uint64, _vp, _vp, _vp, _vp]
        q = np.ascontiguousarray(queries, dtype=np.float32)
        b, k = q.shape[0], int(rparams.k)
        al = candidates.ids if isinstance(candidates, RestrictedVectorCandidates) else np.ascontiguousarray(candidates, dtype=np.uint64)
        ids = np.zeros((b, k), np.uint64); sc = np.zeros((b, k), np.float32); cnt = np.zeros(b, np.uint32); st = np.zeros(b, np.uint32)
        _check(L.hvx_shard_group_search_restricted_batch(self._g, _ptr(q), b, C.byref(rparams), _ptr(al), al.size, _ptr(ids), _ptr(sc), _ptr(cnt), _ptr(st)))
        return ids, sc, cnt, st

    def close(self):
        if self._g:
            lib().hvx_shard_group_free(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batcher:
    """hvx_batcher: coalesces concurrent single-query `search` calls (one per operator invocation in the reference,
    access/search/storage.rs:140-163) into batched launches.  `search` may be called from many threads."""

    def __init__(self, index: "ValidatedVectorReadIndex", params: SearchParams, max_batch: int = 0, max_wait_us: int = 0):
        self._index = index  # keeps the index alive
        self.k = params.k
        h = _vp()
        _check(lib().hvx_batcher_new(index._h, C.byref(params._c()), max_batch, max_wait_us, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().hvx_batcher_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search(self, query):
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        if q.size != self._index.dim:
            raise HelixDbError(ERR_DIMENSION, f"expected dimension {self._index.dim}, got {q.size}")
        ids = np.zeros(self.k, np.uint64); sc = np.zeros(self.k, np.float32); cnt = C.c_uint32(0)
        _check(lib().hvx_batcher_search(self._h, _ptr(q), _ptr(ids), _ptr(sc), C.byref(cnt)))
        return [SearchResult(int(i), s) for i, s in zip(ids[: cnt.value], sc[: cnt.value])]

    # ---- the non-blocking form (hvx_batcher_submit / _poll / _wait / _eventfd) ----
    def submit(self, query):
        """-> ticket, or None when the open batch is full (HVX_ERR_BUSY: submit again after a completion)"""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        if q.size != self._index.dim:
            raise HelixDbError(ERR_DIMENSION, f"expected dimension {self._index.dim}, got {q.size}")
        t = BatcherTicket()
        rc = lib().hvx_batcher_submit(self._h, _ptr(q), C.byref(t))
        if rc == ERR_BUSY:
            return None
        _check(rc)
        return t

    def _take(self, rc, ids, sc, cnt):
        if rc == PENDING:
            return None
        _check(rc)
        return [SearchResult(int(i), s) for i, s in zip(ids[: cnt.value], sc[: cnt.value])]

    def poll(self, ticket):
        """-> results, or None while the ticket's batch is in flight; raises the query's own rejection"""
        ids = np.zeros(self.k, np.uint64); sc = np.zeros(self.k, np.float32); cnt = C.c_uint32(0)
        return self._take(lib().hvx_batcher_poll(self._h, C.byref(ticket), _ptr(ids), _ptr(sc), C.byref(cnt)), ids, sc, cnt)

    def wait(self, ticket, timeout_us=0xFFFFFFFF):
        ids = np.zeros(self.k, np.uint64); sc = np.zeros(self.k, np.float32); cnt = C.c_uint32(0)
        return self._take(lib().hvx_batcher_wait(self._h, C.byref(ticket), int(timeout_us), _ptr(ids), _ptr(sc), C.byref(cnt)), ids, sc, cnt)

    def refresh(self):
        """hvx_batcher_refresh: the lanes adopt rows appended to the image since the batcher was created"""
        L = lib()
        L.hvx_batcher_refresh.restype = C.c_int
        L.hvx_batcher_refresh.argtypes = [_vp]
        _check(L.hvx_batcher_refresh(self._h))

    def eventfd(self):
        fd = lib().hvx_batcher_eventfd(self._h)
        if fd < 0:
            raise HelixDbError(ERR_DEVICE, lib().hvx_last_error().decode())
        return fd

    def stats(self):
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().hvx_batcher_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"batches": int(a.value), "queries": int(b.value), "full_batches": int(c.value)}

    def lane_times(self):
        """hvx_batcher_lane_times: milliseconds the dispatcher lanes spent in each state since creation (summed over the lanes)."""
        t = BatcherTimes()
        _check(lib().hvx_batcher_lane_times(self._h, C.byref(t)))
        return {"lanes": int(t.lanes), **{f[0][:-3] + "_ms": getattr(t, f[0]) / 1e6 for f in BatcherTimes._fields_ if f[0].endswith("_ns")}}


class RestrictedBatcher(Batcher):
    """hvx_batcher_new_restricted: the operator's `Some(candidates) => index.search_restricted(..)` branch (storage.rs:140-163) -- every caller
    brings its own candidate ids; a batch of callers is ONE launch of the exact scan.  poll / wait / eventfd / stats as for Batcher."""

    def __init__(self, index: "ValidatedVectorReadIndex", rparams: RestrictedParams, max_batch: int = 0, max_wait_us: int = 0, lanes: int = 0,
                 max_ids_per_query: int = 0):
        self._index = index
        self.k = int(rparams.k)
        h = _vp()
        _check(lib().hvx_batcher_new_restricted(index._h, C.byref(rparams), max_batch, max_wait_us, lanes, max_ids_per_query, C.byref(h)))
        self._h = h

    def _args(self, query, candidates):
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        if q.size != self._index.dim:
            raise HelixDbError(ERR_DIMENSION, f"expected dimension {self._index.dim}, got {q.size}")
        al = candidates.ids if isinstance(candidates, RestrictedVectorCandidates) else np.ascontiguousarray(candidates, dtype=np.uint64).reshape(-1)
        return q, al

    def search(self, query, candidates):
        q, al = self._args(query, candidates)
        ids = np.zeros(self.k, np.uint64); sc = np.zeros(self.k, np.float32); cnt = C.c_uint32(0)
        _check(lib().hvx_batcher_search_restricted(self._h, _ptr(q), _ptr(al) if al.size else None, al.size, _ptr(ids), _ptr(sc), C.byref(cnt)))
        return [SearchResult(int(i), s) for i, s in zip(ids[: cnt.value], sc[: cnt.value])]

    def submit(self, query, candidates):
        q, al = self._args(query, candidates)
        t = BatcherTicket()
        rc = lib().hvx_batcher_submit_restricted(self._h, _ptr(q), _ptr(al) if al.size else None, al.size, C.byref(t))
        if rc == ERR_BUSY:
            return None
        _check(rc)
        return t


def _visit_arrays(cap):
    return (np.zeros(cap, np.uint64), np.zeros(cap, np.uint32), np.zeros(cap, np.uint64), np.zeros(cap, np.uint64), np.zeros(cap, np.uint32))


def _visits_of(arrs, c):
    nodes, depths, parents, edges, against = arrs
    visits = [(int(nodes[i]), int(depths[i])) for i in range(c)]
    disc = [(int(parents[i]), int(edges[i]), int(against[i])) for i in range(c) if parents[i] != np.uint64(0xFFFFFFFFFFFFFFFF)]
    return visits, disc


def traverse_host(n_nodes, out_offsets, out_targets, edge_labels, seeds, max_depth, direction=DIR_BOTH, allowed_labels=(), hub_degree=0,
                  depth_first=False):
    """Graph::traverse entirely on the host (hvx_traverse_host): no device, no Graph handle.  Returns (visits, edges)."""
    off = np.ascontiguousarray(out_offsets, dtype=np.uint64)
    tgt = np.ascontiguousarray(out_targets, dtype=np.uint64)
    lab = None if edge_labels is None else np.ascontiguousarray(edge_labels, dtype=np.uint32)
    s = np.ascontiguousarray(seeds, dtype=np.uint64)
    al = np.ascontiguousarray(list(allowed_labels), dtype=np.uint32)
    cap = max(int(n_nodes), 1)
    arrs = _visit_arrays(cap)
    cnt = C.c_uint64(0)
    _check(lib().hvx_traverse_host(int(n_nodes), tgt.size, _ptr(off), _ptr(tgt), _ptr(lab), 1 if depth_first else 0, _ptr(s), s.size,
                                   max_depth, direction, _ptr(al), al.size, hub_degree, cap, *[_ptr(a) for a in arrs], C.byref(cnt)))
    return _visits_of(arrs, cnt.value)


class Graph:
    """Device CSR for the graph prefilter (crates/graph-algorithms Graph::traverse; interpreter expand)."""

    def __init__(self, n_nodes, out_offsets, out_targets, edge_labels=None, device=-1):
        off = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        tgt = np.ascontiguousarray(out_targets, dtype=np.uint64)
        lab = None if edge_labels is None else np.ascontiguousarray(edge_labels, dtype=np.uint32)
        self.n = int(n_nodes)
        h = _vp()
        _check(lib().hvx_csr_import(self.n, tgt.size, _ptr(off), _ptr(tgt), _ptr(lab), device, C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().hvx_csr_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def traverse(self, seeds, max_depth, direction=DIR_BOTH, allowed_labels=(), hub_degree=0, include_seeds=True):
        s = np.ascontiguousarray(seeds, dtype=np.uint64)
        lab = np.ascontiguousarray(list(allowed_labels), dtype=np.uint32)
        words = np.zeros((self.n + 63) // 64, np.uint64)
        depth = np.zeros(max(self.n, 1), np.uint32)
        _check(lib().hvx_traverse_filter(self._h, _ptr(s), s.size, max_depth, direction, _ptr(lab), lab.size,
                                         hub_degree, 1 if include_seeds else 0, _ptr(words), _ptr(depth)))
        return words, depth[: self.n]

    def traverse_ordered(self, seeds, max_depth, direction=DIR_BOTH, allowed_labels=(), hub_degree=0):
        """Graph::traverse (BreadthFirst) with its visit order and discovery edges (traversal.rs:216-261):
        returns (visits, edges) -- visits = [(node, depth)] in discovery order, edges[i] = (node whose arc reached the
        i-th non-seed visit, index of the stored edge in the imported out_targets, 1 if followed against its direction)."""
        s = np.ascontiguousarray(seeds, dtype=np.uint64)
        lab = np.ascontiguousarray(list(allowed_labels), dtype=np.uint32)
        cap = max(self.n, 1)
        nodes = np.zeros(cap, np.uint64); depths = np.zeros(cap, np.uint32); parents = np.zeros(cap, np.uint64)
        edges = np.zeros(cap, np.uint64); against = np.zeros(cap, np.uint32)
        cnt = C.c_uint64(0)
        _check(lib().hvx_traverse_ordered(self._h, _ptr(s), s.size, max_depth, direction, _ptr(lab), lab.size, hub_degree, cap,
                                          _ptr(nodes), _ptr(depths), _ptr(parents), _ptr(edges), _ptr(against), C.byref(cnt)))
        c = cnt.value
        visits = [(int(nodes[i]), int(depths[i])) for i in range(c)]
        disc = [(int(parents[i]), int(edges[i]), int(against[i])) for i in range(c) if parents[i] != np.uint64(0xFFFFFFFFFFFFFFFF)]
        return visits, disc

    def traverse_depth_first(self, seeds, max_depth, direction=DIR_BOTH, allowed_labels=(), hub_degree=0):
        """Graph::traverse with TraversalStrategy::DepthFirst (traversal.rs:263-309; host-side, hvx_traverse_dfs); returns
        (visits, edges) shaped like traverse_ordered."""
        s = np.ascontiguousarray(seeds, dtype=np.uint64)
        lab = np.ascontiguousarray(list(allowed_labels), dtype=np.uint32)
        cap = max(self.n, 1)
        arrs = _visit_arrays(cap)
        cnt = C.c_uint64(0)
        _check(lib().hvx_traverse_dfs(self._h, _ptr(s), s.size, max_depth, direction, _ptr(lab), lab.size, hub_degree, cap,
                                      *[_ptr(a) for a in arrs], C.byref(cnt)))
        return _visits_of(arrs, cnt.value)

    def expand(self, rows, direction=DIR_OUT, allowed_labels=()):
        s = np.ascontiguousarray(rows, dtype=np.uint64)
        lab = np.ascontiguousarray(list(allowed_labels), dtype=np.uint32)
        words = np.zeros((self.n + 63) // 64, np.uint64)
        _check(lib().hvx_expand_filter(self._h, _ptr(s), s.size, direction, _ptr(lab), lab.size, _ptr(words)))
        return words


class SimHasher:
    """unaligned_vector/simhash.rs SimHasher::new_with_seed / hash_from_slice, batched on the device."""

    def __init__(self, dim, seed=42, device=-1):
        self.dim = int(dim)
        h = _vp()
        _check(lib().hvx_simhasher_new(self.dim, int(seed), device, C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().hvx_simhasher_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def hash_batch(self, vectors) -> np.ndarray:
        v = np.ascontiguousarray(vectors, dtype=np.float32).reshape(-1, self.dim)
        out = np.zeros(v.shape[0], np.uint64)
        _check(lib().hvx_simhash_batch(self._h, _ptr(v), v.shape[0], _ptr(out)))
        return out

    def hash(self, vector) -> int:
        return int(self.hash_batch(np.asarray(vector, np.float32).reshape(1, -1))[0])


# ---- what the reference's restricted planner would do (restricted.rs:40-56,196-260,426-453,321-342) --------------------
# Host mirrors of the planner the library applies in hvx_search_restricted_batch_params (so that a host can see / log the plan
# of a request, and so that the planning rules stay pinned by the reference's tests on the CPU).
MAX_RESTRICTED_CANDIDATES = 1_000_000
EXACT_CARDINALITY_THRESHOLD = 256
EXACT_VECTOR_BYTES_THRESHOLD = 4 * 1024 * 1024
FILTERED_BEAM_PERCENT = 150
FILTERED_SAMPLED_SEEDS = 64
FILTERED_DIRECTORY_SEEDS = 256
MAX_RESTRICTED_RESULT_COUNT = 800
FILTERED_VECTOR_PAYLOAD_LIMIT = MAX_RESTRICTED_RESULT_COUNT


def restricted_result_count(requested: int, candidate_count: int) -> int:
    """RestrictedResultCount::try_new (restricted.rs:200-213): clamp to the population FIRST, then enforce the 800 limit."""
    count = min(int(requested), int(candidate_count))
    if count <= 0:
        raise HelixDbError(ERR_K_RANGE, "result count must be non-zero")
    if count > MAX_RESTRICTED_RESULT_COUNT:
        raise HelixDbError(ERR_K_RANGE, f"restricted vector search result count {count} is above the maximum {MAX_RESTRICTED_RESULT_COUNT}")
    return count


def restricted_execution_plan(candidate_count: int, dimension: int, params: SearchParams, beam_percent: int = FILTERED_BEAM_PERCENT):
    """restricted_execution_plan_with_beam_percent (restricted.rs:426-453) + FilteredGraphBudgets::with_beam_percent (:230-259).
    Returns {"plan": "exact" | "filtered_graph", "k": .., and for the filtered plan the row / payload / seed budgets}."""
    n = int(candidate_count)
    if n <= 0:
        raise HelixDbError(ERR_INVARIANT, "an empty candidate set never reaches planning")
    if n > MAX_RESTRICTED_CANDIDATES:
        raise HelixDbError(ERR_CANDIDATE_LIMIT, f"restricted vector search accepts at most {MAX_RESTRICTED_CANDIDATES} unique candidates")
    k = restricted_result_count(params.k, n)
    if n <= EXACT_CARDINALITY_THRESHOLD and n * int(dimension) * 4 <= EXACT_VECTOR_BYTES_THRESHOLD:
        return {"plan": "exact", "k": k}
    if beam_percent <= 0:
        raise ValueError("filtered beam percent is nonzero")
    ef_filtered = min(max(params.ef * beam_percent // 100, k * 4), n)
    vector_payloads = min(FILTERED_VECTOR_PAYLOAD_LIMIT, n)
    assert vector_payloads >= k
    return {"plan": "filtered_graph", "k": k, "ef_filtered": ef_filtered, "routing_rows": ef_filtered * 16,
            "bridge_rows": ef_filtered * 8, "vector_payloads": vector_payloads,
            "sampled_seeds": min(FILTERED_SAMPLED_SEEDS, n), "directory_seeds": min(FILTERED_DIRECTORY_SEEDS, n)}


def deterministic_sample_ids(sorted_ids, limit: int):
    """NonEmptyCandidateSet::deterministic_sample_ids (restricted.rs:321-342): evenly spaced ranks of the ascending id list."""
    ids = np.asarray(sorted_ids, dtype=np.uint64)
    n = ids.size
    count = min(int(limit), n)
    if count == n:
        return ids.tolist()
    if count == 1:
        return [int(ids[0])]
    last = n - 1
    return [int(ids[(s * last) // (count - 1)]) for s in range(count)]


def adaptive_threshold_table(configured: int, failure: float) -> np.ndarray:
    """The 64-step table the kernels evaluate policy.rs:577-599 with (threshold(delta) = #{t : delta <= brk[t-1]})."""
    brk = np.zeros(64, np.float32)
    _check(lib().hvx_adaptive_threshold_table(configured, np.float32(failure), _ptr(brk)))
    return brk


CURRENT_SCORE, METRIC_DISTANCE = 0, 1  # result.rs:69-81 DistanceOutputVersion
_UNITS = {COSINE: "HalfCosineScore", EUCLIDEAN: "SquaredEuclideanScore", MANHATTAN: "ManhattanDistance"}


def materialize_distance(score, metric: int, version: int = CURRENT_SCORE):
    """TypedVectorSearchResult::materialize_distance (result.rs:143-176): (value f32, unit label).  Existing responses use
    CURRENT_SCORE; METRIC_DISTANCE takes exactly one square root of the squared-Euclidean score and never doubles the
    half-cosine."""
    s = np.float32(score)
    if not np.isfinite(s) or s < 0:  # DistanceScore::try_new (parameters.rs:243-274)
        raise HelixDbError(ERR_INVARIANT, "distance score must be finite and non-negative")
    if metric == EUCLIDEAN and version == METRIC_DISTANCE:
        return np.float32(np.sqrt(s)), "EuclideanDistance"
    return s, _UNITS[metric]


def unique_restricted_rows(rows, element_type: str = "node"):
    """interpreter/access/restricted_vector.rs:14-38: first row wins per element id; rows keyed in id order.

    `rows` are dicts with "current" = ("node" | "edge", id).  Raises Query errors exactly where the reference does."""
    rows_by_id = {}
    for row in rows:
        current = row.get("current")
        if current is None:
            raise HelixDbError(ERR_K_RANGE, "vector_search expected rows with a current graph element")
        kind, eid = current
        if kind != element_type:
            raise HelixDbError(ERR_K_RANGE, "vector_search index kind does not match the input stream")
        rows_by_id.setdefault(int(eid), row)
    return dict(sorted(rows_by_id.items()))  # BTreeMap iteration order


def materialize_restricted_results(rows_by_id: dict, results):
    """restricted_vector.rs:40-65: re-attach the upstream rows in RANK order and add `$distance` = score as F64."""
    ranked = []
    pending = dict(rows_by_id)
    for r in results:
        row = pending.pop(int(r.entity_id), None)
        if row is None:
            raise HelixDbError(ERR_INVARIANT, "restricted vector search returned an ID outside its exact bitmap")
        row = dict(row)
        vp = dict(row.get("virtual_properties", {}))
        vp["$distance"] = float(np.float64(np.float32(r.score)))
        row["virtual_properties"] = vp
        ranked.append(row)
    return ranked


def order_code_from_simhash_bits(bits: int) -> int:
    return int(lib().hvx_order_code_from_simhash_bits(int(bits)))


def decode_layer0_row(value: bytes):
    """values/vectors.rs:187-210 decode_layer0_neighbors_and_simhash -> (ids, simhash or None)."""
    ids = np.zeros(max(len(value) // 8 + 1, 1), np.uint64)
    cnt, sh, has = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0)
    _check(lib().hvx_decode_layer0_row(value, len(value), _ptr(ids), ids.size, C.byref(cnt), C.byref(sh), C.byref(has)))
    return ids[: cnt.value].tolist(), (int(sh.value) if has.value else None)


def component_limit(metric: int, dim: int) -> float:
    """VectorComponentLimit::try_new (domain.rs:26-78) as the library applies it; inf for cosine."""
    return float(lib().hvx_component_limit(metric, dim))


def decode_simhash_row(value: bytes) -> int:
    """values/vectors/simhash.rs:46-61 decode_simhash."""
    out = C.c_uint64(0)
    _check(lib().hvx_decode_simhash_row(value, len(value), C.byref(out)))
    return int(out.value)


def decode_entry_candidate_layer(value: bytes) -> int:
    """values/vectors/entry.rs:31-47 decode_entry_candidate_layer."""
    out = C.c_uint32(0)
    _check(lib().hvx_decode_entry_candidate_layer(value, len(value), C.byref(out)))
    return int(out.value)


def decode_upper_row(value: bytes):
    """values/vectors/neighbors.rs:79-110 decode_upper_neighbors."""
    ids = np.zeros(max(len(value) // 8 + 1, 1), np.uint64)
    cnt = C.c_uint32(0)
    _check(lib().hvx_decode_upper_row(value, len(value), _ptr(ids), ids.size, C.byref(cnt)))
    return ids[: cnt.value].tolist()


def parse_vector_key(key: bytes):
    """keys/vectors.rs: -> dict(kind, index_id, node_id, order_code, layer) or None."""
    ix, nd, oc, ly = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
    kind = lib().hvx_parse_vector_key(key, len(key), C.byref(ix), C.byref(nd), C.byref(oc), C.byref(ly))
    if not kind:
        return None
    return dict(kind=int(kind), index_id=int(ix.value), node_id=int(nd.value), order_code=int(oc.value), layer=int(ly.value))


class Hydrator:
    """Collects persisted rows (value bytes as stored by HelixDB) and imports them as a device index."""

    def __init__(self, dim, metric):
        self.dim, self.metric = dim, metric
        h = _vp()
        _check(lib().hvx_hydrator_new(dim, metric, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().hvx_hydrator_free(self._h)
            self._h = None

    def add_item(self, node_id, value: bytes):
        _check(lib().hvx_hydrator_add_item(self._h, int(node_id), value, len(value)))

    def add_layer0_row(self, node_id, value: bytes):
        _check(lib().hvx_hydrator_add_layer0_row(self._h, int(node_id), value, len(value)))

    def add_upper_row(self, node_id, layer, value: bytes):
        _check(lib().hvx_hydrator_add_upper_row(self._h, int(node_id), int(layer), value, len(value)))

    def set_metadata(self, value: bytes):
        """entry point / top layer from the index metadata row (rkyv VectorIndexMetadata), checked against the dimension"""
        _check(lib().hvx_hydrator_set_metadata(self._h, value, len(value)))

    def set_entry(self, entry_point, max_layer):
        _check(lib().hvx_hydrator_set_entry(self._h, int(entry_point), int(max_layer)))

    def finish(self, *, m=16, m0=32, float_kernel=KERNEL_AVX_FMA, device=-1, max_batch=1024, dtype=F32, reserve_rows=0, reserve_upper_rows=0):
        d = _Desc(dim=self.dim, metric=self.metric, dtype=dtype, float_kernel=float_kernel, n=0, m=m, m0=m0, has_entry=0,
                  max_layer=0, entry_point=0, shard_id_lo=0, shard_id_hi=0, device=device, max_batch=max_batch)
        h = _vp()
        if reserve_rows or reserve_upper_rows:
            L = lib()
            L.hvx_hydrator_finish_reserve.restype = C.c_int
            L.hvx_hydrator_finish_reserve.argtypes = [_vp, C.POINTER(_Desc), C.c_uint64, C.c_uint64, C.POINTER(_vp)]
            _check(L.hvx_hydrator_finish_reserve(self._h, C.byref(d), int(reserve_rows), int(reserve_upper_rows), C.byref(h)))
        else:
            _check(lib().hvx_hydrator_finish(self._h, C.byref(d), C.byref(h)))
        return ValidatedVectorReadIndex(h, self.dim, self.metric, -1)
