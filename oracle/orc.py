"""ctypes binding of the CPU oracle (oracle/hvx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product library (helix-db_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libhvx_oracle.so")

COSINE, L2SQ, L1 = 0, 1, 2
K_SCALAR, K_SSE, K_AVX, K_AVX_FMA, K_NEON, K_AVX_FMA_HW = 0, 1, 2, 3, 4, 5
OK, ERR_DIMENSION, ERR_NONFINITE, ERR_ZERO_NORM, ERR_MAGNITUDE, ERR_K_RANGE, ERR_CANDIDATE_LIMIT, \
    ERR_DEVICE, ERR_INVARIANT = range(9)


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc"))]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None

f32p = C.POINTER(C.c_float)
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u16p = C.POINTER(C.c_uint16)


class Stats(C.Structure):
    _fields_ = [("expansion_steps", C.c_uint32), ("neighbors_examined", C.c_uint32),
                ("vectors_loaded", C.c_uint32), ("distance_computations", C.c_uint32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


SIMHASH_ALWAYS, SIMHASH_ADAPTIVE, SIMHASH_OFF = 0, 1, 2
SAMPLING_EXHAUSTIVE, SAMPLING_FIXED, SAMPLING_ADAPTIVE = 0, 1, 2
BYPASS_READY, BYPASS_BYPASSING, BYPASS_COOLING = 0, 1, 2
TRIGGER_NONE, TRIGGER_READ_BUDGET, TRIGGER_LOW_YIELD, TRIGGER_BOTH = 0, 1, 2, 3


class _Rec(C.Structure):
    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class SearchParams(_Rec):
    """orc_search_params: SearchParams (mod.rs:410-621) + the index-level config it resolves against."""
    _fields_ = [("k", C.c_uint32), ("ef", C.c_uint32), ("simhash_mode", C.c_uint32),
                ("pre_simhash_sampling_ratio_override", C.c_float),
                ("bypass_min_frontier", C.c_uint32), ("bypass_window_expansions", C.c_uint32),
                ("bypass_min_filter_rate", C.c_float), ("read_budget_multiplier", C.c_uint32),
                ("simhash_sampling_ratio_override", C.c_float), ("simhash_failure_prob_override", C.c_float),
                ("simhash_threshold", C.c_uint32), ("sampling_ratio", C.c_float),
                ("adaptive_enabled", C.c_uint32), ("adaptive_failure_prob", C.c_float),
                ("resident_simhash", C.c_uint32), ("use_fixed_seed", C.c_uint32), ("fixed_seed", C.c_uint64)]

    @classmethod
    def new(cls, k, **over):
        p = cls()
        lib().orc_search_params_default(C.byref(p), k)
        for key, v in over.items():
            assert hasattr(p, key), key
            setattr(p, key, v)
        return p


class AdaptiveStats(_Rec):
    _fields_ = [(n, C.c_uint32) for n in (
        "expansion_steps", "neighbors_examined", "vectors_loaded", "distance_computations",
        "simhash_filtered", "simhash_examined", "simhash_missing_hash",
        "simhash_passed_before_sampling", "simhash_passed_after_sampling",
        "pre_simhash_sample_kept", "pre_simhash_sample_dropped",
        "simhash_bypass_expansions", "simhash_skipped_candidates",
        "simhash_bypass_trigger_budget", "simhash_bypass_trigger_low_yield",
        "txn_get_simhash_filter", "active_simhash_threshold_sum", "active_simhash_threshold_samples",
        "effective_beam_len_sum", "effective_beam_len_samples", "active_sampling_ratio_samples",
        "rng_words")] + [("active_sampling_ratio_sum", C.c_double)]


RESTRICTED_EXACT, RESTRICTED_FILTERED = 1, 2
TERM_EXHAUSTED, TERM_BEAM_COMPLETE, TERM_ROUTING_BUDGET, TERM_BRIDGE_BUDGET, TERM_VECTOR_BUDGET = 1, 2, 3, 4, 5


class RestrictedPlan(_Rec):  # orc_restricted_plan_t: RestrictedExecutionPlan + FilteredGraphBudgets (restricted.rs:216-281)
    _fields_ = [("strategy", C.c_uint32), ("k", C.c_uint32), ("ef_filtered", C.c_uint64), ("routing_rows", C.c_uint64),
                ("bridge_rows", C.c_uint64), ("vector_payloads", C.c_uint64), ("sampled_seeds", C.c_uint64), ("directory_seeds", C.c_uint64)]


class RestrictedStats(_Rec):  # orc_restricted_stats: RestrictedSearchStats (restricted.rs:147-166)
    _fields_ = [("strategy", C.c_uint32), ("termination", C.c_uint32)] + [(n, C.c_uint64) for n in (
        "ef_filtered", "directory_scan_calls", "directory_rows", "directory_decoded_bytes", "directory_hits", "simhash_row_requests",
        "companion_row_requests", "routing_rows", "bridge_rows", "bridge_frontier_pushes", "neighbor_multi_get_calls",
        "vector_payload_requests", "vector_bytes", "distance_computations")]


class PolicyInput(_Rec):
    _fields_ = [("metric", C.c_uint32), ("simhash_mode", C.c_uint32), ("configured_threshold", C.c_uint32),
                ("sampling_ratio", C.c_float), ("pre_sampling_override", C.c_float),
                ("adaptive_enabled", C.c_uint32), ("failure", C.c_float),
                ("bypass_from_deployed", C.c_uint32), ("bypass_ef", C.c_uint32),
                ("bypass_min_frontier", C.c_uint32), ("bypass_window_expansions", C.c_uint32),
                ("bypass_min_filter_rate", C.c_float), ("bypass_read_budget_multiplier", C.c_uint32),
                ("topk_ready", C.c_uint32), ("ef", C.c_uint32), ("search_frontier_len", C.c_uint32),
                ("candidate_frontier_len", C.c_uint32), ("current", C.c_float), ("delta", C.c_float),
                ("state", C.c_uint32), ("state_remaining", C.c_uint32),
                ("simhash_filter_reads", C.c_uint64), ("window_examined", C.c_uint64),
                ("window_filtered", C.c_uint64), ("window_expansions", C.c_uint64)]


class PolicyDecision(_Rec):
    _fields_ = [("fetch_missing", C.c_uint32), ("filter_cached", C.c_uint32), ("has_threshold", C.c_uint32),
                ("threshold", C.c_uint32), ("pre_sampling_kind", C.c_uint32),
                ("pre_sampling_probability", C.c_float), ("sampling_kind", C.c_uint32),
                ("sampling_probability", C.c_float), ("base_sampling_probability", C.c_float),
                ("bypassed", C.c_uint32), ("next_state", C.c_uint32), ("next_state_remaining", C.c_uint32),
                ("trigger", C.c_uint32)]


class Rng(_Rec):
    _fields_ = [("seed", C.c_uint64), ("key", C.c_uint32 * 8), ("buf", C.c_uint32 * 16), ("block", C.c_uint64),
                ("pos", C.c_uint32), ("ready", C.c_int), ("words", C.c_uint32)]

    @classmethod
    def seeded(cls, seed):
        r = cls()
        lib().orc_rng_seed(C.byref(r), seed)
        return r

    def should_sample(self, ratio):
        return bool(lib().orc_rng_should_sample(C.byref(self), np.float32(ratio)))

    def choose_index(self, n):
        v = int(lib().orc_rng_choose_index(C.byref(self), n))
        return None if v < 0 else v


def policy_decide(**kw):
    """Layer0Policy::from_deployed(..)[.with_adaptive_bypass(..)].decide(context) -- see PolicyInput."""
    inp = PolicyInput()
    inp.pre_sampling_override = -1.0
    for key, v in kw.items():
        assert hasattr(inp, key), key
        setattr(inp, key, v)
    out = PolicyDecision()
    lib().orc_policy_decide(C.byref(inp), C.byref(out))
    return out


def candidate_probability(kind, probability, similarity_bits, threshold=None):
    return np.float32(lib().orc_candidate_probability(kind, np.float32(probability), similarity_bits,
                                                      0 if threshold is None else 1, threshold or 0))


def pre_sampling_decision(base_ratio, frontier, ef):
    p = C.c_float(0)
    kind = lib().orc_pre_sampling_decision(np.float32(base_ratio), frontier, ef, C.byref(p))
    return int(kind), np.float32(p.value)


def adaptive_sampling_ratio(base, search_frontier_len, ef, current, delta):
    return np.float32(lib().orc_adaptive_sampling_ratio(np.float32(base), search_frontier_len, ef,
                                                        np.float32(current), np.float32(delta)))


def adaptive_threshold(topk_ready, delta, configured, failure):
    return int(lib().orc_adaptive_threshold(int(topk_ready), np.float32(delta), configured, np.float32(failure)))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.orc_euclidean.restype = C.c_float
    L.orc_euclidean.argtypes = [f32p, f32p, C.c_uint32, C.c_int]
    L.orc_dot.restype = C.c_float
    L.orc_dot.argtypes = [f32p, f32p, C.c_uint32, C.c_int]
    L.orc_manhattan.restype = C.c_float
    L.orc_manhattan.argtypes = [f32p, f32p, C.c_uint32]
    L.orc_euclidean_avxfma_hw.restype = C.c_float
    L.orc_euclidean_avxfma_hw.argtypes = [f32p, f32p, C.c_uint32]
    L.orc_dot_avxfma_hw.restype = C.c_float
    L.orc_dot_avxfma_hw.argtypes = [f32p, f32p, C.c_uint32]
    L.orc_euclidean_sse_hw.restype = C.c_float
    L.orc_euclidean_sse_hw.argtypes = [f32p, f32p, C.c_uint32]
    L.orc_dot_sse_hw.restype = C.c_float
    L.orc_dot_sse_hw.argtypes = [f32p, f32p, C.c_uint32]
    L.orc_have_sse_hw.restype = C.c_int
    L.orc_have_avxfma_hw.restype = C.c_int
    L.orc_scaled_l2_norm.restype = C.c_double
    L.orc_scaled_l2_norm.argtypes = [f32p, C.c_uint32]
    L.orc_header.restype = C.c_float
    L.orc_header.argtypes = [C.c_int, f32p, C.c_uint32]
    L.orc_distance.restype = C.c_float
    L.orc_distance.argtypes = [C.c_int, C.c_int, f32p, C.c_float, f32p, C.c_float, C.c_uint32]
    L.orc_component_limit.restype = C.c_float
    L.orc_component_limit.argtypes = [C.c_int, C.c_uint32]
    L.orc_validate_vector.restype = C.c_int
    L.orc_validate_vector.argtypes = [C.c_int, f32p, C.c_uint32, C.c_uint32, u32p]
    L.orc_distance_score.restype = C.c_int
    L.orc_distance_score.argtypes = [f32p]
    L.orc_select_layer_from_uniform.restype = C.c_uint16
    L.orc_select_layer_from_uniform.argtypes = [C.c_float, C.c_float]
    L.orc_default_ml_for_m.restype = C.c_float
    L.orc_default_ml_for_m.argtypes = [C.c_uint32]
    L.orc_index_new.restype = C.c_void_p
    L.orc_index_new.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_index_free.argtypes = [C.c_void_p]
    L.orc_index_insert.restype = C.c_int
    L.orc_index_insert.argtypes = [C.c_void_p, C.c_uint64, f32p, C.c_uint16]
    L.orc_index_delete.restype = C.c_int
    L.orc_index_delete.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int)]
    L.orc_index_is_live.restype = C.c_int
    L.orc_index_is_live.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_index_prune_candidates.restype = C.c_int
    L.orc_index_prune_candidates.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
    L.orc_index_seed.restype = C.c_int
    L.orc_index_seed.argtypes = [C.c_void_p, C.c_uint64, u64p, f32p, u64p, u64p, u16p, u64p, u64p,
                                 C.c_int, C.c_uint64, C.c_uint16]
    L.orc_index_count.restype = C.c_uint64
    L.orc_index_count.argtypes = [C.c_void_p]
    L.orc_index_entry.restype = C.c_int
    L.orc_index_entry.argtypes = [C.c_void_p, u64p, u16p]
    L.orc_index_export_sizes.restype = C.c_uint64
    L.orc_index_export_sizes.argtypes = [C.c_void_p, u64p, u64p, u64p]
    L.orc_index_export.restype = C.c_int
    L.orc_index_export.argtypes = [C.c_void_p, u64p, f32p, u64p, u64p, u16p, u64p, u64p]
    L.orc_search.restype = C.c_int
    L.orc_search.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, u64p, f32p, u32p,
                             C.POINTER(Stats)]
    L.orc_search_batch_mt.restype = C.c_int
    L.orc_search_batch_mt.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      u64p, f32p, u32p, C.POINTER(Stats)]
    L.orc_flat_search.restype = C.c_int
    L.orc_flat_search.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32, u64p, C.c_uint64, u64p,
                                  f32p, u32p]
    L.orc_flat_search_matrix.restype = C.c_int
    L.orc_flat_search_matrix.argtypes = [C.c_int, C.c_int, f32p, C.c_uint64, C.c_uint32, f32p,
                                         C.c_uint32, u64p, f32p, u32p]
    _extra_signatures(L)
    _lib = L
    return L


def _extra_signatures(L):
    """Signatures of the optional oracle translation units (simhash, restricted, traverse)."""
    if hasattr(L, "orc_simhash_planes"):
        L.orc_simhash_planes.restype = C.c_int
        L.orc_simhash_planes.argtypes = [C.c_uint32, C.c_uint64, f32p]
        L.orc_simhash_hash.restype = C.c_uint64
        L.orc_simhash_hash.argtypes = [f32p, f32p, C.c_uint32]
        L.orc_order_code.restype = C.c_uint64
        L.orc_order_code.argtypes = [C.c_uint64]
        L.orc_stdrng_u32.restype = None
        L.orc_stdrng_u32.argtypes = [C.c_uint64, u32p, C.c_uint32]
        L.orc_simhash_collisions.restype = C.c_uint32
        L.orc_simhash_collisions.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_query_seed.restype = C.c_uint64
        L.orc_query_seed.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    if hasattr(L, "orc_policy_decide"):
        L.orc_search_params_default.restype = None
        L.orc_search_params_default.argtypes = [C.POINTER(SearchParams), C.c_uint32]
        L.orc_policy_decide.restype = None
        L.orc_policy_decide.argtypes = [C.POINTER(PolicyInput), C.POINTER(PolicyDecision)]
        L.orc_candidate_probability.restype = C.c_float
        L.orc_candidate_probability.argtypes = [C.c_uint32, C.c_float, C.c_uint32, C.c_int, C.c_uint32]
        L.orc_pre_sampling_decision.restype = C.c_uint32
        L.orc_pre_sampling_decision.argtypes = [C.c_float, C.c_uint32, C.c_uint32, f32p]
        L.orc_adaptive_sampling_ratio.restype = C.c_float
        L.orc_adaptive_sampling_ratio.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_float]
        L.orc_adaptive_threshold.restype = C.c_uint32
        L.orc_adaptive_threshold.argtypes = [C.c_uint32, C.c_float, C.c_uint32, C.c_float]
        L.orc_rng_seed.restype = None
        L.orc_rng_seed.argtypes = [C.POINTER(Rng), C.c_uint64]
        L.orc_rng_should_sample.restype = C.c_int
        L.orc_rng_should_sample.argtypes = [C.POINTER(Rng), C.c_float]
        L.orc_rng_choose_index.restype = C.c_int64
        L.orc_rng_choose_index.argtypes = [C.POINTER(Rng), C.c_uint64]
        L.orc_index_set_simhash.restype = C.c_int
        L.orc_index_set_simhash.argtypes = [C.c_void_p, C.c_uint64, u64p]
        L.orc_index_get_simhash.restype = C.c_int
        L.orc_index_get_simhash.argtypes = [C.c_void_p, u64p]
        L.orc_index_query_simhash.restype = C.c_uint64
        L.orc_index_query_simhash.argtypes = [C.c_void_p, f32p]
        L.orc_search_params_run.restype = C.c_int
        L.orc_search_params_run.argtypes = [C.c_void_p, f32p, C.c_uint32, C.POINTER(SearchParams), u64p, f32p, u32p,
                                            C.POINTER(AdaptiveStats)]
        L.orc_search_params_batch_mt.restype = C.c_int
        L.orc_search_params_batch_mt.argtypes = [C.c_void_p, f32p, C.c_uint32, C.POINTER(SearchParams), C.c_uint32,
                                                 u64p, f32p, u32p, C.POINTER(AdaptiveStats)]
    if hasattr(L, "orc_traverse"):
        L.orc_traverse.restype = C.c_int64
    if hasattr(L, "orc_search_restricted"):
        L.orc_search_restricted.restype = C.c_int
        L.orc_search_restricted.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, u64p, C.c_uint64, C.c_uint32, C.c_int,
                                            u64p, f32p, u32p, C.POINTER(RestrictedStats)]
        L.orc_restricted_filter_aware_search.restype = C.c_int
        L.orc_restricted_filter_aware_search.argtypes = [C.c_void_p, f32p, C.c_uint32, u64p, C.c_uint64, C.POINTER(RestrictedPlan), C.c_int,
                                                         u64p, f32p, u32p, C.POINTER(RestrictedStats)]
        L.orc_restricted_plan.restype = None
        L.orc_restricted_plan.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(RestrictedPlan)]
        L.orc_deterministic_sample_ranks.restype = C.c_uint64
        L.orc_deterministic_sample_ranks.argtypes = [C.c_uint64, C.c_uint64, u64p]


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


def euclidean(a, b, kernel=K_AVX_FMA):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().orc_euclidean(pa, pb, a.size, kernel))


def dot(a, b, kernel=K_AVX_FMA):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().orc_dot(pa, pb, a.size, kernel))


def manhattan(a, b):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().orc_manhattan(pa, pb, a.size))


def header(metric, v):
    v, pv = _f(v)
    return np.float32(lib().orc_header(metric, pv, v.size))


def distance(metric, a, b, kernel=K_AVX_FMA):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().orc_distance(metric, kernel, pa, header(metric, a), pb, header(metric, b), a.size))


def validate(metric, v, dim):
    v, pv = _f(v)
    bad = C.c_uint32(0)
    rc = lib().orc_validate_vector(metric, pv, v.size, dim, C.byref(bad))
    return rc, int(bad.value)


class Index:
    """Oracle index: the reference's VectorIndex<D> restated over dense arrays."""

    def __init__(self, dim, metric, kernel=K_AVX_FMA, m=16, m0=32, ef_construction=200):
        self.dim, self.metric, self.kernel = dim, metric, kernel
        self.m, self.m0, self.efc = m, m0, ef_construction
        self._h = lib().orc_index_new(dim, metric, kernel, m, m0, ef_construction)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_index_free(self._h)
            self._h = None

    def insert(self, node_id, vector, level):
        v, pv = _f(vector)
        assert v.size == self.dim
        return lib().orc_index_insert(self._h, int(node_id), pv, int(level))

    def delete(self, node_id):
        """stage_delete (mutation.rs:1606-2055): (rc, existed); an unknown id is (OK, False)"""
        ex = C.c_int(0)
        rc = lib().orc_index_delete(self._h, int(node_id), C.byref(ex))
        return rc, bool(ex.value)

    def is_live(self, node_id):
        return bool(lib().orc_index_is_live(self._h, int(node_id)))

    def prune_candidates(self, owner_id, cand_ids, maxn):
        """the prune of add_bidirectional_link for one row: ids that survive select_diverse + backfill, in selection order"""
        c = np.ascontiguousarray(cand_ids, dtype=np.uint64)
        out = np.zeros(max(c.size, 1), np.uint64)
        n = C.c_uint32(0)
        rc = lib().orc_index_prune_candidates(self._h, int(owner_id), c.ctypes.data_as(C.c_void_p), c.size, int(maxn),
                                              out.ctypes.data_as(C.c_void_p), C.byref(n))
        return rc, out[: n.value].copy()

    def seed(self, node_ids, vectors, l0_offsets, l0_neighbors, level=None, up_offsets=None,
             up_neighbors=None, entry_point=None, max_layer=0):
        ids = np.ascontiguousarray(node_ids, dtype=np.uint64)
        vec, pvec = _f(vectors)
        o0 = np.ascontiguousarray(l0_offsets, dtype=np.uint64)
        n0 = np.ascontiguousarray(l0_neighbors, dtype=np.uint64)
        lv = None if level is None else np.ascontiguousarray(level, dtype=np.uint16)
        uo = np.zeros(1, np.uint64) if up_offsets is None else np.ascontiguousarray(up_offsets, dtype=np.uint64)
        un = np.zeros(1, np.uint64) if up_neighbors is None else np.ascontiguousarray(up_neighbors, dtype=np.uint64)
        return lib().orc_index_seed(
            self._h, ids.size, ids.ctypes.data_as(u64p), pvec, o0.ctypes.data_as(u64p),
            n0.ctypes.data_as(u64p), None if lv is None else lv.ctypes.data_as(u16p),
            uo.ctypes.data_as(u64p), un.ctypes.data_as(u64p), 0 if entry_point is None else 1,
            0 if entry_point is None else int(entry_point), int(max_layer))

    @property
    def count(self):
        return int(lib().orc_index_count(self._h))

    def entry(self):
        e, ml = C.c_uint64(0), C.c_uint16(0)
        if not lib().orc_index_entry(self._h, C.byref(e), C.byref(ml)):
            return None
        return int(e.value), int(ml.value)

    def export(self):
        """Arrays in the hvx_index_import layout (ids ascending)."""
        e0, ur, ue = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        n = int(lib().orc_index_export_sizes(self._h, C.byref(e0), C.byref(ur), C.byref(ue)))
        out = dict(
            node_ids=np.zeros(n, np.uint64), vectors=np.zeros((n, self.dim), np.float32),
            l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(max(e0.value, 1), np.uint64),
            level=np.zeros(max(n, 1), np.uint16), up_offsets=np.zeros(ur.value + 1, np.uint64),
            up_neighbors=np.zeros(max(ue.value, 1), np.uint64))
        lib().orc_index_export(
            self._h, out["node_ids"].ctypes.data_as(u64p), out["vectors"].ctypes.data_as(f32p),
            out["l0_offsets"].ctypes.data_as(u64p), out["l0_neighbors"].ctypes.data_as(u64p),
            out["level"].ctypes.data_as(u16p), out["up_offsets"].ctypes.data_as(u64p),
            out["up_neighbors"].ctypes.data_as(u64p))
        out["l0_neighbors"] = out["l0_neighbors"][: e0.value]
        out["up_neighbors"] = out["up_neighbors"][: ue.value]
        out["level"] = out["level"][:n]
        ent = self.entry()
        out["entry_point"], out["max_layer"] = (ent if ent else (None, 0))
        return out

    def search(self, query, k, ef, with_stats=False):
        q, pq = _f(query)
        ids = np.zeros(max(k, 1), np.uint64)
        sc = np.zeros(max(k, 1), np.float32)
        cnt = C.c_uint32(0)
        st = Stats()
        rc = lib().orc_search(self._h, pq, q.size, k, ef, ids.ctypes.data_as(u64p),
                              sc.ctypes.data_as(f32p), C.byref(cnt), C.byref(st))
        res = (rc, ids[: cnt.value].copy(), sc[: cnt.value].copy())
        return res + (st.as_dict(),) if with_stats else res

    def search_batch(self, queries, k, ef, threads=1):
        """nq independent searches on `threads` pthreads; returns (rc, ids[nq,k], scores, counts, stats list)."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        nq = q.shape[0]
        ids = np.zeros((nq, k), np.uint64); sc = np.zeros((nq, k), np.float32)
        cnt = np.zeros(nq, np.uint32); st = (Stats * nq)()
        rc = lib().orc_search_batch_mt(self._h, q.ctypes.data_as(f32p), nq, k, ef, threads,
                                       ids.ctypes.data_as(u64p), sc.ctypes.data_as(f32p),
                                       cnt.ctypes.data_as(u32p), st)
        return rc, ids, sc, cnt, [s.as_dict() for s in st]

    def set_simhash(self, seed=42, node_hashes=None):
        """Per-node SimHash rows: computed with SimHasher(dim, seed) or given in ascending-id order."""
        ph = None
        if node_hashes is not None:
            h = np.ascontiguousarray(node_hashes, dtype=np.uint64)
            assert h.size == self.count
            ph = h.ctypes.data_as(u64p)
        rc = lib().orc_index_set_simhash(self._h, seed, ph)
        assert rc == OK, rc

    def get_simhash(self):
        out = np.zeros(self.count, np.uint64)
        rc = lib().orc_index_get_simhash(self._h, out.ctypes.data_as(u64p))
        assert rc == OK, rc
        return out

    def query_simhash(self, query):
        q, pq = _f(query)
        return int(lib().orc_index_query_simhash(self._h, pq))

    def search_params(self, query, params, with_stats=False):
        """SearchSession::run with full SearchParams (strict or non-strict arm as the params dictate)."""
        q, pq = _f(query)
        k = max(int(params.k), 1)
        ids = np.zeros(k, np.uint64); sc = np.zeros(k, np.float32)
        cnt = C.c_uint32(0); st = AdaptiveStats()
        rc = lib().orc_search_params_run(self._h, pq, q.size, C.byref(params), ids.ctypes.data_as(u64p),
                                         sc.ctypes.data_as(f32p), C.byref(cnt), C.byref(st))
        res = (rc, ids[: cnt.value].copy(), sc[: cnt.value].copy())
        return res + (st.as_dict(),) if with_stats else res

    def search_params_batch(self, queries, params, threads=1):
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        nq, k = q.shape[0], int(params.k)
        ids = np.zeros((nq, k), np.uint64); sc = np.zeros((nq, k), np.float32)
        cnt = np.zeros(nq, np.uint32); st = (AdaptiveStats * nq)()
        rc = lib().orc_search_params_batch_mt(self._h, q.ctypes.data_as(f32p), nq, C.byref(params), threads,
                                              ids.ctypes.data_as(u64p), sc.ctypes.data_as(f32p),
                                              cnt.ctypes.data_as(u32p), st)
        return rc, ids, sc, cnt, [s.as_dict() for s in st]

    def search_restricted(self, query, k, ef, allowed, beam_percent=0, directory=True):
        """VectorIndex::search_restricted (restricted.rs:466-613): Exact below 256 ids / 4 MiB, else the filter-aware walk.
        Returns (rc, ids, scores, stats dict)."""
        q, pq = _f(query)
        al = np.ascontiguousarray(allowed, dtype=np.uint64)
        ids = np.zeros(max(k, 1), np.uint64); sc = np.zeros(max(k, 1), np.float32)
        cnt = C.c_uint32(0); st = RestrictedStats()
        rc = lib().orc_search_restricted(self._h, pq, q.size, k, ef, al.ctypes.data_as(u64p), al.size, beam_percent, 1 if directory else 0,
                                         ids.ctypes.data_as(u64p), sc.ctypes.data_as(f32p), C.byref(cnt), C.byref(st))
        return rc, ids[: cnt.value].copy(), sc[: cnt.value].copy(), st.as_dict()

    def restricted_filter_aware_search(self, query, allowed, *, k, ef_filtered, routing_rows, bridge_rows, vector_payloads, sampled_seeds,
                                       directory_seeds, directory=False):
        """restricted_filter_aware_search with explicit FilteredGraphBudgets, as the reference's tests call it."""
        q, pq = _f(query)
        al = np.ascontiguousarray(allowed, dtype=np.uint64)
        plan = RestrictedPlan(strategy=RESTRICTED_FILTERED, k=k, ef_filtered=ef_filtered, routing_rows=routing_rows, bridge_rows=bridge_rows,
                              vector_payloads=vector_payloads, sampled_seeds=sampled_seeds, directory_seeds=directory_seeds)
        ids = np.zeros(max(k, 1), np.uint64); sc = np.zeros(max(k, 1), np.float32)
        cnt = C.c_uint32(0); st = RestrictedStats()
        rc = lib().orc_restricted_filter_aware_search(self._h, pq, q.size, al.ctypes.data_as(u64p), al.size, C.byref(plan), 1 if directory else 0,
                                                      ids.ctypes.data_as(u64p), sc.ctypes.data_as(f32p), C.byref(cnt), C.byref(st))
        return rc, ids[: cnt.value].copy(), sc[: cnt.value].copy(), st.as_dict()

    def flat(self, query, k, allowed=None):
        q, pq = _f(query)
        ids = np.zeros(max(k, 1), np.uint64)
        sc = np.zeros(max(k, 1), np.float32)
        cnt = C.c_uint32(0)
        if allowed is None:
            pa, na = None, 0
        else:
            al = np.ascontiguousarray(allowed, dtype=np.uint64)
            pa, na = al.ctypes.data_as(u64p), al.size
        rc = lib().orc_flat_search(self._h, pq, q.size, k, pa, na, ids.ctypes.data_as(u64p),
                                   sc.ctypes.data_as(f32p), C.byref(cnt))
        return rc, ids[: cnt.value].copy(), sc[: cnt.value].copy()


def flat_matrix(metric, rows, query, k, kernel=K_AVX_FMA):
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    q, pq = _f(query)
    ids = np.zeros(max(k, 1), np.uint64)
    sc = np.zeros(max(k, 1), np.float32)
    cnt = C.c_uint32(0)
    rc = lib().orc_flat_search_matrix(metric, kernel, rows.ctypes.data_as(f32p), rows.shape[0],
                                      rows.shape[1], pq, k, ids.ctypes.data_as(u64p),
                                      sc.ctypes.data_as(f32p), C.byref(cnt))
    return rc, ids[: cnt.value].copy(), sc[: cnt.value].copy()


def restricted_plan(candidates, dim, k, ef, beam_percent=0):
    """restricted_execution_plan_with_beam_percent (restricted.rs:426-453) -> dict"""
    p = RestrictedPlan()
    lib().orc_restricted_plan(candidates, dim, k, ef, beam_percent, C.byref(p))
    return p.as_dict()


def deterministic_sample_ranks(candidates, limit):
    out = np.zeros(max(min(candidates, limit), 1), np.uint64)
    n = lib().orc_deterministic_sample_ranks(candidates, limit, out.ctypes.data_as(u64p))
    return out[:n].tolist()


def select_layer(ml, uniform):
    return int(lib().orc_select_layer_from_uniform(np.float32(ml), np.float32(uniform)))


def default_ml(m):
    return np.float32(lib().orc_default_ml_for_m(m))


class SimHasher:
    """unaligned_vector/simhash.rs SimHasher::new_with_seed + hash_from_slice."""

    def __init__(self, dim, seed=42):
        self.dim = dim
        self.planes = np.zeros((64, dim), np.float32)
        rc = lib().orc_simhash_planes(dim, seed, self.planes.ctypes.data_as(f32p))
        assert rc == OK

    def hash(self, vec):
        v, pv = _f(vec)
        assert v.size == self.dim
        return int(lib().orc_simhash_hash(self.planes.ctypes.data_as(f32p), pv, self.dim))


def order_code(bits):
    return int(lib().orc_order_code(bits))


def breadth_first_depths(n, out_offsets, out_targets, labels, seeds, max_depth, direction=2, allowed_labels=(), hub_degree=0):
    """crates/graph-algorithms/src/algorithms/traversal.rs:216-261 `breadth_first` + :311-318 `suppresses_hub`, restated
    for what the device returns: the visited set with its minimum depths (both are independent of the arc order, so the
    id-ordered merge of `ArcIter::Both`, model.rs:680-725, does not matter here).  direction: 0 Out, 1 In, 2 Both;
    hub_degree 0 = HubExpansionPolicy::ExpandAll; degree = out + in arcs (traversal.rs:347-356).  Small cases only."""
    from collections import deque
    out = [[] for _ in range(n)]
    inc = [[] for _ in range(n)]
    for u in range(n):
        for a in range(int(out_offsets[u]), int(out_offsets[u + 1])):
            lab = None if labels is None else int(labels[a])
            out[u].append((int(out_targets[a]), lab))
            inc[int(out_targets[a])].append((u, lab))
    depth, seed_list, queue = {}, [], deque()
    for s in seeds:                                    # duplicates collapse, first occurrence wins (traversal.rs:203-210)
        s = int(s)
        if not 0 <= s < n:
            raise KeyError(f"unknown node {s}")
        if s not in depth:
            depth[s] = 0
            queue.append(s)
            seed_list.append(s)
    allowed = set(int(x) for x in allowed_labels)
    while queue:
        u = queue.popleft()
        if depth[u] >= max_depth:
            continue
        if u not in seed_list and hub_degree and len(out[u]) + len(inc[u]) >= hub_degree:
            continue                                   # emitted, never expanded
        arcs = (out[u] if direction in (0, 2) else []) + (inc[u] if direction in (1, 2) else [])
        for v, lab in arcs:
            if allowed and lab not in allowed:
                continue
            if v not in depth:
                depth[v] = depth[u] + 1
                queue.append(v)
    return depth


def breadth_first(n, out_offsets, out_targets, labels, seeds, max_depth, direction=2, allowed_labels=(), hub_degree=0):
    """`Graph::traverse` with TraversalStrategy::BreadthFirst, restated with its ORDER: crates/graph-algorithms/src/algorithms/
    traversal.rs:197-261 (seeds deduplicated in first-occurrence order, FIFO queue, a node is marked and recorded when it is
    first reached, the arc that reached it is its discovery edge), :311-318 (`suppresses_hub`), and the arc order of
    model.rs:635-725: a node's outgoing row, its incoming row, or -- direction Both -- the two rows merged by neighbour
    (compare_arcs' first key; rows are sorted by it, node indexes follow node ids), the outgoing arc first on equal neighbours
    (the reference breaks that tie by graphify key, then edge id -- both arcs reach the same node, so only WHICH of the two
    edges is reported can differ), incoming self-loops skipped.  The incoming row of v lists its sources in ascending order,
    parallel edges in their outgoing-row order.
    Returns (visits, edges): visits = [(node, depth)] in discovery order; edges[i] belongs to the i-th non-seed visit:
    (current node, index of the stored edge in the outgoing arc array, 0 = followed along the edge / 1 = against it)."""
    from collections import deque
    out = [[] for _ in range(n)]
    inc = [[] for _ in range(n)]
    for u in range(n):
        for a in range(int(out_offsets[u]), int(out_offsets[u + 1])):
            lab = None if labels is None else int(labels[a])
            out[u].append((int(out_targets[a]), lab, a))
    for u in range(n):
        for (v, lab, a) in out[u]:
            inc[v].append((u, lab, a))
    for u in range(n):
        assert all(out[u][i][0] <= out[u][i + 1][0] for i in range(len(out[u]) - 1)), "outgoing rows must be sorted by neighbour"

    def arcs(u):
        if direction == 0:
            return [(v, lab, a, 0) for (v, lab, a) in out[u]]
        if direction == 1:
            return [(v, lab, a, 1) for (v, lab, a) in inc[u]]
        merged, i, j = [], 0, 0
        o, c = out[u], [x for x in inc[u] if x[0] != u]          # incoming self-loops are skipped (model.rs:691-696)
        while i < len(o) or j < len(c):
            if j >= len(c) or (i < len(o) and o[i][0] <= c[j][0]):
                merged.append(o[i] + (0,)); i += 1
            else:
                merged.append(c[j] + (1,)); j += 1
        return merged

    depth, visits, edges, seed_set, queue = {}, [], [], set(), deque()
    for s in seeds:
        s = int(s)
        if not 0 <= s < n:
            raise KeyError(f"unknown node {s}")
        if s not in depth:
            depth[s] = 0
            seed_set.add(s)
            queue.append(s)
            visits.append((s, 0))
    allowed = set(int(x) for x in allowed_labels)
    while queue:
        u = queue.popleft()
        if depth[u] >= max_depth:
            continue
        if u not in seed_set and hub_degree and len(out[u]) + len(inc[u]) >= hub_degree:
            continue
        for (v, lab, a, against) in arcs(u):
            if (allowed and lab not in allowed) or v in depth:
                continue
            depth[v] = depth[u] + 1
            queue.append(v)
            visits.append((v, depth[v]))
            edges.append((u, a, against))
    return visits, edges


def depth_first(n, out_offsets, out_targets, labels, seeds, max_depth, direction=2, allowed_labels=(), hub_degree=0):
    """`Graph::traverse` with TraversalStrategy::DepthFirst, restated: crates/graph-algorithms/src/algorithms/traversal.rs:263-309.
    Seeds are pushed in reverse (so the first seed is visited first) and marked when pushed; a node is recorded when it is
    POPPED; its unvisited, label-allowed neighbours -- one arc per neighbour, the first in arc order (`discovered.insert`) --
    are marked and pushed in reverse arc order, each with the arc that scheduled it as its discovery edge.  Arc order,
    hub policy and the returned shapes as in `breadth_first` above."""
    out = [[] for _ in range(n)]
    inc = [[] for _ in range(n)]
    for u in range(n):
        for a in range(int(out_offsets[u]), int(out_offsets[u + 1])):
            lab = None if labels is None else int(labels[a])
            out[u].append((int(out_targets[a]), lab, a))
    for u in range(n):
        for (v, lab, a) in out[u]:
            inc[v].append((u, lab, a))

    def arcs(u):
        if direction == 0:
            return [(v, lab, a, 0) for (v, lab, a) in out[u]]
        if direction == 1:
            return [(v, lab, a, 1) for (v, lab, a) in inc[u]]
        merged, i, j = [], 0, 0
        o, c = out[u], [x for x in inc[u] if x[0] != u]
        while i < len(o) or j < len(c):
            if j >= len(c) or (i < len(o) and o[i][0] <= c[j][0]):
                merged.append(o[i] + (0,)); i += 1
            else:
                merged.append(c[j] + (1,)); j += 1
        return merged

    seed_list = []
    for s in seeds:
        s = int(s)
        if not 0 <= s < n:
            raise KeyError(f"unknown node {s}")
        if s not in seed_list:
            seed_list.append(s)
    allowed = set(int(x) for x in allowed_labels)
    visited, stack, visits, edges = set(), [], [], []
    for s in reversed(seed_list):
        if s not in visited:
            visited.add(s)
            stack.append((s, 0, None))
    while stack:
        u, d, disc = stack.pop()
        visits.append((u, d))
        if disc is not None:
            edges.append(disc)
        if d >= max_depth or (u not in seed_list and hub_degree and len(out[u]) + len(inc[u]) >= hub_degree):
            continue
        chosen, seen = [], set()
        for (v, lab, a, against) in arcs(u):
            if v in visited or (allowed and lab not in allowed) or v in seen:
                continue
            seen.add(v)
            chosen.append((v, a, against))
        for (v, a, against) in reversed(chosen):
            visited.add(v)
            stack.append((v, d + 1, (u, a, against)))
    return visits, edges
