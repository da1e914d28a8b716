"""Shared helpers of the restricted-walk tests: the reference's fixtures for `restricted_filter_aware_search`
(crates/db/tests/production_support/vector/restricted.rs, cited per builder), random member / bridge topologies, and the
host twin of the kernel's algorithm body (tests/native/walk_twin.cpp = helix-db_amd/csrc/hvx_walk_core.h compiled by g++).
Test infrastructure only."""
import ctypes as C
import math
import os
import subprocess

import numpy as np

import fixtures as fx

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TWIN_SRC = os.path.join(HERE, "native", "walk_twin.cpp")
CORE_HDR = os.path.join(ROOT, "helix-db_amd", "csrc", "hvx_walk_core.h")
TWIN_LIB = os.path.join(HERE, "native", "_build", "libwalk_twin.so")

COUNTER_FIELDS = ("termination", "directory_scan_calls", "directory_rows", "directory_decoded_bytes", "directory_hits",
                  "simhash_row_requests", "companion_row_requests", "routing_rows", "bridge_rows", "bridge_frontier_pushes",
                  "neighbor_multi_get_calls", "vector_payload_requests", "vector_bytes", "distance_computations")


class Plan(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("k", "ef_filtered", "routing_rows", "bridge_rows", "vector_payloads", "sampled_seeds",
                                          "directory_seeds", "directory_enabled", "n_sample")]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in COUNTER_FIELDS]


class TwinArgs(C.Structure):
    _fields_ = [("n", C.c_uint32), ("s0", C.c_uint32), ("dim", C.c_uint32), ("entry", C.c_uint32), ("has_entry", C.c_uint32),
                ("b_cap", C.c_uint32), ("variant", C.c_uint32), ("qhash", C.c_uint64), ("order_seed", C.c_uint64),
                ("l0", C.c_void_p), ("node_hash", C.c_void_p), ("dir_code", C.c_void_p), ("dir_row", C.c_void_p),
                ("allowed", C.c_void_p), ("sample_rows", C.c_void_p), ("plan", Plan)]


DIST_FN = C.CFUNCTYPE(C.c_float, C.c_uint32)
_twin = None


def twin_lib():
    global _twin
    if _twin is not None:
        return _twin
    os.makedirs(os.path.dirname(TWIN_LIB), exist_ok=True)
    newest = max(os.path.getmtime(TWIN_SRC), os.path.getmtime(CORE_HDR))
    if not os.path.exists(TWIN_LIB) or os.path.getmtime(TWIN_LIB) < newest:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", TWIN_LIB, TWIN_SRC])
    L = C.CDLL(TWIN_LIB)
    L.walk_twin_run.restype = C.c_uint32
    L.walk_twin_run.argtypes = [C.POINTER(TwinArgs), DIST_FN, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(Counters)]
    L.walk_twin_prefix_offset.restype = C.c_uint32
    L.walk_twin_prefix_offset.argtypes = [C.c_uint32]
    L.walk_twin_order_code.restype = C.c_uint64
    L.walk_twin_order_code.argtypes = [C.c_uint64]
    _twin = L
    return L


class Image:
    """The arrays the device keeps for one index, derived from an oracle index (ids ascending = row order)."""

    def __init__(self, orc, ix, metric, kernel=None):
        self.orc, self.ix, self.metric = orc, ix, metric
        self.kernel = orc.K_AVX_FMA if kernel is None else kernel
        e = ix.export()
        self.ids = e["node_ids"]
        self.vec = e["vectors"]
        self.n, self.dim = self.vec.shape
        off, nb = e["l0_offsets"].astype(np.int64), e["l0_neighbors"]
        deg = np.diff(off)
        self.s0 = max(32, int((deg.max() if self.n else 0) + 31) // 32 * 32)
        self.l0 = np.full((self.n, self.s0), 0xFFFFFFFF, np.uint32)
        for i in range(self.n):
            r = np.searchsorted(self.ids, nb[off[i]:off[i + 1]])
            self.l0[i, :r.size] = np.sort(r)
        self.node_hash = ix.get_simhash()
        codes = np.array([orc.order_code(int(h)) for h in self.node_hash], np.uint64)
        order = np.lexsort((self.ids, codes))
        self.dir_code = np.ascontiguousarray(codes[order])
        self.dir_row = np.ascontiguousarray(order.astype(np.uint32))
        ent = e["entry_point"]
        self.has_entry = ent is not None
        self.entry = int(np.searchsorted(self.ids, ent)) if self.has_entry else 0

    def rows_of(self, ids):
        ids = np.asarray(ids, np.uint64)
        pos = np.searchsorted(self.ids, ids)
        pos = np.minimum(pos, max(self.n - 1, 0))
        ok = (self.ids[pos] == ids) if self.n else np.zeros(ids.size, bool)
        return np.where(ok, pos, 0xFFFFFFFF).astype(np.uint32)

    def candidate_inputs(self, allowed, sampled_seeds):
        """canonical candidate list -> (count incl. unknown ids, membership bitmap over rows, sample rows)"""
        al = np.unique(np.asarray(allowed, np.uint64))
        rows = self.rows_of(al)
        bm = np.zeros((self.n + 31) // 32 + 1, np.uint32)
        known = rows[rows != 0xFFFFFFFF]
        np.bitwise_or.at(bm, known >> 5, np.uint32(1) << (known & 31).astype(np.uint32))
        ranks = self.orc.deterministic_sample_ranks(int(al.size), int(sampled_seeds)) if al.size and sampled_seeds else []
        sample = rows[np.asarray(ranks, np.int64)] if len(ranks) else np.zeros(0, np.uint32)
        return int(al.size), bm, np.ascontiguousarray(sample, np.uint32)


def default_plan(orc, candidates, dim, k, ef, beam_percent=0):
    p = orc.restricted_plan(candidates, dim, k, ef, beam_percent)
    return {f: int(p[f]) for f in ("k", "ef_filtered", "routing_rows", "bridge_rows", "vector_payloads", "sampled_seeds", "directory_seeds")}, int(p["strategy"])


def run_twin(img, query, allowed, plan, directory, variant=0, order_seed=1):
    """-> (bad, ids, score_bits, counters dict)"""
    L = twin_lib()
    orc = img.orc
    q = np.ascontiguousarray(query, np.float32)
    na, bm, sample = img.candidate_inputs(allowed, min(plan["sampled_seeds"], 1 << 30))
    qhash = img.ix.query_simhash(q)
    qh = orc.header(img.metric, q)
    hdrs = [None] * img.n

    def dist(row):
        if hdrs[row] is None:
            hdrs[row] = orc.header(img.metric, img.vec[row])
        return float(orc.lib().orc_distance(img.metric, img.kernel, q.ctypes.data_as(orc.f32p), qh, img.vec[row].ctypes.data_as(orc.f32p),
                                            hdrs[row], img.dim))

    cb = DIST_FN(dist)
    a = TwinArgs(n=img.n, s0=img.s0, dim=img.dim, entry=img.entry, has_entry=1 if img.has_entry else 0,
                 b_cap=plan["bridge_rows"], variant=variant, qhash=qhash, order_seed=order_seed,
                 l0=img.l0.ctypes.data, node_hash=img.node_hash.ctypes.data, dir_code=img.dir_code.ctypes.data,
                 dir_row=img.dir_row.ctypes.data, allowed=bm.ctypes.data, sample_rows=sample.ctypes.data if sample.size else 0,
                 plan=Plan(k=plan["k"], ef_filtered=plan["ef_filtered"], routing_rows=plan["routing_rows"], bridge_rows=plan["bridge_rows"],
                           vector_payloads=plan["vector_payloads"], sampled_seeds=plan["sampled_seeds"],
                           directory_seeds=plan["directory_seeds"], directory_enabled=1 if directory else 0, n_sample=sample.size))
    keys = np.zeros(1024, np.uint64)
    n = C.c_uint32(0)
    st = Counters()
    bad = L.walk_twin_run(C.byref(a), cb, keys.ctypes.data, C.byref(n), C.byref(st))
    assert bad != 99, "twin refused the configuration"
    keys = keys[: n.value]
    top = keys[: min(n.value, plan["ef_filtered"])][: plan["k"]]
    rows = ((top >> np.uint64(1)) & np.uint64(0x7FFFFFFF)).astype(np.int64)
    bits = (top >> np.uint64(32)).astype(np.uint32)
    return bad, img.ids[rows], bits, {f: int(getattr(st, f)) for f in COUNTER_FIELDS}


def assert_same_as_oracle(got_ids, got_bits, got_counters, want, ef_filtered=None):
    """`want` = (rc, ids, scores, stats) of the oracle; every RestrictedSearchStats counter and the termination must agree"""
    rc, ids, scores, st = want
    assert rc == 0, rc
    assert got_ids.tolist() == ids.tolist(), (got_ids.tolist(), ids.tolist())
    assert np.asarray(got_bits, np.uint32).tolist() == scores.view(np.uint32).tolist()
    for f in COUNTER_FIELDS:
        assert got_counters[f] == st[f], (f, got_counters[f], st[f], got_counters, st)
    if ef_filtered is not None:
        assert st["ef_filtered"] == ef_filtered


# ------------------------------------------------------------------------------------------------ fixtures
def circle8(entity_id, n, dim=8):
    """tests/production_support/vector/restricted.rs:146-152 vector_for: (cos, sin)(tau * id / n) as f32, zero padded"""
    v = np.zeros(dim, np.float32)
    a = math.tau * entity_id / n
    v[0], v[1] = np.float32(math.cos(a)), np.float32(math.sin(a))
    return v


def seed_circle(orc, n, dim=8):
    """tests :171-243 seed_index: ids 1..n, Cosine, skip_neighbors rows, entry point 1"""
    ids = np.arange(1, n + 1, dtype=np.uint64)
    vec = np.stack([circle8(i, n, dim) for i in range(1, n + 1)])
    offs, nbrs = [0], []
    for i in range(1, n + 1):
        nbrs.extend(fx.skip_neighbors(i, n))
        offs.append(len(nbrs))
    ix = orc.Index(dim, orc.COSINE, m=32, m0=64)
    assert ix.seed(ids, vec, np.array(offs, np.uint64), np.array(nbrs, np.uint64), entry_point=1, max_layer=0) == orc.OK
    ix.set_simhash(42)
    return ix


def gulf(orc, metric):
    """tests :313-367 seed_three_edge_filtered_gulf: 1 -> 2 -> 3 -> 1001, entry 1"""
    ix = orc.Index(2, metric, m=16, m0=32)
    ids = np.array([1, 2, 3, 1001], np.uint64)
    vec = np.array([[0, 1], [0, 1], [0, 1], [1, 0]], np.float32)
    assert ix.seed(ids, vec, np.array([0, 1, 2, 3, 3], np.uint64), np.array([2, 3, 1001], np.uint64), entry_point=1, max_layer=0) == orc.OK
    ix.set_simhash(42)
    return ix


def competing_bridges(orc):
    """tests :369-423: entry 1 -> {2, 3}; 2 -> 1001, 3 -> 1002"""
    ix = orc.Index(2, orc.COSINE, m=16, m0=32)
    ids = np.array([1, 2, 3, 1001, 1002], np.uint64)
    vec = np.array([[0, 1], [1, 0], [-1, 0], [1, 0], [1, 0]], np.float32)
    assert ix.seed(ids, vec, np.array([0, 2, 3, 4, 4, 4], np.uint64), np.array([2, 3, 1001, 1002], np.uint64), entry_point=1, max_layer=0) == orc.OK
    ix.set_simhash(42)
    return ix


def identical_rows(orc, n=300, dim=8):
    """tests :245-311: n identical vectors, empty graph"""
    ids = np.arange(1, n + 1, dtype=np.uint64)
    vec = np.zeros((n, dim), np.float32)
    vec[:, 0] = 1.0
    ix = orc.Index(dim, orc.COSINE, m=16, m0=32)
    assert ix.seed(ids, vec, np.zeros(n + 1, np.uint64), np.zeros(0, np.uint64), entry_point=1, max_layer=0) == orc.OK
    ix.set_simhash(42)
    return ix


def random_graph(orc, n, dim, metric, seed, m0=32, id_gap=False):
    """A built HNSW over Gaussian rows (the oracle's insert_hnsw); ids either 1..n or with gaps (unknown candidate ids)."""
    rng = np.random.default_rng(seed)
    vec = rng.standard_normal((n, dim)).astype(np.float32)
    ids = (np.arange(n, dtype=np.uint64) * (3 if id_gap else 1)) + 1
    ix = orc.Index(dim, metric, m=m0 // 2, m0=m0, ef_construction=64)
    ml = float(orc.default_ml(m0 // 2))
    for i in range(n):
        u = float(rng.random())
        assert ix.insert(int(ids[i]), vec[i], int(orc.select_layer(ml, max(u, 1e-9)))) == orc.OK
    ix.set_simhash(42)
    return ix, ids, vec
