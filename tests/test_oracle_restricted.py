"""Pins the oracle's restatement of the restricted-search planner and the filter-aware walk (oracle/hvx_oracle_restricted.inc;
crates/db/src/search/vector/restricted.rs:196-260,303-462,528-1148) with the reference's own fixtures for this path:
crates/db/tests/production_support/vector/restricted.rs (cited per test).  CPU only."""
import math

import numpy as np

import fixtures as fx


def circle8(entity_id, n, dim=8):
    """restricted.rs (tests) :146-152 vector_for: (cos, sin)(tau * id / n) as f32, zero padded to `dim`"""
    v = np.zeros(dim, np.float32)
    a = math.tau * entity_id / n
    v[0], v[1] = np.float32(math.cos(a)), np.float32(math.sin(a))
    return v


def seed_circle(orc, n, dim=8):
    """tests :171-243 seed_index: ids 1..n, Cosine, skip_neighbors rows, entry point 1, SimHash directory enabled"""
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


def exact_ids(orc, ix, q, allowed, k):
    rc, ids, _ = ix.flat(q, k, allowed=np.asarray(allowed, np.uint64))
    assert rc == orc.OK
    return ids.tolist()


def test_admission_bounds_exact_work_by_cardinality_or_bytes(orc):
    """tests :459-553"""
    P = orc.restricted_plan
    assert P(256, 1536, 10, 100)["strategy"] == orc.RESTRICTED_EXACT
    assert P(256, 5000, 10, 100)["strategy"] == orc.RESTRICTED_FILTERED
    assert P(257, 2, 10, 100)["strategy"] == orc.RESTRICTED_FILTERED
    b = P(1000, 1536, 10, 100)
    assert (b["strategy"], b["ef_filtered"], b["sampled_seeds"], b["directory_seeds"], b["vector_payloads"]) == (orc.RESTRICTED_FILTERED, 150, 64, 256, 800)
    assert (b["routing_rows"], b["bridge_rows"]) == (150 * 16, 150 * 8)
    for mult, want in ((2, 200), (4, 400)):
        assert P(1000, 1536, 10, 100, beam_percent=mult * 100)["ef_filtered"] == want
    for pct, want in ((100, 100), (150, 150), (200, 200)):
        b = P(1000, 1536, 10, 100, beam_percent=pct)
        assert (b["ef_filtered"], b["sampled_seeds"], b["directory_seeds"], b["vector_payloads"]) == (want, 64, 256, 800)
    ranks = orc.deterministic_sample_ranks(100_000, 256)
    assert len(ranks) == 256 and all(a < b for a, b in zip(ranks, ranks[1:])) and ranks[0] == 0 and ranks[-1] == 99_999
    assert orc.deterministic_sample_ranks(1, 1) == [0]
    assert orc.deterministic_sample_ranks(2, 1) == [0]          # from_ids([3, 7]).sample(1) == [3]
    assert orc.deterministic_sample_ranks(2, 8) == [0, 1]       # ... .sample(8) == [3, 7]


def test_result_count_clamps_before_enforcing_the_payload_limit(orc):
    """tests :555-591: k is clamped to the candidate count first, then must be <= 800"""
    ix = seed_circle(orc, 1200)
    q = circle8(5, 1200)
    rc, ids, _, _ = ix.search_restricted(q, 800, 800, np.arange(1, 1001))
    assert rc == orc.OK and len(ids) <= 800
    rc, ids, _, _ = ix.search_restricted(q, 1000, 1000, np.arange(1, 801))       # MAX + 200 over 800 candidates -> 800
    assert rc == orc.OK
    rc, _, _, _ = ix.search_restricted(q, 801, 801, np.arange(1, 1001))         # MAX + 1 over 1 000 candidates -> rejected
    assert rc == orc.ERR_K_RANGE
    p = orc.restricted_plan(1000, 1536, 800, 800)
    assert p["k"] == 800 and p["vector_payloads"] == 800


def test_candidate_states_deduplicate_reject_overflow_and_keep_empty_explicit(orc):
    """tests :593-610 + :642-676: empty -> no results; duplicates collapse; 1 000 001 unique ids are rejected"""
    ix = seed_circle(orc, 64)
    q = circle8(3, 64)
    rc, ids, _, st = ix.search_restricted(q, 10, 64, np.zeros(0, np.uint64))
    assert rc == orc.OK and ids.size == 0 and st["strategy"] == 0
    rc, ids, _, st = ix.search_restricted(q, 10, 64, [7, 7, 3])
    assert rc == orc.OK and sorted(ids.tolist()) == [3, 7] and st["strategy"] == orc.RESTRICTED_EXACT
    rc, _, _, _ = ix.search_restricted(q, 10, 64, np.arange(0, 1_000_001, dtype=np.uint64))
    assert rc == orc.ERR_CANDIDATE_LIMIT


def gulf(orc, metric):
    """tests :313-367 seed_three_edge_filtered_gulf: 1 -> 2 -> 3 -> 1001, entry 1, no directory"""
    ix = orc.Index(2, metric, m=16, m0=32)
    ids = np.array([1, 2, 3, 1001], np.uint64)
    vec = np.array([[0, 1], [0, 1], [0, 1], [1, 0]], np.float32)
    assert ix.seed(ids, vec, np.array([0, 1, 2, 3, 3], np.uint64), np.array([2, 3, 1001], np.uint64), entry_point=1, max_layer=0) == orc.OK
    ix.set_simhash(42)
    return ix


def test_directoryless_acorn_crosses_a_three_edge_filtered_gulf_without_nonmember_vectors(orc):
    """tests :964-993, :1022-1045: three rejected bridge rows lead to the only member; exactly one payload is read"""
    allowed = np.arange(1000, 1257, dtype=np.uint64)
    for metric in (orc.COSINE, orc.L2SQ, orc.L1):
        ix = gulf(orc, metric)
        rc, ids, _, st = ix.search_restricted([1.0, 0.0], 10, 100, allowed, directory=False)
        assert rc == orc.OK and ids.tolist() == [1001] and st["strategy"] == orc.RESTRICTED_FILTERED
        if metric == orc.COSINE:
            assert st["directory_scan_calls"] == 0 and st["bridge_rows"] == 3 and st["bridge_frontier_pushes"] >= st["bridge_rows"]
            assert st["vector_payload_requests"] == 1 and st["distance_computations"] == 1


def test_simhash_guides_one_bounded_bridge_toward_the_relevant_disconnected_region(orc):
    """tests :369-423, :1047-1101: two rejected neighbours of the entry; the one whose SimHash is closer to the query is
    expanded first and its member neighbour takes the single payload of the budget"""
    ix = orc.Index(2, orc.COSINE, m=16, m0=32)
    ids = np.array([1, 2, 3, 1001, 1002], np.uint64)
    vec = np.array([[0, 1], [1, 0], [-1, 0], [1, 0], [1, 0]], np.float32)
    assert ix.seed(ids, vec, np.array([0, 2, 3, 4, 4, 4], np.uint64), np.array([2, 3, 1001, 1002], np.uint64), entry_point=1, max_layer=0) == orc.OK
    ix.set_simhash(42)
    rc, got, _, st = ix.restricted_filter_aware_search([1.0, 0.0], np.arange(1001, 1258), k=1, ef_filtered=1, routing_rows=2, bridge_rows=2,
                                                       vector_payloads=1, sampled_seeds=0, directory_seeds=0)
    assert rc == orc.OK and got.tolist() == [1001]
    assert st["bridge_rows"] == 2 and st["vector_payload_requests"] == 1 and st["distance_computations"] == 1 and st["bridge_frontier_pushes"] >= 3


def test_explicit_filtered_budgets_record_the_exact_termination_reason(orc):
    """tests :1103-1224"""
    ix = gulf(orc, orc.COSINE)
    allowed = np.arange(1000, 1257)
    for budgets, want in ((dict(ef_filtered=1, routing_rows=0, bridge_rows=1, vector_payloads=1), orc.TERM_ROUTING_BUDGET),
                          (dict(ef_filtered=1, routing_rows=4, bridge_rows=0, vector_payloads=1), orc.TERM_BRIDGE_BUDGET),
                          (dict(ef_filtered=1, routing_rows=4, bridge_rows=2, vector_payloads=0), orc.TERM_VECTOR_BUDGET)):
        rc, got, _, st = ix.restricted_filter_aware_search([1.0, 0.0], allowed, k=1, sampled_seeds=0, directory_seeds=0, **budgets)
        assert rc == orc.OK and got.size == 0 and st["termination"] == want
        assert st["routing_rows"] <= budgets["routing_rows"] and st["bridge_rows"] <= budgets["bridge_rows"]
        assert st["vector_payload_requests"] <= budgets["vector_payloads"]
    rc, got, _, st = ix.restricted_filter_aware_search([1.0, 0.0], np.arange(1, 258), k=1, ef_filtered=1, routing_rows=4, bridge_rows=2,
                                                       vector_payloads=0, sampled_seeds=0, directory_seeds=0)
    assert got.size == 0 and st["termination"] == orc.TERM_VECTOR_BUDGET


def test_directory_entries_seed_vectors_without_re_reading_point_simhash_rows(orc):
    """tests :245-311, :935-961: 300 identical vectors, empty graph, directory on: >= 256 directory hits and exactly the 64
    sampled seeds cost a SimHash read (directory entries carry their own key)"""
    n, dim = 300, 8
    ids = np.arange(1, n + 1, dtype=np.uint64)
    vec = np.zeros((n, dim), np.float32)
    vec[:, 0] = 1.0
    ix = orc.Index(dim, orc.COSINE, m=16, m0=32)
    assert ix.seed(ids, vec, np.zeros(n + 1, np.uint64), np.zeros(0, np.uint64), entry_point=1, max_layer=0) == orc.OK
    ix.set_simhash(42)
    rc, got, _, st = ix.search_restricted(circle8(7, n, dim), 10, 100, ids)
    assert rc == orc.OK and got.size > 0
    assert st["directory_hits"] >= 256 and st["simhash_row_requests"] == 64
    assert st["directory_scan_calls"] <= 64 and st["directory_scan_calls"] % 8 == 0
    assert st["directory_rows"] <= 65_536 and st["directory_decoded_bytes"] <= 4 * 1024 * 1024


def test_exact_and_filter_aware_paths_enforce_membership_and_recall_budgets(orc):
    """tests :1226-1285: 512 x 8-D circle fixture, ef 64, k 10: the small set is answered exactly; allowed = ids not divisible by
    3 takes the filtered walk with recall@10 >= 0.95 inside every budget"""
    n, dim, k = 512, 8, 10
    ix = seed_circle(orc, n, dim)
    small = list(range(1, 65)) + [10, 10, 9999]
    q = circle8(17, n, dim)
    rc, got, _, st = ix.search_restricted(q, k, 64, small)
    assert rc == orc.OK and st["strategy"] == orc.RESTRICTED_EXACT
    assert all(i in set(small) for i in got.tolist())
    assert got.tolist() == exact_ids(orc, ix, q, small, k)
    allowed = [i for i in range(1, n + 1) if i % 3 != 0]
    matched = observed = 0
    for qid in (1, 43, 87, 129, 211, 307, 401, 509):
        q = circle8(qid, n, dim)
        rc, got, _, st = ix.search_restricted(q, k, 64, allowed)
        assert rc == orc.OK and st["strategy"] == orc.RESTRICTED_FILTERED
        exact = set(exact_ids(orc, ix, q, allowed, k))
        matched += sum(1 for i in got.tolist() if i in exact)
        observed += k
        assert all(i % 3 != 0 for i in got.tolist())
        assert st["directory_scan_calls"] <= 64 and st["directory_rows"] <= 65_536 and st["directory_decoded_bytes"] <= 4 * 1024 * 1024
        assert st["routing_rows"] <= st["ef_filtered"] * 16 and st["bridge_rows"] <= st["ef_filtered"] * 8
        assert st["vector_payload_requests"] <= st["ef_filtered"] * 8
        assert st["distance_computations"] == st["vector_payload_requests"]
    assert matched / observed >= 0.95, matched / observed
