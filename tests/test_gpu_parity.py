"""Parity of the gfx950 path (through the C ABI) against the CPU oracle: ids, f32 score BITS and the
SearchStats counters must be identical.  Run with `-m gpu` on an MI355X."""
import os
import numpy as np
import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hv():
    import pyhvx
    pyhvx.lib()  # fails loudly if the HIP extension is missing
    return pyhvx


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


_ORACLE_CACHE = {}


def build_oracle(orc, data, metric, levels, m=16, m0=32, efc=100, kernel=None, ids=None, cached=False):
    """the oracle's sequential insertion of `data`; cached=True (tests that only READ the index, parametrised over device kernel paths):
    the same rows, levels and parameters are inserted once per session -- the CPU build was most of the GPU suite's 9 minutes (round 6)"""
    key = None
    if cached:
        import hashlib
        key = (hashlib.sha1(np.ascontiguousarray(data).tobytes()).hexdigest(), hashlib.sha1(np.ascontiguousarray(levels).tobytes()).hexdigest(),
               metric, m, m0, efc, kernel, None if ids is None else hashlib.sha1(np.ascontiguousarray(ids).tobytes()).hexdigest())
        if key in _ORACLE_CACHE:
            return _ORACLE_CACHE[key]
    ix = orc.Index(data.shape[1], metric, kernel=orc.K_AVX_FMA if kernel is None else kernel, m=m, m0=m0,
                   ef_construction=efc)
    for i in range(data.shape[0]):
        nid = i if ids is None else int(ids[i])
        assert ix.insert(nid, data[i], int(levels[i])) == orc.OK
    if key is not None:
        if len(_ORACLE_CACHE) >= 3:
            _ORACLE_CACHE.pop(next(iter(_ORACLE_CACHE)))
        _ORACLE_CACHE[key] = ix
    return ix


def assert_hnsw_equal(orc, hv, oix, gix, queries, k, ef):
    ids, sc, cnt, stats = gix.search_batch(queries, hv.SearchParams(k).with_ef(ef))
    tot = dict(expansion_steps=0, neighbors_examined=0, vectors_loaded=0, distance_computations=0)
    for qi in range(queries.shape[0]):
        rc, oid, osc, ost = oix.search(queries[qi], k, ef, with_stats=True)
        assert rc == orc.OK
        assert cnt[qi] == oid.size, f"query {qi}: count {cnt[qi]} vs oracle {oid.size}"
        assert ids[qi, :cnt[qi]].tolist() == oid.tolist(), f"query {qi}: ids differ"
        assert bits(sc[qi, :cnt[qi]]).tolist() == bits(osc).tolist(), f"query {qi}: score bits differ"
        for key in tot:
            tot[key] += ost[key]
    for key in tot:
        assert stats[key] == tot[key], f"{key}: device {stats[key]} vs oracle {tot[key]}"
    assert stats["tie_overflow_queries"] == 0


# --- reference golden: src/search/vector/index.rs:2318-2412 ---
def test_phase0_golden_on_device(orc, hv):
    oix = orc.Index(2, orc.COSINE, m=4, m0=8, ef_construction=16)
    for (nid, v), lvl in zip([(1, [1, 0]), (2, [0, 1]), (3, [-1, 0]), (4, [0, -1])], [0, 1, 2, 0]):
        oix.insert(nid, v, lvl)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=2, metric=hv.COSINE, m=4, m0=8)
    ids, sc, cnt, st = gix.search_batch(np.array([[1.0, 0.0]], np.float32), hv.SearchParams(4).with_ef(16))
    assert cnt[0] == 4 and ids[0].tolist() == [1, 2, 4, 3]
    assert bits(sc[0]).tolist() == bits([0.0, 0.5, 0.5, 1.0]).tolist()
    assert (st["expansion_steps"], st["neighbors_examined"], st["vectors_loaded"], st["distance_computations"]) == (4, 12, 3, 4)
    # single-query operator surface
    res = gix.search([1.0, 0.0], hv.SearchParams(4).with_ef(16))
    assert [r.entity_id for r in res] == [1, 2, 4, 3]


# --- reference golden: src/index_lifecycle/vector/driver.rs:5132-5212 ---
def test_driver_bruteforce_equivalence_on_device(orc, hv):
    vecs = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [4, 0, 0], [0, 5, 0], [0, 0, 6], [7, 7, 7]], np.float32)
    q = np.array([[0.25, 0.5, 0.75]], np.float32)
    oix = build_oracle(orc, vecs, orc.L2SQ, [0, 1, 0, 0, 2, 0, 1, 0], efc=200)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=3, metric=hv.EUCLIDEAN)
    assert_hnsw_equal(orc, hv, oix, gix, q, 8, 8)
    fid, fsc, fcnt, _ = gix.flat_search_batch(q, 8)
    hid, hsc, hcnt, _ = gix.search_batch(q, hv.SearchParams(8).with_ef(8))
    assert fid.tolist() == hid.tolist() and bits(fsc).tolist() == bits(hsc).tolist()


# --- reference golden: tests/production_support/vector/restricted.rs:710-800 ---
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_exact_restricted_scan_on_device(orc, hv, metric):
    ids = np.array([1, 2, 3], np.uint64)
    vec = np.array([[1, 0], [1, 0], [-1, 0]], np.float32)
    gix = hv.ValidatedVectorReadIndex.managed(dim=2, metric=metric, node_ids=ids, vectors=vec,
                                              l0_offsets=np.zeros(4, np.uint64), l0_neighbors=np.zeros(0, np.uint64),
                                              entry_point=1)
    res = gix.search_restricted([1.0, 0.0], hv.SearchParams(10), hv.RestrictedVectorCandidates.from_ids([3, 2, 1]))
    assert [r.entity_id for r in res] == [1, 2, 3]
    res = gix.search_restricted([1.0, 0.0], hv.SearchParams(10), hv.RestrictedVectorCandidates.from_ids([1, 9999]))
    assert [r.entity_id for r in res] == [1]
    assert gix.search_restricted([1.0, 0.0], hv.SearchParams(10), hv.RestrictedVectorCandidates.from_ids([])) == []


# --- reference fixture: src/search/vector/scale_contracts.rs (hand-made graph, degree up to 34, M0=64) ---
@pytest.mark.parametrize("n", [24, 10_000])
def test_circle_fixture_on_device(orc, hv, n):
    ids, vec, offs, nbrs = fx.circle_index_arrays(n)
    oix = orc.Index(2, orc.COSINE, m=32, m0=64, ef_construction=200)
    assert oix.seed(ids, vec, offs, nbrs, entry_point=1, max_layer=0) == orc.OK
    gix = hv.ValidatedVectorReadIndex.managed(dim=2, metric=hv.COSINE, node_ids=ids, vectors=vec, l0_offsets=offs,
                                              l0_neighbors=nbrs, entry_point=1, max_layer=0, m=32, m0=64)
    q = np.stack(fx.circle_queries(n))
    assert_hnsw_equal(orc, hv, oix, gix, q, 10, 64)
    gid, _, _, _ = gix.search_batch(q, hv.SearchParams(10).with_ef(64))
    tid, _, _, _ = gix.flat_search_batch(q, 10)
    assert fx.recall_at_k(gid, tid) >= 0.995


CASES = [
    # (n, dim, metric, m, m0, efc, ef, k, nq)
    (2000, 32, 1, 16, 32, 100, 64, 10, 64),
    (2000, 100, 1, 16, 32, 100, 128, 10, 64),    # 100 = 3*32 + 4: scalar tail
    (1500, 128, 0, 16, 32, 100, 128, 10, 64),    # cosine
    (1200, 48, 2, 8, 16, 60, 100, 5, 32),        # manhattan (sequential order)
    (1500, 768, 1, 16, 32, 80, 128, 10, 48),     # the bench shape
    (1500, 20, 1, 16, 32, 80, 200, 20, 32),      # dim < 32: scalar path, R=4 beam
    (1500, 64, 0, 16, 32, 80, 400, 50, 16),      # R=8 beam
]


@pytest.mark.parametrize("n,dim,metric,m,m0,efc,ef,k,nq", CASES)
def test_hnsw_bit_exact_vs_oracle(orc, hv, n, dim, metric, m, m0, efc, ef, k, nq):
    rng = np.random.default_rng(1000 + dim + metric)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    lv = fx.draw_levels(n, m, seed=dim)
    ids = np.arange(n, dtype=np.uint64) * 3 + 7  # non-contiguous external ids
    oix = build_oracle(orc, data, metric, lv, m=m, m0=m0, efc=efc, ids=ids)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=metric, m=m, m0=m0)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    assert_hnsw_equal(orc, hv, oix, gix, q, k, ef)


WAVE_CASES = [
    # (n, dim, metric, m, m0, efc, ef, k, nq) -- shapes served by the one-wavefront-per-query kernel
    (1500, 128, 1, 16, 32, 80, 128, 10, 48),
    (1200, 256, 0, 16, 32, 80, 100, 10, 32),
    (1200, 384, 1, 16, 32, 80, 128, 10, 32),     # 12 chunks: unrolled since round 6 (MiniLM-class embeddings)
    (1000, 384, 0, 16, 32, 60, 100, 10, 24),
    (1200, 512, 1, 16, 32, 60, 128, 10, 32),
    (1500, 768, 0, 16, 32, 80, 128, 10, 32),
    (1000, 1024, 1, 8, 16, 60, 64, 10, 24),
    (1000, 1536, 1, 16, 32, 60, 100, 10, 24),
    (2500, 128, 1, 16, 32, 80, 300, 50, 24),     # R=6 beam
    (3000, 128, 1, 32, 64, 100, 160, 10, 24),    # 64-id rows
]


@pytest.mark.parametrize("path", ["wave", "general", "wave-spill", "pair", "pair-spill", "pair1"])
@pytest.mark.parametrize("n,dim,metric,m,m0,efc,ef,k,nq", WAVE_CASES)
def test_hnsw_kernels_agree_with_oracle(orc, hv, path, n, dim, metric, m, m0, efc, ef, k, nq):
    """All three HNSW kernels (one wavefront per query: hvx_hnsw_wave.h; owner + gatherer wavefront pair: hvx_hnsw_pair.h, round 4;
    the general hvx_hnsw.hip) and the LDS-table -> HBM-bitmap spill paths give the oracle's ids, score bits and counters."""
    rng = np.random.default_rng(77 + dim + metric)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    lv = fx.draw_levels(n, m, seed=dim + 1)
    oix = build_oracle(orc, data, metric, lv, m=m, m0=m0, efc=efc, cached=True)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=metric, m=m, m0=m0)
    if path == "general":
        gix.set_option(hv.OPT_HNSW_GENERAL_KERNEL, 1)
    # (the default picks the pair kernel for one-query-per-SIMD handles; 2 = three gatherer wavefronts where built, 3 = one)
    gix.set_option(hv.OPT_HNSW_PAIR, 3 if path == "pair1" else 2 if path.startswith("pair") else 1)
    if path.endswith("-spill"):
        gix.set_option(hv.OPT_WAVE_LOG2CAP, 8)  # 256-slot table: spills after ~128 visited ids
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    assert_hnsw_equal(orc, hv, oix, gix, q, k, ef)
    assert_hnsw_equal(orc, hv, oix, gix, q[: nq // 2], k, ef)  # second launch: visited state handed back clean


FLOAT_KERNELS = ["scalar", "sse", "avx", "avx_fma", "neon"]


def _kernel_pair(orc, hv, name):
    return {"scalar": (orc.K_SCALAR, hv.KERNEL_SCALAR), "sse": (orc.K_SSE, hv.KERNEL_SSE), "avx": (orc.K_AVX, hv.KERNEL_AVX),
            "avx_fma": (orc.K_AVX_FMA, hv.KERNEL_AVX_FMA), "neon": (orc.K_NEON, hv.KERNEL_NEON)}[name]


@pytest.mark.parametrize("metric,dim", [(1, 96), (0, 88), (1, 24), (0, 200), (1, 12)])
@pytest.mark.parametrize("kernel", FLOAT_KERNELS)
def test_other_float_kernels(orc, hv, kernel, metric, dim):
    """Every FloatSimd kernel of spaces/simple.rs:45-112 has its summation tree on the device (round 5: SSE and NEON -- 4 x 4-lane
    accumulators, 16 floats per iteration, hsum128 vs vaddvq; the reference's CI runs on aarch64): strict HNSW search, the non-strict
    production-default arm, the exact scan and the restricted exact scan return the oracle's ids, score BITS and counters under each
    of them.  Dimensions: a multiple of 32, 16-multiples with both kinds of tail (88 = 64 + 24 for AVX, 80 + 8 for the 128-bit
    kernels), 16 <= dim < 32 (only the 128-bit kernels vectorise), a large one, and dim < 16 (scalar everywhere)."""
    n = 900
    rng = np.random.default_rng(5 + dim)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    lv = fx.draw_levels(n, 16, seed=3)
    ok, hk = _kernel_pair(orc, hv, kernel)
    oix = build_oracle(orc, data, metric, lv, kernel=ok, efc=64)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=metric, float_kernel=hk)
    q = rng.standard_normal((16, dim)).astype(np.float32)
    assert_hnsw_equal(orc, hv, oix, gix, q, 10, 64)
    # exact scan + restricted exact scan (restricted.rs:753-835)
    fid, fsc, fcnt, _ = gix.flat_search_batch(q, 10)
    allowed = np.arange(3, n, 7, dtype=np.uint64)
    rid, rsc, rcnt = gix.search_restricted_batch(q, hv.SearchParams(10), hv.RestrictedVectorCandidates.from_ids(allowed))
    for qi in range(q.shape[0]):
        rc, tid, tsc = oix.flat(q[qi], 10)
        assert fid[qi, :fcnt[qi]].tolist() == tid.tolist() and bits(fsc[qi, :fcnt[qi]]).tolist() == bits(tsc).tolist()
        rc, tid, tsc = orc.flat_matrix(metric, data[allowed.astype(np.int64)], q[qi], 10, kernel=ok)
        assert rid[qi, :rcnt[qi]].tolist() == allowed[tid].tolist() and bits(rsc[qi, :rcnt[qi]]).tolist() == bits(tsc).tolist()
    # the production-default arm (SearchParams::new(k): pre-sampling + RNG; SimHash filter for cosine) on the generic build
    cfg = hv.SimHashConfig.default()
    oix.set_simhash(int(cfg.seed))
    gix.set_simhash(cfg)
    assert_params_equal(orc, hv, oix, gix, q, hv.SearchParams.new(10), cfg)


# --- BASELINE config #1: 10k x 128 f32 from the reference generator, flat k=10 ---
def test_flat_scan_config1_bit_exact(orc, hv):
    n = 10_000
    m = fx.lifecycle_matrix(n)
    ids = np.arange(n, dtype=np.uint64)
    gix = hv.ValidatedVectorReadIndex.managed(dim=128, metric=hv.EUCLIDEAN, node_ids=ids, vectors=m,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64))
    qs = [fx.lifecycle_vector(0)]
    for j in range(1, 64):
        v = fx.lifecycle_vector((1_000_003 * j) % n).copy()
        v[j] += np.float32(1e-3)
        qs.append(v)
    q = np.stack(qs)
    gid, gsc, gcnt, _ = gix.flat_search_batch(q, 10)
    one_id, one_sc, _, _ = gix.flat_search_batch(q[:1], 10)       # single query (the config as quoted)
    assert one_id[0].tolist() == gid[0].tolist() and one_id[0, 0] == 0 and one_sc[0, 0] == 0.0
    for qi in range(q.shape[0]):
        rc, oid, osc = orc.flat_matrix(orc.L2SQ, m, q[qi], 10)
        assert rc == orc.OK and gcnt[qi] == 10
        assert gid[qi].tolist() == oid.tolist()
        assert bits(gsc[qi]).tolist() == bits(osc).tolist()


@pytest.mark.parametrize("metric,dim,n,k", [(0, 128, 3000, 10), (2, 40, 3000, 7), (1, 768, 5000, 100), (1, 7, 300, 400)])
def test_flat_scan_metrics_and_k(orc, hv, metric, dim, n, k):
    rng = np.random.default_rng(dim)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    data[7] = data[3]  # exact duplicate row: tie broken by id
    ids = np.arange(n, dtype=np.uint64) + 100
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=metric, node_ids=ids, vectors=data,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64))
    q = rng.standard_normal((9, dim)).astype(np.float32)
    q[0] = data[3]
    gid, gsc, gcnt, _ = gix.flat_search_batch(q, k)
    for qi in range(q.shape[0]):
        rc, oid, osc = orc.flat_matrix(metric, data, q[qi], k)
        assert gcnt[qi] == oid.size == min(k, n)
        assert (gid[qi, :gcnt[qi]] - 100).tolist() == oid.tolist()
        assert bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
    assert gid[0, :2].tolist() == [103, 107]


def test_cosine_extremes_take_f64_fallback(orc, hv):
    """distance/cosine.rs:128-141: f32::MAX and subnormal rows stay finite through the f64 path."""
    fmax = np.finfo(np.float32).max
    tiny = np.array([1], np.uint32).view(np.float32)[0]
    data = np.array([[fmax, fmax], [tiny, tiny], [1.0, 0.0], [-1.0, 0.5]], np.float32)
    ids = np.arange(4, dtype=np.uint64)
    gix = hv.ValidatedVectorReadIndex.managed(dim=2, metric=hv.COSINE, node_ids=ids, vectors=data,
                                              l0_offsets=np.zeros(5, np.uint64), l0_neighbors=np.zeros(0, np.uint64))
    q = np.array([[fmax, fmax], [tiny, tiny], [0.3, -0.2]], np.float32)
    gid, gsc, gcnt, _ = gix.flat_search_batch(q, 4)
    for qi in range(3):
        rc, oid, osc = orc.flat_matrix(orc.COSINE, data, q[qi], 4)
        assert gid[qi].tolist() == oid.tolist()
        assert bits(gsc[qi]).tolist() == bits(osc).tolist()
    assert gsc[0, 0] <= np.finfo(np.float32).eps


def test_query_validation_statuses(orc, hv):
    n, dim = 64, 8
    rng = np.random.default_rng(0)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64)
    zeros = dict(l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), entry_point=0)
    l2 = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=ids, vectors=data, **zeros)
    cs = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.COSINE, node_ids=ids, vectors=data, **zeros)
    q = rng.standard_normal((4, dim)).astype(np.float32)
    q[1, 3] = np.nan
    q[2, :] = 0.0
    q[3, 0] = np.float32(orc.lib().orc_component_limit(orc.L2SQ, dim)) * np.float32(2)
    *_, st = l2.search_batch(q, hv.SearchParams(3), per_query_status=True)
    assert st.tolist() == [hv.OK, hv.ERR_NONFINITE, hv.OK, hv.ERR_MAGNITUDE]
    ids_c, _, cnt_c, _, st = cs.search_batch(q, hv.SearchParams(3), per_query_status=True)
    assert st.tolist() == [hv.OK, hv.ERR_NONFINITE, hv.ERR_ZERO_NORM, hv.OK]
    assert cnt_c[1] == 0 and cnt_c[2] == 0
    with pytest.raises(hv.HelixDbError) as e:
        cs.search_batch(q, hv.SearchParams(3))
    assert e.value.status == hv.ERR_NONFINITE and e.value.is_invalid_vector_input()
    # rejected queries keep their validation status through the exact-scan entry points too (flat, restricted, fused
    # prefilter): their NaN scores are not an invariant violation (flat_select_kernel clobbered the status)
    *_, st = cs.flat_search_batch(q, 3, per_query_status=True)
    assert st.tolist() == [hv.OK, hv.ERR_NONFINITE, hv.ERR_ZERO_NORM, hv.OK]
    *_, st = l2.flat_search_batch(q, 3, per_query_status=True)
    assert st.tolist() == [hv.OK, hv.ERR_NONFINITE, hv.OK, hv.ERR_MAGNITUDE]
    with pytest.raises(hv.HelixDbError) as e:
        cs.flat_search_batch(q, 3)
    assert e.value.status == hv.ERR_NONFINITE and e.value.is_invalid_vector_input()
    with pytest.raises(hv.HelixDbError) as e:
        cs.search_restricted_batch(q[2:3], hv.SearchParams(3), hv.RestrictedVectorCandidates.from_ids(np.arange(20)))
    assert e.value.status == hv.ERR_ZERO_NORM and e.value.is_invalid_vector_input()
    with pytest.raises(hv.HelixDbError) as e:
        l2.search_batch(q[:, :4], hv.SearchParams(3))
    assert e.value.status == hv.ERR_DIMENSION
    with pytest.raises(hv.HelixDbError) as e:
        hv.SearchParams(5).with_ef(4)
    assert e.value.status == hv.ERR_K_RANGE
    # stored rows are validated at import (decode_item_borrowed, mod.rs:889-949)
    bad = data.copy(); bad[5, 2] = np.inf
    with pytest.raises(hv.HelixDbError) as e:
        hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=ids, vectors=bad, **zeros)
    assert e.value.status == hv.ERR_NONFINITE
    # empty index: no results, not an error (search.rs:1128)
    empty = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.zeros(0, np.uint64),
                                                vectors=np.zeros((0, dim), np.float32), l0_offsets=np.zeros(1, np.uint64),
                                                l0_neighbors=np.zeros(0, np.uint64))
    assert empty.search(q[0], hv.SearchParams(3)) == []


def test_restricted_wide_set_matches_flat_oracle(orc, hv):
    """Candidate sets above the reference's 256-id exact threshold: the device scans them exactly;
    result must equal the oracle's exact scan over the same allowed ids (recall 1.0 >= the
    reference's 0.95 gate, tests/production_support/vector/restricted.rs:1226-1285)."""
    n, dim = 4000, 64
    rng = np.random.default_rng(9)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    lv = np.zeros(n, np.uint16)
    oix = orc.Index(dim, orc.L2SQ)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    assert oix.seed(ids, data, np.zeros(n + 1, np.uint64), np.zeros(0, np.uint64), entry_point=1) == orc.OK
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=ids, vectors=data,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64),
                                              entry_point=1)
    allowed = np.array([i for i in range(1, n + 1) if i % 3 != 0] + [10, 10, 99999], np.uint64)
    q = rng.standard_normal((6, dim)).astype(np.float32)
    gid, gsc, gcnt = gix.search_restricted_batch(q, hv.SearchParams(10).with_ef(64), hv.RestrictedVectorCandidates.from_ids(allowed))
    for qi in range(6):
        rc, oid, osc = oix.flat(q[qi], 10, allowed=allowed)
        assert gid[qi, :gcnt[qi]].tolist() == oid.tolist()
        assert bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
    # per-query candidate lists
    offs = np.array([0, 3, 3, 10], np.uint64)
    al = np.array([5, 9, 2, 100, 101, 102, 103, 104, 105, 106], np.uint64)
    gid, _, gcnt = gix.search_restricted_batch(q[:3], hv.SearchParams(10), al, offsets=offs)
    assert gcnt.tolist() == [3, 0, 7]
    assert sorted(gid[0, :3].tolist()) == [2, 5, 9]
    with pytest.raises(hv.HelixDbError) as e:
        gix.search_restricted_batch(q[:1], hv.SearchParams(900).with_ef(900), np.arange(1, 2001, dtype=np.uint64))
    assert e.value.status == hv.ERR_K_RANGE


def test_traverse_and_expand_match_reference_semantics(orc, hv):
    """crates/graph-algorithms/src/algorithms/traversal.rs:216-318 (visited set + BFS depth are
    order-independent) and interpreter expand (access/expand.rs:16-80)."""
    rng = np.random.default_rng(4)
    n, e = 500, 3000
    src = np.sort(rng.integers(0, n, e)); tgt = rng.integers(0, n, e).astype(np.uint64)
    off = np.zeros(n + 1, np.uint64); np.add.at(off, src + 1, 1); off = np.cumsum(off).astype(np.uint64)
    lab = rng.integers(0, 4, e).astype(np.uint32)
    g = hv.Graph(n, off, tgt, lab)
    for seeds, md, direction, allowed, hub in [([3], 2, 0, [], 0), ([3, 77, 3], 3, 2, [1, 2], 0), ([10, 11], 4, 1, [], 12),
                                               ([0], 0, 2, [], 0), ([5, 6, 7], 50, 0, [0], 0)]:
        words, depth = g.traverse(seeds, md, direction, allowed, hub, include_seeds=True)
        ref = orc.breadth_first_depths(n, off.astype(np.int64), tgt, lab, seeds, md, direction, allowed, hub)
        got = set(np.nonzero(np.unpackbits(words.view(np.uint8), bitorder="little"))[0].tolist())
        assert got == set(ref.keys())
        for v, d in ref.items():
            assert depth[v] == d
        words2, _ = g.traverse(seeds, md, direction, allowed, hub, include_seeds=False)
        got2 = set(np.nonzero(np.unpackbits(words2.view(np.uint8), bitorder="little"))[0].tolist())
        assert got2 == set(ref.keys()) - set(seeds)
    rows = [3, 4, 5]
    w = g.expand(rows, hv.DIR_OUT)
    exp = set()
    for u in rows:
        exp |= set(int(x) for x in tgt[int(off[u]):int(off[u + 1])])
    assert set(np.nonzero(np.unpackbits(w.view(np.uint8), bitorder="little"))[0].tolist()) == exp
    with pytest.raises(hv.HelixDbError):
        g.traverse([n + 5], 1)


def test_ordered_traversal_matches_the_reference_visit_order_and_discovery_edges(orc, hv):
    """hvx_traverse_ordered vs the oracle's ordered restatement of breadth_first (traversal.rs:216-261, model.rs:635-725):
    the visits in discovery order with their depths, and for every non-seed visit the arc that reached it (which node,
    which stored edge, along or against it) -- multigraph rows (parallel edges with different labels), self-loops,
    every direction, label allow-sets, the hub policy, several seeds in a given order, depth caps incl. unbounded."""
    names = ["a", "b", "c", "hub", "leaf", "leaf2", "leaf3"]             # the reference's own fixture first (traversal.rs:576-615)
    e = [(0, 1), (1, 2), (1, 3), (3, 4), (3, 5), (3, 6)]
    off = np.zeros(8, np.uint64); tgt = np.array([t for _, t in e], np.uint64)
    for s_, _ in e:
        off[s_ + 1:] += 1
    g = hv.Graph(7, off, tgt, None)
    visits, edges = g.traverse_ordered([0], 2)
    assert [(names[v], d) for v, d in visits] == [("a", 0), ("b", 1), ("c", 2), ("hub", 2)] and len(edges) == 3
    assert (visits, edges) == orc.breadth_first(7, off.astype(np.int64), tgt, None, [0], 2)
    rng = np.random.default_rng(12)
    for n, deg, nlab in [(300, 4, 3), (2000, 6, 4), (5000, 2, 1), (64, 40, 2)]:
        rows = [np.sort(rng.integers(0, n, rng.integers(0, 2 * deg + 1))) for _ in range(n)]   # sorted rows, parallel edges, self-loops
        off = np.zeros(n + 1, np.uint64); off[1:] = np.cumsum([len(r) for r in rows])
        tgt = np.concatenate(rows).astype(np.uint64) if off[-1] else np.zeros(0, np.uint64)
        lab = rng.integers(0, nlab, int(off[-1])).astype(np.uint32)
        g = hv.Graph(n, off, tgt, lab)
        cases = [([3], 2, 0, [], 0), ([3, 7, 3, 1], 3, 2, [1, 2], 0), ([10, 11], 4, 1, [], 2 * deg + 3), ([0], 0, 2, [], 0),
                 ([5, 6, 7], 60, 0, [0], 0), ([n - 1], 1000, 2, [], 0), ([2, 1], 5, 2, [0], 3 * deg)]
        for seeds, md, direction, allowed, hub in cases:
            visits, edges = g.traverse_ordered(seeds, md, direction, allowed, hub)
            rv, re_ = orc.breadth_first(n, off.astype(np.int64), tgt, lab, seeds, md, direction, allowed, hub)
            assert visits == rv, f"n={n} case {(seeds, md, direction, allowed, hub)}: visit order differs"
            assert edges == re_, f"n={n} case {(seeds, md, direction, allowed, hub)}: discovery edges differ"
            # DepthFirst (traversal.rs:263-309) runs on the host over a mirror of the device CSR
            assert g.traverse_depth_first(seeds, md, direction, allowed, hub) == orc.depth_first(n, off.astype(np.int64), tgt, lab, seeds, md,
                                                                                                  direction, allowed, hub)
    unsorted = hv.Graph(3, np.array([0, 2, 2, 2], np.uint64), np.array([2, 1], np.uint64), None)
    with pytest.raises(hv.HelixDbError) as ex:                            # rows that are not in the reference's order fail loudly
        unsorted.traverse_ordered([0], 1)
    assert ex.value.status == hv.ERR_UNSUPPORTED
    with pytest.raises(hv.HelixDbError):
        g.traverse_ordered([10**6], 1)


@pytest.mark.parametrize("g,b,k", [(5, 33, 10), (8, 64, 10), (8, 17, 100), (2, 1024, 10)])
def test_merge_topk_device_matches_candidate_order(orc, hv, g, b, k):
    """hvx_merge_topk_device (the N>1 merge after the all-gather) vs the Candidate-order checker,
    including equal scores on different shards, short lists and empty shards; world sizes up to the node's 8 shards (round 6:
    payloads and merges of world = 8 had only run with g = 5), k = 10 and 100."""
    import torch
    rng = np.random.default_rng(3 + g + k)
    gix = hv.ValidatedVectorReadIndex.managed(dim=4, metric=hv.EUCLIDEAN, node_ids=np.arange(4, dtype=np.uint64),
                                              vectors=np.zeros((4, 4), np.float32), l0_offsets=np.zeros(5, np.uint64),
                                              l0_neighbors=np.zeros(0, np.uint64))
    ids = np.zeros((g, b, k), np.uint64); sc = np.zeros((g, b, k), np.float32); cnt = np.zeros((g, b), np.int32)
    for s_ in range(g):
        for q in range(b):
            c = int(rng.integers(0, k + 1)) if (s_ + q) % 7 else 0
            vals = np.sort(rng.integers(0, 6, c).astype(np.float32) * np.float32(0.25))  # many ties
            idv = rng.choice(4000, c, replace=False).astype(np.uint64) * g + s_            # unique across shards
            order = np.lexsort((idv, vals))
            sc[s_, q, :c] = vals[order]; ids[s_, q, :c] = idv[order]; cnt[s_, q] = c
    dev = torch.device("cuda")
    t = lambda a, dt: torch.from_numpy(a.view(dt) if a.dtype == np.uint64 else a).to(dev)
    o_ids = torch.zeros(b, k, dtype=torch.int64, device=dev); o_sc = torch.zeros(b, k, dtype=torch.float32, device=dev)
    o_cnt = torch.zeros(b, dtype=torch.int32, device=dev)
    gix.merge_topk_device(g, b, k, t(ids, np.int64), t(sc, None), t(cnt, None), o_ids, o_sc, o_cnt)
    gix.sync()
    r_ids, r_sc, r_cnt = fx.merge_topk_reference(ids, sc, cnt, k)
    got_i, got_s, got_c = o_ids.cpu().numpy().view(np.uint64), o_sc.cpu().numpy(), o_cnt.cpu().numpy()
    assert got_c.tolist() == r_cnt.tolist()
    for q in range(b):
        assert got_i[q, :r_cnt[q]].tolist() == r_ids[q, :r_cnt[q]].tolist()
        assert bits(got_s[q, :r_cnt[q]]).tolist() == bits(r_sc[q, :r_cnt[q]]).tolist()
    # the same lists as ONE packed exchange buffer (what pyhvx/shard.py all-gathers): [g][ids | scores | counts | pad]
    from pyhvx import shard
    payload = shard.payload_bytes(b, k)
    assert payload == hv.lib().hvx_topk_payload_bytes(b, k) and payload % 8 == 0
    packed = np.zeros((g, payload), np.uint8)
    for s_ in range(g):
        packed[s_, : b * k * 8] = ids[s_].view(np.uint8).reshape(-1)
        packed[s_, b * k * 8: b * k * 12] = sc[s_].view(np.uint8).reshape(-1)
        packed[s_, b * k * 12: b * k * 12 + b * 4] = cnt[s_].view(np.uint8).reshape(-1)
    o_ids.zero_(); o_sc.zero_(); o_cnt.zero_()
    gix.merge_topk_packed_device(g, b, k, torch.from_numpy(packed).to(dev), o_ids, o_sc, o_cnt)
    gix.sync()
    got_i, got_s, got_c = o_ids.cpu().numpy().view(np.uint64), o_sc.cpu().numpy(), o_cnt.cpu().numpy()
    assert got_c.tolist() == r_cnt.tolist()
    for q in range(b):
        assert got_i[q, :r_cnt[q]].tolist() == r_ids[q, :r_cnt[q]].tolist()
        assert bits(got_s[q, :r_cnt[q]]).tolist() == bits(r_sc[q, :r_cnt[q]]).tolist()


@pytest.mark.parametrize("n,dim,metric,ef,k", [(1500, 128, 1, 128, 10), (1200, 768, 1, 128, 10), (1200, 256, 0, 100, 10),
                                               (2000, 128, 1, 300, 20)])
@pytest.mark.parametrize("occupancy", [1, 2])
def test_hnsw_bf16_rows_bit_exact_vs_oracle_on_rounded_vectors(orc, hv, n, dim, metric, ef, k, occupancy):
    """BASELINE config #4 storage: rows rounded to bf16 once at import; distances in f32 on the rounded
    values in the reference's summation order == the oracle run on the rounded vectors, bit for bit."""
    rng = np.random.default_rng(500 + dim)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    rounded = fx.round_bf16(data)
    assert not np.array_equal(rounded, data)
    lv = fx.draw_levels(n, 16, seed=dim + 5)
    oix = build_oracle(orc, rounded, metric, lv, efc=80)
    ex = oix.export()
    ex["vectors"] = data  # the device does the rounding
    gix = hv.ValidatedVectorReadIndex.from_export(ex, dim=dim, metric=metric, dtype=hv.BF16)
    gix.set_occupancy(occupancy)
    q = rng.standard_normal((24, dim)).astype(np.float32)  # queries stay f32
    assert_hnsw_equal(orc, hv, oix, gix, q, k, ef)
    with pytest.raises(hv.HelixDbError) as e:  # shapes the bf16 kernel does not serve fail loudly
        hv.ValidatedVectorReadIndex.managed(dim=40, metric=metric, node_ids=np.arange(4, dtype=np.uint64),
                                            vectors=np.ones((4, 40), np.float32), l0_offsets=np.zeros(5, np.uint64),
                                            l0_neighbors=np.zeros(0, np.uint64), dtype=hv.BF16)
    assert e.value.status == hv.ERR_UNSUPPORTED


@pytest.mark.parametrize("n,dim,metric,k,b", [(5000, 128, 1, 10, 9), (3000, 256, 0, 7, 5), (20000, 768, 1, 100, 130),
                                              (50, 128, 1, 10, 3), (4000, 512, 0, 10, 17)])
def test_bf16_exact_scan_on_matrix_cores_is_bit_exact(orc, hv, n, dim, metric, k, b):
    """Exact scan over bf16 rows = MFMA candidate generation + reference-order re-rank + certificate
    (hvx_flat_mfma.hip): ids and score bits equal the oracle's exact scan over the rounded vectors."""
    rng = np.random.default_rng(900 + dim + n)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    data[7] = data[3]  # exact duplicate row: tie broken by id
    rounded = fx.round_bf16(data)
    ids = np.arange(n, dtype=np.uint64) + 11
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=metric, node_ids=ids, vectors=data, dtype=hv.BF16,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64),
                                              max_batch=max(b, 16))
    q = rng.standard_normal((b, dim)).astype(np.float32)
    q[0] = data[3]
    gid, gsc, gcnt, st = gix.flat_search_batch(q, k)
    for qi in range(b):
        rc, oid, osc = orc.flat_matrix(metric, rounded, q[qi], k)
        assert gcnt[qi] == oid.size == min(k, n)
        assert (gid[qi, :gcnt[qi]] - 11).tolist() == oid.tolist(), f"query {qi}"
        assert bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
    assert gid[0, :2].tolist() == [14, 18]


def test_bf16_exact_scan_never_guesses_on_dense_near_ties(orc, hv):
    """2000 rows within the error bound of each other: the certificate cannot be issued with 64
    candidates; the widened pass (1023) or a loud error are the only acceptable outcomes."""
    n, dim = 3000, 128
    rng = np.random.default_rng(5)
    base = rng.standard_normal(dim).astype(np.float32)
    data = np.tile(base, (n, 1)) + rng.standard_normal((n, dim)).astype(np.float32) * np.float32(1e-3)
    data[2000:] = rng.standard_normal((n - 2000, dim)).astype(np.float32)
    rounded = fx.round_bf16(data)
    ids = np.arange(n, dtype=np.uint64)
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=ids, vectors=data, dtype=hv.BF16,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64))
    q = (base + np.float32(0.01)).reshape(1, dim).astype(np.float32)
    try:
        gid, gsc, gcnt, _ = gix.flat_search_batch(q, 10)
    except hv.HelixDbError as e:
        assert e.status == hv.ERR_INVARIANT and "certificate" in str(e)
        return
    rc, oid, osc = orc.flat_matrix(orc.L2SQ, rounded, q[0], 10)
    assert gid[0].tolist() == oid.tolist() and bits(gsc[0]).tolist() == bits(osc).tolist()


@pytest.mark.parametrize("n,dim,metric,k,b", [(5000, 128, 1, 10, 9), (3000, 256, 0, 7, 5), (20000, 768, 1, 10, 130),
                                              (4000, 1536, 1, 10, 17)])
def test_fp8_exact_scan_is_bit_exact_on_the_dequantised_rows(orc, hv, n, dim, metric, k, b):
    """BASELINE config #5 storage: e4m3fn codes + one f32 scale per row.  The index IS the dequantised rows
    fl32(scale * decode(code)); the MFMA scan + reference-order re-rank + certificate returns the oracle's exact
    scan over exactly those values (ids and score bits)."""
    rng = np.random.default_rng(1300 + dim + n)
    data = (rng.standard_normal((n, dim)) * rng.uniform(0.2, 3.0, (n, 1))).astype(np.float32)
    data[7] = data[3]
    deq = fx.quantize_fp8_rows(data)
    ids = np.arange(n, dtype=np.uint64) + 11
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=metric, node_ids=ids, vectors=data, dtype=hv.FP8_E4M3,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64),
                                              max_batch=max(b, 16))
    q = rng.standard_normal((b, dim)).astype(np.float32)
    q[0] = data[3]
    gid, gsc, gcnt, _ = gix.flat_search_batch(q, k)
    for qi in range(b):
        rc, oid, osc = orc.flat_matrix(metric, deq, q[qi], k)
        assert gcnt[qi] == oid.size
        assert (gid[qi, :gcnt[qi]] - 11).tolist() == oid.tolist(), f"query {qi}"
        assert bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
    assert gid[0, :2].tolist() == [14, 18]
    # quantisation loss is a property of the dtype, reported separately: recall of the fp8 answer vs the f32 rows
    hits = 0
    for qi in range(b):
        rc, tid, _ = orc.flat_matrix(metric, data, q[qi], k)
        hits += len(set((gid[qi, :gcnt[qi]] - 11).tolist()) & set(tid.tolist()))
    assert hits / (b * k) > 0.5
    with pytest.raises(hv.HelixDbError) as e:  # HNSW over fp8 rows is not built: fails loudly
        gix.search_batch(q[:1], hv.SearchParams(k))
    assert e.value.status == hv.ERR_UNSUPPORTED


def test_simhash_on_device_matches_reference_known_answer_and_oracle(orc, hv):
    """a16: the device SimHasher reproduces the reference KAT (simhash_registry.rs:344-362) and the oracle on
    random vectors (incl. a dimension that is not a multiple of anything), and the order code (simhash.rs:314-329)."""
    assert hv.SimHasher(3, 42).hash([1.0, 2.0, 3.0]) == 0x6D91_A757_8862_6786
    rng = np.random.default_rng(8)
    for dim, seed in [(3, 42), (128, 42), (768, 7), (100, 123456789)]:
        dh, oh = hv.SimHasher(dim, seed), orc.SimHasher(dim, seed)
        v = rng.standard_normal((257, dim)).astype(np.float32)
        v[5] = 0.0  # all dots are 0 -> no bit set
        got = dh.hash_batch(v)
        assert got.tolist() == [oh.hash(v[i]) for i in range(v.shape[0])]
        assert got[5] == 0
    for bits in (0, (1 << 64) - 1, 1 << 63, 1 << 47, 1 << 31, 1 << 15, 0x6D91_A757_8862_6786):
        assert hv.order_code_from_simhash_bits(bits) == orc.order_code(bits)


def test_hydration_from_persisted_rows_equals_direct_import(orc, hv):
    """an index hydrated from HelixDB's persisted value bytes (item rows, layer-0 rows in both tagged
    formats, upper rows in historical distance order, a dangling neighbour) searches exactly like the oracle."""
    n, dim = 600, 128
    rng = np.random.default_rng(21)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    lv = fx.draw_levels(n, 16, seed=9)
    ids = np.arange(n, dtype=np.uint64) * 5 + 3
    oix = build_oracle(orc, data, orc.L2SQ, lv, efc=60, ids=ids)
    ex = oix.export()
    hy = hv.Hydrator(dim, hv.EUCLIDEAN)
    be = lambda xs: b"".join(int(x).to_bytes(8, "big") for x in xs)
    up_row = 0
    for i in range(n):
        nid = int(ex["node_ids"][i])
        hy.add_item(nid, np.float32(0).tobytes() + ex["vectors"][i].tobytes())     # Euclidean header = bias 0.0
        nb = ex["l0_neighbors"][int(ex["l0_offsets"][i]):int(ex["l0_offsets"][i + 1])].tolist()
        if i % 3 == 0:
            row = bytes([0x13, 0x01]) + len(nb).to_bytes(4, "big") + (0x1234 + i).to_bytes(8, "little") + be(nb)
        elif i % 3 == 1:
            row = bytes([0x12]) + (len(nb) + 1).to_bytes(4, "big") + be(nb + [999_999_999])  # dangling id: no vector row
        else:
            row = bytes([0x12]) + len(nb).to_bytes(4, "big") + be(nb)
        hy.add_layer0_row(nid, row)
        for layer in range(1, int(ex["level"][i]) + 1):
            unb = ex["up_neighbors"][int(ex["up_offsets"][up_row]):int(ex["up_offsets"][up_row + 1])].tolist()
            hy.add_upper_row(nid, layer, len(unb).to_bytes(4, "big") + be(unb[::-1]))  # historical (unsorted) order
            up_row += 1
    hy.set_entry(ex["entry_point"], ex["max_layer"])
    gix = hy.finish()
    q = rng.standard_normal((16, dim)).astype(np.float32)
    assert_hnsw_equal(orc, hv, oix, gix, q, 10, 64)
    # round 6: the same rows hydrated WITH spare capacity (hvx_hydrator_finish_reserve) take the host's later writes: 40 more nodes
    # inserted in the reference's order -- the grown image searches like the oracle that inserted the same nodes
    more = rng.standard_normal((40, dim)).astype(np.float32)
    mlv = fx.draw_levels(40, 16, seed=10)
    mids = np.arange(40, dtype=np.uint64) + (int(ids[-1]) + 7)
    gix2 = hy.finish(reserve_rows=40, reserve_upper_rows=int(mlv.sum()))
    for i in range(40):
        assert oix.insert(int(mids[i]), more[i], int(mlv[i])) == orc.OK
    gix2.insert_batch(mids, more, mlv, ef_construction=60, sequential=True)
    assert gix2.live_rows() == n + 40
    assert_hnsw_equal(orc, hv, oix, gix2, np.vstack([q, more[:4]]), 10, 64)


# ---------------------------------------------------------------------------------------------------------------
# Non-strict layer-0 arms: SimHash filter, pre/post sampling, adaptive bypass, query RNG.
# Oracle: oracle/hvx_oracle_adaptive.inc (pinned by the reference's policy / search-mode tests, tests/test_oracle_adaptive.py).
# ---------------------------------------------------------------------------------------------------------------
def _oracle_params(orc, p, cfg):
    none = lambda v: -1.0 if v is None else float(v)
    return orc.SearchParams.new(
        p.k, ef=p.ef, simhash_mode=p.simhash_mode,
        pre_simhash_sampling_ratio_override=none(p.pre_simhash_sampling_ratio_override),
        bypass_min_frontier=p.simhash_bypass_min_frontier, bypass_window_expansions=p.simhash_bypass_window_expansions,
        bypass_min_filter_rate=p.simhash_bypass_min_filter_rate, read_budget_multiplier=p.simhash_read_budget_multiplier,
        simhash_sampling_ratio_override=none(p.simhash_sampling_ratio_override),
        simhash_failure_prob_override=none(p.simhash_failure_prob_override),
        simhash_threshold=cfg.simhash_threshold, sampling_ratio=cfg.sampling_ratio,
        adaptive_enabled=cfg.adaptive_enabled, adaptive_failure_prob=cfg.adaptive_failure_prob,
        resident_simhash=int(cfg.resident_snapshot))  # resident snapshot (memory_store.rs:329-335) or uncached handle


ADAPTIVE_STAT_KEYS = ("expansion_steps", "neighbors_examined", "vectors_loaded", "distance_computations",
                      "simhash_filtered", "simhash_examined", "simhash_passed_before_sampling",
                      "simhash_passed_after_sampling", "pre_simhash_sample_kept", "pre_simhash_sample_dropped",
                      "simhash_bypass_expansions", "simhash_skipped_candidates", "simhash_bypass_trigger_budget",
                      "simhash_bypass_trigger_low_yield", "active_simhash_threshold_sum",
                      "active_simhash_threshold_samples", "effective_beam_len_sum", "effective_beam_len_samples",
                      "active_sampling_ratio_samples", "rng_words", "txn_get_simhash_filter", "active_sampling_ratio_sum")


def assert_params_equal(orc, hv, oix, gix, queries, p, cfg):
    ids, sc, cnt, st, per_query, stats = gix.search_batch_with_stats(queries, p)
    op = _oracle_params(orc, p, cfg)
    agg = {}
    for qi in range(queries.shape[0]):
        rc, oid, osc, ost = oix.search_params(queries[qi], op, with_stats=True)
        assert rc == orc.OK and st[qi] == 0
        assert ids[qi, :cnt[qi]].tolist() == oid.tolist(), f"query {qi}: ids differ"
        assert bits(sc[qi, :cnt[qi]]).tolist() == bits(osc).tolist(), f"query {qi}: score bits differ"
        for key in ADAPTIVE_STAT_KEYS:
            assert per_query[qi][key] == ost[key], f"query {qi}: {key} device {per_query[qi][key]} vs oracle {ost[key]}"
            agg[key] = agg.get(key, 0) + ost[key]
    assert stats["tie_overflow_queries"] == 0
    return agg


def _default(hv):
    return hv.SearchParams.new(10)


def _always_post_sampling(hv):   # fixed threshold, explicit pre 1.0 => the post-filter Bernoulli stage draws
    return hv.SearchParams.new(10).with_ef(64).with_simhash_mode(hv.SIMHASH_ALWAYS).with_pre_simhash_sampling_ratio(1.0) \
        .with_simhash_sampling_ratio(0.5)


def _adaptive_post_sampling(hv):  # candidate_probability weighting by similarity bits
    return hv.SearchParams.new(10).with_ef(48).with_pre_simhash_sampling_ratio(1.0).with_simhash_sampling_ratio(0.35)


def _throughput(hv):
    return hv.SearchParams.throughput_profile_floor_92(10)


def _bypass_happy(hv):            # index.rs:2515-2530: every filtering epoch re-triggers the low-yield bypass
    return hv.SearchParams.new(5).with_ef(16).with_pre_simhash_sampling_ratio(0.25).with_simhash_sampling_ratio(0.5) \
        .with_simhash_failure_prob(0.5).with_simhash_bypass_tuning(1, 1, 1.0, 1)


def _defer_all(hv):               # ratio 0 defers every neighbour and disables the fallback: the entry is the result
    return hv.SearchParams.new(10).with_ef(32).with_simhash_mode(hv.SIMHASH_ALWAYS).with_pre_simhash_sampling_ratio(1.0) \
        .with_simhash_sampling_ratio(0.0)


def _fallback(hv):                # ratio 0.03: most wide frontiers sample nobody and take the choose_index fallback
    return hv.SearchParams.new(10).with_ef(32).with_simhash_mode(hv.SIMHASH_ALWAYS).with_pre_simhash_sampling_ratio(1.0) \
        .with_simhash_sampling_ratio(0.03)


def _pre_zero(hv):                # pre-sampling ratio 0: the pre-stage fallback picks one neighbour per wide frontier
    return hv.SearchParams.new(10).with_ef(24).with_pre_simhash_sampling_ratio(0.0)


def _wide_beam(hv):
    return hv.SearchParams.new(50).with_ef(200)


def _ef400(hv):
    return hv.SearchParams.new(100).with_ef(400)


def _ef800(hv):
    return hv.SearchParams.new(400).with_ef(800)


ADAPTIVE_CASES = [
    # (name, metric, n, dim, m, m0, params, config overrides)
    ("default-cos", 0, 3000, 128, 16, 32, _default, {}),
    ("default-l2", 1, 3000, 128, 16, 32, _default, {}),
    ("default-cos-768", 0, 1500, 768, 16, 32, _default, {}),
    ("always-post", 0, 2500, 256, 16, 32, _always_post_sampling, {"simhash_threshold": 30}),
    ("adaptive-post", 0, 2500, 128, 32, 64, _adaptive_post_sampling, {"sampling_ratio": 0.35}),
    ("throughput-l2", 1, 2500, 128, 32, 64, _throughput, {}),
    ("throughput-cos", 0, 2500, 128, 32, 64, _throughput, {}),
    ("bypass", 0, 2000, 128, 16, 32, _bypass_happy, {"simhash_threshold": 20, "sampling_ratio": 0.5}),
    ("defer-all", 0, 2000, 128, 16, 32, _defer_all, {"simhash_threshold": 28}),
    ("fallback", 0, 2000, 128, 16, 32, _fallback, {"simhash_threshold": 28}),
    ("fallback-l2", 1, 2000, 128, 32, 64, _fallback, {}),
    ("pre-zero", 1, 2000, 128, 32, 64, _pre_zero, {}),
    ("fixed-not-adaptive", 0, 2000, 128, 16, 32, _default, {"adaptive_enabled": 0, "simhash_threshold": 34}),
    ("wide-beam", 0, 3000, 128, 16, 32, _wide_beam, {}),
    ("threshold-0", 0, 2000, 128, 16, 32, _default, {"simhash_threshold": 0}),
    # shapes served by the GENERIC build of the non-strict arms (any dimension / metric / summation tree, ef <= 800)
    ("gen-dim100-l2", 1, 2500, 100, 16, 32, _default, {}),            # 3 chunks + a 4-element scalar tail
    ("gen-dim100-cos", 0, 2500, 100, 16, 32, _default, {}),
    ("dim384-cos", 0, 2000, 384, 16, 32, _default, {}),               # NK = 12: unrolled since round 6 (the GENERIC build served it before)
    ("dim384-l2-throughput", 1, 2000, 384, 16, 32, _throughput, {}),
    ("gen-dim20-cos", 0, 2000, 20, 16, 32, _default, {}),             # dim < 32: the scalar kernel
    ("gen-manhattan", 2, 2500, 96, 16, 32, _default, {}),             # sampling stays live for Manhattan (policy.rs:89-101)
    ("gen-manhattan-throughput", 2, 2500, 40, 32, 64, _throughput, {}),
    ("gen-ef400-cos", 0, 4000, 128, 16, 32, _ef400, {}),              # R = 7 beam on a shape the unrolled builds stop at 352
    ("gen-ef800-l2", 1, 4000, 64, 16, 32, _ef800, {}),                # R = 13 beam, the restricted path's k limit
    ("gen-always-post-dim100", 0, 2500, 100, 16, 32, _always_post_sampling, {"simhash_threshold": 30}),
]


@pytest.mark.parametrize("occupancy", [1, 2])
@pytest.mark.parametrize("name,metric,n,dim,m,m0,mk,over", ADAPTIVE_CASES, ids=[c[0] for c in ADAPTIVE_CASES])
def test_non_strict_arms_match_oracle(orc, hv, name, metric, n, dim, m, m0, mk, over, occupancy):
    """Production-default (`SearchParams::new(k)`) and every other non-strict configuration: ids, score bits, all
    SearchStats counters of the filter / sampling / bypass stages and the number of RNG words drawn equal the oracle's --
    on the one-query-per-SIMD builds and (round 5) on the two-per-SIMD builds of the unrolled f32 shapes (256-word RNG window,
    visited table sized to the 20 KiB LDS share; the generic shapes keep their one-per-SIMD build under either setting)."""
    if occupancy == 2 and name.startswith("gen-"):
        pytest.skip("generic shapes have no two-per-SIMD build: the handle setting does not change their launch")
    rng = np.random.default_rng(1000 + n + dim + metric)
    centers = rng.standard_normal((24, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 24, n)] + 0.7 * rng.standard_normal((n, dim))).astype(np.float32)
    lv = fx.draw_levels(n, m, seed=dim + 3)
    oix = build_oracle(orc, data, metric, lv, m=m, m0=m0, efc=80)
    cfg = hv.SimHashConfig.default(**over)
    oix.set_simhash(int(cfg.seed))
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=metric, m=m, m0=m0)
    gix.set_simhash(cfg)
    gix.set_occupancy(occupancy)
    assert gix.get_simhash().tolist() == oix.get_simhash().tolist()      # device SimHasher == oracle, every row
    q = (centers[rng.integers(0, 24, 40)] + 0.7 * rng.standard_normal((40, dim))).astype(np.float32)
    p = mk(hv)
    agg = assert_params_equal(orc, hv, oix, gix, q, p, cfg)
    assert_params_equal(orc, hv, oix, gix, q[:7], p, cfg)               # second launch: clean visited state
    # the case exercises what its name says
    if name in ("default-cos", "default-cos-768", "throughput-cos", "wide-beam", "fixed-not-adaptive"):
        assert agg["simhash_examined"] > 0 and agg["simhash_filtered"] > 0
    if metric != 0:  # the SimHash filter is cosine-only (policy.rs:67-91); sampling stays live
        assert agg["simhash_examined"] == 0
    if name in ("always-post", "adaptive-post"):
        assert agg["pre_simhash_sample_kept"] == 0 and agg["rng_words"] > 0
        assert agg["simhash_passed_after_sampling"] < agg["simhash_passed_before_sampling"]
    if name.startswith("throughput") or name.startswith("default"):
        assert agg["pre_simhash_sample_dropped"] > 0
    if name == "bypass":
        assert agg["simhash_bypass_expansions"] > 0 and agg["simhash_bypass_trigger_low_yield"] > 0
    if name == "defer-all":
        assert agg["simhash_passed_after_sampling"] == 0 and agg["rng_words"] == 0 and agg["distance_computations"] == 40
    if name.startswith("fallback"):
        assert agg["simhash_passed_after_sampling"] > 0 and agg["rng_words"] > agg["simhash_passed_before_sampling"] // 2
    if name == "pre-zero":
        assert agg["pre_simhash_sample_dropped"] > 0 and agg["pre_simhash_sample_kept"] > 0


@pytest.mark.parametrize("occupancy", [1, 2])
@pytest.mark.parametrize("metric,dim", [(0, 256), (1, 128), (0, 768), (1, 1024), (1, 384)])
def test_non_strict_arms_over_bf16_rows(orc, hv, metric, dim, occupancy):
    """Config #4 storage under the production-default params: the oracle runs on the rounded vectors (its SimHash rows
    are those of the rounded vectors too, as the device computes them from the stored bf16 values)."""
    rng = np.random.default_rng(77 + dim)
    n = 2500
    data = rng.standard_normal((n, dim)).astype(np.float32)
    rounded = fx.round_bf16(data)
    lv = fx.draw_levels(n, 16, seed=dim + 9)
    oix = build_oracle(orc, rounded, metric, lv, efc=80, cached=True)  # (read-only here; set_simhash(42) is idempotent)
    oix.set_simhash(42)
    ex = oix.export()
    ex["vectors"] = data  # the device does the rounding
    gix = hv.ValidatedVectorReadIndex.from_export(ex, dim=dim, metric=metric, dtype=hv.BF16)
    cfg = hv.SimHashConfig.default()
    gix.set_simhash(cfg)
    gix.set_occupancy(occupancy)   # round 5: two-per-SIMD builds of the non-strict arms over bf16 rows too
    assert gix.get_simhash().tolist() == oix.get_simhash().tolist()
    q = rng.standard_normal((32, dim)).astype(np.float32)
    agg = assert_params_equal(orc, hv, oix, gix, q, hv.SearchParams.new(10), cfg)
    assert agg["pre_simhash_sample_dropped"] > 0 and (metric == 1 or agg["simhash_filtered"] > 0)
    assert_params_equal(orc, hv, oix, gix, q, hv.SearchParams.throughput_profile_floor_92(10), cfg)


@pytest.mark.parametrize("occupancy", [1, 2])
@pytest.mark.parametrize("spill", [False, True])
@pytest.mark.parametrize("mult,ef", [(1, 100), (3, 100), (1, 24)])
def test_uncached_handle_read_accounting_and_budget_bypass(orc, hv, spill, mult, ef, occupancy):
    """hvx_simhash_config.resident_snapshot = 0: every SimHash row a query sees for the first time in a filtering epoch is
    one stable-view read (memory_store.rs:338-347); once ef x multiplier reads are spent the read-budget trigger
    (policy.rs:266) opens bypass windows.  Reads, triggers, results equal the oracle's uncached accounting -- also when
    the visited table has spilled to the HBM bitmaps."""
    rng = np.random.default_rng(606 + mult + ef)
    n, dim = 2500, 128
    data = rng.standard_normal((n, dim)).astype(np.float32)
    oix = build_oracle(orc, data, 0, fx.draw_levels(n, 16, seed=12), efc=80)
    oix.set_simhash(42)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=0)
    gix.set_occupancy(occupancy)
    if spill:
        gix.set_option(hv.OPT_WAVE_LOG2CAP, 8)
    cfg = hv.SimHashConfig.default(resident_snapshot=0, simhash_threshold=30)
    gix.set_simhash(cfg)
    q = rng.standard_normal((32, dim)).astype(np.float32)
    p = hv.SearchParams.new(10).with_ef(ef).with_simhash_bypass_tuning(8, 3, 0.05, mult)
    agg = assert_params_equal(orc, hv, oix, gix, q, p, cfg)
    assert agg["txn_get_simhash_filter"] > 0
    if mult == 1:
        assert agg["simhash_bypass_trigger_budget"] > 0 and agg["simhash_bypass_expansions"] > 0
    # the same index with resident accounting answers differently once the budget has fired
    cfg_r = hv.SimHashConfig.default(resident_snapshot=1, simhash_threshold=30)
    gix.set_simhash(cfg_r)
    agg_r = assert_params_equal(orc, hv, oix, gix, q, p, cfg_r)
    assert agg_r["txn_get_simhash_filter"] == 0 and agg_r["simhash_bypass_trigger_budget"] == 0


@pytest.mark.parametrize("occupancy", [1, 2])
def test_non_strict_arms_spill_path_and_given_hashes(orc, hv, occupancy):
    """LDS visited table -> HBM bitmap spill inside the non-strict arms (visited TEST and late insert both take the
    bitmap), with the SimHash rows handed over by the host instead of recomputed."""
    rng = np.random.default_rng(4242)
    n, dim = 2500, 128
    data = rng.standard_normal((n, dim)).astype(np.float32)
    lv = fx.draw_levels(n, 16, seed=8)
    oix = build_oracle(orc, data, 0, lv, efc=80)
    oix.set_simhash(42)
    cfg = hv.SimHashConfig.default()
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=0)
    gix.set_simhash(cfg, node_hashes=oix.get_simhash())
    gix.set_occupancy(occupancy)
    gix.set_option(hv.OPT_WAVE_LOG2CAP, 8)
    q = rng.standard_normal((24, dim)).astype(np.float32)
    agg = assert_params_equal(orc, hv, oix, gix, q, hv.SearchParams.new(10), cfg)
    assert agg["distance_computations"] > 24 * 128  # far more visited ids than the 256-slot table holds


@pytest.mark.parametrize("metric,dim,ef,k", [(1, 128, 1000, 10), (0, 96, 2500, 1000), (2, 40, 993, 50)])
def test_beams_beyond_the_kernel_limit_are_answered_by_the_exact_scan(orc, hv, metric, dim, ef, k):
    """SearchBeamWidth::try_new has no upper bound (parameters.rs:118-133).  ef + 32 > 1024 exceeds every beam the HNSW kernels hold:
    the call is answered by the exact scan of the index (VERDICT r4 missing #6) -- the true top-k, in Candidate order, with the
    per-query validation statuses of the search path -- for the strict arm and for SearchParams alike; k beyond the scan's limit
    still fails loudly."""
    rng = np.random.default_rng(31 + dim)
    n = 3000
    data = rng.standard_normal((n, dim)).astype(np.float32)
    oix = build_oracle(orc, data, metric, fx.draw_levels(n, 16, seed=5), efc=64)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=metric)
    q = rng.standard_normal((9, dim)).astype(np.float32)
    q[4, 3] = np.nan
    gix.set_simhash()
    for params in (hv.SearchParams(k).with_ef(ef), hv.SearchParams.new(k).with_ef(ef)):
        ids, sc, cnt, st, status = gix.search_batch(q, params, per_query_status=True)
        assert status[4] == hv.ERR_NONFINITE and cnt[4] == 0
        for qi in (0, 1, 2, 3, 5, 6, 7, 8):
            rc, tid, tsc = oix.flat(q[qi], k)
            assert rc == orc.OK and status[qi] == 0 and cnt[qi] == min(k, n)
            assert ids[qi, :cnt[qi]].tolist() == tid.tolist()
            assert sc[qi, :cnt[qi]].view(np.uint32).tolist() == tsc.view(np.uint32).tolist()
        assert st["distance_computations"] == 8 * n
    with pytest.raises(hv.HelixDbError) as e:
        gix.search_batch(q[:1], hv.SearchParams(1025).with_ef(1025))
    assert e.value.status == hv.ERR_UNSUPPORTED


def test_strict_params_route_to_the_strict_kernel_and_validation(orc, hv):
    rng = np.random.default_rng(99)
    n, dim = 1500, 128
    data = rng.standard_normal((n, dim)).astype(np.float32)
    oix = build_oracle(orc, data, 1, fx.draw_levels(n, 16, seed=1), efc=60)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=1)
    q = rng.standard_normal((8, dim)).astype(np.float32)
    # Off + pre 1.0 through the params entry point == hvx_search_batch (no SimHash rows needed)
    strict = hv.SearchParams(10).with_ef(64)
    ids, sc, cnt, st, per_query, _ = gix.search_batch_with_stats(q, strict)
    ids2, sc2, cnt2, _ = gix.search_batch(q, strict)
    assert ids.tolist() == ids2.tolist() and bits(sc).tolist() == bits(sc2).tolist()
    assert all(d["rng_words"] == 0 and d["effective_beam_len_samples"] == 0 for d in per_query)
    # "layer-zero filtering or sampling requires a query fingerprint" (search.rs:289-295)
    with pytest.raises(hv.HelixDbError) as e:
        gix.search_batch(q, hv.SearchParams.new(10))
    assert e.value.status == hv.ERR_INVARIANT
    gix.set_simhash()
    bad = hv.SearchParams.new(10)
    bad.simhash_bypass_window_expansions = 0
    with pytest.raises(hv.HelixDbError) as e:
        gix.search_batch(q, bad)
    assert e.value.status == hv.ERR_K_RANGE
    # invalid queries keep their per-query status in the non-strict arms too
    q2 = q.copy()
    q2[3, 5] = np.nan
    ids, sc, cnt, stats, st = gix.search_batch(q2, hv.SearchParams.new(10), per_query_status=True)
    assert st.tolist() == [0, 0, 0, hv.ERR_NONFINITE, 0, 0, 0, 0] and cnt[3] == 0


@pytest.mark.parametrize("contiguous", [True, False])
def test_fused_prefilter_search_equals_the_two_call_form_and_the_oracle(orc, hv, contiguous):
    """hvx_prefilter_search_batch (hop -> device bitmap -> device row list -> exact restricted scan) against
    hvx_expand_filter / hvx_traverse_filter + hvx_search_restricted_batch and the oracle's exact scan over the allowed
    ids (restricted.rs:753-835); vectors exist only for part of the graph's nodes (ids without a vector are omitted)."""
    rng = np.random.default_rng(31)
    n_nodes, e, dim = 4000, 20000, 64
    src = np.sort(rng.integers(0, n_nodes, e)); tgt = rng.integers(0, n_nodes, e).astype(np.uint64)
    off = np.zeros(n_nodes + 1, np.uint64); np.add.at(off, src + 1, 1); off = np.cumsum(off).astype(np.uint64)
    lab = rng.integers(0, 3, e).astype(np.uint32)
    g = hv.Graph(n_nodes, off, tgt, lab)
    # vectors on a subset of the nodes: a contiguous id range, or every third id (binary-search id map on the device)
    vec_ids = np.arange(500, 3500, dtype=np.uint64) if contiguous else np.arange(0, n_nodes, 3, dtype=np.uint64)
    data = rng.standard_normal((vec_ids.size, dim)).astype(np.float32)
    oix = orc.Index(dim, orc.L2SQ)
    assert oix.seed(vec_ids, data, np.zeros(vec_ids.size + 1, np.uint64), np.zeros(0, np.uint64)) == orc.OK
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=vec_ids, vectors=data,
                                              l0_offsets=np.zeros(vec_ids.size + 1, np.uint64),
                                              l0_neighbors=np.zeros(0, np.uint64), max_batch=16)
    q = rng.standard_normal((9, dim)).astype(np.float32)
    p = hv.SearchParams(10)
    for seeds, kw in [(np.arange(40, 90), dict()),                                           # one-hop expand
                      (np.arange(0, 400), dict(allowed_labels=[1], direction=hv.DIR_BOTH)),
                      ([7, 8, 9], dict(traverse=True, max_depth=3, direction=hv.DIR_OUT)),   # BFS, seeds included
                      ([7, 8, 9], dict(traverse=True, max_depth=2, direction=hv.DIR_BOTH, include_seeds=False, hub_degree=14))]:
        ids, sc, cnt, ncand, _ = gix.prefilter_search_batch(g, q, p, seeds, **kw)
        if kw.get("traverse"):
            words, _ = g.traverse(seeds, kw["max_depth"], kw["direction"], kw.get("allowed_labels", ()), kw.get("hub_degree", 0),
                                  include_seeds=kw.get("include_seeds", True))
        else:
            words = g.expand(seeds, kw.get("direction", hv.DIR_OUT), kw.get("allowed_labels", ()))
        cand = hv.RestrictedVectorCandidates.from_bitmap_words(words)
        assert ncand == len(cand) and ncand > 10
        ids2, sc2, cnt2 = gix.search_restricted_batch(q, p, cand)
        assert cnt.tolist() == cnt2.tolist() and ids.tolist() == ids2.tolist() and bits(sc).tolist() == bits(sc2).tolist()
        for qi in range(q.shape[0]):
            rc, oid, osc = oix.flat(q[qi], 10, allowed=cand.ids)
            assert rc == orc.OK and ids[qi, :cnt[qi]].tolist() == oid.tolist() and bits(sc[qi, :cnt[qi]]).tolist() == bits(osc).tolist()
    # an empty stream expands to nothing; a hop that reaches only vector-less nodes returns no rows
    ids, sc, cnt, ncand, _ = gix.prefilter_search_batch(g, q, p, [])
    assert ncand == 0 and cnt.sum() == 0


def test_async_kernel_timing_ring():
    """hvx_index_timing_begin / _collect: un-synchronised device-pointer searches bracketed by a ring of HIP-event pairs.
    Runs in a child process because torch (device buffers) must initialise its HIP runtime before the library does."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
torch.cuda.init()
sys.path[:0] = [%r, %r, %r]
import pyhvx as hv, orc, fixtures as fx
rng = np.random.default_rng(3)
n, dim = 1200, 128
data = rng.standard_normal((n, dim)).astype(np.float32)
oix = orc.Index(dim, orc.L2SQ, ef_construction=60)
lv = fx.draw_levels(n, 16, seed=2)
for i in range(n):
    assert oix.insert(i, data[i], int(lv[i])) == orc.OK
gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=1, max_batch=16)
dev = torch.device("cuda", 0)
q = torch.from_numpy(rng.standard_normal((16, dim)).astype(np.float32)).to(dev)
d_ids = torch.zeros(16, 10, dtype=torch.int64, device=dev); d_sc = torch.zeros(16, 10, dtype=torch.float32, device=dev)
d_cnt = torch.zeros(16, dtype=torch.int32, device=dev); d_st = torch.zeros(16, dtype=torch.int32, device=dev)
gix.set_stream(torch.cuda.current_stream(dev).cuda_stream)
gix.timing_begin(5)
for _ in range(7):  # two more calls than slots: the surplus is simply not timed
    gix.search_batch_device(q, 10, 64, d_ids, d_sc, d_cnt, d_st)
ms = gix.timing_collect(8)
assert ms.size == 5 and (ms > 0).all() and (ms < 50).all(), ms
assert gix.timing_collect(8).size == 0  # disarmed
rc, oid, _ = oix.search(q[3].cpu().numpy(), 10, 64)
assert d_ids[3].cpu().numpy().astype(np.uint64)[: oid.size].tolist() == oid.tolist()
print("TIMING-OK")
''' % tuple(__import__("os").path.join(fx.ROOT, d) for d in ("helix-db_amd", "oracle", "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "TIMING-OK" in out.stdout, out.stdout + out.stderr


def test_batcher_coalesces_concurrent_single_query_callers(orc, hv):
    """hvx_batcher: 48 threads issue single-query searches; every caller gets exactly the rows a direct
    batch call returns (strict and production-default params), launches are shared, a rejected query fails alone."""
    import threading
    rng = np.random.default_rng(17)
    n, dim = 2000, 128
    data = rng.standard_normal((n, dim)).astype(np.float32)
    oix = build_oracle(orc, data, 0, fx.draw_levels(n, 16, seed=4), efc=60)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=0, max_batch=64)
    gix.set_simhash()
    q = rng.standard_normal((480, dim)).astype(np.float32)
    for params in (hv.SearchParams(10).with_ef(64), hv.SearchParams.new(10)):
        want_ids, want_sc, want_cnt, _ = gix.search_batch(q, params)
        bt = hv.Batcher(gix, params, max_batch=32, max_wait_us=2000)
        got = [None] * q.shape[0]
        errors = []
        start = threading.Barrier(48)  # every caller's first query arrives at once: the first batches are shared for certain

        def worker(t):
            try:
                start.wait()
                for i in range(t, q.shape[0], 48):
                    got[i] = bt.search(q[i])
            except Exception as e:  # pragma: no cover
                errors.append(e)

        th = [threading.Thread(target=worker, args=(t,)) for t in range(48)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errors, errors
        for i in range(q.shape[0]):
            assert [r.entity_id for r in got[i]] == want_ids[i, :want_cnt[i]].tolist()
            assert bits([r.score for r in got[i]]).tolist() == bits(want_sc[i, :want_cnt[i]]).tolist()
        st = bt.stats()
        # a batch closes as soon as a device lane is free (no timer), so its size follows the load: 48 Python threads behind the GIL
        # are a trickle (1.5-3 queries per launch measured); the C++ harness (scripts/bench_batcher.cpp, 1 024 callers) sees 200-400
        assert st["queries"] == q.shape[0] and st["batches"] < q.shape[0], st  # launches were shared
        bad = q[0].copy()
        bad[5] = np.inf
        with pytest.raises(hv.HelixDbError) as e:
            bt.search(bad)
        assert e.value.status == hv.ERR_NONFINITE
        assert [r.entity_id for r in bt.search(q[1])] == want_ids[1, :want_cnt[1]].tolist()
        bt.close()


def test_batcher_non_blocking_submit_poll_eventfd(orc, hv):
    """Round 5: ValidatedVectorReadIndex::search is an async fn (read_index.rs:81-102).  ONE host thread keeps hundreds
    of queries in flight through hvx_batcher_submit / _poll and sleeps on the batcher's eventfd between completions (select() here;
    tokio's AsyncFd in the reference host): every ticket returns exactly the rows of a direct batch call, a full open batch answers
    HVX_ERR_BUSY instead of blocking, a rejected query fails alone at its poll, hvx_batcher_wait times out with the ticket still valid,
    and freeing the batcher with tickets nobody polls any more does not hang."""
    import os, select, time
    rng = np.random.default_rng(23)
    n, dim = 2000, 128
    data = rng.standard_normal((n, dim)).astype(np.float32)
    oix = build_oracle(orc, data, 1, fx.draw_levels(n, 16, seed=4), efc=60)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=1, max_batch=64)
    gix.set_simhash()
    q = rng.standard_normal((600, dim)).astype(np.float32)
    q[77, 3] = np.nan
    for params in (hv.SearchParams(10).with_ef(64), hv.SearchParams.new(10)):
        want_ids, want_sc, want_cnt, _, want_st = gix.search_batch(q, params, per_query_status=True)
        bt = hv.Batcher(gix, params, max_batch=64, max_wait_us=300)
        fd = bt.eventfd()
        assert fd >= 0 and bt.eventfd() == fd
        pending, got, busy, nxt, window = {}, {}, 0, 0, 200
        t_end = time.time() + 60
        while len(got) < q.shape[0] and time.time() < t_end:
            while nxt < q.shape[0] and len(pending) < window:      # keep `window` tickets in flight
                t = bt.submit(q[nxt])
                if t is None:                                       # the open batch (64 slots) is full: not an error, not a block
                    busy += 1
                    break
                pending[nxt] = t
                nxt += 1
            r, _, _ = select.select([fd], [], [], 0.05)
            if r:
                try:
                    os.read(fd, 8)
                except BlockingIOError:
                    pass
            for i in list(pending):
                try:
                    res = bt.poll(pending[i])
                except hv.HelixDbError as e:
                    res = e
                if res is not None:
                    got[i] = res
                    del pending[i]
        assert len(got) == q.shape[0] and not pending
        # (200 tickets against 64-slot batches: submit says BUSY whenever the open batch is full at that instant -- how often depends
        #  on how fast the lanes close batches, so it is handled above but not required here; the forced case follows)
        for i in range(q.shape[0]):
            if want_st[i]:
                assert isinstance(got[i], hv.HelixDbError) and got[i].status == hv.ERR_NONFINITE and i == 77
                continue
            assert [r_.entity_id for r_ in got[i]] == want_ids[i, :want_cnt[i]].tolist()
            assert bits([r_.score for r_ in got[i]]).tolist() == bits(want_sc[i, :want_cnt[i]]).tolist()
        st = bt.stats()
        assert st["queries"] == q.shape[0] and st["batches"] < q.shape[0] // 4, st      # one thread, yet the launches carry many queries
        # a full open batch answers BUSY, never blocks: a second batcher with ONE lane whose device is kept busy by a long search
        if params.requires_query_simhash():
            slow = hv.Batcher(gix, params, max_batch=4, max_wait_us=200000)
            held = [slow.submit(q[i]) for i in range(4)]             # fills the first batch (launched at once: expectation 1 ... or later)
            more, saw_busy = [], False
            for i in range(4, 64):                                    # keep submitting: with batches of four, some submit meets a full open batch
                t = slow.submit(q[i])
                if t is None:
                    saw_busy = True
                    break
                more.append((i, t))
            for i, t in list(enumerate(held)) + more:
                res = slow.wait(t)
                assert [r_.entity_id for r_ in res] == want_ids[i, :want_cnt[i]].tolist()
            slow.close()
            assert saw_busy or len(more) == 60                        # either outcome is legal; neither blocked
        # wait with a time-out: a ticket whose batch cannot be complete yet stays valid
        t = bt.submit(q[5])
        first = bt.wait(t, timeout_us=1)
        res = first if first is not None else bt.wait(t)
        assert [r_.entity_id for r_ in res] == want_ids[5, :want_cnt[5]].tolist()
        # tickets that are never polled: close() must come back (0.2 s of grace per lane), later submits fail loudly
        for i in range(40):
            assert bt.submit(q[i]) is not None
        t0 = time.time()
        bt.close()
        assert time.time() - t0 < 20


@pytest.mark.parametrize("metric", [0, 1])
def test_duplicate_vectors_ties_are_broken_by_id_in_both_arms(orc, hv, metric):
    """20 exact copies of each of 120 vectors: every score occurs 20 times, so the (score, id) order (model.rs:55-61)
    decides pops, evictions and the k cut everywhere; strict arm and production-default params against the oracle."""
    rng = np.random.default_rng(2024 + metric)
    base = rng.standard_normal((120, 128)).astype(np.float32)
    data = np.repeat(base, 20, axis=0)[rng.permutation(2400)]
    n = data.shape[0]
    oix = build_oracle(orc, data, metric, fx.draw_levels(n, 16, seed=6), efc=80)
    oix.set_simhash(42)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=128, metric=metric)
    cfg = hv.SimHashConfig.default()
    gix.set_simhash(cfg)
    q = np.concatenate([base[:12] + np.float32(0.05) * rng.standard_normal((12, 128)).astype(np.float32), base[12:20]])
    assert_hnsw_equal(orc, hv, oix, gix, q, 25, 64)     # k cuts through a group of equal scores
    assert_hnsw_equal(orc, hv, oix, gix, q, 10, 100)
    assert_params_equal(orc, hv, oix, gix, q, hv.SearchParams.new(10), cfg)
    assert_params_equal(orc, hv, oix, gix, q, hv.SearchParams.new(25).with_ef(48), cfg)


@pytest.mark.parametrize("occupancy", [1, 2])
def test_ties_beyond_the_beam_slack_are_rerun_with_a_wider_beam(orc, hv, occupancy):
    """70 exact copies of each of 40 vectors: a beam of ef = 128 entries (+ 64 of slack in the R = 3 build) evicts equal-score
    candidates it may still have to expand, so the kernel flags those queries and the re-run launch repeats them with the
    R = 6 beam; results, score bits and counters then equal the oracle's and no query stays flagged."""
    rng = np.random.default_rng(99)
    base = rng.standard_normal((40, 128)).astype(np.float32)
    data = np.repeat(base, 70, axis=0)[rng.permutation(2800)]
    n = data.shape[0]
    oix = build_oracle(orc, data, orc.L2SQ, fx.draw_levels(n, 16, seed=2), efc=80)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=128, metric=hv.EUCLIDEAN)
    gix.set_occupancy(occupancy)
    q = np.concatenate([base[:16], base[16:24] + np.float32(0.01) * rng.standard_normal((8, 128)).astype(np.float32)])
    assert_hnsw_equal(orc, hv, oix, gix, q, 10, 128)   # asserts tie_overflow_queries == 0
    assert_hnsw_equal(orc, hv, oix, gix, q, 100, 128)
    oix.set_simhash(42)
    cfg = hv.SimHashConfig.default()
    gix.set_simhash(cfg)
    assert_params_equal(orc, hv, oix, gix, q, hv.SearchParams.new(10).with_ef(128), cfg)


@pytest.mark.parametrize("metric,dim,n,k,b", [(1, 512, 40000, 10, 512), (0, 256, 80000, 25, 512), (1, 768, 60000, 100, 500)])
def test_f32_exact_scan_on_matrix_cores_is_bit_exact(orc, hv, metric, dim, n, k, b):
    """Whole-corpus exact scans over f32 rows (dim >= 256) with enough work (b x n x dim >= 2^33) generate their candidates on the matrix
    cores (rows split into bf16 hi + lo on the fly), re-rank them in the reference's summation order and certify the
    result; it must equal the oracle's exact scan -- and the VALU exact-scan kernel -- bit for bit."""
    rng = np.random.default_rng(dim + n)
    centers = rng.standard_normal((32, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 32, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
    data[7] = data[3]  # exact duplicate row: tie broken by id
    ids = np.arange(n, dtype=np.uint64) + 11
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=metric, node_ids=ids, vectors=data,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64),
                                              max_batch=512)
    q = (centers[rng.integers(0, 32, b)] + 0.5 * rng.standard_normal((b, dim))).astype(np.float32)
    q[0] = data[3]
    gix.flat_search_batch(q, k)                       # first use loads the kernels: not the run that is timed
    gid, gsc, gcnt, stats = gix.flat_search_batch(q, k)
    assert gix.last_scan_path() & (hv.PATH_MFMA_128 | hv.PATH_TILE_256 | hv.PATH_SMALL_BATCH)
    gix.set_option(hv.OPT_FLAT_FORCE_VALU, 1)
    gix.flat_search_batch(q, k)
    vid, vsc, vcnt, vstats = gix.flat_search_batch(q, k)
    assert gix.last_scan_path() == hv.PATH_VALU
    gix.set_option(hv.OPT_FLAT_FORCE_VALU, 0)
    assert gid.tolist() == vid.tolist() and bits(gsc).tolist() == bits(vsc).tolist() and gcnt.tolist() == vcnt.tolist()
    print(f"f32 exact scan {b} x {n} x {dim}: MFMA path {stats['device_ms']:.3f} ms, VALU kernel {vstats['device_ms']:.3f} ms")
    if dim == 768:
        assert stats["device_ms"] < vstats["device_ms"]      # and that is the point of it
    for qi in range(0, b, max(1, b // 24)):
        rc, oid, osc = orc.flat_matrix(metric, data, q[qi], k, kernel=orc.K_AVX_FMA_HW)
        assert rc == orc.OK and (gid[qi, :gcnt[qi]] - 11).tolist() == oid.tolist(), f"query {qi}"
        assert bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()


def test_f32_exact_scan_falls_back_to_the_valu_kernel_on_dense_near_ties(orc, hv):
    """5 000 rows within a few ulps of each other: no certificate can separate the k-th score from the rest, so the
    batch must be answered by the exact VALU scan -- never by a guess, never by an error."""
    rng = np.random.default_rng(8)
    n, dim, b, k = 20000, 512, 1024, 10           # b x n x dim = 2^33.3: takes the matrix-core path first
    base = rng.standard_normal(dim).astype(np.float32)
    data = np.tile(base, (n, 1))
    data[:, 0] += (np.arange(n) % 7).astype(np.float32) * np.float32(1e-6)
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64),
                                              vectors=data, l0_offsets=np.zeros(n + 1, np.uint64),
                                              l0_neighbors=np.zeros(0, np.uint64), max_batch=b)
    q = rng.standard_normal((b, dim)).astype(np.float32)
    gid, gsc, gcnt, _ = gix.flat_search_batch(q, k)
    for qi in (0, 17, 500, 1023):
        rc, oid, osc = orc.flat_matrix(orc.L2SQ, data, q[qi], k, kernel=orc.K_AVX_FMA_HW)
        assert gid[qi].tolist() == oid.tolist() and bits(gsc[qi]).tolist() == bits(osc).tolist()


def test_f32_exact_scan_answers_only_the_uncertified_queries_with_the_valu_kernel(orc, hv):
    """A corpus with one blob of 3 000 near-identical rows: queries that land in the blob cannot be certified and are
    re-answered (alone) by the exact VALU scan, the other ~1 000 queries of the batch keep their matrix-core result."""
    rng = np.random.default_rng(21)
    n, dim, b, k = 20000, 512, 1024, 10
    centers = rng.standard_normal((16, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 16, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
    blob = rng.standard_normal(dim).astype(np.float32) * np.float32(3.0)
    data[5000:8000] = blob
    data[5000:8000, 0] += (np.arange(3000) % 11).astype(np.float32) * np.float32(1e-6)
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64),
                                              vectors=data, l0_offsets=np.zeros(n + 1, np.uint64),
                                              l0_neighbors=np.zeros(0, np.uint64), max_batch=b)
    q = (centers[rng.integers(0, 16, b)] + 0.5 * rng.standard_normal((b, dim))).astype(np.float32)
    hard = [3, 100, 511, 512, 777, 1023]
    for i in hard:
        q[i] = blob + np.float32(0.01) * rng.standard_normal(dim).astype(np.float32)
    gid, gsc, gcnt, _ = gix.flat_search_batch(q, k)
    assert gix.last_scan_path() & hv.PATH_VALU_FALLBACK_QUERIES and not gix.last_scan_path() & hv.PATH_VALU  # those queries only
    for qi in hard + [0, 1, 99, 640, 1022]:
        rc, oid, osc = orc.flat_matrix(orc.L2SQ, data, q[qi], k, kernel=orc.K_AVX_FMA_HW)
        assert gcnt[qi] == k and gid[qi].tolist() == oid.tolist(), f"query {qi}"
        assert bits(gsc[qi]).tolist() == bits(osc).tolist()
    assert all(5000 <= int(x) < 8000 for x in gid[3])


@pytest.mark.parametrize("dtype_name,metric,dim", [("bf16", 1, 128), ("bf16", 0, 256), ("fp8", 1, 128)])
def test_restricted_scan_over_bf16_and_fp8_rows(orc, hv, dtype_name, metric, dim):
    """search_restricted (restricted.rs:753-835) over quantised storage: the matrix-core pipeline gathers the tile's rows
    through the candidate row list; results equal the oracle's exact scan over the allowed ids on the stored values."""
    rng = np.random.default_rng(900 + dim + metric)
    n, k, b = 6000, 10, 9
    data = (rng.standard_normal((n, dim)) * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
    stored = fx.round_bf16(data) if dtype_name == "bf16" else fx.quantize_fp8_rows(data)
    ids = np.arange(n, dtype=np.uint64) * 2 + 5                       # non-contiguous external ids
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=metric, node_ids=ids, vectors=data,
                                              dtype=hv.BF16 if dtype_name == "bf16" else hv.FP8_E4M3,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), max_batch=16)
    oix = orc.Index(dim, metric)
    assert oix.seed(ids, stored, np.zeros(n + 1, np.uint64), np.zeros(0, np.uint64)) == orc.OK
    q = rng.standard_normal((b, dim)).astype(np.float32)
    for size in (37, 900):
        allowed = np.concatenate([rng.choice(ids, size, replace=False), np.array([4, 10**9], np.uint64)])  # + ids without a vector
        cand = hv.RestrictedVectorCandidates.from_ids(allowed)
        gid, gsc, gcnt = gix.search_restricted_batch(q, hv.SearchParams(k), cand)
        for qi in range(b):
            rc, oid, osc = oix.flat(q[qi], k, allowed=allowed)
            assert rc == orc.OK and gcnt[qi] == oid.size == k
            assert gid[qi, :k].tolist() == oid.tolist(), f"{size} candidates, query {qi}"
            assert bits(gsc[qi, :k]).tolist() == bits(osc).tolist()


def test_large_restricted_scan_over_f32_rows_takes_the_matrix_cores(orc, hv):
    """A 30 000-candidate restricted scan x 512 queries x 768 dims is above the work threshold: row-list gather on the
    matrix-core pipeline, equal to the VALU kernel and the oracle bit for bit."""
    rng = np.random.default_rng(33)
    n, dim, b, k = 60000, 768, 512, 10
    centers = rng.standard_normal((24, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 24, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64), vectors=data,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), max_batch=b)
    q = (centers[rng.integers(0, 24, b)] + 0.5 * rng.standard_normal((b, dim))).astype(np.float32)
    allowed = np.sort(rng.choice(n, 30000, replace=False)).astype(np.uint64)
    cand = hv.RestrictedVectorCandidates.from_ids(allowed)
    gid, gsc, gcnt = gix.search_restricted_batch(q, hv.SearchParams(k), cand)
    assert gix.last_scan_path() & (hv.PATH_MFMA_128 | hv.PATH_TILE_256 | hv.PATH_SMALL_BATCH)
    gix.set_option(hv.OPT_FLAT_FORCE_VALU, 1)
    vid, vsc, vcnt = gix.search_restricted_batch(q, hv.SearchParams(k), cand)
    assert gix.last_scan_path() == hv.PATH_VALU
    assert gid.tolist() == vid.tolist() and bits(gsc).tolist() == bits(vsc).tolist() and gcnt.tolist() == vcnt.tolist()
    sub = data[allowed.astype(np.int64)]
    for qi in (0, 100, 511):
        rc, oid, osc = orc.flat_matrix(orc.L2SQ, sub, q[qi], k, kernel=orc.K_AVX_FMA_HW)
        assert allowed[oid.astype(np.int64)].tolist() == gid[qi].tolist() and bits(osc).tolist() == bits(gsc[qi]).tolist()


# --- the 256 x 256 filtered contraction kernel (hvx_flat_tile.hip) ---
TILE_CASES = [("bf16", 1, 768, 30000, 10, 300), ("bf16", 0, 256, 20001, 25, 70), ("fp8", 1, 1536, 12000, 10, 130),
              ("fp8", 0, 128, 20000, 10, 257), ("f32", 1, 512, 40000, 10, 512), ("f32", 0, 768, 30000, 100, 400)]


@pytest.mark.parametrize("dtype_name,metric,dim,n,k,b", TILE_CASES)
def test_exact_scan_through_the_256_tile_kernel(orc, hv, dtype_name, metric, dim, n, k, b, tile_build=0):
    """Every slice after the first chunk of an exact scan runs on the 256 x 256 filtered kernel (global_load_lds staging,
    swizzled LDS image, fp8 codes widened in registers, f32 rows through their bf16 shadow).  With 2 048-row first chunks
    a small corpus takes several slices (ragged last row tile, padded query tile); the answer must equal the 128 x 128
    kernel's and the oracle's exact scan over the stored values, ids and score bits."""
    rng = np.random.default_rng(7000 + dim + n)
    centers = rng.standard_normal((32, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 32, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
    if dtype_name == "fp8":
        data *= rng.uniform(0.2, 3.0, (n, 1)).astype(np.float32)
    data[n - 5] = data[3]                                             # a duplicate in the last (ragged) row tile
    stored = {"bf16": fx.round_bf16, "fp8": fx.quantize_fp8_rows, "f32": lambda x: x}[dtype_name](data)
    dt = {"bf16": hv.BF16, "fp8": hv.FP8_E4M3, "f32": hv.F32}[dtype_name]
    ids = np.arange(n, dtype=np.uint64) + 11
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=metric, node_ids=ids, vectors=data, dtype=dt,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), max_batch=b)
    q = (centers[rng.integers(0, 32, b)] + 0.5 * rng.standard_normal((b, dim))).astype(np.float32)
    q[0] = data[3]
    q[b - 1, 5] = np.nan                                              # a rejected query in the padded query tile
    gix.set_option(hv.OPT_FLAT_FIRST_CHUNK, 2048)
    gix.set_option(hv.OPT_FLAT_TILE_BUILD, tile_build)
    gix.set_option(hv.OPT_FLAT_NO_SMALLB, 1)                           # (batches of <= 128 queries would take the small-batch kernel)
    gid, gsc, gcnt, _, gst = gix.flat_search_batch(q, k, per_query_status=True)
    assert gix.last_scan_path() & hv.PATH_TILE_256, gix.last_scan_path()   # the kernel ran
    if gix.last_scan_path() & hv.PATH_PAIR_OVERFLOW_REPEAT:                # (a 2 048-row first chunk gives loose thresholds: a slice may
        gix.set_option(hv.OPT_FLAT_FIRST_CHUNK, 8192)                       #  overflow its pair buffer and be repeated unfiltered --
        gid, gsc, gcnt, _, gst = gix.flat_search_batch(q, k, per_query_status=True)   # then check the tile kernel's OWN answer too)
        assert gix.last_scan_path() & hv.PATH_TILE_256 and not gix.last_scan_path() & hv.PATH_PAIR_OVERFLOW_REPEAT, gix.last_scan_path()
    gix.set_option(hv.OPT_FLAT_NO_TILE, 1)
    oid_, osc_, ocnt_, _, ost_ = gix.flat_search_batch(q, k, per_query_status=True)
    assert not gix.last_scan_path() & hv.PATH_TILE_256 and gix.last_scan_path() & hv.PATH_MFMA_128
    assert gid.tolist() == oid_.tolist() and bits(gsc).tolist() == bits(osc_).tolist() and gcnt.tolist() == ocnt_.tolist()
    assert gst.tolist() == ost_.tolist() and gst[b - 1] == 2 and not gst[: b - 1].any()
    assert gcnt[b - 1] == 0 and gcnt[0] == k
    kern = orc.K_AVX_FMA_HW if dtype_name == "f32" else None
    for qi in list(range(0, b - 1, max(1, b // 16))) + [b - 2]:
        rc, oid, osc = orc.flat_matrix(metric, stored, q[qi], k, **({"kernel": kern} if kern is not None else {}))
        assert rc == orc.OK and (gid[qi, :gcnt[qi]] - 11).tolist() == oid.tolist(), f"query {qi}"
        assert bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
    assert sorted((gid[0, :2] - 11).tolist()) == [3, n - 5]


@pytest.mark.parametrize("dtype_name,metric,dim,n,k,b", [TILE_CASES[0], TILE_CASES[2]])
def test_exact_scan_through_the_512_thread_tile_build(orc, hv, dtype_name, metric, dim, n, k, b):
    """HVX_OPT_FLAT_TILE_BUILD = 1: the one-workgroup-per-CU build (256 x 256 tiles, two 64-deep LDS buffers; in tuning builds
    it carries the measurement switches) -- same answers as the default build and the oracle."""
    test_exact_scan_through_the_256_tile_kernel(orc, hv, dtype_name, metric, dim, n, k, b, tile_build=1)


@pytest.mark.parametrize("dtype_name,metric,dim,n,k,b", TILE_CASES)
def test_exact_scan_through_the_role_split_tile_build(orc, hv, dtype_name, metric, dim, n, k, b):
    """HVX_OPT_FLAT_TILE_BUILD = 2 (round 4): 512 threads on a 256 x 256 tile with the two wavefronts of every SIMD in opposite roles
    (one issues its MFMAs at priority while the other has its fragment reads and LDS-DMA copies in flight; raw barriers, counted
    waits) -- bf16 / fp8 / f32-shadow rows, ragged tiles, a rejected query: the same answers as the 128 x 128 kernel and the oracle."""
    test_exact_scan_through_the_256_tile_kernel(orc, hv, dtype_name, metric, dim, n, k, b, tile_build=2)


@pytest.mark.parametrize("dtype_name,metric,dim,n,k,b", [("fp8", 1, 1536, 12000, 10, 130), ("fp8", 0, 128, 20000, 10, 257), ("fp8", 1, 768, 25000, 10, 300),
                                                         ("fp8", 1, 256, 30000, 25, 64)])
def test_exact_scan_through_the_mx_fp8_tile_build(orc, hv, dtype_name, metric, dim, n, k, b):
    """HVX_OPT_FLAT_TILE_BUILD = 3 (round 4): fp8 rows x the query as two e4m3 pieces on v_mfma_scale_f32_32x32x64_f8f6f4 (the
    query's power-of-two scale in the instruction's block scale, no widening of the codes) -- the same answers as the 128 x 128 kernel
    (every query) and the oracle; a scan that drops a true neighbour in its filtered epilogue would differ."""
    test_exact_scan_through_the_256_tile_kernel(orc, hv, dtype_name, metric, dim, n, k, b, tile_build=3)


@pytest.mark.parametrize("dtype_name,metric,dim,n,k,b", [("fp8", 1, 1536, 12000, 10, 130), ("fp8", 0, 128, 20000, 10, 257)])
def test_exact_scan_through_the_bf16_widening_fp8_tile_build(orc, hv, dtype_name, metric, dim, n, k, b):
    """HVX_OPT_FLAT_TILE_BUILD = 4: fp8 codes widened to bf16 in registers (the default until round 4; the MX build's A/B partner)."""
    test_exact_scan_through_the_256_tile_kernel(orc, hv, dtype_name, metric, dim, n, k, b, tile_build=4)


def test_mx_scaled_mfma_operand_layout():
    """tests/native/mx_probe.hip: the operand layout and the per-row block scale the MX build relies on, on this device."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "native", "_bin", "mx_probe")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "helix-db_amd", "csrc"), "probe"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "test 0: 0 of 1024" in r.stdout and "test 1: 0 of 1024" in r.stdout, r.stdout[-1500:]


def test_exact_scan_lanes_share_one_bf16_shadow(orc, hv):
    """Forked handles (execution lanes) of an f32 index scan through ONE bf16 shadow of the rows, whichever lane builds it."""
    rng = np.random.default_rng(77)
    n, dim, b, k = 30000, 512, 600, 10
    centers = rng.standard_normal((32, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 32, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64), vectors=data,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), max_batch=b)
    lane = gix.fork()
    q = (centers[rng.integers(0, 32, b)] + 0.5 * rng.standard_normal((b, dim))).astype(np.float32)
    lid, lsc, lcnt, _ = lane.flat_search_batch(q, k)       # the fork builds the shadow ...
    assert lane.last_scan_path() & hv.PATH_TILE_256
    gid, gsc, gcnt, _ = gix.flat_search_batch(q, k)        # ... the parent finds it
    assert gix.last_scan_path() & hv.PATH_TILE_256
    assert lid.tolist() == gid.tolist() and bits(lsc).tolist() == bits(gsc).tolist()
    lane.close()
    gid2, gsc2, _, _ = gix.flat_search_batch(q, k)         # and keeps it after the lane is gone
    assert gid2.tolist() == gid.tolist() and bits(gsc2).tolist() == bits(gsc).tolist()
    for qi in range(0, b, 60):
        rc, oid, osc = orc.flat_matrix(orc.L2SQ, data, q[qi], k, kernel=orc.K_AVX_FMA_HW)
        assert rc == orc.OK and gid[qi].tolist() == oid.tolist() and bits(gsc[qi]).tolist() == bits(osc).tolist()


def test_restricted_scan_through_the_256_tile_kernel(orc, hv):
    """The 256 x 256 kernel gathers its row tile through the candidate row list (restricted scans over bf16 rows)."""
    rng = np.random.default_rng(71)
    n, dim, b, k = 40000, 256, 100, 10
    data = rng.standard_normal((n, dim)).astype(np.float32)
    stored = fx.round_bf16(data)
    ids = np.arange(n, dtype=np.uint64) * 3 + 1
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=ids, vectors=data, dtype=hv.BF16,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), max_batch=b)
    oix = orc.Index(dim, orc.L2SQ)
    assert oix.seed(ids, stored, np.zeros(n + 1, np.uint64), np.zeros(0, np.uint64)) == orc.OK
    q = rng.standard_normal((b, dim)).astype(np.float32)
    allowed = rng.choice(ids, 9001, replace=False)
    cand = hv.RestrictedVectorCandidates.from_ids(allowed)
    gix.set_option(hv.OPT_FLAT_FIRST_CHUNK, 1024)
    gix.set_option(hv.OPT_FLAT_NO_SMALLB, 1)
    gix.set_option(hv.OPT_RESTRICTED_DIRECT, 1)  # (a set this small would take the one-launch reference-order kernel, round 6)
    gid, gsc, gcnt = gix.search_restricted_batch(q, hv.SearchParams(k), cand)
    assert gix.last_scan_path() & hv.PATH_TILE_256
    for qi in range(0, b, 7):
        rc, oid, osc = oix.flat(q[qi], k, allowed=allowed)
        assert rc == orc.OK and gid[qi, :k].tolist() == oid.tolist() and bits(gsc[qi, :k]).tolist() == bits(osc).tolist()


# --- execution lanes (hvx_index_fork) and the two-queries-per-SIMD build of the wave kernel ---
@pytest.mark.parametrize("n,dim,metric,m,m0,efc,ef,k,nq", WAVE_CASES)
def test_two_queries_per_simd_build_equals_oracle(orc, hv, n, dim, metric, m, m0, efc, ef, k, nq):
    """hvx_index_set_occupancy(2): same ids, score bits and SearchStats counters as the oracle on every wave-kernel shape."""
    rng = np.random.default_rng(4000 + dim + metric)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    lv = fx.draw_levels(n, m, seed=dim + 1)
    oix = build_oracle(orc, data, metric, lv, m=m, m0=m0, efc=efc)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=metric, m=m, m0=m0)
    gix.set_occupancy(2)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    assert_hnsw_equal(orc, hv, oix, gix, q, k, ef)


def test_two_queries_per_simd_build_spill_path(orc, hv):
    """The half-LDS build spills to the HBM bitmap at 7/8 of its (smaller) table: forced with a 256-slot table."""
    rng = np.random.default_rng(77)
    n, dim = 2500, 128
    data = rng.standard_normal((n, dim)).astype(np.float32)
    oix = build_oracle(orc, data, orc.L2SQ, fx.draw_levels(n, 16, seed=5), efc=80)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=hv.EUCLIDEAN)
    gix.set_option(hv.OPT_WAVE_LOG2CAP, 8)
    gix.set_occupancy(2)
    q = rng.standard_normal((32, dim)).astype(np.float32)
    assert_hnsw_equal(orc, hv, oix, gix, q, 10, 128)
    assert_hnsw_equal(orc, hv, oix, gix, q, 10, 128)  # the bitmap was handed back zeroed


def test_forked_lanes_share_the_image_and_outlive_their_parent(orc, hv):
    import threading
    rng = np.random.default_rng(91)
    n, dim = 3000, 128
    data = rng.standard_normal((n, dim)).astype(np.float32)
    oix = build_oracle(orc, data, orc.COSINE, fx.draw_levels(n, 16, seed=9), efc=80)
    oix.set_simhash(42)
    root = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=hv.COSINE)
    root.set_simhash()
    lanes = [root.fork() for _ in range(3)]
    lanes[1].set_occupancy(2)
    with pytest.raises(hv.HelixDbError):
        lanes[0].set_simhash()  # the SimHash rows belong to the image: root handle only
    q = rng.standard_normal((96, dim)).astype(np.float32)
    want = root.search_batch(q, hv.SearchParams(10).with_ef(128))
    want_p = root.search_batch(q, hv.SearchParams.new(10))
    want_f = root.flat_search_batch(q, 10)
    root.close()  # lanes keep the image alive
    out = [None] * len(lanes)

    def work(i):
        res = []
        for _ in range(5):
            a = lanes[i].search_batch(q, hv.SearchParams(10).with_ef(128))
            b = lanes[i].search_batch(q, hv.SearchParams.new(10))
            c = lanes[i].flat_search_batch(q, 10)
            res.append((a, b, c))
        out[i] = res

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(lanes))]
    [t.start() for t in th]
    [t.join() for t in th]
    for res in out:
        for a, b, c in res:
            for got, exp in ((a, want), (b, want_p), (c, want_f)):
                assert got[0].tolist() == exp[0].tolist() and bits(got[1]).tolist() == bits(exp[1]).tolist()
                assert got[2].tolist() == exp[2].tolist()
    for qi in range(0, 96, 7):  # and the lanes agree with the oracle
        rc, oid, osc = oix.search(q[qi], 10, 128)
        assert out[1][0][0][0][qi, :len(oid)].tolist() == oid.tolist()
    for ln in lanes:
        ln.close()


@pytest.mark.parametrize("dtype_name", ["f32", "bf16", "fp8"])
def test_import_from_device_memory_equals_import_from_host(hv, dtype_name):
    """hvx_index_import accepts rows already resident on a device; the copy is ordered before the import's own kernels
    (rounding / quantisation, validation, packing), so the image equals the one built from host rows -- sized so that
    the copy is still in flight when the first kernel would otherwise start."""
    import torch
    rng = np.random.default_rng(123)
    n, dim = 400_000, 256
    data = rng.standard_normal((n, dim)).astype(np.float32)
    dt = {"f32": hv.F32, "bf16": hv.BF16, "fp8": hv.FP8_E4M3}[dtype_name]
    kw = dict(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64), l0_offsets=np.zeros(n + 1, np.uint64),
              l0_neighbors=np.zeros(0, np.uint64), dtype=dt, max_batch=64)
    host = hv.ValidatedVectorReadIndex.managed(vectors=data, **kw)
    xd = torch.from_numpy(data).to("cuda:0")
    torch.cuda.synchronize()
    devi = hv.ValidatedVectorReadIndex.managed(vectors=xd, **kw)
    q = data[n - 64:] + np.float32(0.01)  # nearest rows sit at the END of the array: the last bytes a copy delivers
    a = host.flat_search_batch(q, 10)
    b = devi.flat_search_batch(q, 10)
    assert a[0].tolist() == b[0].tolist() and bits(a[1]).tolist() == bits(b[1]).tolist()
    assert all(int(a[0][i, 0]) == n - 64 + i for i in range(64))


def test_c_abi_example_runs_and_agrees_with_the_oracle(orc, tmp_path):
    """examples/c_abi_example.c, compiled as C99 against include/helix_vec.h, RUN on the device: its three answers (strict
    search with distance-computation count, production-default params with RNG word count, restricted search) equal the
    oracle's on the same 8-node index."""
    import os, re, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "c_abi_example"
    cmd = ["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_example.c"), "-L",
           os.path.join(root, "helix-db_amd"), "-lhelix_vec_gfx950", f"-Wl,-rpath,{os.path.join(root, 'helix-db_amd')}", "-o", str(exe)]
    assert subprocess.run(cmd, capture_output=True, text=True).returncode == 0
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    out = run.stdout
    n, dim, k = 8, 128, 3
    vec = np.zeros((n, dim), np.float32)
    vec[:, 0] = np.arange(n)
    ids = np.arange(n, dtype=np.uint64) + 100
    offs, nbrs = [0], []
    for i in range(n):
        if i > 0:
            nbrs.append(100 + i - 1)
        if i + 1 < n:
            nbrs.append(100 + i + 1)
        offs.append(len(nbrs))
    oix = orc.Index(dim, orc.L2SQ, m=16, m0=32)
    assert oix.seed(ids, vec, np.array(offs, np.uint64), np.array(nbrs, np.uint64), entry_point=100, max_layer=0) == orc.OK
    q = np.zeros(dim, np.float32)
    q[0] = np.float32(5.2)
    rc, oid, osc, ost = oix.search(q, k, 16, with_stats=True)
    m = re.search(r"strict : (\d+) results, nearest id (\d+) score (\S+), (\d+) distance computations", out)
    assert m, out
    assert (int(m.group(1)), int(m.group(2)), int(m.group(4))) == (len(oid), int(oid[0]), ost["distance_computations"])
    assert np.float32(float(m.group(3))) == np.float32(float("%g" % osc[0]))
    oix.set_simhash(42)
    rc, pid, psc, pst = oix.search_params(q, orc.SearchParams.new(k), with_stats=True)
    m = re.search(r"default: (\d+) results, nearest id (\d+), (\d+) RNG words drawn", out)
    assert m and (int(m.group(1)), int(m.group(2)), int(m.group(3))) == (len(pid), int(pid[0]), pst["rng_words"]), out
    rc, rid, rsc = oix.flat(q, k, allowed=np.array([100, 103, 107], np.uint64))
    m = re.search(r"restricted: (\d+) results, nearest allowed id (\d+)", out)
    assert m and (int(m.group(1)), int(m.group(2))) == (len(rid), int(rid[0])), out
    assert oix.delete(105) == (orc.OK, True) and oix.delete(9999) == (orc.OK, False)          # the example's write
    rc, did, dsc = oix.search(q, k, 16)
    m = re.search(r"after delete: (\d+) deleted, (\d+) missing, (\d+) live rows, nearest id (\d+), contains\(105\) = (\d)", out)
    assert m and [int(g) for g in m.groups()] == [1, 1, oix.count, int(did[0]), 0], out


def test_restricted_device_scan_dominates_the_reference_filter_aware_walk(orc, hv):
    """Row a11: above 256 candidates the reference answers with an ACORN-style walk (restricted.rs:837-1148, recall gate 0.95);
    the device scans the allowed rows exactly.  On the reference's own fixture (tests/production_support/vector/restricted.rs:
    1226-1285: 512 x 8-D circle, ef 64, k 10, allowed = ids not divisible by 3) the device's answer IS the exact top-k, so its
    recall is 1.0 >= the walk's (restated in the oracle, pinned by tests/test_oracle_restricted.py) >= 0.95."""
    import math
    n, dim, k = 512, 8, 10

    def vec(i):
        v = np.zeros(dim, np.float32)
        a = math.tau * i / n
        v[0], v[1] = np.float32(math.cos(a)), np.float32(math.sin(a))
        return v

    ids = np.arange(1, n + 1, dtype=np.uint64)
    data = np.stack([vec(i) for i in range(1, n + 1)])
    offs, nbrs = [0], []
    for i in range(1, n + 1):
        nbrs.extend(fx.skip_neighbors(i, n))
        offs.append(len(nbrs))
    offs, nbrs = np.array(offs, np.uint64), np.array(nbrs, np.uint64)
    oix = orc.Index(dim, orc.COSINE, m=32, m0=64)
    assert oix.seed(ids, data, offs, nbrs, entry_point=1, max_layer=0) == orc.OK
    oix.set_simhash(42)
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.COSINE, node_ids=ids, vectors=data, l0_offsets=offs, l0_neighbors=nbrs,
                                              entry_point=1, max_layer=0, m=32, m0=64)
    allowed = np.array([i for i in range(1, n + 1) if i % 3 != 0], np.uint64)
    q = np.stack([vec(i) for i in (1, 43, 87, 129, 211, 307, 401, 509)])
    gid, gsc, gcnt = gix.search_restricted_batch(q, hv.SearchParams(k).with_ef(64), hv.RestrictedVectorCandidates.from_ids(allowed))
    walk_hits = dev_hits = 0
    for qi in range(q.shape[0]):
        rc, exact, esc = oix.flat(q[qi], k, allowed=allowed)
        assert gid[qi, :gcnt[qi]].tolist() == exact.tolist() and bits(gsc[qi, :gcnt[qi]]).tolist() == bits(esc).tolist()
        rc, wid, _, st = oix.search_restricted(q[qi], k, 64, allowed)
        assert rc == orc.OK and st["strategy"] == orc.RESTRICTED_FILTERED
        walk_hits += len(set(wid.tolist()) & set(exact.tolist()))
        dev_hits += len(set(gid[qi, :gcnt[qi]].tolist()) & set(exact.tolist()))
    assert dev_hits == 8 * k and dev_hits >= walk_hits and walk_hits / (8.0 * k) >= 0.95


def test_shard_group_runs_search_allgather_merge_in_one_call(orc, hv):
    """hvx_shard_group_*: a group of ONE rank with a real RCCL communicator (ncclCommInitRank + ncclAllGather on this GPU):
    the one-call step returns exactly what a direct search returns, for strict search, repeated calls and smaller batches."""
    import torch
    rng = np.random.default_rng(321)
    n, dim, b, k, ef = 4000, 128, 96, 10, 64
    data = rng.standard_normal((n, dim)).astype(np.float32)
    oix = build_oracle(orc, data, orc.L2SQ, fx.draw_levels(n, 16, seed=4), efc=80)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=hv.EUCLIDEAN, max_batch=128)
    lane = gix.fork()
    grp = hv.ShardGroup(lane, hv.ShardGroup.unique_id(), 0, 1, 128, 16)
    dev = torch.device("cuda", 0)
    q = rng.standard_normal((b, dim)).astype(np.float32)
    dq = torch.from_numpy(q).to(dev)
    ids = torch.zeros(b, k, dtype=torch.int64, device=dev); sc = torch.zeros(b, k, dtype=torch.float32, device=dev)
    cnt = torch.zeros(b, dtype=torch.int32, device=dev)
    want = gix.search_batch(q, hv.SearchParams(k).with_ef(ef))
    for _ in range(3):
        grp.search_batch_device(dq, k, ef, ids, sc, cnt)
    lane.sync()
    assert ids.cpu().numpy().astype(np.uint64).tolist() == want[0].tolist()
    assert bits(sc.cpu().numpy()).tolist() == bits(want[1]).tolist() and cnt.cpu().numpy().tolist() == want[2].tolist()
    grp.search_batch_device(dq[:7], k, ef, ids, sc, cnt)  # a smaller batch: another payload geometry
    lane.sync()
    assert ids[:7].cpu().numpy().astype(np.uint64).tolist() == want[0][:7].tolist()
    with pytest.raises(hv.HelixDbError):
        grp.search_batch_device(dq, 17, 64, ids, sc, cnt)  # k beyond the group's max_k
    # round 5: a second execution lane of the rank joins the SAME communicator (hvx_shard_group_attach); steps of both lanes
    # alternate, their all-gathers share the rank's exchange stream; either group may be freed first
    assert hv.rccl_version() >= 2000
    lane2 = gix.fork()
    grp2 = grp.attach(lane2)
    ids2 = torch.zeros(b, k, dtype=torch.int64, device=dev); sc2 = torch.zeros(b, k, dtype=torch.float32, device=dev)
    cnt2 = torch.zeros(b, dtype=torch.int32, device=dev)
    for _ in range(4):
        grp.search_batch_device(dq, k, ef, ids, sc, cnt)
        grp2.search_batch_device(dq, k, ef, ids2, sc2, cnt2)
    lane.sync(); lane2.sync()
    for got_i, got_s, got_c in ((ids, sc, cnt), (ids2, sc2, cnt2)):
        assert got_i.cpu().numpy().astype(np.uint64).tolist() == want[0].tolist()
        assert bits(got_s.cpu().numpy()).tolist() == bits(want[1]).tolist() and got_c.cpu().numpy().tolist() == want[2].tolist()
    grp.close()                                            # the primary goes first: the communicator stays with grp2
    grp2.search_batch_device(dq, k, ef, ids2, sc2, cnt2)
    lane2.sync()
    assert ids2.cpu().numpy().astype(np.uint64).tolist() == want[0].tolist()
    grp2.close()
    lane.close()
    lane2.close()


# --- the one-pass small-batch exact scan (hvx_flat_smallb.hip): b <= 128 queries, rows held in registers ---
@pytest.mark.parametrize("dtype_name,metric,dim,n,k,b", [("f32", 1, 1536, 9000, 10, 32), ("f32", 0, 768, 20011, 10, 7), ("f32", 1, 256, 40000, 25, 64),
                                                         ("f32", 1, 512, 30000, 10, 100), ("f32", 1, 256, 33000, 100, 128), ("f32", 1, 768, 6001, 10, 1),
                                                         ("bf16", 1, 768, 20000, 10, 33), ("bf16", 0, 1536, 5000, 10, 32), ("bf16", 1, 256, 9000, 10, 128)])
def test_small_batch_exact_scan_is_bit_exact(orc, hv, dtype_name, metric, dim, n, k, b):
    """Batches of <= 128 queries over >= 2^22 row elements take the register-resident one-pass kernel (candidates) + sliced
    selection + exact re-rank + certificate: ids and score bits equal the oracle's exact scan and the VALU kernel's, for whole
    scans and restricted (row-list) scans, ragged row counts, a rejected query and a duplicate row."""
    rng = np.random.default_rng(dim + n + b)
    centers = rng.standard_normal((32, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 32, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
    data[7] = data[3]
    stored = fx.round_bf16(data) if dtype_name == "bf16" else data
    ids = np.arange(n, dtype=np.uint64) + 11
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=metric, node_ids=ids, vectors=data, dtype=hv.BF16 if dtype_name == "bf16" else hv.F32,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), max_batch=max(b, 16))
    q = (centers[rng.integers(0, 32, b)] + 0.5 * rng.standard_normal((b, dim))).astype(np.float32)
    q[0] = data[3]
    if b > 2:
        q[b - 1, 5] = np.nan
    gid, gsc, gcnt, _, gst = gix.flat_search_batch(q, k, per_query_status=True)
    assert gix.last_scan_path() & hv.PATH_SMALL_BATCH and not gix.last_scan_path() & (hv.PATH_MFMA_128 | hv.PATH_TILE_256), gix.last_scan_path()
    kern = {"kernel": orc.K_AVX_FMA_HW} if dtype_name == "f32" else {}
    for qi in range(b):
        if b > 2 and qi == b - 1:
            assert gst[qi] == hv.ERR_NONFINITE and gcnt[qi] == 0
            continue
        rc, oid, osc = orc.flat_matrix(metric, stored, q[qi], k, **kern)
        assert gst[qi] == 0 and (gid[qi, :gcnt[qi]] - 11).tolist() == oid.tolist(), f"query {qi}"
        assert bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
    assert sorted((gid[0, :2] - 11).tolist()) == [3, 7]
    # the same through the other kernels of this handle
    gix.set_option(hv.OPT_FLAT_NO_SMALLB, 1)
    oid_, osc_, ocnt_, _, ost_ = gix.flat_search_batch(q, k, per_query_status=True)
    assert not gix.last_scan_path() & hv.PATH_SMALL_BATCH
    assert gid.tolist() == oid_.tolist() and bits(gsc).tolist() == bits(osc_).tolist() and gcnt.tolist() == ocnt_.tolist() and gst.tolist() == ost_.tolist()
    gix.set_option(hv.OPT_FLAT_NO_SMALLB, 0)
    # restricted: a candidate row list (ragged length, not a multiple of 32)
    gix.set_option(hv.OPT_RESTRICTED_DIRECT, 1)  # (few queries over a set this small would take the one-launch reference-order kernel, round 6)
    allowed = np.sort(rng.choice(ids, min(n - 5, max(2 ** 22 // dim + 77, 3001)), replace=False))
    good = q[: max(1, b - 1)] if b > 2 else q
    rid, rsc, rcnt = gix.search_restricted_batch(good, hv.SearchParams(k), hv.RestrictedVectorCandidates.from_ids(allowed))
    assert gix.last_scan_path() & hv.PATH_SMALL_BATCH
    sub = stored[(allowed - 11).astype(np.int64)]
    for qi in range(0, good.shape[0], max(1, good.shape[0] // 8)):
        rc, oid, osc = orc.flat_matrix(metric, sub, good[qi], k, **kern)
        assert allowed[oid.astype(np.int64)].tolist() == rid[qi, :rcnt[qi]].tolist() and bits(osc).tolist() == bits(rsc[qi, :rcnt[qi]]).tolist()


@pytest.mark.parametrize("dtype_name,metric,dim,n,k,b", [("f32", 1, 1536, 9000, 10, 32), ("f32", 1, 768, 20011, 10, 7), ("f32", 0, 512, 9000, 10, 32),
                                                         ("f32", 1, 512, 12000, 10, 1), ("f32", 1, 1024, 5003, 25, 17), ("f32", 0, 1536, 4000, 10, 3),
                                                         ("f32", 1, 256, 20000, 10, 31), ("bf16", 1, 768, 20000, 10, 32), ("bf16", 0, 1536, 5000, 10, 9),
                                                         ("bf16", 1, 128, 40001, 10, 32), ("bf16", 1, 256, 20000, 10, 3), ("bf16", 1, 1024, 9000, 10, 32)])
def test_ring_build_of_the_small_batch_scan(orc, hv, dtype_name, metric, dim, n, k, b):
    """Batches of <= 32 queries take the LDS-ring build (rows through per-wavefront rings of 128-byte lines, queries in registers,
    depth split over the wavefronts of a workgroup): its candidates are certified by the ONE-pass attempt (a wrong dot product would
    send the scan on to the next attempt -- the path word says which attempts ran, and it equals the register-fragment build's), ids
    and score bits equal the oracle's and the register-fragment build's, for whole scans, the full split and ragged, scattered
    restricted row lists."""
    rng = np.random.default_rng(dim * 7 + n + b)
    centers = rng.standard_normal((32, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 32, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
    data[7] = data[3]
    stored = fx.round_bf16(data) if dtype_name == "bf16" else data
    ids = np.arange(n, dtype=np.uint64) + 11
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=metric, node_ids=ids, vectors=data, dtype=hv.BF16 if dtype_name == "bf16" else hv.F32,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), max_batch=32)
    q = (centers[rng.integers(0, 32, b)] + 0.5 * rng.standard_normal((b, dim))).astype(np.float32)
    q[0] = data[3]
    kern = {"kernel": orc.K_AVX_FMA_HW} if dtype_name == "f32" else {}
    results = {}
    for build in (0, 2):
        gix.set_option(hv.OPT_FLAT_NO_SMALLB, build)
        gid, gsc, gcnt, _, gst = gix.flat_search_batch(q, k, per_query_status=True)
        path = gix.last_scan_path()
        assert path & hv.PATH_SMALL_BATCH and not path & (hv.PATH_MFMA_128 | hv.PATH_TILE_256 | hv.PATH_WIDENED | hv.PATH_VALU_FALLBACK_QUERIES), (build, path)
        results[build] = (gid.tolist(), bits(gsc).tolist(), gcnt.tolist(), gst.tolist(), path)  # (path: the same attempts certified both builds)
    assert results[0] == results[2]
    gix.set_option(hv.OPT_FLAT_NO_SMALLB, 0)
    for qi in range(b):
        rc, oid, osc = orc.flat_matrix(metric, stored, q[qi], k, **kern)
        assert gst[qi] == 0 and (gid[qi, :gcnt[qi]] - 11).tolist() == oid.tolist(), f"query {qi}"
        assert bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
    # the full split (hi + lo parts) on the ring build
    gix.set_option(hv.OPT_FLAT_NO_FAST, 1)
    fid, fsc, fcnt, _, fst = gix.flat_search_batch(q, k, per_query_status=True)
    path = gix.last_scan_path()
    assert path & hv.PATH_SMALL_BATCH and path & hv.PATH_FULL_SPLIT and not path & (hv.PATH_WIDENED | hv.PATH_VALU_FALLBACK_QUERIES), path
    assert (fid.tolist(), bits(fsc).tolist(), fcnt.tolist()) == (gid.tolist(), bits(gsc).tolist(), gcnt.tolist())
    gix.set_option(hv.OPT_FLAT_NO_FAST, 0)
    # restricted row lists: ragged and scattered (lists below 2^22 row elements take other kernels)
    gix.set_option(hv.OPT_RESTRICTED_DIRECT, 1)  # (... and few queries over a small list the one-launch reference-order kernel, round 6)
    for size in (max(2 ** 22 // dim + 77, 3001), max(2 ** 22 // dim + 1, 33)):
        allowed = np.sort(rng.choice(ids, min(n - 5, size), replace=False))
        rid, rsc, rcnt = gix.search_restricted_batch(q, hv.SearchParams(k), hv.RestrictedVectorCandidates.from_ids(allowed))
        path = gix.last_scan_path()
        assert path & hv.PATH_SMALL_BATCH and not path & (hv.PATH_WIDENED | hv.PATH_VALU_FALLBACK_QUERIES), path
        sub = stored[(allowed - 11).astype(np.int64)]
        for qi in range(0, b, max(1, b // 6)):
            rc, oid, osc = orc.flat_matrix(metric, sub, q[qi], k, **kern)
            assert allowed[oid.astype(np.int64)].tolist() == rid[qi, :rcnt[qi]].tolist() and bits(osc).tolist() == bits(rsc[qi, :rcnt[qi]]).tolist()


@pytest.mark.parametrize("cfg", ["0 768 1000 32 0 0", "0 1024 1000 32 0 1", "2 512 1000 32 0 0", "0 1536 1000 32 0 0", "2 256 1000 32 0 1", "0 768 1000 32 1 1",
                                 "2 1536 3000 32 0 1", "2 1536 3000 5 1 0", "0 128 999 7 1 1", "2 768 1001 32 1 1", "2 1024 77 1 0 0", "2 2048 500 32 1 0",
                                 "0 384 500 32 1 0", "2 128 5000 32 0 1", "0 256 33 32 0 0", "2 1536 20000 32 0 0"])
def test_small_batch_kernels_raw_dot_products(cfg):
    """The candidate kernels themselves (both builds of hvx_flat_smallb.hip), outside the pipeline whose exact re-rank and
    certificate would hide an approximate score that is wrong on the high side: every raw dot product against a double-precision
    dot product of the operands the build multiplies (kind, dim, rows, queries, full split, scattered row list).  This harness
    found the one-MFMA-step-short accumulator store of the first ring build (tests/native/smallq_probe.hip)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "native", "_bin", "smallq_probe")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "helix-db_amd", "csrc"), "probe"])
    r = subprocess.run([exe] + cfg.split(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ring build bad 0" in r.stdout and "register build bad 0" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]


def test_small_batch_scan_uses_the_bf16_shadow_once_it_exists(orc, hv):
    """An f32 index whose image already holds the bf16 shadow of its rows (built by an earlier large scan) streams the shadow --
    half the bytes -- in the small-batch kernel too; same answers."""
    rng = np.random.default_rng(99)
    n, dim, k = 30000, 512, 10
    centers = rng.standard_normal((32, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 32, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64), vectors=data,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), max_batch=600)
    qs = (centers[rng.integers(0, 32, 24)] + 0.5 * rng.standard_normal((24, dim))).astype(np.float32)
    a_id, a_sc, a_cnt, _ = gix.flat_search_batch(qs, k)                # f32 rows
    assert gix.last_scan_path() & hv.PATH_SMALL_BATCH
    qb = (centers[rng.integers(0, 32, 600)] + 0.5 * rng.standard_normal((600, dim))).astype(np.float32)
    gix.flat_search_batch(qb, k)                                        # a large scan builds the shadow
    assert gix.last_scan_path() & hv.PATH_TILE_256
    b_id, b_sc, b_cnt, _ = gix.flat_search_batch(qs, k)                # now the shadow
    assert gix.last_scan_path() & hv.PATH_SMALL_BATCH
    assert a_id.tolist() == b_id.tolist() and bits(a_sc).tolist() == bits(b_sc).tolist()
    for qi in range(0, 24, 5):
        rc, oid, osc = orc.flat_matrix(orc.L2SQ, data, qs[qi], k, kernel=orc.K_AVX_FMA_HW)
        assert a_id[qi].tolist() == oid.tolist() and bits(a_sc[qi]).tolist() == bits(osc).tolist()


@pytest.mark.parametrize("dtype_name,dim", [("f32", 100), ("bf16", 128), ("fp8", 256)])
def test_read_rows_returns_the_stored_values(orc, hv, dtype_name, dim):
    """hvx_index_read_rows_device hands back exactly the values the index computes distances on: the rows (f32), the RNE-rounded
    values (bf16), fl32(scale x decode(code)) (fp8) -- equal to the numpy twins of the import's rounding / quantiser."""
    import torch
    rng = np.random.default_rng(dim)
    n = 3000
    data = (rng.standard_normal((n, dim)) * rng.uniform(0.01, 30.0, (n, 1))).astype(np.float32)
    stored = {"bf16": fx.round_bf16, "fp8": fx.quantize_fp8_rows, "f32": lambda x: x}[dtype_name](data)
    dt = {"bf16": hv.BF16, "fp8": hv.FP8_E4M3, "f32": hv.F32}[dtype_name]
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64) * 3, vectors=data, dtype=dt,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64))
    out = torch.empty(1000, dim, dtype=torch.float32, device="cuda")
    gix.read_rows_device(1500, 1000, out)
    assert (out.cpu().numpy().view(np.uint32) == stored[1500:2500].view(np.uint32)).all()
    with pytest.raises(hv.HelixDbError):
        gix.read_rows_device(2500, 1000, out)


def test_shard_group_entry_points_and_status(orc, hv):
    """Every step of hvx_shard_group_* (strict, full SearchParams, exact scan, restricted) on a one-rank group with a real RCCL
    communicator equals the direct call, and the per-query status travels with the payload: a NaN query comes back with its
    validation status and no rows (search.rs:1120-1125) instead of count 0 under HVX_OK (VERDICT r2 missing #2)."""
    import torch
    rng = np.random.default_rng(654)
    n, dim, b, k, ef = 4000, 128, 40, 10, 64
    data = rng.standard_normal((n, dim)).astype(np.float32)
    oix = build_oracle(orc, data, orc.L2SQ, fx.draw_levels(n, 16, seed=4), efc=80)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=hv.EUCLIDEAN, max_batch=64)
    gix.set_simhash()
    lane = gix.fork()
    grp = hv.ShardGroup(lane, hv.ShardGroup.unique_id(), 0, 1, 64, 16)
    dev = torch.device("cuda", 0)
    q = rng.standard_normal((b, dim)).astype(np.float32)
    q[5, 3] = np.nan
    dq = torch.from_numpy(q).to(dev)

    def bufs():
        return (torch.zeros(b, k, dtype=torch.int64, device=dev), torch.zeros(b, k, dtype=torch.float32, device=dev),
                torch.zeros(b, dtype=torch.int32, device=dev), torch.full((b,), 77, dtype=torch.int32, device=dev))

    def same(got, want_ids, want_sc, want_cnt):
        lane.sync()
        ids, sc, cnt, st = (t.cpu().numpy() for t in got)
        assert st[5] == hv.ERR_NONFINITE and cnt[5] == 0 and not np.delete(st, 5).any()
        for i in range(b):
            if i == 5:
                continue
            c = int(want_cnt[i])
            assert cnt[i] == c and ids[i, :c].astype(np.uint64).tolist() == want_ids[i, :c].tolist() and bits(sc[i, :c]).tolist() == bits(want_sc[i, :c]).tolist()

    w = gix.search_batch(q, hv.SearchParams(k).with_ef(ef), per_query_status=True)
    g = bufs(); grp.search_batch_device(dq, k, ef, *g); same(g, w[0], w[1], w[2])
    p = hv.SearchParams.new(k)
    pi, ps, pc = bufs()[:3]; pst = torch.zeros(b, dtype=torch.int32, device=dev)
    gix.search_batch_params_device(dq, p, pi, ps, pc, pst)
    gix.sync()
    g = bufs(); grp.search_batch_params_device(dq, p, *g); same(g, pi.cpu().numpy().astype(np.uint64), ps.cpu().numpy(), pc.cpu().numpy())
    f = gix.flat_search_batch(q, k, per_query_status=True)
    g = bufs(); grp.flat_search_batch_device(dq, k, *g); same(g, f[0], f[1], f[2])
    # restricted: the group slices the id list by the shard's id range (all of it here) and plans like the direct call
    allowed = np.sort(rng.choice(np.arange(n, dtype=np.uint64), 900, replace=False))
    rp = hv.RestrictedParams.new(k, 100)
    d = gix.search_restricted_batch_params(q, rp, allowed)
    r_ids, r_sc, r_cnt, r_st = grp.search_restricted_batch(q, rp, allowed)
    assert r_st.tolist() == d[3].tolist() and r_st[5] == hv.ERR_NONFINITE and r_cnt.tolist() == d[2].tolist()
    for i in range(b):
        c = int(r_cnt[i])
        assert r_ids[i, :c].tolist() == d[0][i, :c].tolist() and bits(r_sc[i, :c]).tolist() == bits(d[1][i, :c]).tolist()
    with pytest.raises(hv.HelixDbError) as e:
        grp.search_restricted_batch(q[:1], hv.RestrictedParams.new(801, 801), np.arange(n, dtype=np.uint64))
    assert e.value.status == hv.ERR_K_RANGE
    grp.close()
    lane.close()


# ------------------------------------------------------------------------------------------------------------------------
# Round 4
# ------------------------------------------------------------------------------------------------------------------------
WIDE_CASES = [
    # (n, dim, metric, ef, k, dtype) -- strict searches with ef 353 .. 800: the 448 / 832-entry register beams of the wave kernel
    (6000, 128, 1, 400, 50, "f32"),      # 448-entry beam
    (6000, 128, 1, 416, 10, "f32"),      # its largest ef
    (6000, 128, 0, 800, 100, "f32"),     # 832-entry beam, cosine
    (4000, 768, 1, 512, 10, "f32"),      # the bench dimension
    (3000, 1536, 1, 600, 10, "f32"),
    (6000, 128, 1, 500, 20, "bf16"),     # bf16 rows refused ef > 352 before (hvx_api.hip, round 3)
    (4000, 768, 0, 800, 10, "bf16"),
]


@pytest.mark.parametrize("n,dim,metric,ef,k,dtype", WIDE_CASES)
def test_wide_beams_on_the_wave_kernel_equal_the_oracle(orc, hv, n, dim, metric, ef, k, dtype):
    """search.rs:267-1067 has no beam limit (parameters.rs:118-133 asks only ef >= k).  Rounds 1-3 served ef > 352 with the
    four-wavefront general kernel (f32) or not at all (bf16); round 4 runs them on the one-wavefront-per-query kernel with 448- and
    832-entry register beams and the 8 192-slot visited table (spilling to the exact bitmap): ids, score bits and all four
    SearchStats counters equal the oracle's, on both occupancy settings of the handle (the wide builds are one query per SIMD)."""
    rng = np.random.default_rng(4100 + dim + ef)
    centres = rng.standard_normal((24, dim)).astype(np.float32)
    data = (centres[rng.integers(0, 24, n)] + 0.6 * rng.standard_normal((n, dim))).astype(np.float32)
    lv = fx.draw_levels(n, 16, seed=ef)
    stored = fx.round_bf16(data) if dtype == "bf16" else data
    oix = build_oracle(orc, stored, metric, lv, efc=80)
    ex = oix.export()
    ex["vectors"] = data
    gix = hv.ValidatedVectorReadIndex.from_export(ex, dim=dim, metric=metric, dtype=hv.BF16 if dtype == "bf16" else hv.F32)
    q = (centres[rng.integers(0, 24, 24)] + 0.6 * rng.standard_normal((24, dim))).astype(np.float32)
    for occ in (1, 2):
        gix.set_occupancy(occ)
        assert_hnsw_equal(orc, hv, oix, gix, q, k, ef)
    gix.set_option(hv.OPT_WAVE_LOG2CAP, 9)   # 512-slot table: every query continues on the HBM bitmap
    assert_hnsw_equal(orc, hv, oix, gix, q, k, ef)
    gix.set_option(hv.OPT_WAVE_LOG2CAP, 0)
    assert_hnsw_equal(orc, hv, oix, gix, q[:5], k, ef)   # bitmap handed back clean


def test_tie_rerun_reaches_the_832_entry_beam(orc, hv):
    """The re-run chain of the unrolled strict builds is 192 -> 384 -> 832 entries (round 3: 192 -> 384 only): 300 exact copies of
    each of 10 vectors at ef = 256 overflow the 384-entry beam's slack (128) on equal scores; the listed queries are searched again
    with 832 entries and then equal the oracle; nothing stays flagged.  And a launch on a corpus WITHOUT duplicates lists nothing."""
    rng = np.random.default_rng(4200)
    base = rng.standard_normal((10, 128)).astype(np.float32)
    data = np.repeat(base, 300, axis=0)[rng.permutation(3000)]
    n = data.shape[0]
    oix = build_oracle(orc, data, orc.L2SQ, fx.draw_levels(n, 16, seed=8), efc=80)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=128, metric=hv.EUCLIDEAN)
    q = np.concatenate([base[:6], base[6:10] + np.float32(0.01) * rng.standard_normal((4, 128)).astype(np.float32)])
    for occ in (1, 2):
        gix.set_occupancy(occ)
        assert_hnsw_equal(orc, hv, oix, gix, q, 10, 256)    # asserts tie_overflow_queries == 0
        assert_hnsw_equal(orc, hv, oix, gix, q, 10, 256)    # the list was handed back empty: a second call behaves the same


def test_restricted_auto_plan_beyond_the_walk_limits_is_answered_exactly(orc, hv):
    """k >= 301 (or ef > 800) gives bridge_rows = 8 x max(1.5 ef, 4 k) > 9 600 under the reference's own budget rule
    (restricted.rs:230-259); the reference serves k up to 800.  The AUTO plan then takes the exact gathered scan (strategy EXACT in
    the stats, a superset of the walk in recall) instead of HVX_ERR_UNSUPPORTED; bf16 rows, which the walk does not read, likewise.
    Explicit FILTERED keeps failing loudly."""
    rng = np.random.default_rng(4300)
    n, dim = 5000, 64
    data = rng.standard_normal((n, dim)).astype(np.float32)
    oix = build_oracle(orc, data, orc.L2SQ, fx.draw_levels(n, 16, seed=3), efc=60)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=hv.EUCLIDEAN)
    gix.set_simhash()
    allowed = np.sort(rng.choice(np.arange(n, dtype=np.uint64), 2000, replace=False))
    q = rng.standard_normal((5, dim)).astype(np.float32)
    ids, sc, cnt, st, rs, _ = gix.search_restricted_batch_params(q, hv.RestrictedParams.new(400, 400), allowed, want_stats=True)
    assert all(r["strategy"] == hv.RESTRICTED_EXACT for r in rs)
    for i in range(5):
        rc, oid, osc = oix.flat(q[i], 400, allowed=allowed)
        assert cnt[i] == 400 and ids[i].tolist() == oid.tolist() and bits(sc[i]).tolist() == bits(osc).tolist()
    with pytest.raises(hv.HelixDbError) as e:
        gix.search_restricted_batch_params(q, hv.RestrictedParams.new(400, 400, strategy=hv.RESTRICTED_FILTERED), allowed)
    assert e.value.status == hv.ERR_UNSUPPORTED


def test_shard_step_joins_the_exchange_when_the_local_search_fails(orc, hv):
    """ADVICE r3 / VERDICT r3 weak #8: a rank whose local search fails must still enter the all-gather (the others would block in it).
    On a one-rank group with a real RCCL communicator: a restricted step whose local plan fails (FILTERED without SimHash rows) returns
    the local status AFTER the exchange ran, and a device-output step that fails locally (non-strict params without SimHash rows)
    leaves HVX_SHARD_RANK_FAILED | status in every query's merged status word with count 0."""
    import torch
    rng = np.random.default_rng(4400)
    n, dim, b, k = 3000, 128, 8, 10
    data = rng.standard_normal((n, dim)).astype(np.float32)
    oix = build_oracle(orc, data, orc.L2SQ, fx.draw_levels(n, 16, seed=4), efc=60)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=hv.EUCLIDEAN, max_batch=16)   # NO SimHash rows
    grp = hv.ShardGroup(gix, hv.ShardGroup.unique_id(), 0, 1, 16, 16)
    q = rng.standard_normal((b, dim)).astype(np.float32)
    allowed = np.sort(rng.choice(np.arange(n, dtype=np.uint64), 900, replace=False))
    with pytest.raises(hv.HelixDbError) as e:
        grp.search_restricted_batch(q, hv.RestrictedParams.new(k, 100, strategy=hv.RESTRICTED_FILTERED), allowed)
    assert e.value.status == hv.ERR_INVARIANT and "SimHash" in str(e.value)
    dev = torch.device("cuda", 0)
    dq = torch.from_numpy(q).to(dev)
    g = (torch.zeros(b, k, dtype=torch.int64, device=dev), torch.zeros(b, k, dtype=torch.float32, device=dev),
         torch.full((b,), 5, dtype=torch.int32, device=dev), torch.zeros(b, dtype=torch.int32, device=dev))
    with pytest.raises(hv.HelixDbError):
        grp.search_batch_params_device(dq, hv.SearchParams.new(k), *g)        # needs SimHash rows: fails locally, exchange still runs
    gix.sync()
    assert (g[3].cpu().numpy() & 0x100).all() and not g[2].cpu().numpy().any()
    # the group is still usable afterwards
    w = gix.search_batch(q, hv.SearchParams(k).with_ef(64))
    g2 = (torch.zeros(b, k, dtype=torch.int64, device=dev), torch.zeros(b, k, dtype=torch.float32, device=dev),
          torch.zeros(b, dtype=torch.int32, device=dev), torch.zeros(b, dtype=torch.int32, device=dev))
    grp.search_batch_device(dq, k, 64, *g2)
    gix.sync()
    assert g2[0].cpu().numpy().astype(np.uint64).tolist() == w[0].tolist() and not g2[3].cpu().numpy().any()
    grp.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_pair_kernel_on_duplicates_ties_rejected_queries_and_bf16(orc, hv, dtype):
    """The owner / gatherer kernel where its speculation is most exposed: duplicate vectors (equal scores decide the next pop by id:
    the prediction misses and the pass is discarded), ties beyond the beam slack (the queries join the re-run list and come back
    from the one-wavefront kernel), a rejected query in the batch (its workgroup leaves before any barrier), k > ef / 2, a batch
    larger than the CU count, bf16 rows.  Default selector: a handle with one query per SIMD takes the pair kernel."""
    rng = np.random.default_rng(4500)
    base = rng.standard_normal((60, 128)).astype(np.float32)
    data = np.repeat(base, 40, axis=0)[rng.permutation(2400)]
    data[::7] += np.float32(0.3) * rng.standard_normal((data[::7].shape[0], 128)).astype(np.float32)
    n = data.shape[0]
    stored = fx.round_bf16(data) if dtype == "bf16" else data
    oix = build_oracle(orc, stored, orc.L2SQ, fx.draw_levels(n, 16, seed=12), efc=80)
    ex = oix.export()
    ex["vectors"] = data
    gix = hv.ValidatedVectorReadIndex.from_export(ex, dim=128, metric=hv.EUCLIDEAN, dtype=hv.BF16 if dtype == "bf16" else hv.F32, max_batch=2048)
    gix.set_occupancy(1)
    q = np.concatenate([base[:20], base[20:40] + np.float32(0.02) * rng.standard_normal((20, 128)).astype(np.float32),
                        rng.standard_normal((24, 128)).astype(np.float32)])
    assert_hnsw_equal(orc, hv, oix, gix, q, 10, 64)
    assert_hnsw_equal(orc, hv, oix, gix, q, 60, 100)
    assert_hnsw_equal(orc, hv, oix, gix, q, 10, 300)    # 384-entry beam
    big = np.tile(q, (20, 1))[:1200]                    # more workgroups than the chip holds at once
    assert_hnsw_equal(orc, hv, oix, gix, big, 10, 48)
    qb = q[:8].copy()
    qb[3, 5] = np.nan
    ids, sc, cnt, _, st = gix.search_batch(qb, hv.SearchParams(10).with_ef(64), per_query_status=True)
    assert st[3] == hv.ERR_NONFINITE and cnt[3] == 0
    for i in (0, 1, 2, 4, 7):
        rc, oid, osc = oix.search(qb[i], 10, 64)
        assert ids[i, :cnt[i]].tolist() == oid.tolist() and bits(sc[i, :cnt[i]]).tolist() == bits(osc).tolist()


def test_batcher_free_with_callers_in_flight_does_not_hang(orc, hv):
    """callers that leave on shutdown must account for their slot, or a dispatcher draining that buffer spins forever
    and hvx_batcher_free hangs in join().  24 threads keep searching while the batcher is freed: every call either returns the direct
    call's rows or fails with "shutting down", and close() returns."""
    import threading
    import time
    rng = np.random.default_rng(18)
    n, dim = 1500, 128
    data = rng.standard_normal((n, dim)).astype(np.float32)
    oix = build_oracle(orc, data, 1, fx.draw_levels(n, 16, seed=6), efc=60)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=1, max_batch=32)
    q = rng.standard_normal((64, dim)).astype(np.float32)
    want_ids, _, want_cnt, _ = gix.search_batch(q, hv.SearchParams(10).with_ef(64))
    for _round in range(3):
        bt = hv.Batcher(gix, hv.SearchParams(10).with_ef(64), max_batch=32)
        lk = threading.Lock()
        state = {"closing": False}
        wrong, done = [], [0]

        def worker(t):
            i = t
            while True:
                with lk:              # no call STARTS once the batcher is being freed (calling into a freed handle is the host's bug);
                    if state["closing"]:   # calls already inside it are the ones this test is about
                        return
                try:
                    got = bt.search(q[i % 64])
                    if [r.entity_id for r in got] != want_ids[i % 64, :want_cnt[i % 64]].tolist():
                        wrong.append(i)
                    done[0] += 1
                except hv.HelixDbError as e:
                    if "shutting down" not in str(e):
                        wrong.append(str(e))
                    return
                i += 24

        th = [threading.Thread(target=worker, args=(t,)) for t in range(24)]
        [t.start() for t in th]
        time.sleep(0.05)

        def close_it():
            with lk:
                state["closing"] = True
            bt.close()

        closer = threading.Thread(target=close_it)
        closer.start()
        closer.join(timeout=30)
        assert not closer.is_alive(), "hvx_batcher_free hangs while callers are blocked"
        [t.join(timeout=30) for t in th]
        assert not any(t.is_alive() for t in th) and not wrong, wrong[:3]


def test_shard_group_bounded_wait_and_abort(orc, hv):
    """hvx_shard_group_wait / _abort (round 6, VERDICT r5 weak #10: nothing bounded a hung collective): a step that is still running when
    the time is up answers "not yet" (False) instead of parking the host, the same wait with a generous bound answers True and the results
    are there; after an abort the group refuses further steps loudly until it is re-formed."""
    import torch
    rng = np.random.default_rng(12)
    n, dim, b, k = 200000, 256, 512, 10
    data = rng.standard_normal((n, dim)).astype(np.float32)
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64), vectors=data,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), entry_point=0, max_batch=b)
    dev = torch.device("cuda")
    try:
        grp = hv.ShardGroup(gix, hv.ShardGroup.unique_id(), 0, 1, b, k)   # one rank, WITH a communicator: the collective really runs
    except hv.HelixDbError:
        pytest.skip("RCCL is not loadable on this box")
    dq = torch.from_numpy(rng.standard_normal((b, dim)).astype(np.float32)).to(dev)
    ids = torch.zeros(b, k, dtype=torch.int64, device=dev); sc = torch.zeros(b, k, dtype=torch.float32, device=dev)
    cnt = torch.zeros(b, dtype=torch.int32, device=dev); st = torch.zeros(b, dtype=torch.int32, device=dev)
    assert grp.wait(0) is True                                           # nothing enqueued yet
    for _ in range(20):                                                  # a queue of exact scans: milliseconds of device work
        grp.flat_search_batch_device(dq, k, ids, sc, cnt, st)
    assert grp.wait(0) is False                                          # bounded: the host gets its answer at once
    assert grp.wait(60000) is True
    want = gix.flat_search_batch(dq.cpu().numpy(), k)
    assert ids.cpu().numpy().astype(np.uint64).tolist() == want[0].tolist()
    grp.abort()
    with pytest.raises(hv.HelixDbError) as e:
        grp.flat_search_batch_device(dq, k, ids, sc, cnt, st)
    assert e.value.status == hv.ERR_DEVICE
    grp.close()
    gix.close()
