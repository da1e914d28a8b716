"""The one-launch restricted exact scan (csrc/hvx_restricted_exact.hip, round 6) through the C ABI against the oracle's
restricted_exact_scan (restricted.rs:753-835, :661-704, :615-659; RestrictedVectorCandidates is a set: :303-371): ids and f32 score BITS,
for a candidate set shared by a batch and for every query with ITS OWN candidate list (the operator's shape,
execution/interpreter/access/search/storage.rs:140-163) -- unknown ids, duplicates, empty lists, rejected queries, deleted rows,
non-contiguous ids, every summation tree and metric, bf16 rows; plus the device plan of HVX_RESTRICTED_AUTO next to the reference's."""
import numpy as np
import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hv():
    import pyhvx
    pyhvx.lib()
    return pyhvx


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def pair(orc, hv, n, dim, metric, kernel=None, dtype="f32", seed=0, sparse_ids=False, max_batch=64):
    """an oracle index and its device image over the same rows (no graph: exact scans never touch it)"""
    rng = np.random.default_rng(seed)
    centres = rng.standard_normal((12, dim)).astype(np.float32)
    data = (centres[rng.integers(0, 12, n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
    if dtype == "bf16":
        data = fx.round_bf16(data)
    ids = (np.sort(rng.choice(np.arange(10 * n, dtype=np.uint64), n, replace=False)) if sparse_ids else np.arange(n, dtype=np.uint64)) + np.uint64(7)
    ok, hk = kernel or (orc.K_AVX_FMA, hv.KERNEL_AVX_FMA)
    oix = orc.Index(dim, metric, kernel=ok)
    off = np.zeros(n + 1, np.uint64)
    assert oix.seed(ids, data, off, np.zeros(0, np.uint64), entry_point=int(ids[0])) == orc.OK
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=metric, node_ids=ids, vectors=data, l0_offsets=off, l0_neighbors=np.zeros(0, np.uint64),
                                              entry_point=int(ids[0]), float_kernel=hk, max_batch=max_batch,
                                              dtype=hv.BF16 if dtype == "bf16" else hv.F32)
    queries = (centres[rng.integers(0, 12, 40)] + 0.3 * rng.standard_normal((40, dim))).astype(np.float32)
    return oix, gix, ids, data, queries, rng


SHAPES = [
    # (n, dim, metric, kernel pair name, dtype, k, sparse ids)
    (3000, 768, 1, "avx_fma", "f32", 10, False),     # the bench dimension, Euclidean
    (2500, 1536, 1, "avx_fma", "f32", 10, True),     # config #3's dimension, ids that need the binary search
    (2000, 128, 0, "avx_fma", "f32", 17, False),     # cosine
    (1500, 384, 1, "avx_fma", "f32", 5, False),      # 12 chunks: a load group of six
    (900, 3072, 0, "avx_fma", "f32", 10, False),     # 96 chunks
    (1200, 72, 2, "avx_fma", "f32", 10, False),      # Manhattan: sequential, the any-shape build
    (1500, 100, 1, "avx", "f32", 10, True),          # AVX without FMA, a scalar tail of 4
    (1500, 88, 0, "neon", "f32", 12, False),         # the 128-bit trees
    (1500, 40, 1, "sse", "f32", 10, False),
    (1000, 24, 1, "scalar", "f32", 64, False),       # the widest k the kernel keeps
    (2500, 768, 1, "avx_fma", "bf16", 10, False),    # config #4's rows
    (2000, 256, 0, "avx_fma", "bf16", 10, True),
    (1500, 1024, 0, "avx_fma", "bf16", 10, False),   # cosine over bf16 rows at 32 chunks: one query per tile
]


def kernels(orc, hv):
    return {"avx_fma": (orc.K_AVX_FMA, hv.KERNEL_AVX_FMA), "avx": (orc.K_AVX, hv.KERNEL_AVX), "sse": (orc.K_SSE, hv.KERNEL_SSE),
            "neon": (orc.K_NEON, hv.KERNEL_NEON), "scalar": (orc.K_SCALAR, hv.KERNEL_SCALAR)}


@pytest.mark.parametrize("n,dim,metric,kern,dtype,k,sparse", SHAPES)
def test_shared_set_equals_the_oracle_and_the_older_pipeline(orc, hv, n, dim, metric, kern, dtype, k, sparse):
    """one candidate set for the batch (the fused prefilter's shape): the one-launch kernel, forced, against the oracle's exact scan and
    against the matrix-core / VALU pipeline it replaces for small sets; batches of 1, 3 (a ragged tile) and 40 (several tiles, and more
    than max_batch would allow in one chunk when max_batch is 32)"""
    oix, gix, ids, data, q, rng = pair(orc, hv, n, dim, metric, kernels(orc, hv)[kern], dtype, seed=dim + n, sparse_ids=sparse, max_batch=32)
    picked = rng.choice(ids, n // 2, replace=False)
    allowed = np.concatenate([picked, picked[:50], np.array([1, 2, 3, 10 ** 12], np.uint64)])  # duplicates and ids that hold no vector
    rng.shuffle(allowed)
    p = hv.SearchParams(k).with_ef(max(k, 64))
    for b in (1, 3, 40):
        gix.set_option(hv.OPT_RESTRICTED_DIRECT, 2)
        ids_d, sc_d, cnt_d = gix.search_restricted_batch(q[:b], p, allowed)
        assert gix.last_scan_path() == hv.PATH_DIRECT
        gix.set_option(hv.OPT_RESTRICTED_DIRECT, 1)
        ids_o, sc_o, cnt_o = gix.search_restricted_batch(q[:b], p, allowed)
        assert gix.last_scan_path() != hv.PATH_DIRECT
        for i in range(b):
            rc, want_ids, want_sc = oix.flat(q[i], k, allowed=allowed)
            assert rc == orc.OK
            assert ids_d[i, : cnt_d[i]].tolist() == want_ids.tolist() and bits(sc_d[i, : cnt_d[i]]).tolist() == bits(want_sc).tolist()
            assert ids_o[i, : cnt_o[i]].tolist() == want_ids.tolist() and bits(sc_o[i, : cnt_o[i]]).tolist() == bits(want_sc).tolist()
    gix.close()


@pytest.mark.parametrize("n,dim,metric,kern,dtype,k,sparse", [SHAPES[0], SHAPES[1], SHAPES[2], SHAPES[5], SHAPES[6], SHAPES[10], SHAPES[12]])
def test_every_query_with_its_own_candidate_list(orc, hv, n, dim, metric, kern, dtype, k, sparse):
    """hvx_search_restricted_batch_params with allowed_offsets: ONE launch answers the batch; lists of 0 .. 1 500 ids in caller order with
    duplicates and unknown ids, fewer candidates than k, an empty list (no result and NO validation: restricted.rs:539-541), a rejected
    query (fails alone), under the device plan (AUTO) and the forced exact strategy; RestrictedSearchStats.strategy says EXACT"""
    oix, gix, ids, data, q, rng = pair(orc, hv, n, dim, metric, kernels(orc, hv)[kern], dtype, seed=7 * dim + n, sparse_ids=sparse, max_batch=16)
    b = 40
    q = q[:b].copy()
    lens = rng.integers(1, 1500, b)
    lens[3], lens[9], lens[21] = 0, 4, 1  # empty, fewer than k, a single candidate
    q[5, 0] = np.nan                      # rejected (ValidatedMetricVector::try_new)
    q[21] = np.nan                        # rejected, single candidate
    lists = []
    for i in range(b):
        own = rng.choice(ids, int(lens[i]), replace=True)  # duplicates
        if i % 4 == 0 and lens[i]:
            own = np.concatenate([own, np.array([0, 10 ** 15], np.uint64)])  # ids that hold no vector
        lists.append(own.astype(np.uint64))
    lists[3] = np.zeros(0, np.uint64)
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
    flat = np.concatenate(lists) if off[-1] else np.zeros(0, np.uint64)
    for rp in (hv.RestrictedParams.auto(k, 100), hv.RestrictedParams.new(k, 100, strategy=hv.RESTRICTED_EXACT)):
        got_ids, got_sc, got_cnt, got_st, rs, stats = gix.search_restricted_batch_params(q, rp, flat, offsets=off, want_stats=True)
        assert gix.last_scan_path() == hv.PATH_DIRECT
        assert stats["distance_computations"] == int(off[-1]) and stats["device_ms"] > 0.0
        for i in range(b):
            if len(lists[i]) == 0:
                assert got_cnt[i] == 0 and got_st[i] == 0 and rs[i]["strategy"] == 0
                continue
            rc, want_ids, want_sc = oix.flat(q[i], k, allowed=lists[i])
            if i in (5, 21):
                assert rc == orc.ERR_NONFINITE and got_st[i] == hv.ERR_NONFINITE and got_cnt[i] == 0 and rs[i]["strategy"] == 0
                continue
            assert rc == orc.OK and got_st[i] == 0 and rs[i]["strategy"] == hv.RESTRICTED_EXACT
            assert got_ids[i, : got_cnt[i]].tolist() == want_ids.tolist(), i
            assert bits(got_sc[i, : got_cnt[i]]).tolist() == bits(want_sc).tolist(), i
    # the older path (list by list) returns the same rows
    gix.set_option(hv.OPT_RESTRICTED_DIRECT, 1)
    o_ids, o_sc, o_cnt, o_st, _ = gix.search_restricted_batch_params(q, hv.RestrictedParams.new(k, 100, strategy=hv.RESTRICTED_EXACT), flat, offsets=off)
    assert o_cnt.tolist() == got_cnt.tolist() and o_st.tolist() == got_st.tolist()
    for i in range(b):
        assert o_ids[i, : o_cnt[i]].tolist() == got_ids[i, : got_cnt[i]].tolist() and bits(o_sc[i, : o_cnt[i]]).tolist() == bits(got_sc[i, : got_cnt[i]]).tolist()
    gix.close()


def test_device_plan_next_to_the_reference_plan(orc, hv):
    """HVX_RESTRICTED_AUTO = the device's plan: exact while the candidate rows take <= HVX_OPT_RESTRICTED_EXACT_MIB (default 1 GiB); the
    reference's line (256 ids / 4 MiB, restricted.rs:426-453) stays selectable and is what RestrictedParams.new asks for"""
    n, dim, k = 3000, 64, 10
    rng = np.random.default_rng(5)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    lv = fx.draw_levels(n, 16, seed=2)
    oix = orc.Index(dim, orc.L2SQ, m=16, m0=32, ef_construction=64)
    for i in range(n):
        assert oix.insert(i, data[i], int(lv[i])) == orc.OK
    oix.set_simhash(42)
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=hv.EUCLIDEAN, max_batch=32)
    gix.set_simhash()
    q = rng.standard_normal((4, dim)).astype(np.float32)
    allowed = np.arange(0, n, 2, dtype=np.uint64)  # 1 500 ids: above the reference's 256
    a = gix.search_restricted_batch_params(q, hv.RestrictedParams.auto(k, 64), allowed)
    r = gix.search_restricted_batch_params(q, hv.RestrictedParams.new(k, 64), allowed)
    assert all(s["strategy"] == hv.RESTRICTED_EXACT for s in a[4]) and all(s["strategy"] == hv.RESTRICTED_FILTERED for s in r[4])
    for i in range(4):
        rc, want_ids, want_sc = oix.flat(q[i], k, allowed=allowed)
        assert a[0][i, : a[2][i]].tolist() == want_ids.tolist() and bits(a[1][i, : a[2][i]]).tolist() == bits(want_sc).tolist()
        rc, oid, osc, ost = oix.search_restricted(q[i], k, 64, allowed)
        assert r[0][i, : r[2][i]].tolist() == oid.tolist() and bits(r[1][i, : r[2][i]]).tolist() == bits(osc).tolist()
    # the line moves with the option: 3 000 x 64 x 4 B = 750 KiB of rows stay exact under a 1 MiB limit, 3 000 rows of a limit of ... MiB is
    # the unit, so the other side of the line needs more rows than this index has -- covered by the C3 leg of bench.py (1 GiB = 174 762 rows
    # of 1536 floats); here: the option is accepted and the plan stays exact below it
    gix.set_option(hv.OPT_RESTRICTED_EXACT_MIB, 1)
    a2 = gix.search_restricted_batch_params(q, hv.RestrictedParams.auto(k, 64), np.arange(n, dtype=np.uint64))
    assert all(s["strategy"] == hv.RESTRICTED_EXACT for s in a2[4])
    gix.close()


def test_deleted_rows_and_fused_prefilter_take_the_one_launch_scan(orc, hv):
    """a deleted node is no candidate (its id maps to nothing: mutation.rs:1708-1745); the fused hop + kNN call (where_() -> vector_search)
    lands on the same kernel and the same rows"""
    n, dim, k = 2000, 128, 10
    rng = np.random.default_rng(11)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    lv = fx.draw_levels(n, 16, seed=3)
    ids = np.arange(n, dtype=np.uint64)
    gix, _ = hv.ValidatedVectorReadIndex.build(dim=dim, metric=hv.EUCLIDEAN, node_ids=ids, vectors=data, levels=lv, m=16, m0=32, ef_construction=64,
                                               sequential=True, search_max_batch=32)
    oix = orc.Index(dim, orc.L2SQ, m=16, m0=32, ef_construction=64)
    for i in range(n):
        assert oix.insert(i, data[i], int(lv[i])) == orc.OK
    gone = rng.choice(ids, 60, replace=False)
    gix.delete_batch(gone)
    for g in gone:
        assert oix.delete(int(g)) == (orc.OK, True)
    q = rng.standard_normal((8, dim)).astype(np.float32)
    lists = [rng.choice(ids, 400, replace=False).astype(np.uint64) for _ in range(8)]
    lists[2] = np.concatenate([lists[2], gone])  # deleted ids in the list
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
    got = gix.search_restricted_batch_params(q, hv.RestrictedParams.auto(k, 64), np.concatenate(lists), offsets=off)
    assert gix.last_scan_path() == hv.PATH_DIRECT
    for i in range(8):
        rc, want_ids, want_sc = oix.flat(q[i], k, allowed=lists[i])
        assert got[0][i, : got[2][i]].tolist() == want_ids.tolist() and bits(got[1][i, : got[2][i]]).tolist() == bits(want_sc).tolist()
        assert not set(got[0][i, : got[2][i]].tolist()) & set(gone.tolist())
    # fused: node i -> (i + n/2) mod n, sources 0..299 => candidates n/2 .. n/2 + 299
    tgt = ((ids + np.uint64(n // 2)) % np.uint64(n)).astype(np.uint64)
    g = hv.Graph(n, np.arange(n + 1, dtype=np.uint64), tgt)
    src = np.arange(300, dtype=np.uint64)
    f_ids, f_sc, f_cnt, ncand, rs, _ = gix.prefilter_search_batch_params(g, q, hv.RestrictedParams.auto(k, 64), src, direction=hv.DIR_OUT)
    assert ncand == 300 and gix.last_scan_path() == hv.PATH_DIRECT and all(s["strategy"] == hv.RESTRICTED_EXACT for s in rs)
    cand = np.arange(n // 2, n // 2 + 300, dtype=np.uint64)
    for i in range(8):
        rc, want_ids, want_sc = oix.flat(q[i], k, allowed=cand)
        assert f_ids[i, : f_cnt[i]].tolist() == want_ids.tolist() and bits(f_sc[i, : f_cnt[i]]).tolist() == bits(want_sc).tolist()
    gix.close()


@pytest.mark.parametrize("dtype,metric,dim,n,k,b,spread", [("f32", 1, 1536, 9000, 10, 32, 0.05), ("f32", 1, 768, 20011, 64, 7, 0.5), ("f32", 0, 512, 12000, 10, 32, 0.02),
                                                           ("f32", 1, 256, 40000, 25, 64, 0.5), ("f32", 1, 1024, 6000, 1, 1, 0.001), ("bf16", 1, 768, 20000, 10, 33, 0.1),
                                                           ("bf16", 0, 1536, 5000, 17, 32, 0.5), ("f32", 1, 256, 50000, 10, 128, 0.01)])
def test_exact_tail_equals_the_certificate_pipeline(orc, hv, dtype, metric, dim, n, k, b, spread):
    """hvx_flat_tail.hip: the small-batch contraction's approximate dot products turned into the exact answer by ONE more launch (filter by
    the shared k-th exact score, re-score in the reference's order) -- ids and score bits equal the oracle's and the selection / re-rank /
    certificate pipeline of rounds 3-5, on rows clustered so tightly (spread 0.001 - 0.05) that the one-pass bound covers whole clusters,
    with duplicate rows (ties broken by id), a rejected query, whole scans and scattered row lists."""
    rng = np.random.default_rng(dim + n + b + k)
    centers = rng.standard_normal((32, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 32, n)] + spread * rng.standard_normal((n, dim))).astype(np.float32)
    data[7] = data[3]
    data[100:140] = data[99]  # forty copies of one row: more equal scores than k
    stored = fx.round_bf16(data) if dtype == "bf16" else data
    ids = np.arange(n, dtype=np.uint64) + 11
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=metric, node_ids=ids, vectors=data, dtype=hv.BF16 if dtype == "bf16" else hv.F32,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), max_batch=max(b, 16))
    q = (centers[rng.integers(0, 32, b)] + spread * rng.standard_normal((b, dim))).astype(np.float32)
    q[0] = data[99]
    if b > 2:
        q[b - 1, 5] = np.nan
    kern = {"kernel": orc.K_AVX_FMA_HW} if dtype == "f32" else {}
    gid, gsc, gcnt, _, gst = gix.flat_search_batch(q, k, per_query_status=True)
    assert gix.last_scan_path() & hv.PATH_EXACT_TAIL and gix.last_scan_path() & hv.PATH_SMALL_BATCH
    gix.set_option(hv.OPT_FLAT_NO_TAIL, 1)
    oid_, osc_, ocnt_, _, ost_ = gix.flat_search_batch(q, k, per_query_status=True)
    assert not gix.last_scan_path() & hv.PATH_EXACT_TAIL
    gix.set_option(hv.OPT_FLAT_NO_TAIL, 0)
    assert gid.tolist() == oid_.tolist() and bits(gsc).tolist() == bits(osc_).tolist() and gcnt.tolist() == ocnt_.tolist() and gst.tolist() == ost_.tolist()
    for qi in range(b):
        if b > 2 and qi == b - 1:
            assert gst[qi] == hv.ERR_NONFINITE and gcnt[qi] == 0
            continue
        rc, oid, osc = orc.flat_matrix(metric, stored, q[qi], k, **kern)
        assert gst[qi] == 0 and (gid[qi, :gcnt[qi]] - 11).tolist() == oid.tolist(), f"query {qi}"
        assert bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
    # a scattered row list (the restricted scan's shape); twice, so that the self-cleaning counters are used again
    allowed = np.sort(rng.choice(ids, min(n - 5, max(2 ** 22 // dim + 77, 3001)), replace=False))
    good = q[: max(1, b - 1)] if b > 2 else q
    sub = stored[(allowed - 11).astype(np.int64)]
    gix.set_option(hv.OPT_RESTRICTED_DIRECT, 1)  # (small sets would take the one-launch reference-order kernel)
    for _ in range(2):
        rid, rsc, rcnt = gix.search_restricted_batch(good, hv.SearchParams(k), hv.RestrictedVectorCandidates.from_ids(allowed))
        assert gix.last_scan_path() & hv.PATH_EXACT_TAIL
        for qi in range(0, good.shape[0], max(1, good.shape[0] // 8)):
            rc, oid, osc = orc.flat_matrix(metric, sub, good[qi], k, **kern)
            assert allowed[oid.astype(np.int64)].tolist() == rid[qi, :rcnt[qi]].tolist() and bits(osc).tolist() == bits(rsc[qi, :rcnt[qi]]).tolist()
    gix.close()


def test_batching_operator_for_the_prefiltered_branch(orc, hv):
    """hvx_batcher_new_restricted / _search_restricted / _submit_restricted: 64 threads, each with ITS OWN candidate ids per call
    (storage.rs:140-163 `Some(candidates) => index.search_restricted(..)`), coalesced into launches of the one-launch exact scan; every caller
    gets the oracle's rows for ITS set; an empty list, a rejected query (fails alone), a list beyond the slot size (refused, not truncated),
    the non-blocking form, and the unrestricted entry points refusing a restricted batcher."""
    import threading
    n, dim, k = 4000, 256, 10
    oix, gix, ids, data, q, rng = pair(orc, hv, n, dim, 1, None, "f32", seed=77, sparse_ids=True, max_batch=64)
    # (a 10-ms window: the callers are Python threads -- a 200-us one left every call alone in its batch in 1 of 7 runs of the suite, and the
    # test asserts that calls ARE coalesced)
    bt = hv.RestrictedBatcher(gix, hv.RestrictedParams.auto(k, 100), max_batch=64, max_wait_us=10000, lanes=2, max_ids_per_query=1500)
    threads, per = 16, 12
    jobs = [[(rng.standard_normal(dim).astype(np.float32), rng.choice(ids, int(rng.integers(1, 1500)), replace=True).astype(np.uint64)) for _ in range(per)]
            for _ in range(threads)]
    out = [[None] * per for _ in range(threads)]
    errs = []

    def work(t):
        try:
            for i, (qq, al) in enumerate(jobs[t]):
                out[t][i] = bt.search(qq, al)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    for t in range(threads):
        for i, (qq, al) in enumerate(jobs[t]):
            rc, want_ids, want_sc = oix.flat(qq, k, allowed=al)
            got = out[t][i]
            assert [r.entity_id for r in got] == want_ids.tolist() and bits(np.array([r.score for r in got], np.float32)).tolist() == bits(want_sc).tolist()
    st = bt.stats()
    # (whether blocking Python threads meet in one batch is a matter of timing -- a lane dispatches as soon as it holds as many queries as
    # its previous batch: in 2 of 9 runs of the whole suite every call went alone; the burst of tickets below MUST coalesce)
    assert st["queries"] == threads * per and st["batches"] <= threads * per
    assert bt.search(q[0], np.zeros(0, np.uint64)) == []                      # empty set: nothing, no validation
    bad = q[1].copy(); bad[3] = np.inf
    assert bt.search(bad, np.zeros(0, np.uint64)) == []
    with pytest.raises(hv.HelixDbError) as e:
        bt.search(bad, ids[:10])
    assert e.value.status == hv.ERR_NONFINITE
    with pytest.raises(hv.HelixDbError) as e:
        bt.search(q[0], np.arange(1501, dtype=np.uint64))                    # beyond max_ids_per_query: refused
    assert e.value.status == hv.ERR_UNSUPPORTED
    tickets = []
    before = bt.stats()
    for i in range(20):
        tk = bt.submit(q[i], ids[i * 50:(i + 1) * 50 + 500])
        assert tk is not None
        tickets.append(tk)
    for i, tk in enumerate(tickets):
        got = bt.wait(tk)
        rc, want_ids, want_sc = oix.flat(q[i], k, allowed=ids[i * 50:(i + 1) * 50 + 500])
        assert [r.entity_id for r in got] == want_ids.tolist()
    after = bt.stats()
    assert after["queries"] - before["queries"] == 20 and after["batches"] - before["batches"] < 20   # twenty tickets in a burst: coalesced
    with pytest.raises(hv.HelixDbError):
        hv.Batcher.search(bt, q[0])                                           # the unrestricted entry point refuses this batcher
    with pytest.raises(hv.HelixDbError):                                      # the reference plan walks above 256 ids: not batched per caller
        hv.RestrictedBatcher(gix, hv.RestrictedParams.new(k, 100), max_batch=64, max_ids_per_query=1500)
    bt.close()
    gix.close()
