"""Delete of nodes from a live device image (hvx_index_delete_batch, csrc/hvx_delete.hip; VERDICT r4 missing #4) against the oracle's
restatement of the reference's delete path (orc_index_delete == mutation.rs:1606-2055: stage_delete_with_metadata, delete_from_layer,
remove_edge_from_neighbor, relink_neighbor)."""
import numpy as np
import pytest

import fixtures as fx
from test_oracle_delete import rows_by_id

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hv():
    import pyhvx
    pyhvx.lib()
    return pyhvx


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def graph_rows(g, ids, skip=()):
    """hvx_index_export_graph (one entry per row SLOT, deleted ones included with empty rows) -> {id: (level, l0 row, [upper rows])} of
    the live nodes"""
    out, r = {}, 0
    skip = set(skip)
    for t, nid in enumerate(ids.tolist()):
        lv = int(g["level"][t])
        l0 = g["l0_neighbors"][int(g["l0_offsets"][t]):int(g["l0_offsets"][t + 1])].tolist()
        up = [g["up_neighbors"][int(g["up_offsets"][r + l]):int(g["up_offsets"][r + l + 1])].tolist() for l in range(lv)]
        r += lv
        if nid in skip:
            assert l0 == [] and all(u == [] for u in up), f"deleted node {nid} still holds neighbour rows"
            continue
        out[nid] = (lv, l0, up)
    return out


def assert_same_graph(gix, oix, ids, deleted):
    ex = oix.export()
    want = rows_by_id(ex)
    got = graph_rows(gix.export_graph(), ids, deleted)
    assert set(got) == set(want)
    bad = [nid for nid in want if got[nid] != want[nid]]
    assert not bad, f"{len(bad)} nodes differ, first {bad[:4]}: device {got[bad[0]]} oracle {want[bad[0]]}"
    g = gix.export_graph()
    ent = oix.entry()
    if ent is None:
        assert g["entry_point"] is None
    else:
        assert (g["entry_point"], g["max_layer"]) == ent


@pytest.mark.parametrize("n,dim,metric,m,m0,efc,kern,relink", [(1600, 128, 1, 16, 32, 100, "avx_fma", "steps"), (1200, 256, 0, 16, 32, 80, "avx_fma", "steps"),
                                                               (1000, 72, 2, 8, 16, 60, "avx_fma", "steps"), (1200, 768, 1, 16, 32, 100, "avx_fma", "steps"),
                                                               (900, 88, 1, 8, 16, 60, "neon", "one_wavefront"),
                                                               (1100, 128, 0, 16, 32, 80, "avx_fma", "two_launch_steps")])
def test_sequential_deletes_equal_the_oracles_row_for_row(orc, hv, n, dim, metric, m, m0, efc, kern, relink):
    """VectorIndex::delete one node at a time (mutation.rs:1606-1774): after three batches of deletes (scattered ids, the entry point
    twice, ids that are unknown, an id twice in one batch) every layer-0 row, every upper row, the entry point and the top layer of
    the live nodes equal the oracle's; deleted ids are gone from the HNSW searches (strict + the production default arm), the exact
    scan, restricted candidate sets and hvx_index_contains; the statistics count what happened; a fork adopts the owner's generation at its next launch
    (in-place rewrites cannot be hidden from it); rows appended AFTER the deletes link exactly as the oracle's do.  Every relink path of the device (one fused launch per
    source with all its prunes' distance matrices evaluated up front; two launches per source; the one-wavefront kernel with the lazy
    select_diverse) produces the same rows."""
    ok, hk = {"avx_fma": (orc.K_AVX_FMA, hv.KERNEL_AVX_FMA), "neon": (orc.K_NEON, hv.KERNEL_NEON)}[kern]
    rng = np.random.default_rng(4200 + dim + metric + n)
    n_all = n + 120
    data = rng.standard_normal((n_all, dim)).astype(np.float32)
    lv = fx.draw_levels(n_all, m, seed=n + 3)
    ids = np.arange(n_all, dtype=np.uint64) * 3 + 7
    gix, st = hv.ValidatedVectorReadIndex.build(dim=dim, metric=metric, node_ids=ids[:n], vectors=data[:n], levels=lv[:n], m=m, m0=m0,
                                                ef_construction=efc, sequential=True, float_kernel=hk, reserve_rows=120,
                                                reserve_upper_rows=int(lv[n:].sum()))
    oix = orc.Index(dim, metric, kernel=ok, m=m, m0=m0, ef_construction=efc)
    for i in range(n):
        assert oix.insert(int(ids[i]), data[i], int(lv[i])) == orc.OK
    assert_same_graph(gix, oix, ids[:n], ())
    if relink == "one_wavefront":
        gix.set_option(hv.OPT_DELETE_SEQUENTIAL, 1)
    elif relink == "two_launch_steps":   # the steps rows wider than 34 ids take (default: one fused launch per relinked row)
        gix.set_option(hv.OPT_DELETE_SEQUENTIAL, 2)
    gix.set_simhash()
    lane = gix.fork()
    seq0 = gix.visible_seq()
    perm = rng.permutation(n)
    deleted = []
    batches = [perm[: n // 20], perm[n // 20: n // 8], perm[n // 8: n // 5]]
    for bi, b in enumerate(batches):
        want = [int(x) for x in ids[b] if int(x) not in deleted]               # (an earlier batch may have taken one of them as its entry point)
        ent = oix.entry()[0]
        if ent not in want and ent not in deleted:
            want.insert(len(want) // 2, ent)                                  # the entry point itself (mutation.rs:1756-1767)
        extra = [10 ** 12 + bi, want[0]]                                       # an unknown id; an id already deleted by this batch
        if deleted:
            extra.append(deleted[0])                                           # ... and one deleted by an earlier batch
        moves = 0
        for d in want:
            moves += int(oix.entry() is not None and oix.entry()[0] == d)
            assert oix.delete(d) == (orc.OK, True)
        stb = gix.delete_batch(np.asarray(want + extra, np.uint64))
        deleted += want
        assert stb["requested"] == len(want) + len(extra) and stb["deleted"] == len(want) and stb["missing"] == len(extra)
        assert stb["entry_moves"] == moves and stb["relinked_rows"] > 0
        assert gix.live_rows() == n - len(deleted) == oix.count and gix.rows() == n and gix.visible_seq() == seq0 + bi + 1
        assert_same_graph(gix, oix, ids[:n], deleted)
    dset = set(deleted)
    assert not gix.contains(deleted[0]) and gix.contains(int(ids[perm[-1]])) and not gix.contains(5)
    audit = gix.audit_graph()
    assert audit["out_of_range_ids"] == 0 and audit["unsorted_entries"] == 0 and audit["self_loops"] == 0 and audit["nodes"] == n - len(deleted)
    assert audit["degree_overflow_rows"] == 0
    # searches: ids and score bits equal the oracle's on the relinked graph, nothing deleted comes back
    oix.set_simhash(42)
    q = np.vstack([rng.standard_normal((20, dim)).astype(np.float32), data[[int(x) for x in perm[:4]]]])   # four queries ARE deleted vectors
    gid, gsc, gcnt, _ = gix.search_batch(q, hv.SearchParams(10).with_ef(64))
    pid, psc, pcnt, _ = gix.search_batch(q, hv.SearchParams.new(10))
    fid, fsc, fcnt, _ = gix.flat_search_batch(q, 10)
    op = orc.SearchParams.new(10)
    allowed = np.asarray(sorted(set(int(x) for x in ids[rng.permutation(n)[: n // 3]]) | set(deleted[:30])), np.uint64)
    rid, rsc, rcnt = gix.search_restricted_batch(q, hv.SearchParams(10), hv.RestrictedVectorCandidates.from_ids(allowed))
    for qi in range(q.shape[0]):
        rc, oid, osc = oix.search(q[qi], 10, 64)
        assert gid[qi, :gcnt[qi]].tolist() == oid.tolist() and bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
        rc, oid, osc = oix.search_params(q[qi], op)
        assert pid[qi, :pcnt[qi]].tolist() == oid.tolist() and bits(psc[qi, :pcnt[qi]]).tolist() == bits(osc).tolist()
        rc, tid, tsc = oix.flat(q[qi], 10)
        assert fid[qi, :fcnt[qi]].tolist() == tid.tolist() and bits(fsc[qi, :fcnt[qi]]).tolist() == bits(tsc).tolist()
        rc, tid, tsc = oix.flat(q[qi], 10, allowed=allowed)
        assert rid[qi, :rcnt[qi]].tolist() == tid.tolist() and bits(rsc[qi, :rcnt[qi]]).tolist() == bits(tsc).tolist()
        assert not (set(gid[qi, :gcnt[qi]].tolist()) | set(fid[qi, :fcnt[qi]].tolist()) | set(rid[qi, :rcnt[qi]].tolist())) & dset
    # the fork: deletes rewrite shared rows in place, so its old generation cannot be served any more (its entry point may be a node
    # whose rows are now empty: ADVICE r5) -- it still SHOWS the old generation until it is used, and adopts the owner's at its next
    # launch without an explicit hvx_index_refresh (round 6)
    assert lane.live_rows() == n and lane.visible_seq() == seq0
    lid, lsc, lcnt, _ = lane.search_batch(q, hv.SearchParams(10).with_ef(64))
    assert lid.tolist() == gid.tolist() and bits(lsc).tolist() == bits(gsc).tolist()
    assert lane.live_rows() == n - len(deleted) and lane.visible_seq() == gix.visible_seq()
    lane.refresh()
    assert lane.live_rows() == n - len(deleted)
    lid, lsc, lcnt, _ = lane.search_batch(q, hv.SearchParams(10).with_ef(64))
    assert lid.tolist() == gid.tolist() and bits(lsc).tolist() == bits(gsc).tolist()
    fl_id, fl_sc, fl_cnt, _ = lane.flat_search_batch(q, 10)
    assert fl_id.tolist() == fid.tolist()
    with pytest.raises(hv.HelixDbError):
        lane.delete_batch(ids[:1])                                            # a fork does not own the image
    lane.close()
    # rows appended after the deletes: VectorIndex::insert on the relinked graph
    for i in range(n, n_all):
        assert oix.insert(int(ids[i]), data[i], int(lv[i])) == orc.OK
    gix.insert_batch(ids[n:], data[n:], lv[n:], ef_construction=efc, sequential=True)
    assert gix.rows() == n_all and gix.live_rows() == n_all - len(deleted)
    assert_same_graph(gix, oix, ids, deleted)
    gid, gsc, gcnt, _ = gix.search_batch(q, hv.SearchParams(10).with_ef(64))
    fid, fsc, fcnt, _ = gix.flat_search_batch(q, 10)
    for qi in range(q.shape[0]):
        rc, oid, osc = oix.search(q[qi], 10, 64)
        assert gid[qi, :gcnt[qi]].tolist() == oid.tolist() and bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
        rc, tid, tsc = oix.flat(q[qi], 10)
        assert fid[qi, :fcnt[qi]].tolist() == tid.tolist()
    # ... and a few of the new rows deleted again
    again = [int(x) for x in ids[n + 5: n + 25]]
    for d in again:
        assert oix.delete(d) == (orc.OK, True)
    assert gix.delete_batch(np.asarray(again, np.uint64))["deleted"] == len(again)
    assert_same_graph(gix, oix, ids, deleted + again)


def test_deletes_on_an_imported_image_the_walk_the_prefilter_and_the_batcher(orc, hv):
    """An image imported from persisted rows (hvx_index_import layout, no spare capacity) takes deletes too.  Afterwards the restricted
    filter-aware walk (SimHash directory rebuilt without the deleted rows: mutation.rs:1718-1722) equals the oracle's walk counter for
    counter, the fused prefilter path maps deleted ids to nothing, a batcher's lanes serve the new generation after hvx_batcher_refresh,
    and deleting EVERY node leaves an index that answers with no rows."""
    rng = np.random.default_rng(77)
    n, dim, m, m0, efc = 2000, 128, 16, 32, 100
    data = rng.standard_normal((n, dim)).astype(np.float32)
    lv = fx.draw_levels(n, m, seed=5)
    ids = np.arange(n, dtype=np.uint64) + 100
    oix = orc.Index(dim, orc.L2SQ, m=m, m0=m0, ef_construction=efc)
    for i in range(n):
        assert oix.insert(int(ids[i]), data[i], int(lv[i])) == orc.OK
    gix = hv.ValidatedVectorReadIndex.from_export(oix.export(), dim=dim, metric=1, max_batch=64)
    gix.set_simhash()
    oix.set_simhash(42)
    bt = hv.Batcher(gix, hv.SearchParams(10).with_ef(64), max_batch=32, max_wait_us=500)
    dels = [int(x) for x in ids[rng.permutation(n)[:300]]]
    for d in dels:
        assert oix.delete(d)[0] == orc.OK
    st = gix.delete_batch(np.asarray(dels, np.uint64))
    assert st["deleted"] == 300 and st["missing"] == 0 and st["seconds"] > 0
    assert_same_graph(gix, oix, ids, dels)
    assert gix.get_simhash().shape[0] == n                       # (SimHash rows stay per row slot on the device)
    # the planned restricted search: a group big enough for the filter-aware walk, deleted ids among the candidates
    allowed = np.asarray(sorted(set(int(x) for x in ids[rng.permutation(n)[:900]]) | set(dels[:100])), np.uint64)
    q = rng.standard_normal((8, dim)).astype(np.float32)
    rp = hv.RestrictedParams.new(10, 100)
    cand = hv.RestrictedVectorCandidates.from_ids(allowed)
    wid, wsc, wcnt, wst, wrs = gix.search_restricted_batch_params(q, rp, cand)
    assert wrs[0]["strategy"] == hv.RESTRICTED_FILTERED
    for qi in range(q.shape[0]):
        rc, oid, osc, ost = oix.search_restricted(q[qi], 10, 100, allowed)
        assert rc == orc.OK and wid[qi, :wcnt[qi]].tolist() == oid.tolist() and bits(wsc[qi, :wcnt[qi]]).tolist() == bits(osc).tolist()
        assert wrs[qi] == ost, (wrs[qi], ost)        # every RestrictedSearchStats counter and the termination reason
        assert not set(wid[qi, :wcnt[qi]].tolist()) & set(dels)
    # fused prefilter: a graph whose hop reaches deleted nodes as well
    n_nodes = n + 200
    e = 9000
    src = np.sort(rng.integers(0, n_nodes, e)); tgt = rng.integers(0, n_nodes, e).astype(np.uint64)
    off = np.zeros(n_nodes + 1, np.uint64); np.add.at(off, src + 1, 1); off = np.cumsum(off).astype(np.uint64)
    g = hv.Graph(n_nodes, off, tgt, rng.integers(0, 3, e).astype(np.uint32))
    seeds = np.arange(100, 400)
    pids, psc, pcnt, ncand, _ = gix.prefilter_search_batch(g, q, hv.SearchParams(10), seeds)
    words = g.expand(seeds, hv.DIR_OUT, ())
    c2 = hv.RestrictedVectorCandidates.from_bitmap_words(words)
    assert ncand == len(c2) and set(c2.ids.tolist()) & set(dels)
    for qi in range(q.shape[0]):
        rc, oid, osc = oix.flat(q[qi], 10, allowed=c2.ids)
        assert pids[qi, :pcnt[qi]].tolist() == oid.tolist() and bits(psc[qi, :pcnt[qi]]).tolist() == bits(osc).tolist()
    # the batcher created before the deletes
    bt.refresh()
    for qi in range(q.shape[0]):
        res = bt.search(q[qi])
        rc, oid, osc = oix.search(q[qi], 10, 64)
        assert [r_.entity_id for r_ in res] == oid.tolist()
    bt.close()
    # everything goes
    rest = [int(x) for x in ids.tolist() if int(x) not in set(dels)]
    for d in rest:
        assert oix.delete(d)[0] == orc.OK
    st = gix.delete_batch(np.asarray(rest, np.uint64))
    assert st["deleted"] == len(rest) and gix.live_rows() == 0 and oix.count == 0 and oix.entry() is None
    gid, gsc, gcnt, _ = gix.search_batch(q, hv.SearchParams(10).with_ef(64))
    assert gcnt.sum() == 0
    fid, fsc, fcnt, _ = gix.flat_search_batch(q, 10)
    assert fcnt.sum() == 0


def test_upserts_equal_the_oracles_delete_plus_insert(orc, hv):
    """VectorInsertContract::Upsert (index.rs:2018-2060: an insert of an id the index holds deletes it first): new vectors for live
    ids, for an id deleted earlier, and for a fresh id above the image -- rows, entry point and searches equal the oracle's
    delete + insert of the same ids with the same levels (a revived node keeps the level of its slot).  Refused without side
    effects: an invalid vector anywhere in the batch, an id between the image's ids that never had a slot."""
    rng = np.random.default_rng(515)
    n, dim, m, m0, efc = 1200, 128, 16, 32, 100
    data = rng.standard_normal((n + 40, dim)).astype(np.float32)
    lv = fx.draw_levels(n + 1, m, seed=77)
    ids = np.arange(n + 1, dtype=np.uint64) * 4 + 10
    gix, _ = hv.ValidatedVectorReadIndex.build(dim=dim, metric=1, node_ids=ids[:n], vectors=data[:n], levels=lv[:n], m=m, m0=m0, ef_construction=efc,
                                               sequential=True, reserve_rows=4, reserve_upper_rows=int(lv[n]) + 1)
    oix = orc.Index(dim, orc.L2SQ, m=m, m0=m0, ef_construction=efc)
    for i in range(n):
        assert oix.insert(int(ids[i]), data[i], int(lv[i])) == orc.OK
    gone = int(ids[321])
    assert oix.delete(gone) == (orc.OK, True) and gix.delete_batch([gone])["deleted"] == 1
    before = gix.export_graph()
    with pytest.raises(hv.HelixDbError) as e:                                  # NaN in the second vector: nothing of the batch happens
        gix.upsert_batch(ids[[5, 6]], np.vstack([data[n], data[n + 1] * np.float32(np.nan)]))
    assert e.value.status == hv.ERR_NONFINITE
    with pytest.raises(hv.HelixDbError) as e:                                  # an id in a gap of the image's ids: no slot to put it in
        gix.upsert_batch([int(ids[10]) + 1], data[n:n + 1])
    assert e.value.status == hv.ERR_UNSUPPORTED
    after = gix.export_graph()
    assert all(np.array_equal(before[k_], after[k_]) for k_ in ("l0_offsets", "l0_neighbors", "up_offsets", "up_neighbors")) and gix.live_rows() == n - 1
    # live ids (the entry point among them), the deleted id, a fresh id
    ent = oix.entry()[0]
    targets = [int(x) for x in ids[rng.permutation(n)[:26]] if int(x) not in (gone, ent)][:20] + [ent, gone, int(ids[n])]
    level_of = {int(ids[i]): int(lv[i]) for i in range(n + 1)}
    for t, nid in enumerate(targets):
        if oix.is_live(nid):
            assert oix.delete(nid) == (orc.OK, True)
        assert oix.insert(nid, data[n + 1 + t], level_of[nid]) == orc.OK
    st = gix.upsert_batch(np.asarray(targets, np.uint64), data[n + 1: n + 1 + len(targets)], [level_of[x] for x in targets], ef_construction=efc)
    assert st["nodes"] == len(targets) and gix.live_rows() == n + 1 == oix.count and gix.rows() == n + 1
    assert_same_graph(gix, oix, ids, ())
    assert all(gix.contains(x) for x in targets)
    q = np.vstack([rng.standard_normal((12, dim)).astype(np.float32), data[n + 1: n + 5]])   # four queries ARE upserted vectors
    gid, gsc, gcnt, _ = gix.search_batch(q, hv.SearchParams(10).with_ef(64))
    fid, fsc, fcnt, _ = gix.flat_search_batch(q, 10)
    for qi in range(q.shape[0]):
        rc, oid, osc = oix.search(q[qi], 10, 64)
        assert gid[qi, :gcnt[qi]].tolist() == oid.tolist() and bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
        rc, tid, tsc = oix.flat(q[qi], 10)
        assert fid[qi, :fcnt[qi]].tolist() == tid.tolist() and bits(fsc[qi, :fcnt[qi]]).tolist() == bits(tsc).tolist()
    assert fid[12, 0] == targets[0] and fsc[12, 0] == 0.0                       # the new vector answers under the old id


def test_an_upsert_into_a_slot_refreshes_the_caches_of_the_matrix_core_scan(orc, hv):
    """The exact scan on the matrix cores keeps |x|^2 per row (per handle) and a bf16 shadow of the f32 rows (per image).  A vector
    replaced IN PLACE by an upsert must reach both before the next scan: the 64-query streaming kernel reads candidates from the
    shadow, and its certificate is computed from the row norms (found by bench.py's upsert phase: 0 of 64 new vectors came back
    as their ids' nearest neighbours before the epoch on the image existed)."""
    rng = np.random.default_rng(99)
    n, dim = 40000, 256
    data = rng.standard_normal((n, dim)).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64) + 1
    gix, _ = hv.ValidatedVectorReadIndex.build(dim=dim, metric=1, node_ids=ids, vectors=data, levels=fx.draw_levels(n, 16, seed=4), m=16, m0=32,
                                               ef_construction=100, search_max_batch=1024)
    warm = data[rng.integers(0, n, 1024)] + np.float32(0.01)
    gix.flat_search_batch(warm, 10)                      # 1 024 queries: the tile kernels build the shadow and the row norms
    rows_at = [5, 77, 4000, n - 1]
    newv = (rng.standard_normal((4, dim)) * 3).astype(np.float32)
    gix.upsert_batch(ids[rows_at], newv)
    rows = data.copy()
    rows[rows_at] = newv
    q = np.vstack([newv, rng.standard_normal((60, dim)).astype(np.float32)])
    fid, fsc, fcnt, _ = gix.flat_search_batch(q, 10)     # 64 queries: the one-pass streaming kernel over the shadow
    for qi in range(q.shape[0]):
        rc, oid, osc = orc.flat_matrix(orc.L2SQ, rows, q[qi], 10, kernel=orc.K_AVX_FMA_HW)
        assert rc == orc.OK and fid[qi, :fcnt[qi]].tolist() == (oid.astype(np.int64) + 1).tolist() and bits(fsc[qi, :fcnt[qi]]).tolist() == bits(osc).tolist()
    assert fid[:4, 0].tolist() == ids[rows_at].tolist() and (fsc[:4, 0] == 0).all()


@pytest.mark.parametrize("n,dim,metric", [(1400, 128, 1), (1000, 256, 0), (700, 768, 1)])
def test_deletes_on_a_bf16_image_equal_the_oracle_on_the_rounded_rows(orc, hv, n, dim, metric):
    """BASELINE config #4's storage (rows rounded to bf16 once at import, interleaved device layout): hvx_index_delete_batch relinks on
    the resident bf16 rows -- the ranking and every prune's distance matrix in f32 on the rounded values, in the reference's summation
    order -- so rows, entry point and top layer equal the oracle's deletes on the ROUNDED vectors after two batches (scattered ids and
    the entry point); the searches (strict arm + production default), the exact scan and a restricted scan equal the oracle's bit for
    bit afterwards.  Paths that read f32 rows refuse a bf16 image loudly (one-wavefront relinks, two-launch steps)."""
    rng = np.random.default_rng(9100 + dim + metric)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    rounded = fx.round_bf16(data)
    lv = fx.draw_levels(n, 16, seed=n + dim)
    ids = np.arange(n, dtype=np.uint64) * 3 + 7
    oix = orc.Index(dim, metric, kernel=orc.K_AVX_FMA, m=16, m0=32, ef_construction=80)
    for i in range(n):
        assert oix.insert(int(ids[i]), rounded[i], int(lv[i])) == orc.OK
    ex = oix.export()
    assert ex["node_ids"].tolist() == ids.tolist()
    ex["vectors"] = data                                                       # the device does the rounding
    gix = hv.ValidatedVectorReadIndex.from_export(ex, dim=dim, metric=metric, dtype=hv.BF16, m=16, m0=32)
    assert_same_graph(gix, oix, ids, ())
    gix.set_simhash()
    perm = rng.permutation(n)
    deleted = []
    for bi, b in enumerate([perm[: n // 16], perm[n // 16: n // 6]]):
        want = [int(x) for x in ids[b] if int(x) not in deleted]
        ent = oix.entry()[0]
        if ent not in want and ent not in deleted:
            want.insert(len(want) // 3, ent)
        moves = 0
        for d in want:
            moves += int(oix.entry() is not None and oix.entry()[0] == d)
            assert oix.delete(d) == (orc.OK, True)
        stb = gix.delete_batch(np.asarray(want + [10 ** 12 + bi], np.uint64))
        deleted += want
        assert stb["deleted"] == len(want) and stb["missing"] == 1 and stb["entry_moves"] == moves and stb["relinked_rows"] > 0
        assert gix.live_rows() == n - len(deleted) == oix.count
        assert_same_graph(gix, oix, ids, deleted)
    dset = set(deleted)
    oix.set_simhash(42)
    q = np.vstack([rng.standard_normal((16, dim)).astype(np.float32), data[[int(x) for x in perm[:4]]]])
    gid, gsc, gcnt, _ = gix.search_batch(q, hv.SearchParams(10).with_ef(64))
    pid, psc, pcnt, _ = gix.search_batch(q, hv.SearchParams.new(10))
    fid, fsc, fcnt, _ = gix.flat_search_batch(q, 10)
    allowed = np.asarray(sorted(set(int(x) for x in ids[rng.permutation(n)[: n // 3]]) | set(deleted[:20])), np.uint64)
    rid, rsc, rcnt = gix.search_restricted_batch(q, hv.SearchParams(10), hv.RestrictedVectorCandidates.from_ids(allowed))
    op = orc.SearchParams.new(10)
    for qi in range(q.shape[0]):
        rc, oid, osc = oix.search(q[qi], 10, 64)
        assert gid[qi, :gcnt[qi]].tolist() == oid.tolist() and bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
        rc, oid, osc = oix.search_params(q[qi], op)
        assert pid[qi, :pcnt[qi]].tolist() == oid.tolist() and bits(psc[qi, :pcnt[qi]]).tolist() == bits(osc).tolist()
        rc, tid, tsc = oix.flat(q[qi], 10)
        assert fid[qi, :fcnt[qi]].tolist() == tid.tolist() and bits(fsc[qi, :fcnt[qi]]).tolist() == bits(tsc).tolist()
        rc, tid, tsc = oix.flat(q[qi], 10, allowed=allowed)
        assert rid[qi, :rcnt[qi]].tolist() == tid.tolist() and bits(rsc[qi, :rcnt[qi]]).tolist() == bits(tsc).tolist()
        assert not (set(gid[qi, :gcnt[qi]].tolist()) | set(fid[qi, :fcnt[qi]].tolist()) | set(rid[qi, :rcnt[qi]].tolist())) & dset
    live = [int(x) for x in ids if int(x) not in dset]
    for opt in (1, 2):                                                         # the relink paths that read f32 rows
        gix.set_option(hv.OPT_DELETE_SEQUENTIAL, opt)
        with pytest.raises(hv.HelixDbError):
            gix.delete_batch(np.asarray(live[:1], np.uint64))
    gix.set_option(hv.OPT_DELETE_SEQUENTIAL, 0)
    assert gix.live_rows() == n - len(deleted)                                 # ... refused before anything changed
    assert_same_graph(gix, oix, ids, deleted)
    gix.close()


@pytest.mark.parametrize("n,dim,metric,dtype", [(700, 64, 1, "f32"), (600, 128, 0, "f32"), (500, 128, 1, "bf16")])
def test_deletes_on_an_m32_m0_64_graph_equal_the_oracle(orc, hv, n, dim, metric, dtype):
    """The reference's scale fixture runs M = 32 / M0 = 64 (scale_contracts.rs:167-173): rows of up to 64 ids, a source's merged row of up
    to 128, reciprocal lists of 65 -- the WIDE build of the fused step (two candidates per lane, 128-bit masks, matrix rows streamed).
    Rows, entry point and top layer equal the oracle's sequential deletes after two batches; searches equal the oracle's afterwards."""
    rng = np.random.default_rng(6400 + dim + metric)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    rows = fx.round_bf16(data) if dtype == "bf16" else data
    lv = fx.draw_levels(n, 32, seed=n + 1)
    ids = np.arange(n, dtype=np.uint64) * 2 + 3
    oix = orc.Index(dim, metric, kernel=orc.K_AVX_FMA, m=32, m0=64, ef_construction=120)
    for i in range(n):
        assert oix.insert(int(ids[i]), rows[i], int(lv[i])) == orc.OK
    ex = oix.export()
    assert max(np.diff(ex["l0_offsets"].astype(np.int64))) > 34           # rows the narrow build does not serve
    ex["vectors"] = data
    gix = hv.ValidatedVectorReadIndex.from_export(ex, dim=dim, metric=metric, m=32, m0=64, dtype=hv.BF16 if dtype == "bf16" else hv.F32)
    assert_same_graph(gix, oix, ids, ())
    perm = rng.permutation(n)
    deleted = []
    for bi, b in enumerate([perm[: n // 14], perm[n // 14: n // 6]]):
        want = [int(x) for x in ids[b] if int(x) not in deleted]
        ent = oix.entry()[0]
        if ent not in want and ent not in deleted:
            want.insert(len(want) // 2, ent)
        for d in want:
            assert oix.delete(d) == (orc.OK, True)
        stb = gix.delete_batch(np.asarray(want, np.uint64))
        deleted += want
        assert stb["deleted"] == len(want) and stb["relinked_rows"] > 0
        assert_same_graph(gix, oix, ids, deleted)
    q = rng.standard_normal((12, dim)).astype(np.float32)
    gid, gsc, gcnt, _ = gix.search_batch(q, hv.SearchParams(10).with_ef(64))
    for qi in range(q.shape[0]):
        rc, oid, osc = oix.search(q[qi], 10, 64)
        assert gid[qi, :gcnt[qi]].tolist() == oid.tolist() and bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
    gix.close()


@pytest.mark.parametrize("n,dim,metric", [(1200, 128, 1), (900, 256, 0)])
def test_upserts_into_a_bf16_image_equal_the_oracle_on_the_rounded_rows(orc, hv, n, dim, metric):
    """Config #4's storage takes new vectors for the ids it holds (round 6): the vector is rounded to bf16, validated, packed into the
    node's slot, and linked by the one-node steps over the interleaved bf16 rows with the rounded vector as the f32 query of its build
    search -- rows, entry point, searches and the exact scan (row norms refreshed) equal the oracle's delete + insert of the ROUNDED
    vectors.  Live ids (the entry point among them) and an id deleted earlier; a new id needs spare rows (hvx_index_import_reserve: next test)."""
    rng = np.random.default_rng(8800 + dim)
    efc = 80
    data = rng.standard_normal((n + 40, dim)).astype(np.float32)
    rounded = fx.round_bf16(data)
    lv = fx.draw_levels(n, 16, seed=n + 9)
    ids = np.arange(n, dtype=np.uint64) * 4 + 10
    oix = orc.Index(dim, metric, kernel=orc.K_AVX_FMA, m=16, m0=32, ef_construction=efc)
    for i in range(n):
        assert oix.insert(int(ids[i]), rounded[i], int(lv[i])) == orc.OK
    ex = oix.export()
    ex["vectors"] = data[:n]
    gix = hv.ValidatedVectorReadIndex.from_export(ex, dim=dim, metric=metric, dtype=hv.BF16, m=16, m0=32)
    gone = int(ids[77])
    assert oix.delete(gone) == (orc.OK, True) and gix.delete_batch([gone])["deleted"] == 1
    with pytest.raises(hv.HelixDbError):                                       # a new id: this image was imported without spare rows
        gix.upsert_batch([int(ids[-1]) + 4], data[n:n + 1])
    assert gix.live_rows() == n - 1
    ent = oix.entry()[0]
    targets = [int(x) for x in ids[rng.permutation(n)[:20]] if int(x) not in (gone, ent)][:14] + [ent, gone]
    level_of = {int(ids[i]): int(lv[i]) for i in range(n)}
    for t, nid in enumerate(targets):
        if oix.is_live(nid):
            assert oix.delete(nid) == (orc.OK, True)
        assert oix.insert(nid, rounded[n + t], level_of[nid]) == orc.OK
    st = gix.upsert_batch(np.asarray(targets, np.uint64), data[n: n + len(targets)], ef_construction=efc)
    assert st["nodes"] == len(targets) and gix.live_rows() == n == oix.count
    assert_same_graph(gix, oix, ids, ())
    q = np.vstack([rng.standard_normal((10, dim)).astype(np.float32), rounded[n: n + 4]])   # four queries ARE upserted vectors
    gid, gsc, gcnt, _ = gix.search_batch(q, hv.SearchParams(10).with_ef(64))
    fid, fsc, fcnt, _ = gix.flat_search_batch(q, 10)
    for qi in range(q.shape[0]):
        rc, oid, osc = oix.search(q[qi], 10, 64)
        assert gid[qi, :gcnt[qi]].tolist() == oid.tolist() and bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
        rc, tid, tsc = oix.flat(q[qi], 10)
        assert fid[qi, :fcnt[qi]].tolist() == tid.tolist() and bits(fsc[qi, :fcnt[qi]]).tolist() == bits(tsc).tolist()
    assert fid[10, 0] == targets[0]                                              # the new vector answers under the old id
    gix.close()


@pytest.mark.parametrize("n,dim,metric", [(1000, 128, 1), (800, 256, 0)])
def test_inserts_into_a_bf16_image_with_spare_rows_equal_the_oracle(orc, hv, n, dim, metric):
    """hvx_index_import_reserve + hvx_index_insert_batch on a bf16 image: the first 60 % of the rows are hydrated (rounded at import), the
    rest arrive in three insert calls -- rounded to bf16, validated, packed behind the image's rows and linked one node at a time over the
    bf16 rows -- and an appending upsert; rows, entry point, top layer and searches equal the oracle's insertion of the ROUNDED vectors."""
    rng = np.random.default_rng(7300 + dim)
    efc = 80
    data = rng.standard_normal((n + 1, dim)).astype(np.float32)
    rounded = fx.round_bf16(data)
    lv = fx.draw_levels(n + 1, 16, seed=n + 2)
    ids = np.arange(n + 1, dtype=np.uint64) * 2 + 1
    n0 = n * 6 // 10
    oix = orc.Index(dim, metric, kernel=orc.K_AVX_FMA, m=16, m0=32, ef_construction=efc)
    for i in range(n0):
        assert oix.insert(int(ids[i]), rounded[i], int(lv[i])) == orc.OK
    ex = oix.export()
    ex["vectors"] = data[:n0]
    gix = hv.ValidatedVectorReadIndex.from_export(ex, dim=dim, metric=metric, dtype=hv.BF16, m=16, m0=32, reserve_rows=n + 1 - n0,
                                                  reserve_upper_rows=int(lv[n0:].sum()))
    cuts = [n0, n0 + (n - n0) // 3, n0 + 2 * (n - n0) // 3, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        for i in range(a, b):
            assert oix.insert(int(ids[i]), rounded[i], int(lv[i])) == orc.OK
        gix.insert_batch(ids[a:b], data[a:b], lv[a:b], ef_construction=efc)
        assert gix.live_rows() == b == oix.count
        assert_same_graph(gix, oix, ids[:b], ())
    assert oix.insert(int(ids[n]), rounded[n], int(lv[n])) == orc.OK            # a fresh id through the upsert entry point: appended
    gix.upsert_batch(ids[n:n + 1], data[n:n + 1], lv[n:n + 1], ef_construction=efc)
    assert_same_graph(gix, oix, ids, ())
    q = np.vstack([rng.standard_normal((10, dim)).astype(np.float32), rounded[n - 3: n + 1]])
    gid, gsc, gcnt, _ = gix.search_batch(q, hv.SearchParams(10).with_ef(64))
    fid, fsc, fcnt, _ = gix.flat_search_batch(q, 10)
    for qi in range(q.shape[0]):
        rc, oid, osc = oix.search(q[qi], 10, 64)
        assert gid[qi, :gcnt[qi]].tolist() == oid.tolist() and bits(gsc[qi, :gcnt[qi]]).tolist() == bits(osc).tolist()
        rc, tid, tsc = oix.flat(q[qi], 10)
        assert fid[qi, :fcnt[qi]].tolist() == tid.tolist() and bits(fsc[qi, :fcnt[qi]]).tolist() == bits(tsc).tolist()
    gix.close()
