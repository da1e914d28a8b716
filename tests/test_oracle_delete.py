"""The oracle's restatement of the reference's delete path (oracle/hvx_oracle.c orc_index_delete == mutation.rs:1606-2055
stage_delete_with_metadata / delete_from_layer / relink_neighbor).  The reference holds no golden rows for a relinked graph, so
the restatement is pinned on what its own tests assert (crates/db/src/search/vector/index.rs: unknown id :2263, double delete
:2294-2295, stale / deleted entry point repaired :3540-3568, the deleted id never returned :3571-3605) plus the row invariants
every staged row must satisfy (neighbor_set.rs:1-9) -- parity of the RELINKED ROWS themselves stays unpinned, as the header of
the function says.  The device path (hvx_index_delete_batch) is held to this restatement row for row in tests/test_gpu_delete.py."""
import numpy as np
import pytest

import fixtures as fx


def build(orc, n, dim, metric, m, m0, efc, seed, kernel=None, ids=None):
    rng = np.random.default_rng(seed)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    lv = fx.draw_levels(n, m, seed=seed + 1)
    ids = np.arange(n, dtype=np.uint64) * 2 + 5 if ids is None else ids
    ix = orc.Index(dim, metric, kernel=orc.K_AVX_FMA if kernel is None else kernel, m=m, m0=m0, ef_construction=efc)
    for i in range(n):
        assert ix.insert(int(ids[i]), data[i], int(lv[i])) == orc.OK
    return ix, data, lv, ids


def rows_by_id(ex):
    """export -> {id: (level, layer-0 row, [upper rows])}"""
    out, r = {}, 0
    for t, nid in enumerate(ex["node_ids"].tolist()):
        l0 = ex["l0_neighbors"][int(ex["l0_offsets"][t]):int(ex["l0_offsets"][t + 1])].tolist()
        lv = int(ex["level"][t])
        up = [ex["up_neighbors"][int(ex["up_offsets"][r + l]):int(ex["up_offsets"][r + l + 1])].tolist() for l in range(lv)]
        r += lv
        out[nid] = (lv, l0, up)
    return out


def check_rows(rows, m, m0):
    live = set(rows)
    for nid, (lv, l0, up) in rows.items():
        for layer, row in enumerate([l0] + up):
            assert row == sorted(set(row)) and nid not in row          # canonical: ascending, deduped, self-free
            assert len(row) <= (max(m0, 2 * m) if layer == 0 else m)    # degree limit of the layer (mutation.rs:178-196)
            assert set(row) <= live                                     # nothing points at a deleted node
            for t in row:
                assert rows[t][0] >= layer                              # ... or at a node that does not live on this layer


def test_unknown_id_and_double_delete_succeed_and_change_nothing(orc):
    ix, data, lv, ids = build(orc, 300, 16, orc.L2SQ, 8, 16, 40, seed=1)
    before = ix.export()
    assert ix.delete(10 ** 9) == (orc.OK, False) and ix.count == 300          # index.rs:2263
    after = ix.export()
    assert all(np.array_equal(before[k], after[k]) for k in ("node_ids", "l0_offsets", "l0_neighbors", "up_offsets", "up_neighbors"))
    victim = int(ids[17])
    assert ix.delete(victim) == (orc.OK, True) and ix.count == 299 and not ix.is_live(victim)
    once = ix.export()
    assert ix.delete(victim) == (orc.OK, False) and ix.count == 299             # index.rs:2294-2295
    twice = ix.export()
    assert all(np.array_equal(once[k], twice[k]) for k in ("node_ids", "l0_offsets", "l0_neighbors", "up_offsets", "up_neighbors"))


@pytest.mark.parametrize("metric,dim,m,m0", [(1, 24, 8, 16), (0, 32, 6, 12), (2, 12, 4, 8)])
def test_deleted_ids_are_never_returned_and_the_rows_stay_canonical(orc, metric, dim, m, m0):
    ix, data, lv, ids = build(orc, 700, dim, metric, m, m0, 60, seed=10 + metric)
    rng = np.random.default_rng(5)
    dels = [int(x) for x in ids[rng.permutation(700)[:250]]]
    for i, d in enumerate(dels):
        assert ix.delete(d) == (orc.OK, True)
        if i % 50 == 49:
            check_rows(rows_by_id(ix.export()), m, m0)
    assert ix.count == 450
    rows = rows_by_id(ix.export())
    assert not (set(rows) & set(dels))
    check_rows(rows, m, m0)
    q = rng.standard_normal((40, dim)).astype(np.float32)
    hits = 0
    for qi in range(40):
        rc, oid, osc = ix.search(q[qi], 10, 80)
        rc2, tid, tsc = ix.flat(q[qi], 10)
        assert rc == orc.OK and rc2 == orc.OK and not (set(oid.tolist()) & set(dels)) and not (set(tid.tolist()) & set(dels))   # index.rs:3601-3604
        hits += len(set(oid.tolist()) & set(tid.tolist()))
        rc3, rid, rsc = ix.flat(q[qi], 10, allowed=np.asarray(dels[:20] + [int(x) for x in tid[:3]], np.uint64))
        assert set(rid.tolist()) == set(int(x) for x in tid[:3])        # deleted candidates hold no vector: omitted like unindexed ids
    assert hits / 400.0 > 0.97                                           # the relinked graph still navigates


def test_deleting_the_entry_point_moves_it_to_the_best_remaining_candidate(orc):
    ix, data, lv, ids = build(orc, 400, 16, orc.L2SQ, 8, 16, 40, seed=3)
    for _ in range(6):
        entry, top = ix.entry()
        assert ix.delete(entry) == (orc.OK, True)
        rows = rows_by_id(ix.export())
        best = max(v[0] for v in rows.values())
        want = min(nid for nid, v in rows.items() if v[0] == best)       # highest layer first, then ascending id (keys/vectors.rs:1097)
        assert ix.entry() == (want, best) and ix.is_live(want)            # index.rs:3561-3567: the repaired entry point holds an item
        rc, oid, _ = ix.search(data[0], 5, 40)
        assert rc == orc.OK and entry not in oid.tolist()


def test_a_chain_closes_over_a_deleted_link(orc):
    """five points on a line with Mmax = 2 (m = 1): every node links its two neighbours; without the middle node its neighbours must
    find each other through relink_neighbor's candidates (the sources' joint neighbourhood, mutation.rs:1862-1875)"""
    ix = orc.Index(2, orc.L2SQ, m=1, m0=2, ef_construction=10)
    for i in range(5):
        assert ix.insert(i + 1, np.asarray([float(i), 0.0], np.float32), 0) == orc.OK
    assert ix.delete(3) == (orc.OK, True)
    rows = rows_by_id(ix.export())
    assert 4 in rows[2][1] and 2 in rows[4][1]
    for start in (0.0, 4.0):
        rc, oid, _ = ix.search(np.asarray([start, 0.0], np.float32), 4, 10)
        assert sorted(oid.tolist()) == [1, 2, 4, 5]


def test_delete_everything_then_insert_again(orc):
    ix, data, lv, ids = build(orc, 60, 8, orc.L2SQ, 4, 8, 20, seed=8)
    for d in ids.tolist():
        assert ix.delete(int(d)) == (orc.OK, True)
    assert ix.count == 0 and ix.entry() is None
    rc, oid, _ = ix.search(data[0], 3, 10)
    assert rc == orc.OK and oid.size == 0
    assert ix.insert(1000, data[0], 2) == orc.OK and ix.entry() == (1000, 2)   # the first row of an empty index: entry point, empty rows
    assert ix.insert(int(ids[3]), data[3], 0) == orc.OK                        # an upsert's second half: the id again, a fresh row
    rc, oid, _ = ix.search(data[3], 2, 10)
    assert oid.tolist() == [int(ids[3]), 1000]


def test_simhash_rows_and_the_restricted_walk_skip_deleted_nodes(orc):
    ix, data, lv, ids = build(orc, 900, 32, orc.L2SQ, 8, 16, 60, seed=21)
    ix.set_simhash(42)
    dels = [int(x) for x in ids[::7]]
    for d in dels:
        assert ix.delete(d)[0] == orc.OK
    hashes = ix.get_simhash()
    assert hashes.size == ix.count
    allowed = np.asarray(sorted(set(ids[100:700].tolist())), np.uint64)
    q = data[5]
    rc, oid, osc, st = ix.search_restricted(q, 10, 100, allowed)
    assert rc == orc.OK and not (set(oid.tolist()) & set(dels)) and set(oid.tolist()) <= set(allowed.tolist())


# ------------------------------------------------------------------------------------------------------------------------
# A second, independent reading of the same Rust: plain Python containers, one statement per statement of
# mutation.rs:1658-2055 / mod.rs:809-856, distances from the oracle's distance function only.  Small cases, pure-Python loops.
# ------------------------------------------------------------------------------------------------------------------------
class PyGraph:
    def __init__(self, orc, ex, vectors_by_id, metric, kernel, m, m0):
        self.orc, self.metric, self.kernel = orc, metric, kernel
        self.upper_limit, self.layer0_limit = m, max(m0, 2 * m)               # MutationDegreeLimits::try_from_metadata (mutation.rs:181-196)
        self.vec = dict(vectors_by_id)
        self.rows = {}                                                        # (layer, id) -> sorted list
        self.level = {}
        for nid, (lv, l0, up) in rows_by_id(ex).items():
            self.level[nid] = lv
            self.rows[(0, nid)] = list(l0)
            for l, r in enumerate(up):
                self.rows[(l + 1, nid)] = list(r)
        self.entry, self.max_layer = ex["entry_point"], ex["max_layer"]

    def dist(self, a, b):
        return float(self.orc.distance(self.metric, self.vec[a], self.vec[b], kernel=self.kernel))

    def neighbors(self, layer, nid):                                          # load_neighbors_for_mutation: an absent row reads as empty
        return list(self.rows.get((layer, nid), []))

    def stage(self, layer, nid, neighbors):                                   # stage_neighbors_vec_for_mutation: sorted, canonical, within the limit
        neighbors = sorted(neighbors)
        assert len(set(neighbors)) == len(neighbors) and nid not in neighbors
        assert len(neighbors) <= (self.layer0_limit if layer == 0 else self.upper_limit)
        self.rows[(layer, nid)] = neighbors

    def select_diverse(self, candidates, m):                                  # mod.rs:809-856; candidates = [(score, id)] sorted
        selected = []
        for score, c in candidates:
            if len(selected) >= m:
                break
            if c not in self.vec:
                continue
            if all(not (self.dist(c, s) < score) for s in selected):
                selected.append(c)
        if len(selected) < m:
            for score, c in candidates:
                if len(selected) >= m:
                    break
                if c in self.vec and c not in selected:
                    selected.append(c)
        return selected

    def relink_neighbor(self, layer, nb, candidates, maximum):               # mutation.rs:1916-2055
        if nb not in self.vec:
            return
        old = self.neighbors(layer, nb)
        current = list(old)
        cd = sorted((self.dist(nb, c), c) for c in candidates if c != nb and c in self.vec)
        for score, c in cd[:maximum]:
            if c not in current:
                current.append(c)
        if len(current) > maximum:
            ds = sorted((self.dist(nb, c), c) for c in current if c in self.vec)
            current = self.select_diverse(ds, maximum)
        self.stage(layer, nb, current)
        for new in current:
            if new in old:
                continue
            rev = self.neighbors(layer, new)
            if nb in rev:
                continue
            rev.append(nb)
            if len(rev) > maximum:
                assert new in self.vec
                ds = sorted((self.dist(new, c), c) for c in rev if c in self.vec)
                rev = self.select_diverse(ds, maximum)
            self.stage(layer, new, rev)

    def delete_from_layer(self, node, layer, maximum, extra_sources):        # mutation.rs:1819-1888
        outgoing = self.neighbors(layer, node)
        mandatory = sorted(set(n for n in outgoing if n != node))
        affected = sorted(set(mandatory) | set(s for s in extra_sources if s != node))
        if not affected:
            return
        relink = set(mandatory)
        for nb in affected:
            row = self.neighbors(layer, nb)
            if node in row:                                                    # remove_edge_from_neighbor
                row.remove(node)
                self.stage(layer, nb, row)
                relink.add(nb)
        relink = sorted(relink)
        if not relink:
            return
        cands = set(c for c in relink if c != node)
        for nb in relink:
            cands |= set(c for c in self.neighbors(layer, nb) if c != node and c != nb)
        for nb in relink:
            self.relink_neighbor(layer, nb, cands, maximum)

    def delete(self, node):                                                   # stage_delete_with_metadata, mutation.rs:1658-1774
        if node not in self.vec:
            return False
        top = self.level[node]
        reverse = {}
        for (layer, owner), row in self.rows.items():
            if node in row and owner != node:
                reverse.setdefault(layer, []).append(owner)
        layers = set(range(top + 1)) | set(reverse)
        for layer in sorted(layers, reverse=True):
            self.delete_from_layer(node, layer, self.layer0_limit if layer == 0 else self.upper_limit, sorted(reverse.get(layer, [])))
        for layer in range(top + 1):
            self.rows.pop((layer, node), None)
        del self.vec[node]
        del self.level[node]
        if self.entry == node:                                                # find_best_entry_candidate: [inv_layer][node_id] key order
            if self.level:
                best = max(self.level.values())
                self.entry, self.max_layer = min(n for n, l in self.level.items() if l == best), best
            else:
                self.entry, self.max_layer = None, 0
        return True

    def as_rows(self):
        return {nid: (lv, self.rows[(0, nid)], [self.rows[(l, nid)] for l in range(1, lv + 1)]) for nid, lv in self.level.items()}


@pytest.mark.parametrize("metric,dim,m,m0,kern", [(1, 12, 4, 8, "avx_fma"), (0, 40, 5, 10, "avx"), (2, 9, 3, 6, "scalar"), (1, 24, 6, 12, "neon")])
def test_the_c_restatement_of_delete_equals_an_independent_python_reading(orc, metric, dim, m, m0, kern):
    kernel = {"avx_fma": orc.K_AVX_FMA, "avx": orc.K_AVX, "scalar": orc.K_SCALAR, "neon": orc.K_NEON}[kern]
    n = 260
    ix, data, lv, ids = build(orc, n, dim, metric, m, m0, 30, seed=40 + dim, kernel=kernel)
    py = PyGraph(orc, ix.export(), {int(ids[i]): data[i] for i in range(n)}, metric, kernel, m, m0)
    assert py.as_rows() == rows_by_id(ix.export())
    rng = np.random.default_rng(dim)
    order = [int(x) for x in ids[rng.permutation(n)[:90]]]
    order.insert(10, ix.entry()[0])                                            # the entry point among them
    for step, d in enumerate(order):
        want = py.delete(d)
        rc, existed = ix.delete(d)
        assert rc == orc.OK and existed == want
        if step % 6 == 0 or step == len(order) - 1:
            ex = ix.export()
            assert py.as_rows() == rows_by_id(ex), f"rows differ after delete #{step} (node {d})"
            assert (py.entry, py.max_layer) == ((ex["entry_point"], ex["max_layer"]) if ex["entry_point"] is not None else (None, 0))
