import sys, os
sys.path.insert(0, "helix-db_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np, torch
import pyhvx as hv, fixtures as fx
rng = np.random.default_rng(9300)
n0, n, dim, m, m0, efc = 60000, 80000, 256, 16, 32, 100
centres = rng.standard_normal((64, dim)).astype(np.float32)
data = (centres[rng.integers(0, 64, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
lv = fx.draw_levels(n, m, seed=5)
ids = np.arange(n, dtype=np.uint64)
q = (centres[rng.integers(0, 64, 512)] + 0.5 * rng.standard_normal((512, dim))).astype(np.float32)
def rec(ix, tag):
    gid, _, _, _ = ix.search_batch(q, hv.SearchParams(10).with_ef(100))
    fid, _, _, _ = ix.flat_search_batch(q, 10)
    a = ix.audit_graph()
    g = ix.export_graph()
    deg = np.diff(g["l0_offsets"].astype(np.int64))
    print(tag, "recall", round(fx.recall_at_k(gid, fid), 4), "unreachable", a["unreachable_l0"], "deg mean", round(float(deg.mean()), 2), "deg new rows", round(float(deg[n0:].mean()), 2), "deg old", round(float(deg[:n0].mean()), 2), flush=True)
for rep in range(2):
    whole, _ = hv.ValidatedVectorReadIndex.build(dim=dim, metric=1, node_ids=ids, vectors=data, levels=lv, m=m, m0=m0, ef_construction=efc, max_batch=1024, batch_divisor=16, search_max_batch=512)
    rec(whole, f"whole#{rep}")
    whole.close()
for rep, chunks in enumerate(([(n0, n0 + 8000), (n0 + 8000, n)], [(n0, n)], [(a, a + 1000) for a in range(n0, n, 1000)])):
    gix, _ = hv.ValidatedVectorReadIndex.build(dim=dim, metric=1, node_ids=ids[:n0], vectors=data[:n0], levels=lv[:n0], m=m, m0=m0, ef_construction=efc, max_batch=1024, batch_divisor=16, search_max_batch=512, reserve_rows=n - n0)
    for a, b in chunks:
        st = gix.insert_batch(ids[a:b], data[a:b], lv[a:b], ef_construction=efc, max_batch=1024, batch_divisor=16)
    rec(gix, f"grown#{rep} ({len(chunks)} calls, last stats {st})")
    gix.close()
