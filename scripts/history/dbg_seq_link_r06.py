"""round 6 debugging aid: first inserted node after which the sequential device build (fused select / link steps) leaves the oracle's rows"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "helix-db_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import pyhvx as hv, fixtures as fx
import orc
orc.lib()

def rows_of(g, n):
    return [g["l0_neighbors"][int(g["l0_offsets"][i]):int(g["l0_offsets"][i + 1])].tolist() for i in range(n)]

n, dim, metric, m, m0, efc = 1000, 48, 2, 8, 16, 50
rng = np.random.default_rng(7100 + dim + metric + n)
data = rng.standard_normal((n, dim)).astype(np.float32)
lv = fx.draw_levels(n, m, seed=n)
ids = np.arange(n, dtype=np.uint64) * 3 + 5

def diff_at(k):
    oix = orc.Index(dim, metric, kernel=orc.K_AVX_FMA, m=m, m0=m0, ef_construction=efc)
    for i in range(k):
        assert oix.insert(int(ids[i]), data[i], int(lv[i])) == orc.OK
    ex = oix.export()
    gix, st = hv.ValidatedVectorReadIndex.build(dim=dim, metric=metric, node_ids=ids[:k], vectors=data[:k], levels=lv[:k], m=m, m0=m0, ef_construction=efc, sequential=True)
    g = gix.export_graph()
    a, b = rows_of(g, k), rows_of(ex, k)
    bad = [i for i in range(k) if a[i] != b[i]]
    gix.close()
    return bad, a, b

for rep in range(6):
    bad, a, b = diff_at(300)
    print("mode", os.environ.get("HVX_SEQ_DEBUG"), "rep", rep, "bad rows", len(bad), flush=True)
