"""Sequential inserts / upserts on a device-built graph (hvx_index_insert_batch sequential = the reference's one-node-at-a-time order;
hvx_index_upsert_batch), for rocprofv3 --kernel-trace --stats:
    python scripts/seq_insert_bench.py [rows] [dim] [inserts]
Prints one JSON line: us per sequential insert / per upsert."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "helix-db_amd"), os.path.join(ROOT, "tests")]
import pyhvx as hv  # noqa: E402
import fixtures as fx  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
ni = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rng = np.random.default_rng(1)
centers = rng.standard_normal((256, dim)).astype(np.float32)
data = (centers[rng.integers(0, 256, n + ni)] + 0.6 * rng.standard_normal((n + ni, dim)).astype(np.float32)).astype(np.float32)
lv = fx.draw_levels(n + ni, 16, seed=2)
ids = np.arange(n + ni, dtype=np.uint64)
ix, st = hv.ValidatedVectorReadIndex.build(dim=dim, metric=hv.EUCLIDEAN, node_ids=ids[:n], vectors=data[:n], levels=lv[:n], m=16, m0=32, ef_construction=200,
                                            search_max_batch=256, reserve_rows=ni, reserve_upper_rows=int(lv[n:].sum()))
ix.sync()
ix.insert_batch(ids[n:n + 8], data[n:n + 8], lv[n:n + 8], ef_construction=200, sequential=True)   # warm-up
ix.sync()
t0 = time.perf_counter()
ix.insert_batch(ids[n + 8:], data[n + 8:], lv[n + 8:], ef_construction=200, sequential=True)
ix.sync()
t_ins = time.perf_counter() - t0
up = rng.choice(n, 64, replace=False).astype(np.uint64)
newv = (data[up] + 0.05 * rng.standard_normal((64, dim))).astype(np.float32)
t0 = time.perf_counter()
ix.upsert_batch(up, newv, ef_construction=200)
ix.sync()
t_up = time.perf_counter() - t0
print(json.dumps({"rows": n, "dim": dim, "sequential_inserts": ni - 8, "us_per_insert": round(t_ins / (ni - 8) * 1e6, 1), "upserts": 64,
                  "us_per_upsert": round(t_up / 64 * 1e6, 1)}))
