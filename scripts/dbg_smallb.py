import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("helix-db_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import pyhvx as hv
for (dtype_name, metric, dim, n, k, b) in [("bf16", 0, 256, 20001, 25, 70), ("bf16", 1, 256, 20001, 25, 70), ("bf16", 0, 256, 20001, 25, 32), ("f32", 1, 768, 20000, 10, 32), ("f32", 0, 768, 20000, 10, 32), ("bf16", 0, 768, 20000, 10, 32), ("bf16", 0, 256, 20001, 25, 64)]:
    rng = np.random.default_rng(7000 + dim + n)
    centers = rng.standard_normal((32, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 32, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
    dt = {"bf16": hv.BF16, "f32": hv.F32}[dtype_name]
    gix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=metric, node_ids=np.arange(n, dtype=np.uint64), vectors=data, dtype=dt,
                                              l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), max_batch=max(b, 16))
    q = (centers[rng.integers(0, 32, b)] + 0.5 * rng.standard_normal((b, dim))).astype(np.float32)
    a = gix.flat_search_batch(q, k)
    pa = gix.last_scan_path()
    gix.set_option(hv.OPT_FLAT_NO_SMALLB, 1)
    c = gix.flat_search_batch(q, k)
    pc = gix.last_scan_path()
    print(dtype_name, metric, dim, n, k, b, "smallb path", pa, "other path", pc, "same", a[0].tolist() == c[0].tolist(), "ms", a[3]["device_ms"], c[3]["device_ms"], flush=True)
