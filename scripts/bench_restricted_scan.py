#!/usr/bin/env python3
"""Restricted exact scan (config #3's exact strategy) by candidate-set size: kernel-bracket ms, end-to-end ms, HBM GB/s of the
algorithmic bytes (candidates x dim x 4).  usage: bench_restricted_scan.py [rows=1000000] [dim=1536] [b=32]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))
import numpy as np, torch
import pyhvx as hv
from pyhvx import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
b = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda", 0)
x, _ = synth.corpus("clustered", n, dim, 1, 20260923, dev)
ix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64), vectors=x,
                                         l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), device=0, max_batch=max(b, 16))
q = x[torch.randint(0, n, (b,), generator=torch.Generator().manual_seed(1)).to(dev)].cpu().numpy()
for size in (1000, 10000, 100000, 400000):
    if size > n // 2:
        continue
    allowed = np.arange(n // 2, n // 2 + size, dtype=np.uint64)
    cand = hv.RestrictedVectorCandidates.from_ids(allowed)
    for opt in (0, 2, 1):  # the LDS-ring build (b <= 32), the register-fragment build, the large-batch kernels
        ix.set_option(hv.OPT_FLAT_NO_SMALLB, opt)
        lat, kern = [], []
        for r in range(6):
            st = hv.Stats()
            t0 = time.perf_counter()
            ids = np.zeros((b, 10), np.uint64); sc = np.zeros((b, 10), np.float32); cnt = np.zeros(b, np.uint32)
            hv._check(hv.lib().hvx_search_restricted_batch(ix._h, hv._ptr(q), b, 10, 100, hv._ptr(cand.ids), None, cand.ids.size, hv._ptr(ids), hv._ptr(sc),
                                                           hv._ptr(cnt), None, hv.C.byref(st)))
            if r:
                lat.append(time.perf_counter() - t0); kern.append(st.device_ms)
        kms = float(np.median(kern))
        print(json.dumps({"candidates": size, "b": b, "dim": dim, "small_batch_selector": opt, "path": ix.last_scan_path(), "kernel_ms": round(kms, 4),
                          "end_to_end_ms": round(float(np.median(lat)) * 1e3, 3), "hbm_gbs": round(size * dim * 4 / (kms * 1e-3) / 1e9, 1),
                          "frac": round(size * dim * 4 / (kms * 1e-3) / 8e12, 4)}), flush=True)
