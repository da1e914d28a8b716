#!/usr/bin/env python3
"""Round 6: the one-launch restricted exact scan (csrc/hvx_restricted_exact.hip) next to the matrix-core pipeline it replaces for small
candidate sets, on config #3's shape (rows x 1536 f32, Euclidean, clustered stand-in).
  * shared candidate set (the fused prefilter's shape): kernel-bracket ms / end-to-end ms / algorithmic GB/s by set size and batch;
  * every query with its own candidate list (the operator's shape): one launch per batch of b lists of m ids;
  * the fused hop + kNN call for ONE query (the reference issues one query per call).
usage: bench_restricted_direct.py [rows=400000] [dim=1536] [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))
import numpy as np
import torch
import pyhvx as hv
from pyhvx import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
out_path = sys.argv[3] if len(sys.argv) > 3 else None
dev = torch.device("cuda", 0)
x, _ = synth.corpus("clustered", n, dim, 1, 20260923, dev, centres=1024, sigma=0.15)
ids_all = np.arange(n, dtype=np.uint64)
ix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=ids_all, vectors=x, l0_offsets=np.zeros(n + 1, np.uint64),
                                         l0_neighbors=np.zeros(0, np.uint64), entry_point=0, device=0, max_batch=1024)
gen = torch.Generator().manual_seed(1)
qall = x[torch.randint(0, n, (1024,), generator=gen).to(dev)].cpu().numpy()
rec = {"rows": n, "dim": dim, "shared": [], "per_query": [], "fused_single": []}


def timed(fn, rounds=7):
    lat, kern = [], []
    for r in range(rounds + 1):
        st = hv.Stats()
        t0 = time.perf_counter()
        fn(st)
        if r:
            lat.append(time.perf_counter() - t0)
            kern.append(st.device_ms)
    return float(np.median(lat)) * 1e3, float(np.median(kern))


def shared_call(q, allowed, k=10):
    b = q.shape[0]
    ids = np.zeros((b, k), np.uint64); sc = np.zeros((b, k), np.float32); cnt = np.zeros(b, np.uint32)

    def run(st):
        hv._check(hv.lib().hvx_search_restricted_batch(ix._h, hv._ptr(q), b, k, 100, hv._ptr(allowed), None, allowed.size, hv._ptr(ids), hv._ptr(sc),
                                                       hv._ptr(cnt), None, hv.C.byref(st)))
    return run, (ids, sc, cnt)


for size in (100, 1000, 10000, 100000):
    if size > n // 2:
        continue
    allowed = np.arange(n // 2, n // 2 + size, dtype=np.uint64)
    for b in (1, 8, 32):
        row = {"candidates": size, "b": b}
        res = {}
        for name, opt in (("direct", 2), ("pipeline", 1), ("auto", 0)):
            ix.set_option(hv.OPT_RESTRICTED_DIRECT, opt)
            run, outs = shared_call(qall[:b], allowed)
            ms, kms = timed(run)
            res[name] = outs[0].copy()
            alg = size * dim * 4
            row[name] = {"kernel_ms": round(kms, 4), "end_to_end_ms": round(ms, 3), "path": ix.last_scan_path(),
                         "alg_gbs": round(alg / (kms * 1e-3) / 1e9, 1), "frac_of_8TBs": round(alg / (kms * 1e-3) / 8e12, 4)}
        row["same_ids"] = bool((res["direct"] == res["pipeline"]).all() and (res["auto"] == res["pipeline"]).all())
        rec["shared"].append(row)
        print(json.dumps(row), flush=True)

ix.set_option(hv.OPT_RESTRICTED_DIRECT, 0)
rng = np.random.default_rng(3)
for m in (100, 1000, 10000):
    for b in (32, 256, 1024):
        if b * m > 8_000_000:
            continue
        lists = [np.sort(rng.choice(ids_all, m, replace=False)) for _ in range(b)]
        off = (np.arange(b + 1, dtype=np.uint64) * np.uint64(m))
        flat = np.concatenate(lists).astype(np.uint64)
        q = qall[:b]
        ids = np.zeros((b, 10), np.uint64); sc = np.zeros((b, 10), np.float32); cnt = np.zeros(b, np.uint32); stt = np.zeros(b, np.uint32)
        rp = hv.RestrictedParams.auto(10, 100)

        def run(st):
            hv._check(hv.lib().hvx_search_restricted_batch_params(ix._h, hv._ptr(q), b, hv.C.byref(rp), hv._ptr(flat), hv._ptr(off), flat.size, hv._ptr(ids),
                                                                  hv._ptr(sc), hv._ptr(cnt), hv._ptr(stt), None, hv.C.byref(st)))
        ms, kms = timed(run, rounds=5)
        alg = b * m * dim * 4
        row = {"lists": b, "ids_per_list": m, "kernel_ms": round(kms, 4), "end_to_end_ms": round(ms, 3), "path": ix.last_scan_path(),
               "qps_kernel": round(b / (kms * 1e-3), 1), "qps_end_to_end": round(b / (ms * 1e-3), 1), "alg_gbs": round(alg / (kms * 1e-3) / 1e9, 1),
               "frac_of_8TBs": round(alg / (kms * 1e-3) / 8e12, 4)}
        rec["per_query"].append(row)
        print(json.dumps(row), flush=True)

# fused hop + kNN, one query per call (index_lifecycle_scale.rs:1893-1912)
tgt = ((ids_all + np.uint64(n // 2)) % np.uint64(n)).astype(np.uint64)
g = hv.Graph(n, np.arange(n + 1, dtype=np.uint64), tgt)
for size in (100, 1000, 10000, 100000):
    if size > n // 2:
        continue
    src = np.arange(size, dtype=np.uint64)
    for name, opt in (("direct", 0), ("pipeline", 1)):
        ix.set_option(hv.OPT_RESTRICTED_DIRECT, opt)
        one = []
        ix.prefilter_search_batch(g, qall[:1], hv.SearchParams(10).with_ef(100), src, direction=hv.DIR_OUT)
        for qi in range(32):
            t0 = time.perf_counter()
            ix.prefilter_search_batch(g, qall[qi:qi + 1], hv.SearchParams(10).with_ef(100), src, direction=hv.DIR_OUT)
            one.append((time.perf_counter() - t0) * 1e3)
        one.sort()
        row = {"candidates": size, "scan": name, "single_query_end_to_end_ms": {"p50": round(one[15], 4), "p95": round(one[30], 4)}, "path": ix.last_scan_path()}
        rec["fused_single"].append(row)
        print(json.dumps(row), flush=True)
if out_path:
    json.dump(rec, open(out_path, "w"), indent=1)
