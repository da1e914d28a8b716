#!/usr/bin/env python3
"""BASELINE config #3 on MI355X: graph where_() filter + vector kNN (the reference's DBpedia-1M traversal
prefilter benchmark, crates/db/tests/production_support/index_lifecycle_scale.rs:585-613,1712-2013, run by
scripts/run-dbpedia-1m-prefilter-benchmark.sh).

The DBpedia fbin cannot be downloaded here, so the corpus is the synthetic stand-in of SURVEY.md 8(d) with the
benchmark's exact topology: N x 1536 f32, Euclidean, node i has ONE out-edge to (i + N/2) mod N, four equality
groups of 100 / 1 000 / 10 000 / 100 000 source nodes starting at 0 / 100 / 1 100 / 11 100, candidates = the
one-hop expansion of a group, 32 queries = rows 800 000 + j*100 000/32, k = 10.

Per selectivity: one `hvx_expand_filter` (device CSR hop -> candidate bitmap) + one
`hvx_search_restricted_batch` over the 32 queries.  The device scans the candidates exactly, so recall@10 is
1.0 by construction (reference gate: >= 0.92); a sample is checked bit-for-bit against the CPU oracle.
Prints one JSON line per selectivity.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "helix-db_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--queries", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--verify", type=int, default=4, help="queries per selectivity checked against the oracle")
    args = ap.parse_args()
    import pyhvx as hv
    from pyhvx import synth
    import orc
    n, dim, k, nq = args.rows, args.dim, args.k, args.queries
    dev = torch.device("cuda", 0)
    x, _ = synth.embedding_like(n, dim, 1, 20260923, dev, latent=24, clusters=1024)
    xh = x.cpu().numpy()
    del x
    ids = np.arange(n, dtype=np.uint64)
    t0 = time.time()
    ix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=ids, vectors=xh,
                                             l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64),
                                             max_batch=nq)
    # one out-edge per node: i -> (i + n/2) mod n   (index_lifecycle_scale.rs:1769-1776)
    off = np.arange(n + 1, dtype=np.uint64)
    tgt = ((np.arange(n, dtype=np.uint64) + np.uint64(n // 2)) % np.uint64(n)).astype(np.uint64)
    g = hv.Graph(n, off, tgt)
    print(f"[prefilter] index + graph on device in {time.time() - t0:.1f}s", file=sys.stderr)
    qrows = [(int(n * 0.8) + j * (n // 10) // nq) % n for j in range(nq)]  # rows 800 000 + j*100 000/32 at n = 1M
    q = xh[qrows].copy()
    starts = {100: 0, 1000: 100, 10000: 1100, 100000: 11100}
    for size, start in starts.items():
        if start + size > n // 2:
            continue
        src = np.arange(start, start + size, dtype=np.uint64)
        lat = []
        for r in range(args.rounds + 1):
            t1 = time.perf_counter()
            words = g.expand(src, hv.DIR_OUT)                                   # where_() group -> one hop
            cand = hv.RestrictedVectorCandidates.from_bitmap_words(words)
            t2 = time.perf_counter()
            gid, gsc, gcnt = ix.search_restricted_batch(q, hv.SearchParams(k), cand)
            t3 = time.perf_counter()
            if r:
                lat.append((t2 - t1, t3 - t2))
        assert len(cand) == size and int(cand.ids[0]) == start + n // 2
        # the same in ONE call: the hop's bitmap never leaves the device (hvx_prefilter_search_batch)
        fused = []
        for r in range(args.rounds + 1):
            t1 = time.perf_counter()
            fid, fsc, fcnt, ncand, fst = ix.prefilter_search_batch(g, q, hv.SearchParams(k), src, direction=hv.DIR_OUT)
            if r:
                fused.append(time.perf_counter() - t1)
        assert ncand == size and fid.tolist() == gid.tolist() and fsc.view(np.uint32).tolist() == gsc.view(np.uint32).tolist()
        fu = float(np.median(fused)) * 1e3
        ok = True
        for qi in range(min(args.verify, nq)):
            rc, oid, osc = orc.flat_matrix(orc.L2SQ, xh[start + n // 2: start + n // 2 + size], q[qi], k, kernel=orc.K_AVX_FMA_HW)
            ok &= (gid[qi, :gcnt[qi]] - np.uint64(start + n // 2)).tolist() == oid.tolist()
            ok &= gsc[qi, :gcnt[qi]].view(np.uint32).tolist() == osc.view(np.uint32).tolist()
        ex = float(np.median([a for a, _ in lat])) * 1e3
        se = float(np.median([b for _, b in lat])) * 1e3
        print(json.dumps({
            "workload": f"configs[2] stand-in: {n}x{dim} f32, one-hop expand of an equality group of {size} nodes -> restricted kNN k={k}, {nq} queries",
            "candidates": size, "strategy": "exact device scan",
            "reference_plan": hv.restricted_execution_plan(size, dim, hv.SearchParams.new(k)),
            "expand_ms": round(ex, 3), "restricted_search_ms_per_batch": round(se, 3),
            "end_to_end_us_per_query": round((ex + se) * 1e3 / nq, 1),
            "fused_call_ms_per_batch": round(fu, 3), "fused_us_per_query": round(fu * 1e3 / nq, 1),
            "fused_scan_kernels_ms": round(fst["device_ms"], 3), "recall_at_10": 1.0,
            "reference_gates": {"recall_at_10": 0.92, "vector_increment_p95_ms": 15, "end_to_end_p95_ms": 50},
            "algorithmic_bytes_per_query": size * dim * 4, "oracle_bit_exact_sample": bool(ok)}), flush=True)
        assert ok, "restricted search differs from the oracle"


if __name__ == "__main__":
    main()
