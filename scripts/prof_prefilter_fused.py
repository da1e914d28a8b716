#!/usr/bin/env python3
"""Where one fused graph-hop + restricted-kNN call (hvx_prefilter_search_batch, configs[2] topology) spends its time: run ONE group
size `--size` `--iters` times; meant to run under `rocprofv3 --hip-trace --kernel-trace --memory-copy-trace` (per-API / per-kernel
totals / iters = the anatomy of a call).  Prints the median end-to-end ms."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--sizes", default="100,1000,10000,100000")
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    import pyhvx as hv
    from pyhvx import synth
    n, dim, k, nq = args.rows, args.dim, 10, 32
    dev = torch.device("cuda", 0)
    x, _ = synth.embedding_like(n, dim, 1, 20260923, dev, latent=24, clusters=1024)
    ids = np.arange(n, dtype=np.uint64)
    ix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=ids, vectors=x, l0_offsets=np.zeros(n + 1, np.uint64),
                                             l0_neighbors=np.zeros(0, np.uint64), max_batch=nq)
    off = np.arange(n + 1, dtype=np.uint64)
    tgt = ((np.arange(n, dtype=np.uint64) + np.uint64(n // 2)) % np.uint64(n)).astype(np.uint64)
    g = hv.Graph(n, off, tgt)
    qrows = [(int(n * 0.8) + j * (n // 10) // nq) % n for j in range(nq)]
    q = x[qrows].cpu().numpy().copy()
    starts = {100: 0, 1000: 100, 10000: 1100, 100000: 11100}
    for size in [int(s) for s in args.sizes.split(",")]:
        src = np.arange(starts[size], starts[size] + size, dtype=np.uint64)
        t = []
        for r in range(args.iters + 1):
            t1 = time.perf_counter()
            fid, fsc, fcnt, ncand, fst = ix.prefilter_search_batch(g, q, hv.SearchParams(k), src, direction=hv.DIR_OUT)
            if r:
                t.append(time.perf_counter() - t1)
        print(json.dumps({"candidates": size, "iters": args.iters, "fused_call_ms_median": round(float(np.median(t)) * 1e3, 3),
                          "fused_call_ms_min": round(float(np.min(t)) * 1e3, 3), "scan_kernels_ms": round(fst["device_ms"], 3)}), flush=True)


if __name__ == "__main__":
    main()
