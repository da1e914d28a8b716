#!/usr/bin/env python3
"""Throughput of the config #2 step with L execution lanes (hvx_index_fork) x the 1- / 2-queries-per-SIMD kernel build,
in ONE process on one box: every combination answers the same 1 024 queries `steps` times, lane i % L per step, no host
synchronisation inside the timed region; results (ids, score bits, per-query counters) must be identical everywhere.
usage: ab_lanes.py [dataset=embedding|clustered|gaussian] [ef=128] [steps=60]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))

import numpy as np
import torch


def main():
    dataset = sys.argv[1] if len(sys.argv) > 1 else "embedding"
    ef = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    import pyhvx as hv
    from pyhvx import synth
    dev = torch.device("cuda", 0)
    n, dim, b, k = int(os.environ.get("AB_ROWS", 1_000_000)), 768, 1024, 10
    nbq = 16  # distinct query batches, cycled: lanes in flight never gather the same rows (identical batches hit in L2 / MALL and flatter the number)
    x, q_all = synth.corpus(dataset, n, dim, b * nbq, 20260921, dev)
    qs = [q_all[j * b:(j + 1) * b] for j in range(nbq)]
    q = qs[0]
    bf16 = os.environ.get("AB_DTYPE", "f32") == "bf16"   # config #4 storage: rows rounded once, f32 arithmetic on the rounded values
    if bf16:
        x = x.to(torch.bfloat16).to(torch.float32)
    g = synth.build_hnsw_graph(x, m=16, m0=32, level_seed=7)
    ix = hv.ValidatedVectorReadIndex.managed(dtype=hv.BF16 if bf16 else hv.F32,
        dim=dim, metric=hv.EUCLIDEAN, node_ids=g["node_ids"], vectors=x, l0_offsets=g["l0_offsets"],
        l0_neighbors=g["l0_neighbors"], level=g["level"], up_offsets=g["up_offsets"], up_neighbors=g["up_neighbors"],
        entry_point=g["entry_point"], max_layer=g["max_layer"], m=16, m0=32, max_batch=b)
    del x
    max_lanes = int(os.environ.get("AB_MAX_LANES", 4))
    lanes = [ix] + [ix.fork() for _ in range(max_lanes - 1)]
    bufs = [(torch.zeros(b, k, dtype=torch.int64, device=dev), torch.zeros(b, k, dtype=torch.float32, device=dev),
             torch.zeros(b, dtype=torch.int32, device=dev), torch.zeros(b, dtype=torch.int32, device=dev),
             torch.zeros(b, 4, dtype=torch.int32, device=dev)) for _ in range(max_lanes)]
    ref = None
    rows = []
    for occ in (1, 2):
        for ln in lanes:
            ln.set_occupancy(occ)
        for L in range(1, max_lanes + 1):
            for rep in range(2):
                for l in range(L):
                    for t in bufs[l]:
                        t.zero_()
                for i in range(2 * L):  # warm-up
                    l = i % L
                    lanes[l].search_batch_device(q, k, ef, *bufs[l])
                torch.cuda.synchronize()
                for l in range(L):
                    lanes[l].timing_begin(steps)
                t0 = time.perf_counter()
                for i in range(steps):
                    l = i % L
                    lanes[l].search_batch_device(qs[i % nbq] if i < steps - L else q, k, ef, *bufs[l])  # (the last step of every lane answers batch 0: the identity check)
                for l in range(L):
                    lanes[l].sync()
                dt = time.perf_counter() - t0
                kms = np.concatenate([lanes[l].timing_collect(steps) for l in range(L)])
            torch.cuda.synchronize()
            same = True
            for l in range(L):
                cur = [t.clone() for t in bufs[l]]
                if ref is None:
                    ref = cur
                same &= all(bool((cur[i] == ref[i]).all()) for i in (0, 2, 3, 4)) and bool((cur[1].view(torch.int32) == ref[1].view(torch.int32)).all())
            qst = ref[4].cpu().numpy().astype(np.int64)
            alg = qst[:, 3].sum() * dim * (2 if bf16 else 4) + qst[:, 1].sum() * 4 + b * dim * 4
            ms_step = dt * 1e3 / steps
            rows.append({"occ": occ, "lanes": L, "ms_per_step": round(ms_step, 4), "qps": round(b / ms_step * 1e3, 0),
                         "frac_hbm": round(alg / (ms_step * 1e-3) / 8e12, 4), "kernel_ms_mean": round(float(kms.mean()), 4),
                         "identical": same})
            print(json.dumps(rows[-1]), flush=True)
    if bf16:
        print(json.dumps({"dataset": dataset, "dtype": "bf16", "rows": n, "ef": ef, "dist_per_query": float(qst[:, 3].mean()), "rows_table": rows}))
        return
    f = [torch.zeros(b, k, dtype=torch.int64, device=dev), torch.zeros(b, k, dtype=torch.float32, device=dev),
         torch.zeros(b, dtype=torch.int32, device=dev), torch.zeros(b, dtype=torch.int32, device=dev)]
    ix.flat_search_batch_device(q, k, *f)
    torch.cuda.synchronize()
    th, gh = f[0].cpu().numpy(), ref[0].cpu().numpy()
    recall = sum(len(set(gh[i].tolist()) & set(th[i].tolist())) for i in range(b)) / float(b * k)
    print(json.dumps({"recall_at_10": round(recall, 4), "dataset": dataset, "ef": ef, "dist_per_query": float(qst[:, 3].mean()), "p99": int(np.percentile(qst[:, 3], 99)),
                      "max": int(qst[:, 3].max()), "rows": rows}))


if __name__ == "__main__":
    main()
