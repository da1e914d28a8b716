#!/usr/bin/env python3
"""The production-default parameters (SearchParams::new(k)) and the strict arm at the same beam width on execution lanes:
metric x arm x queries-per-SIMD x lanes on the 1M x 768 headline corpus and graph, one process, one box.
usage: ab_ad_lanes.py [steps=24]      (HVX_LIB_PATH selects the library build)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    import pyhvx as hv
    from pyhvx import synth
    dev = torch.device("cuda", 0)
    n, dim, b, k = int(os.environ.get("AB_ROWS", 1_000_000)), int(os.environ.get("AB_DIM", 768)), 1024, 10
    nbq = 16
    x, q_all = synth.corpus("embedding", n, dim, b * nbq, 20260921, dev)
    qs = [q_all[j * b:(j + 1) * b] for j in range(nbq)]
    torch.cuda.synchronize()
    lv = synth.draw_levels(n, 16, 7)
    bix, _ = hv.ValidatedVectorReadIndex.build(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64), vectors=x, levels=lv, m=16, m0=32,
                                               ef_construction=200, max_batch=2048, batch_divisor=32, device=0, search_max_batch=b)
    bix.sync()
    g = bix.export_graph()
    g["node_ids"] = np.arange(n, dtype=np.uint64)
    bix.close()
    max_lanes = 4
    only = os.environ.get("AB_ONLY", "")
    for mname, metric in (("l2", hv.EUCLIDEAN), ("cosine", hv.COSINE)):
        if only and mname not in only:
            continue
        bf16 = os.environ.get("AB_DTYPE", "f32") == "bf16"
        ix = hv.ValidatedVectorReadIndex.managed(dtype=hv.BF16 if bf16 else hv.F32, dim=dim, metric=metric, node_ids=g["node_ids"], vectors=x, l0_offsets=g["l0_offsets"],
                                                 l0_neighbors=g["l0_neighbors"], level=g["level"], up_offsets=g["up_offsets"], up_neighbors=g["up_neighbors"],
                                                 entry_point=g["entry_point"], max_layer=g["max_layer"], m=16, m0=32, max_batch=b)
        ix.set_simhash()
        lanes = [ix] + [ix.fork() for _ in range(max_lanes - 1)]
        bufs = [(torch.zeros(b, k, dtype=torch.int64, device=dev), torch.zeros(b, k, dtype=torch.float32, device=dev),
                 torch.zeros(b, dtype=torch.int32, device=dev), torch.zeros(b, dtype=torch.int32, device=dev),
                 torch.zeros(b, 4, dtype=torch.int32, device=dev)) for _ in range(max_lanes)]
        pp = hv.SearchParams.new(k)
        for arm in ("default", "strict"):
            if only and arm not in only:
                continue
            for occ in ((1,) if os.environ.get("AB_QUICK") else (1, 2)):
                for ln in lanes:
                    ln.set_occupancy(occ)
                for L in ((1,) if os.environ.get("AB_QUICK") else (1, 2, 4)):
                    def step(i):
                        l = i % L
                        q = qs[i % nbq]
                        if arm == "default":
                            lanes[l].search_batch_params_device(q, pp, bufs[l][0], bufs[l][1], bufs[l][2], bufs[l][3], bufs[l][4], None)
                        else:
                            lanes[l].search_batch_device(q, k, pp.ef, *bufs[l])
                    best = None
                    for rep in range(2):
                        for i in range(2 * L):
                            step(i)
                        torch.cuda.synchronize()
                        for l in range(L):
                            lanes[l].timing_begin(steps)
                        t0 = time.perf_counter()
                        for i in range(steps):
                            step(i)
                        for l in range(L):
                            lanes[l].sync()
                        dt = time.perf_counter() - t0
                        kms = np.concatenate([lanes[l].timing_collect(steps) for l in range(L)])
                        if best is None or dt < best[0]:
                            best = (dt, kms)
                    dt, kms = best
                    qst = bufs[0][4].cpu().numpy().astype(np.int64)
                    alg = qst[:, 3].sum() * dim * (2 if os.environ.get('AB_DTYPE', 'f32') == 'bf16' else 4) + qst[:, 1].sum() * 4 + b * dim * 4
                    ms = dt * 1e3 / steps
                    print(json.dumps({"metric": mname, "arm": arm, "occ": occ, "lanes": L, "ms_per_step": round(ms, 4), "qps": round(b / ms * 1e3),
                                      "frac_hbm": round(alg / (ms * 1e-3) / 8e12, 4), "kernel_ms_each": round(float(kms.mean()), 4),
                                      "dist_per_query": round(float(qst[:, 3].mean()), 1)}), flush=True)
        for ln in lanes[1:]:
            ln.close()
        ix.close()


if __name__ == "__main__":
    main()
