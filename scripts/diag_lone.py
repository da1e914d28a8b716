#!/usr/bin/env python3
"""Where does a LONE batch spend its time?  Per-query wavefront lifetimes (in-kernel 100 MHz clock) of the one-wavefront kernel and
the owner / gatherer kernel on the bench workload (1M x 768, ef 128, batch 1024), against each query's distance evaluations."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))
import numpy as np
import torch
import pyhvx as hv
from pyhvx import synth

n, dim, b, k, ef = int(os.environ.get("ROWS", 1_000_000)), 768, 1024, 10, int(os.environ.get("EF", 128))
dev = torch.device("cuda", 0)
x, q = synth.corpus("embedding", n, dim, b, 20260921, dev)
torch.cuda.synchronize()
lv = synth.draw_levels(n, 16, 7)
ix, _ = hv.ValidatedVectorReadIndex.build(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64), vectors=x, levels=lv, m=16, m0=32,
                                          ef_construction=200, device=0, search_max_batch=b)
ix.sync()
ix.set_occupancy(1)
bufs = [torch.zeros(b, k, dtype=torch.int64, device=dev), torch.zeros(b, k, dtype=torch.float32, device=dev),
        torch.zeros(b, dtype=torch.int32, device=dev), torch.zeros(b, dtype=torch.int32, device=dev), torch.zeros(b, 4, dtype=torch.int32, device=dev)]
for name, opt in (("one wavefront per query", 1), ("owner + 1 gatherer", 3), ("owner + 3 gatherers", 0)):
    ix.set_option(hv.OPT_HNSW_PAIR, opt)
    reps = 10
    for _ in range(3):
        ix.search_batch_device(q, k, ef, *bufs, want_stats=False)
        ix.sync()
    ix.timing_begin(reps)
    for _ in range(reps):
        ix.search_batch_device(q, k, ef, *bufs, want_stats=False)
        ix.sync()
    wc = ix.wave_clocks(reps, b).astype(np.int64)[-1]   # (before timing_collect: collecting restarts the ring)
    ms = ix.timing_collect(reps)
    qst = bufs[4].cpu().numpy().astype(np.int64)
    life = (wc[:, 1] - wc[:, 0]) / 100.0           # us
    start = (wc[:, 0] - wc[:, 0].min()) / 100.0
    end = (wc[:, 1] - wc[:, 0].min()) / 100.0
    dc, ex = qst[:, 3], qst[:, 0]
    pc = lambda a, p: round(float(np.percentile(a, p)), 1)
    order = np.argsort(life)
    out = {"kernel": name, "kernel_ms_mean": round(float(ms.mean()), 4), "kernel_ms_min": round(float(ms.min()), 4),
           "lifetime_us": {p: pc(life, p) for p in (1, 10, 50, 90, 99, 100)}, "start_us_p99": pc(start, 99), "end_us": {p: pc(end, p) for p in (50, 90, 99, 100)},
           "us_per_expansion": {p: pc(life / ex, p) for p in (10, 50, 90, 99)},
           "us_per_distance": {p: pc(life / dc, p) for p in (10, 50, 90)},
           "corr_lifetime_distance_computations": round(float(np.corrcoef(life, dc)[0, 1]), 3),
           "corr_lifetime_expansions": round(float(np.corrcoef(life, ex)[0, 1]), 3),
           "slowest_32_queries": {"distance_computations_mean": round(float(dc[order[-32:]].mean()), 1), "expansions_mean": round(float(ex[order[-32:]].mean()), 1),
                                  "lifetime_us_mean": round(float(life[order[-32:]].mean()), 1)},
           "all_queries": {"distance_computations_mean": round(float(dc.mean()), 1), "expansions_mean": round(float(ex.mean()), 1),
                           "lifetime_us_mean": round(float(life.mean()), 1)},
           # bytes still to be read after time t: how saturated is the memory system in the tail?
           "fraction_of_wave_time_after_p50_end": round(float(np.clip(end - np.percentile(end, 50), 0, None).sum() / life.sum()), 3)}
    print(json.dumps(out), flush=True)
