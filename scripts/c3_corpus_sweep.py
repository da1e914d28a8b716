#!/usr/bin/env python3
"""configs[2] stand-in: which synthetic corpus lets the REFERENCE'S OWN plan (restricted.rs:426-453: exact <= 256 ids, the
filter-aware walk above) pass the reference's gates (index_lifecycle_scale.rs:1924-1927,2001-2011: recall@10 >= 0.92 in every
group, <= 800 distance computations)?  Builds 1M x 1536 per variant on the device, runs the benchmark's four groups with the
planned strategy and the exact scan, prints recall / counters per group.  usage: c3_corpus_sweep.py name:latent:clusters:spread ..."""
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

n, dim, k, ef, nq = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("DIM", 1536)), 10, 100, 32
dev = torch.device("cuda", 0)
for spec in sys.argv[1:]:
    name, latent, clusters, spread = spec.split(":")
    kw = {} if name == "clustered" else dict(latent=int(latent), clusters=int(clusters), spread=float(spread))
    x, _ = synth.corpus(name, n, dim, 1, 20260923, dev, **kw)
    torch.cuda.synchronize()
    t0 = time.time()
    ix, _ = hv.ValidatedVectorReadIndex.build(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64), vectors=x, levels=synth.draw_levels(n, 16, 11),
                                              m=16, m0=32, ef_construction=200, device=0, search_max_batch=nq, scatter=os.environ.get('SCATTER', '1') == '1')
    ix.sync()
    tb = time.time() - t0
    ix.set_simhash()
    off = np.arange(n + 1, dtype=np.uint64)
    tgt = ((np.arange(n, dtype=np.uint64) + np.uint64(n // 2)) % np.uint64(n)).astype(np.uint64)
    g = hv.Graph(n, off, tgt)
    q = x[[(int(n * 0.8) + j * (n // 10) // nq) % n for j in range(nq)]].cpu().numpy().copy()
    out = {"corpus": spec, "build_s": round(tb, 1), "groups": []}
    for size, start in ((100, 0), (1000, 100), (10000, 1100), (100000, 11100)):
        src = np.arange(start, start + size, dtype=np.uint64)
        fid, fsc, fcnt, ncand, fst = ix.prefilter_search_batch(g, q, hv.SearchParams(k).with_ef(ef), src, direction=hv.DIR_OUT)
        rp = hv.RestrictedParams.new(k, ef)
        ms = []
        for r in range(4):
            pid, psc, pcnt, ncand, prs, pst = ix.prefilter_search_batch_params(g, q, rp, src, direction=hv.DIR_OUT)
            ms.append(pst["device_ms"])
        rec = sum(len(set(pid[i, :pcnt[i]].tolist()) & set(fid[i, :fcnt[i]].tolist())) for i in range(nq)) / float(nq * k)
        dc = [r_["distance_computations"] for r_ in prs]
        terms = {}
        for r_ in prs:
            terms[str(r_["termination"])] = terms.get(str(r_["termination"]), 0) + 1
        out["groups"].append({"candidates": size, "recall": round(rec, 4), "dc_mean": round(float(np.mean(dc)), 1), "dc_max": int(max(dc)),
                              "directory_rows_max": int(max(r_["directory_rows"] for r_ in prs)), "terminations": terms, "walk_kernel_ms": round(float(np.median(ms[1:])), 3)})
    out["all_pass"] = all(gp["recall"] >= 0.92 and gp["dc_max"] <= 800 for gp in out["groups"])
    print(json.dumps(out), flush=True)
    ix.close()
    del x
    torch.cuda.empty_cache()
