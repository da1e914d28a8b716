#!/usr/bin/env python3
"""A/B of an experimental wave-kernel build against the production one, alternating inside ONE process on one box:
`ab_wave.py HVX_WAVE_PF` times the config #2 step with the variable unset / set, checks that ids, score bits and the
per-query counters are identical, and prints the median kernel times."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))

import numpy as np
import torch


def main():
    var = sys.argv[1] if len(sys.argv) > 1 else "HVX_WAVE_PF"
    ef = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    import pyhvx as hv
    from pyhvx import synth
    dev = torch.device("cuda", 0)
    n, dim, b, k = 1_000_000, 768, 1024, 10
    x, q = synth.embedding_like(n, dim, b, 20260921, dev, latent=16, clusters=1024)
    g = synth.build_hnsw_graph(x, m=16, m0=32, level_seed=7)
    ix = hv.ValidatedVectorReadIndex.managed(
        dim=dim, metric=hv.EUCLIDEAN, node_ids=g["node_ids"], vectors=x.cpu().numpy(), l0_offsets=g["l0_offsets"],
        l0_neighbors=g["l0_neighbors"], level=g["level"], up_offsets=g["up_offsets"], up_neighbors=g["up_neighbors"],
        entry_point=g["entry_point"], max_layer=g["max_layer"], m=16, m0=32, max_batch=b)
    ix.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    bufs = {}
    for v in (0, 1):
        bufs[v] = (torch.zeros(b, k, dtype=torch.int64, device=dev), torch.zeros(b, k, dtype=torch.float32, device=dev),
                   torch.zeros(b, dtype=torch.int32, device=dev), torch.zeros(b, dtype=torch.int32, device=dev),
                   torch.zeros(b, 4, dtype=torch.int32, device=dev))
    ms = {0: [], 1: []}
    for rep in range(6):
        for v in (0, 1):
            if v:
                os.environ[var] = "1"
            else:
                os.environ.pop(var, None)
            ids, sc, cnt, st, qst = bufs[v]
            for it in range(12):
                s = ix.search_batch_device(q, k, ef, ids, sc, cnt, st, qst, want_stats=True)
                if it >= 2 and rep >= 1:
                    ms[v].append(s["device_ms"])
    torch.cuda.synchronize()
    same = all(bool((bufs[0][i] == bufs[1][i]).all()) for i in (0, 2, 3, 4)) and \
        bool((bufs[0][1].view(torch.int32) == bufs[1][1].view(torch.int32)).all())
    out = {"variable": var, "ef": ef, "baseline_ms_median": round(float(np.median(ms[0])), 4), "variant_ms_median": round(float(np.median(ms[1])), 4),
           "baseline_ms_min": round(float(np.min(ms[0])), 4), "variant_ms_min": round(float(np.min(ms[1])), 4),
           "speedup": round(float(np.median(ms[0]) / np.median(ms[1])), 4), "results_and_counters_identical": same}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
