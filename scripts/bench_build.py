#!/usr/bin/env python3
"""Device HNSW build (hvx_index_build) at benchmark scale: build time, then recall@10 / work per query at ef=128 against the
exact scan.  usage: bench_build.py [rows=1000000] [max_batch=2048] [divisor=32] [dataset=embedding] [link_mode=0]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))
import numpy as np
import torch


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    bmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    div = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    dataset = sys.argv[4] if len(sys.argv) > 4 else "embedding"
    link_mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    import pyhvx as hv
    from pyhvx import synth
    dev = torch.device("cuda", 0)
    dim, b, k, ef = 768, 1024, 10, 128
    x, q = synth.corpus(dataset, n, dim, b, 20260921, dev)
    lv = synth.draw_levels(n, 16, 7)
    torch.cuda.synchronize()
    t0 = time.time()
    ix, st = hv.ValidatedVectorReadIndex.build(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64), vectors=x, levels=lv,
                                               m=16, m0=32, ef_construction=200, max_batch=bmax, batch_divisor=div, search_max_batch=b, link_mode=link_mode)
    ix.sync()
    t_build = time.time() - t0
    bufs = [torch.zeros(b, k, dtype=torch.int64, device=dev), torch.zeros(b, k, dtype=torch.float32, device=dev),
            torch.zeros(b, dtype=torch.int32, device=dev), torch.zeros(b, dtype=torch.int32, device=dev), torch.zeros(b, 4, dtype=torch.int32, device=dev)]
    s = ix.search_batch_device(q, k, ef, *bufs, want_stats=True)
    s = ix.search_batch_device(q, k, ef, *bufs, want_stats=True)
    f = [torch.zeros(b, k, dtype=torch.int64, device=dev), torch.zeros(b, k, dtype=torch.float32, device=dev),
         torch.zeros(b, dtype=torch.int32, device=dev), torch.zeros(b, dtype=torch.int32, device=dev)]
    ix.flat_search_batch_device(q, k, *f)
    torch.cuda.synchronize()
    g, t = bufs[0].cpu().numpy(), f[0].cpu().numpy()
    rec = sum(len(set(g[i].tolist()) & set(t[i].tolist())) for i in range(b)) / float(b * k)
    gr = ix.export_graph()
    deg = np.diff(gr["l0_offsets"].astype(np.int64))
    print(json.dumps({"rows": n, "dim": dim, "dataset": dataset, "max_batch": bmax, "divisor": div, "link_mode": link_mode, "build_seconds": round(t_build, 2),
                      "inserts_per_s": round(n / t_build, 1), "batches": st["batches"], "recall_at_10": round(rec, 4),
                      "distance_computations_per_query": round(s["distance_computations"] / b, 1), "kernel_ms": round(s["device_ms"], 4),
                      "degree_mean": round(float(deg.mean()), 2), "degree_full_frac": round(float((deg == 32).mean()), 3), "max_layer": gr["max_layer"]}))


if __name__ == "__main__":
    main()
