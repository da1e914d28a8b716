#!/usr/bin/env python3
"""Kernel time / recall / work of the search under different SearchParams on the config #2 corpus (1M x 768):
strict arm, production default, and AD-kernel configurations that isolate the cost of each non-strict stage."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--metric", default="cosine", choices=["l2", "cosine"])
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import pyhvx as hv
    from pyhvx import synth
    dev = torch.device("cuda", 0)
    n, dim, b, k = args.rows, args.dim, 1024, 10
    x, q = synth.embedding_like(n, dim, b, 20260921, dev, latent=16, clusters=1024)
    if args.metric == "cosine":
        x = torch.nn.functional.normalize(x, dim=1)
        q = torch.nn.functional.normalize(q, dim=1)
    g = synth.build_hnsw_graph(x, m=16, m0=32, level_seed=7)
    ix = hv.ValidatedVectorReadIndex.managed(
        dim=dim, metric=hv.COSINE if args.metric == "cosine" else hv.EUCLIDEAN, node_ids=g["node_ids"], vectors=x.cpu().numpy(),
        l0_offsets=g["l0_offsets"], l0_neighbors=g["l0_neighbors"], level=g["level"], up_offsets=g["up_offsets"],
        up_neighbors=g["up_neighbors"], entry_point=g["entry_point"], max_layer=g["max_layer"], m=16, m0=32, max_batch=b)
    ix.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ix.set_simhash()
    d_ids = torch.zeros(b, k, dtype=torch.int64, device=dev); d_sc = torch.zeros(b, k, dtype=torch.float32, device=dev)
    d_cnt = torch.zeros(b, dtype=torch.int32, device=dev); d_st = torch.zeros(b, dtype=torch.int32, device=dev)
    d_qst = torch.zeros(b, 4, dtype=torch.int32, device=dev)
    d_ast = torch.zeros(b, C.sizeof(hv.AdaptiveStats), dtype=torch.uint8, device=dev)
    t_ids = torch.zeros_like(d_ids)
    ix.flat_search_batch_device(q, k, t_ids, d_sc, d_cnt, d_st)
    torch.cuda.synchronize()
    truth = t_ids.cpu().numpy()
    P = hv.SearchParams
    cases = [
        ("strict ef=100 (strict kernel)", P(k).with_ef(100)),
        ("AD kernel, nothing active: Always, threshold 0 (config), sampling 1.0, pre 1.0", P.new(k).with_simhash_mode(hv.SIMHASH_ALWAYS)
         .with_pre_simhash_sampling_ratio(1.0).with_simhash_sampling_ratio(1.0), {"simhash_threshold": 0}),
        ("AD kernel, sampling only: Off + pre-sampling 0.5", P.new(k).with_simhash_mode(hv.SIMHASH_OFF).with_pre_simhash_sampling_ratio(0.5)),
        ("AD kernel, fixed filter only: Always thr 43, sampling 1.0, pre 1.0", P.new(k).with_simhash_mode(hv.SIMHASH_ALWAYS)
         .with_pre_simhash_sampling_ratio(1.0).with_simhash_sampling_ratio(1.0)),
        ("production default SearchParams::new(10)", P.new(k)),
        ("throughput_profile_floor_92", P.throughput_profile_floor_92(k)),
    ]
    for case in cases:
        name, p = case[0], case[1]
        if len(case) > 2:
            ix.set_simhash(hv.SimHashConfig.default(**case[2]), node_hashes=None)
        ms = []
        ms_stats = []
        for it in range(args.steps + 3):
            st = ix.search_batch_params_device(q, p, d_ids, d_sc, d_cnt, d_st, d_qst, None, want_stats=True)
            if it >= 3:
                ms.append(st["device_ms"])
        for it in range(5):  # the build that also collects the SearchStats of the non-strict stages
            st = ix.search_batch_params_device(q, p, d_ids, d_sc, d_cnt, d_st, d_qst, d_ast, want_stats=True)
            ms_stats.append(st["device_ms"])
        torch.cuda.synchronize()
        got = d_ids.cpu().numpy()
        rec = sum(len(set(got[i].tolist()) & set(truth[i].tolist())) for i in range(b)) / float(b * k)
        qs = d_qst.cpu().numpy().astype(np.int64)
        ast = np.frombuffer(d_ast.cpu().numpy().tobytes(), dtype=np.dtype(hv.AdaptiveStats))
        print(json.dumps({"params": name, "ef": p.ef, "kernel_ms": round(float(np.mean(ms)), 4), "kernel_ms_with_stats": round(float(np.mean(ms_stats[1:])), 4), "recall_at_10": round(rec, 4),
                          "expansions": round(float(qs[:, 0].mean()), 1), "expansions_max": int(qs[:, 0].max()),
                          "distance_computations": round(float(qs[:, 3].mean()), 1), "distance_computations_max": int(qs[:, 3].max()),
                          "simhash_examined": round(float(ast["simhash_examined"].mean()), 1),
                          "simhash_filtered": round(float(ast["simhash_filtered"].mean()), 1),
                          "pre_dropped": round(float(ast["pre_simhash_sample_dropped"].mean()), 1),
                          "bypass_expansions": round(float(ast["simhash_bypass_expansions"].mean()), 2),
                          "rng_words": round(float(ast["rng_words"].mean()), 1)}), flush=True)
        if len(case) > 2:
            ix.set_simhash()


if __name__ == "__main__":
    main()
