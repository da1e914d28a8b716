#!/usr/bin/env python3
"""A/B of the exact-scan tile kernel builds inside ONE process (tuning library: HVX_FLAT_TILE_BUILD is read per launch).
For every build: ms per scan (device timer of the call), and whether the returned ids / score bits equal the default build's.
usage: HVX_LIB_PATH=helix-db_amd/libhelix_vec_gfx950_tuning.so python scripts/tile_ab.py --dtype bf16 --builds 0,4,6"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["fp8", "bf16", "f32"])
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--builds", default="0,4,6")
    args = ap.parse_args()
    import pyhvx as hv
    from pyhvx import synth
    dev = torch.device("cuda", 0)
    n, dim, b, k = args.rows, args.dim, args.batch, args.k
    x, q = synth.embedding_like(n, dim, b, 20260924, dev, latent=24, clusters=4096)
    torch.cuda.synchronize()
    ix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64), vectors=x,
                                             l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64),
                                             max_batch=b, dtype={"fp8": hv.FP8_E4M3, "bf16": hv.BF16, "f32": hv.F32}[args.dtype])
    del x
    ref = None
    useful = 2.0 * b * n * dim
    for build in args.builds.split(","):
        if build == "0":
            os.environ.pop("HVX_FLAT_TILE_BUILD", None)
        else:
            os.environ["HVX_FLAT_TILE_BUILD"] = build
        ids = torch.zeros(b, k, dtype=torch.int64, device=dev); sc = torch.zeros(b, k, device=dev)
        cnt = torch.zeros(b, dtype=torch.int32, device=dev); st = torch.zeros(b, dtype=torch.int32, device=dev)
        ms = []
        try:
            for i in range(args.steps + 1):
                s = ix.flat_search_batch_device(q, k, ids, sc, cnt, st, want_stats=True)
                if i:
                    ms.append(s["device_ms"])
        except Exception as e:  # a failed certificate / device error of an experimental build is a result, not a crash
            print(json.dumps({"dtype": args.dtype, "build": build, "error": str(e)[:200]}), flush=True)
            continue
        got = (ids.cpu().numpy(), sc.cpu().numpy().view(np.uint32), cnt.cpu().numpy())
        if ref is None:
            ref = got
        same = bool(all(np.array_equal(a, r) for a, r in zip(got, ref)))
        m = float(np.mean(ms))
        print(json.dumps({"dtype": args.dtype, "rows": n, "dim": dim, "batch": b, "build": build, "ms_per_scan": round(m, 3), "ms_each": [round(v, 3) for v in ms],
                          "algorithmic_frac_of_bf16_peak": round(useful / m / 1e9 / 2500.0, 4), "equals_first_build": same}), flush=True)


if __name__ == "__main__":
    main()
