#!/usr/bin/env python3
"""Writes the arrays scripts/bench_batcher.cpp loads (synthetic corpus + bulk-built graph, pyhvx.synth), builds the
harness and runs it: concurrent single-query callers with and without the batching operator."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--threads", type=int, default=1024)
    ap.add_argument("--per-thread", type=int, default=100)
    ap.add_argument("--lanes", default="2,3")
    ap.add_argument("--dir", default="/tmp/hvx_batcher")
    ap.add_argument("--modes", default="strict,default")
    ap.add_argument("--no-direct", action="store_true", help="skip the one-call-per-query baseline (about a minute per run)")
    args = ap.parse_args()
    from pyhvx import synth
    dev = torch.device("cuda", 0)
    x, q = synth.embedding_like(args.rows, args.dim, 4096, 20260921, dev, latent=16, clusters=1024)
    g = synth.build_hnsw_graph(x, m=16, m0=32, level_seed=7)
    os.makedirs(args.dir, exist_ok=True)
    np.array([args.rows, args.dim, 16, g["entry_point"], g["max_layer"]], np.uint64).tofile(f"{args.dir}/meta.u64")
    g["node_ids"].astype(np.uint64).tofile(f"{args.dir}/ids.u64")
    x.cpu().numpy().tofile(f"{args.dir}/vectors.f32")
    q.cpu().numpy().tofile(f"{args.dir}/queries.f32")
    for name, dt in (("l0_offsets", np.uint64), ("l0_neighbors", np.uint64), ("up_offsets", np.uint64), ("up_neighbors", np.uint64)):
        g[name].astype(dt).tofile(f"{args.dir}/{name}.u64")
    g["level"].astype(np.uint16).tofile(f"{args.dir}/level.u16")
    del x, q
    torch.cuda.empty_cache()
    exe = os.path.join(ROOT, "scripts", "_bin", "bench_batcher")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "scripts", "bench_batcher.cpp"),
                           "-o", exe, "-L", os.path.join(ROOT, "helix-db_amd"), "-lhelix_vec_gfx950",
                           f"-Wl,-rpath,{os.path.join(ROOT, 'helix-db_amd')}", "-lpthread"])
    for mode in args.modes.split(","):
        for lanes in args.lanes.split(","):
            subprocess.check_call([exe, args.dir, str(args.threads), str(args.per_thread), mode, lanes] + (["nodirect"] if args.no_direct else []))


if __name__ == "__main__":
    main()
