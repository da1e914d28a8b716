"""Why does bench.py's config #4 leg overlap its lanes less than scripts/ab_lanes.py?  Same index, same kernels; toggles."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import pyhvx as hv
from pyhvx import synth
import bench

dev = torch.device("cuda", 0)
n, dim, b, k, ef, nbq = 1_250_000, 768, 1024, 10, 128, 16
x, q_all = synth.corpus("embedding", n, dim, b * nbq, 20260921, dev)
x = x.to(torch.bfloat16).to(torch.float32)
qs = [q_all[j * b:(j + 1) * b] for j in range(nbq)]
g = synth.build_hnsw_graph(x, m=16, m0=32, level_seed=7)
ix = hv.ValidatedVectorReadIndex.managed(dtype=hv.BF16, dim=dim, metric=hv.EUCLIDEAN, node_ids=g["node_ids"], vectors=x, l0_offsets=g["l0_offsets"],
        l0_neighbors=g["l0_neighbors"], level=g["level"], up_offsets=g["up_offsets"], up_neighbors=g["up_neighbors"],
        entry_point=g["entry_point"], max_layer=g["max_layer"], m=16, m0=32, max_batch=b, device=0)
L, steps = 3, 45
handles = [ix] + [ix.fork() for _ in range(L - 1)]
for h in handles: h.set_occupancy(2)
bufs = [bench.out_buffers(b, k, dev) for _ in handles]

def plain(tag, timing=True):
    for i in range(2 * L): handles[i % L].search_batch_device(qs[i % nbq], k, ef, *bufs[i % L])
    torch.cuda.synchronize()
    if timing:
        for h in handles: h.timing_begin(steps)
    t0 = time.perf_counter()
    for i in range(steps): handles[i % L].search_batch_device(qs[i % nbq], k, ef, *bufs[i % L])
    for h in handles: h.sync()
    dt = time.perf_counter() - t0
    kms = np.concatenate([h.timing_collect(steps) for h in handles]) if timing else np.zeros(1)
    print(json.dumps({"variant": tag, "ms_per_step": round(dt * 1e3 / steps, 4), "kernel_ms_each": round(float(kms.mean()), 4)}), flush=True)

plain("plain loop, timing ring")
plain("plain loop, no timing ring", timing=False)
class A: pass
args = A(); args.query_batches = 0; args.steps = steps; args.warmup = 6
ls = bench.LaneSet.__new__(bench.LaneSet)
ls.handles = handles; ls.streams = [torch.cuda.ExternalStream(h.stream(), device=dev) for h in handles]; ls.bufs = bufs; ls.sharded = None
ls.b, ls.k, ls.dev, ls.occ, ls.residency, ls.groups, ls.merged, ls.last_q = b, k, dev, 2, None, None, None, [None] * L
el, span, kms = bench.timed_steps(ls, qs, ef, steps, 6, lambda: None)
print(json.dumps({"variant": "bench.timed_steps", "ms_per_step": round(el * 1e3 / steps, 4), "span_per_step": round(span / steps, 4), "kernel_ms_each": round(float(kms.mean()), 4), "residency": ls.residency}), flush=True)
plain("plain loop again")
# a second f32 index on the device, like hnsw_leg's ix_truth
ix2 = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=g["node_ids"], vectors=x, l0_offsets=np.zeros(n + 1, np.uint64), l0_neighbors=np.zeros(0, np.uint64), device=0, max_batch=b)
plain("plain loop, with a second f32 index alive")
