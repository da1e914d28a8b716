#!/usr/bin/env python3
"""Is the recall plateau of the 1M x 768 `clustered` corpus (SURVEY 8(d)'s stated variant: 1 024 Gaussian centres, sigma 0.15) a property
of the reference's algorithm or of the device builder?  (VERDICT r3 missing #3 / weak #1c.)  The CPU ORACLE builds the graph by
sequential insert_hnsw (mutation.rs:787-895) over the numpy twin of the generator at the bench's size and searches it at ef 128 / 256 /
800: recall@10 against the exact scan, distance evaluations, the layer-0 components.  ~30-40 minutes of one CPU core, no GPU.
usage: oracle_clustered_1m_study.py [rows=1000000] [centres=1024]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "helix-db_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import orc
from pyhvx import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
CENTRES = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
DIM, NQ, K, M, EFC = 768, 256, 10, 16, 200
x, q = synth.clustered_np(N, DIM, NQ, 20260927, centres=CENTRES, sigma=0.15)
lv = synth.draw_levels(N, M, 11)
oix = orc.Index(DIM, orc.L2SQ, kernel=orc.K_AVX_FMA_HW if orc.lib().orc_have_avxfma_hw() else orc.K_AVX_FMA, m=M, m0=2 * M, ef_construction=EFC)
t0 = time.time()
for i in range(N):
    assert oix.insert(i, x[i], int(lv[i])) == orc.OK
    if (i + 1) % 100000 == 0:
        print(f"[{time.time() - t0:.0f}s] {i + 1} rows inserted", file=sys.stderr, flush=True)
build_s = time.time() - t0
ex = oix.export()
deg = np.diff(ex["l0_offsets"].astype(np.int64))
truth = [oix.flat(q[i], K)[1] for i in range(NQ)]
sweep = []
for ef in (128, 256, 800):
    rc, ids, _, cnt, st = oix.search_batch(q, K, ef, threads=8)
    assert rc == orc.OK
    h = sum(len(set(ids[i, :cnt[i]].tolist()) & set(truth[i].tolist())) for i in range(NQ))
    sweep.append({"ef": ef, "recall_at_10": round(h / float(NQ * K), 4),
                  "distance_computations_per_query": round(float(np.mean([s["distance_computations"] for s in st])), 1)})
comps = synth.layer0_components(ex["l0_offsets"], ex["l0_neighbors"], ex["node_ids"], ex["entry_point"])
print(json.dumps({"corpus": f"synth.clustered_np({N}, {DIM}, {NQ}, seed=20260927, centres={CENTRES}, sigma=0.15)",
                  "builder": "oracle orc_index_insert == insert_hnsw (mutation.rs:787-895), sequential, M=16 M0=32 efC=200",
                  "build_seconds_one_core": round(build_s, 1), "ef_sweep": sweep, "layer0_components": comps,
                  "degree_mean": round(float(deg.mean()), 2), "max_layer": int(ex["max_layer"])}))
