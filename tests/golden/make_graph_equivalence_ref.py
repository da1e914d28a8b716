#!/usr/bin/env python3
"""Reference side of bench.py's `graph_equivalence` object.

The benchmark graph is bulk-built (pyhvx/synth.py or the device builder); the reference builds by sequential
`insert_hnsw` (crates/db/src/search/vector/mutation.rs:787-895).  This script runs the ORACLE's restatement of that
insertion (oracle/hvx_oracle.c: orc_index_insert) over a 100 000 x 768 corpus drawn with numpy's PCG64 (identical rows on
every machine), searches it with the oracle at ef = 128, k = 10, and writes what bench.py compares its own builder
against on the same rows and queries: recall@10 vs the exact scan, distance evaluations and expansions per query, and the
layer-0 degree histogram.  ~2-3 minutes of one CPU core; run once, output committed:
    python tests/golden/make_graph_equivalence_ref.py
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "helix-db_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np

N, DIM, NQ, K, EF, M, EFC = 100_000, 768, 256, 10, 128, 16, 200
SEED, LEVEL_SEED = 20260925, 11


CL_CENTRES, CL_SIGMA, CL_SEED = 128, 0.15, 20260926   # ~780 rows per cluster: the regime of the 1M x 768 / 1 024-centre bench corpus


def main():
    import orc
    from pyhvx import synth
    clustered = len(sys.argv) > 1 and sys.argv[1] == "clustered"
    if clustered:  # SURVEY 8(d)'s clustered variant (round 4): is the recall plateau insert_hnsw's or the device builder's?
        x, q = synth.clustered_np(N, DIM, NQ, CL_SEED, centres=CL_CENTRES, sigma=CL_SIGMA)
    else:
        x, q = synth.embedding_like_np(N, DIM, NQ, SEED)
    lv = synth.draw_levels(N, M, LEVEL_SEED)
    oix = orc.Index(DIM, orc.L2SQ, kernel=orc.K_AVX_FMA_HW if orc.lib().orc_have_avxfma_hw() else orc.K_AVX_FMA, m=M, m0=2 * M,
                    ef_construction=EFC)
    t0 = time.time()
    for i in range(N):
        assert oix.insert(i, x[i], int(lv[i])) == orc.OK
    build_s = time.time() - t0
    ex = oix.export()
    deg = np.diff(ex["l0_offsets"].astype(np.int64))
    rc, ids, sc, cnt, st = oix.search_batch(q, K, EF, threads=8)
    assert rc == orc.OK
    hits = 0
    for i in range(NQ):
        rc, tid, _ = oix.flat(q[i], K)
        hits += len(set(ids[i, :cnt[i]].tolist()) & set(tid.tolist()))
    truth = [oix.flat(q[i], K)[1] for i in range(NQ)]
    sweep = []
    for ef in (128, 256, 800):
        rc, s_ids, _, s_cnt, s_st = oix.search_batch(q, K, ef, threads=8)
        assert rc == orc.OK
        h = sum(len(set(s_ids[i, :s_cnt[i]].tolist()) & set(truth[i].tolist())) for i in range(NQ))
        sweep.append({"ef": ef, "recall_at_10": round(h / float(NQ * K), 4),
                      "distance_computations_per_query": round(float(np.mean([s["distance_computations"] for s in s_st])), 1)})
    comps = synth.layer0_components(ex["l0_offsets"], ex["l0_neighbors"], ex["node_ids"], ex["entry_point"])
    name = (f"synth.clustered_np({N}, {DIM}, {NQ}, seed={CL_SEED}, centres={CL_CENTRES}, sigma={CL_SIGMA})" if clustered
            else f"synth.embedding_like_np({N}, {DIM}, {NQ}, seed={SEED})")
    out = {
        "corpus": f"{name}; levels synth.draw_levels(n, {M}, {LEVEL_SEED})",
        "ef_sweep": sweep, "layer0_components": comps,
        "builder": "oracle orc_index_insert == insert_hnsw (mutation.rs:787-895), sequential, M=16 M0=32 efC=200",
        "n": N, "dim": DIM, "queries": NQ, "k": K, "ef": EF, "build_seconds_one_core": round(build_s, 1),
        "recall_at_10": round(hits / float(NQ * K), 4),
        "distance_computations_per_query": round(float(np.mean([s["distance_computations"] for s in st])), 1),
        "expansion_steps_per_query": round(float(np.mean([s["expansion_steps"] for s in st])), 1),
        "degree_mean": round(float(deg.mean()), 2), "degree_histogram": np.bincount(deg, minlength=2 * M + 1).tolist(),
        "max_layer": int(ex["max_layer"]),
    }
    with open(os.path.join(HERE, "graph_equivalence_ref_clustered.json" if clustered else "graph_equivalence_ref.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
