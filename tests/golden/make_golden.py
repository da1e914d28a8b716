#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small seeded inputs + the ORACLE's outputs for them, committed so that the device
path is also checked against frozen vectors (not only against the live oracle), and so that a change of the oracle
itself shows up as a diff.

The reference is Rust and cannot run in this image (no cargo/rustc), so these vectors come from
the C restatement in oracle/ -- which is pinned against the reference's own known answers by tests/test_oracle_golden.py
and tests/test_oracle_adaptive.py (and those known answers are frozen in reference_known_answers.json, transcribed from
the reference's tests with file:line).  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import fixtures as fx  # noqa: E402
import orc  # noqa: E402

STAT_KEYS = ("expansion_steps", "neighbors_examined", "vectors_loaded", "distance_computations")
AD_KEYS = STAT_KEYS + ("simhash_filtered", "simhash_examined", "simhash_passed_before_sampling",
                       "simhash_passed_after_sampling", "pre_simhash_sample_kept", "pre_simhash_sample_dropped",
                       "simhash_bypass_expansions", "simhash_skipped_candidates", "rng_words")


def make(name, metric, n, dim, seed, k, ef):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((12, dim)).astype(np.float32)
    data = (centers[rng.integers(0, 12, n)] + 0.6 * rng.standard_normal((n, dim))).astype(np.float32)
    lv = fx.draw_levels(n, 16, seed=seed + 1)
    ix = orc.Index(dim, metric, m=16, m0=32, ef_construction=60)
    for i in range(n):
        assert ix.insert(1000 + 3 * i, data[i], int(lv[i])) == orc.OK      # non-contiguous external ids
    ix.set_simhash(42)
    ex = ix.export()
    q = (centers[rng.integers(0, 12, 16)] + 0.6 * rng.standard_normal((16, dim))).astype(np.float32)
    out = {kk: np.asarray(v) for kk, v in ex.items() if kk not in ("entry_point", "max_layer")}
    out["entry_point"] = np.uint64(ex["entry_point"]); out["max_layer"] = np.uint32(ex["max_layer"])
    out["metric"] = np.uint32(metric); out["k"] = np.uint32(k); out["ef"] = np.uint32(ef)
    out["queries"] = q
    out["node_simhash"] = ix.get_simhash()
    out["query_simhash"] = np.array([ix.query_simhash(q[i]) for i in range(16)], np.uint64)
    s_ids = np.zeros((16, k), np.uint64); s_sc = np.zeros((16, k), np.float32); s_st = np.zeros((16, len(STAT_KEYS)), np.uint32)
    f_ids = np.zeros((16, k), np.uint64); f_sc = np.zeros((16, k), np.float32)
    p_ids = np.zeros((16, k), np.uint64); p_sc = np.zeros((16, k), np.float32); p_st = np.zeros((16, len(AD_KEYS)), np.uint32)
    p = orc.SearchParams.new(k)
    for i in range(16):
        rc, ids, sc, st = ix.search(q[i], k, ef, with_stats=True)
        assert rc == orc.OK and ids.size == k
        s_ids[i], s_sc[i], s_st[i] = ids, sc, [st[x] for x in STAT_KEYS]
        rc, ids, sc = ix.flat(q[i], k)
        f_ids[i], f_sc[i] = ids, sc
        rc, ids, sc, st = ix.search_params(q[i], p, with_stats=True)
        assert rc == orc.OK and ids.size == k
        p_ids[i], p_sc[i], p_st[i] = ids, sc, [st[x] for x in AD_KEYS]
    out.update(strict_ids=s_ids, strict_score_bits=s_sc.view(np.uint32), strict_stats=s_st,
               flat_ids=f_ids, flat_score_bits=f_sc.view(np.uint32),
               default_ids=p_ids, default_score_bits=p_sc.view(np.uint32), default_stats=p_st)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {kk: (v.shape if hasattr(v, "shape") else v) for kk, v in out.items() if kk.endswith("ids")})


if __name__ == "__main__":
    make("hnsw_l2_400x128", orc.L2SQ, 400, 128, 101, 10, 64)
    make("hnsw_cosine_400x128", orc.COSINE, 400, 128, 202, 10, 64)
