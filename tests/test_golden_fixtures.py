"""Frozen vectors under tests/golden/: the oracle must reproduce them (CPU); the device path must reproduce the .npz
outputs through the C ABI (GPU) -- without the live oracle in the loop."""
import json
import os

import numpy as np
import pytest

import fixtures as fx

GOLD = os.path.join(fx.ROOT, "tests", "golden")
NPZ = ["hnsw_l2_400x128.npz", "hnsw_cosine_400x128.npz"]
STAT_KEYS = ("expansion_steps", "neighbors_examined", "vectors_loaded", "distance_computations")
AD_KEYS = STAT_KEYS + ("simhash_filtered", "simhash_examined", "simhash_passed_before_sampling",
                       "simhash_passed_after_sampling", "pre_simhash_sample_kept", "pre_simhash_sample_dropped",
                       "simhash_bypass_expansions", "simhash_skipped_candidates", "rng_words")


def test_oracle_reproduces_the_reference_known_answers(orc):
    ka = json.load(open(os.path.join(GOLD, "reference_known_answers.json")))
    s = ka["simhash_known_answer"]
    assert orc.SimHasher(s["dim"], s["seed"]).hash(s["vector"]) == int(s["hash_hex"], 16)
    for bits_hex, code_hex in ka["order_code"]["cases"]:
        assert orc.order_code(int(bits_hex, 16)) == int(code_hex, 16)
    p = ka["phase0_pipeline"]
    ix = orc.Index(2, orc.COSINE, m=p["m"], m0=p["m0"], ef_construction=p["ef_construction"])
    for nid, v, lvl in p["inserts"]:
        assert ix.insert(nid, v, lvl) == orc.OK
    assert list(ix.entry()) == p["entry"]
    rc, ids, sc, st = ix.search(p["query"], p["k"], p["ef"], with_stats=True)
    assert rc == orc.OK and ids.tolist() == p["ids"] and st == p["stats"]
    assert sc.view(np.uint32).tolist() == np.array(p["scores"], np.float32).view(np.uint32).tolist()
    m = ka["metric_definitions"]
    assert orc.distance(orc.COSINE, m["cosine"][0], m["cosine"][1]) == np.float32(m["cosine"][2])
    assert orc.distance(orc.L2SQ, m["euclidean"][0], m["euclidean"][1]) == np.float32(m["euclidean"][2])
    assert orc.distance(orc.L1, m["manhattan"][0], m["manhattan"][1]) == np.float32(m["manhattan"][2])
    qs = ka["query_seed"]
    h, e, ef = int(qs["query_simhash_hex"], 16), qs["entry_point"], qs["ef"]
    rotl = lambda v, r: ((v << r) | (v >> (64 - r))) & (2**64 - 1)
    assert orc.lib().orc_query_seed(h, e, ef) == h ^ rotl(e, 17) ^ rotl(ef, 7)
    for case in ka["policy"]["cases"]:
        c = dict(ka["policy"]["context"])
        c.update(case.get("context_override", {}))
        d = orc.policy_decide(**case["policy"], **c)
        for key, want in case["expect"].items():
            assert np.float32(getattr(d, key)) == np.float32(want), (case, key)


def _load(name):
    z = np.load(os.path.join(GOLD, name))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", NPZ)
def test_oracle_reproduces_the_frozen_vectors(orc, name):
    z = _load(name)
    metric, k, ef, dim = int(z["metric"]), int(z["k"]), int(z["ef"]), z["vectors"].shape[1]
    ix = orc.Index(dim, metric, m=16, m0=32)
    assert ix.seed(z["node_ids"], z["vectors"], z["l0_offsets"], z["l0_neighbors"], z["level"], z["up_offsets"], z["up_neighbors"],
                   entry_point=int(z["entry_point"]), max_layer=int(z["max_layer"])) == orc.OK
    ix.set_simhash(42)
    assert ix.get_simhash().tolist() == z["node_simhash"].tolist()
    p = orc.SearchParams.new(k)
    for i, q in enumerate(z["queries"]):
        assert ix.query_simhash(q) == int(z["query_simhash"][i])
        rc, ids, sc, st = ix.search(q, k, ef, with_stats=True)
        assert rc == orc.OK and ids.tolist() == z["strict_ids"][i].tolist()
        assert sc.view(np.uint32).tolist() == z["strict_score_bits"][i].tolist()
        assert [st[x] for x in STAT_KEYS] == z["strict_stats"][i].tolist()
        rc, ids, sc = ix.flat(q, k)
        assert ids.tolist() == z["flat_ids"][i].tolist() and sc.view(np.uint32).tolist() == z["flat_score_bits"][i].tolist()
        rc, ids, sc, st = ix.search_params(q, p, with_stats=True)
        assert rc == orc.OK and ids.tolist() == z["default_ids"][i].tolist()
        assert sc.view(np.uint32).tolist() == z["default_score_bits"][i].tolist()
        assert [st[x] for x in AD_KEYS] == z["default_stats"][i].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NPZ)
def test_device_reproduces_the_frozen_vectors(name):
    import pyhvx as hv
    z = _load(name)
    metric, k, ef, dim = int(z["metric"]), int(z["k"]), int(z["ef"]), z["vectors"].shape[1]
    gix = hv.ValidatedVectorReadIndex.managed(
        dim=dim, metric=metric, node_ids=z["node_ids"], vectors=z["vectors"], l0_offsets=z["l0_offsets"],
        l0_neighbors=z["l0_neighbors"], level=z["level"], up_offsets=z["up_offsets"], up_neighbors=z["up_neighbors"],
        entry_point=int(z["entry_point"]), max_layer=int(z["max_layer"]), m=16, m0=32)
    gix.set_simhash()
    assert gix.get_simhash().tolist() == z["node_simhash"].tolist()
    q = z["queries"]
    ids, sc, cnt, st, per_query, _ = gix.search_batch_with_stats(q, hv.SearchParams(k).with_ef(ef))
    assert (cnt == k).all() and not st.any()
    assert ids.tolist() == z["strict_ids"].tolist() and sc.view(np.uint32).tolist() == z["strict_score_bits"].tolist()
    assert [[d[x] for x in STAT_KEYS] for d in per_query] == z["strict_stats"].tolist()
    fid, fsc, fcnt, _ = gix.flat_search_batch(q, k)
    assert fid.tolist() == z["flat_ids"].tolist() and fsc.view(np.uint32).tolist() == z["flat_score_bits"].tolist()
    ids, sc, cnt, st, per_query, _ = gix.search_batch_with_stats(q, hv.SearchParams.new(k))
    assert (cnt == k).all() and not st.any()
    assert ids.tolist() == z["default_ids"].tolist() and sc.view(np.uint32).tolist() == z["default_score_bits"].tolist()
    assert [[d[x] for x in AD_KEYS] for d in per_query] == z["default_stats"].tolist()
