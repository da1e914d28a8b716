"""Deterministic fixtures restated from the reference's own tests (no reference code is imported).

Each generator cites the reference file:line it follows (paths under /root/reference/crates/db/).
"""
from __future__ import annotations

import math

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MASK64 = (1 << 64) - 1


def lifecycle_vector(entity_id: int, dim: int = 128) -> np.ndarray:
    """tests/production_support/index_lifecycle_scale.rs:410-421 `vector(entity_id)`:
    xorshift(13,7,17) seeded id+0x9e3779b97f4a7c15, component = ((s & 0xffff) - 32768)/32768."""
    state = (entity_id + 0x9E3779B97F4A7C15) & MASK64
    out = np.empty(dim, np.float32)
    for i in range(dim):
        state ^= (state << 13) & MASK64
        state ^= state >> 7
        state ^= (state << 17) & MASK64
        centered = (state & 0xFFFF) - 32768
        out[i] = np.float32(centered) / np.float32(32768.0)
    return out


def lifecycle_matrix(n: int, dim: int = 128) -> np.ndarray:
    """Vectorised `lifecycle_vector` for ids 0..n-1 (bit-identical)."""
    state = (np.arange(n, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    out = np.empty((n, dim), np.float32)
    for i in range(dim):
        state ^= state << np.uint64(13)
        state ^= state >> np.uint64(7)
        state ^= state << np.uint64(17)
        centered = (state & np.uint64(0xFFFF)).astype(np.int64) - 32768
        out[:, i] = centered.astype(np.float32) / np.float32(32768.0)
    return out


def circle_vector(entity_id: int, entity_count: int, dim: int = 2) -> np.ndarray:
    """src/search/vector/scale_contracts.rs:45-49 vector_for (dim 2); the 8-D variant used by
    tests/production_support/vector/restricted.rs pads harmonics -- see circle_vector_nd."""
    angle = math.tau * entity_id / entity_count
    return np.array([np.float32(math.cos(angle)), np.float32(math.sin(angle))], np.float32)


def skip_neighbors(entity_id: int, entity_count: int) -> list[int]:
    """src/search/vector/scale_contracts.rs:52-72: power-of-two ring links, sorted, deduped."""
    out = []
    offset = 1
    while offset < entity_count:
        forward = (entity_id - 1 + offset) % entity_count + 1
        backward = (entity_id - 1 + entity_count - offset % entity_count) % entity_count + 1
        if forward != entity_id:
            out.append(forward)
        if backward != entity_id:
            out.append(backward)
        offset *= 2
    return sorted(set(out))


def circle_index_arrays(entity_count: int):
    """Seed arrays of the scale fixture (scale_contracts.rs:95-155): ids 1..N, entry=1, layer 0 only."""
    ids = np.arange(1, entity_count + 1, dtype=np.uint64)
    ang = (math.tau * ids.astype(np.float64)) / float(entity_count)
    vec = np.stack([np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)], axis=1)
    offs = [0]
    nbrs = []
    for i in range(1, entity_count + 1):
        nb = skip_neighbors(i, entity_count)
        nbrs.extend(nb)
        offs.append(len(nbrs))
    return ids, vec, np.array(offs, np.uint64), np.array(nbrs, np.uint64)


def circle_queries(entity_count: int, query_count: int = 24):
    """scale_contracts.rs:175-180."""
    return [circle_vector(1 + qi * (entity_count // query_count), entity_count) for qi in range(query_count)]


def draw_levels(n: int, m: int, seed: int) -> np.ndarray:
    """Layer draws for synthetic builds: uniform f32 from PCG64(seed) pushed through the reference's
    select_layer_from_uniform (src/search/vector/mod.rs:776-796) -- vectorised restatement."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u = rng.random(n, dtype=np.float32)
    ml = np.float32(1.0) / np.log(np.float32(max(m, 2)))
    u = np.clip(u, np.finfo(np.float32).tiny, np.float32(1.0) - np.finfo(np.float32).eps)
    s = np.floor(-np.log(u).astype(np.float32) * ml)
    s = np.where(np.isfinite(s) & (s > 0), s, 0)
    return np.minimum(s, 63).astype(np.uint16)


def recall_at_k(got_ids, true_ids) -> float:
    hit = 0
    tot = 0
    for g, t in zip(got_ids, true_ids):
        ts = set(int(x) for x in t)
        hit += sum(1 for x in g if int(x) in ts)
        tot += len(t)
    return hit / max(tot, 1)


def merge_topk_reference(ids, scores, counts, k):
    """Checker for the multi-shard merge: per query, the k smallest of the gathered per-shard lists by
    the reference's Candidate order (score asc, then id asc; model.rs:55-61).
    ids/scores [g,b,k'], counts [g,b] -> (ids [b,k] u64, scores [b,k] f32, counts [b])."""
    ids = np.asarray(ids).astype(np.uint64)
    scores = np.asarray(scores, np.float32)
    counts = np.asarray(counts).astype(np.int64)
    g, b, _ = ids.shape
    out_i = np.zeros((b, k), np.uint64)
    out_s = np.zeros((b, k), np.float32)
    out_c = np.zeros(b, np.uint32)
    for q in range(b):
        cand = []
        for s in range(g):
            for t in range(int(counts[s, q])):
                cand.append((float(scores[s, q, t]), int(ids[s, q, t])))
        cand.sort()
        cand = cand[:k]
        out_c[q] = len(cand)
        for t, (sc, i) in enumerate(cand):
            out_i[q, t] = i
            out_s[q, t] = np.float32(sc)
    return out_i, out_s, out_c


def round_bf16(x: np.ndarray) -> np.ndarray:
    """f32 -> nearest bf16 (ties to even) -> f32: the values a bf16-stored index holds."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    lsb = (u >> np.uint64(16)) & np.uint64(1)
    r = ((u + np.uint64(0x7FFF) + lsb) >> np.uint64(16)) << np.uint64(16)
    return (r & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.float32).reshape(np.shape(x))


def quantize_fp8_rows(x: np.ndarray) -> np.ndarray:
    """The values an fp8-e4m3fn-stored index holds (twin of quantize_fp8_kernel, hvx_dtype.hip): per row
    scale = max|x| / 448, code = RNE(x / scale) to e4m3fn, value = fl32(scale * decode(code))."""
    x = np.ascontiguousarray(x, np.float32)
    amax = np.abs(x).max(axis=1).astype(np.float32)
    scale = np.where(amax > 0, amax / np.float32(448.0), np.float32(1.0)).astype(np.float32)
    y = (x / scale[:, None]).astype(np.float32)
    a = np.abs(y)
    a = np.where(a < np.float32(464.0), a, np.float32(448.0)).astype(np.float32)
    _, ex = np.frexp(a)
    e = ex.astype(np.int32) - 1
    e = np.where((a == 0) | (e < -6), -6, e)
    step = np.ldexp(np.float32(1.0), e - 3).astype(np.float32)
    v = (np.rint(a / step) * step).astype(np.float32)
    v = np.minimum(v, np.float32(448.0))
    v = np.where(y < 0, -v, v).astype(np.float32)
    return (scale[:, None] * v).astype(np.float32)
