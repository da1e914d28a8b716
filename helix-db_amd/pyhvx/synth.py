"""Synthetic corpora and a bulk HNSW-format graph builder for the benchmark / parity harness.

The reference builds its graph by sequential insertion (crates/db/src/search/vector/mutation.rs:787-1097;
SURVEY.md Appendix C), which at 1M x 768 takes hours on a CPU.  The search path -- the thing this
repository accelerates -- only needs *a* graph in the reference's persisted format: per node and layer a
canonical row (ascending ids, deduped, self-free, degree <= M0 on layer 0 / M above:
neighbor_set.rs:1-9, values/vectors.rs:97-111), node levels drawn with the reference's layer rule
(mod.rs:776-796) and an entry point on the top layer.  This module produces such a graph in bulk on the
GPU (or on CPU tensors for small tests):

  1. exact kNN candidates per layer (chunked GEMM + top-k, re-scored in f32),
  2. the reference's diversity heuristic `select_diverse` (mod.rs:809-856: keep c iff every kept s has
     dist(c,s) >= dist(c,node); backfill with the closest rejected) vectorised over nodes,
  3. reverse edges merged in and capped at the degree limit (the reference keeps rows symmetric and
     degree-bounded: mutation.rs:1498-1583),
  4. rows canonicalised (ascending ids).

torch is plumbing here (device memory + library GEMM/sort); nothing in this file is on the search path.
"""
from __future__ import annotations

import math

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------
# corpora
# ---------------------------------------------------------------------------------------------
def embedding_like(n: int, dim: int, n_queries: int, seed: int, device, latent: int = 16, clusters: int = 1024,
                   spread: float = 0.6, noise: float = 0.02, chunk: int = 1 << 18):
    """Unit-norm f32 rows with the low intrinsic dimension of learned text embeddings: a Gaussian
    mixture in a `latent`-dimensional space, mapped by a fixed random linear map to `dim`, plus small
    isotropic noise, L2-normalised.  Queries are held-out draws of the same distribution.
    (i.i.d. N(0,1) rows at dim=768 have no neighbour structure: distance concentration makes every ANN
    index, the reference's included, miss the recall gate at any practical ef; see DESIGN.md.)"""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    w = torch.randn(latent, dim, generator=g, device=dev) / math.sqrt(latent)
    centres = torch.randn(clusters, latent, generator=g, device=dev)

    def draw(count):
        out = torch.empty(count, dim, device=dev, dtype=torch.float32)
        for s in range(0, count, chunk):
            c = min(chunk, count - s)
            which = torch.randint(0, clusters, (c,), generator=g, device=dev)
            z = centres[which] + spread * torch.randn(c, latent, generator=g, device=dev)
            x = z @ w + noise * torch.randn(c, dim, generator=g, device=dev)
            out[s:s + c] = x / x.norm(dim=1, keepdim=True)
        return out

    return draw(n), draw(n_queries)


def topic_ordered(n: int, dim: int, n_queries: int, seed: int, device, latent: int = 12, clusters: int = 1024, spread: float = 0.6,
                  noise: float = 0.02, chunk: int = 1 << 18):
    """`embedding_like` rows in CORPUS ORDER: sorted by latent cluster, then along the first latent coordinate inside the cluster.
    The stand-in for a dump whose row order follows its source (DBpedia's entity list is ordered by entity URI: neighbouring rows
    share title prefixes / list pages / categories), i.e. a contiguous id range is topically local -- the property the reference's
    million-row prefilter benchmark leans on: its candidate sets ARE contiguous id ranges (index_lifecycle_scale.rs:592-613,1769-1776)
    and its filter-aware walk reaches members through members.  With ids assigned at random instead (`embedding_like`) a 1 000-id
    range is 0.1 % of every neighbourhood and the reference's own budgets (<= 1 200 bridge rows) end the walk after ~90 scored rows.
    Returns (rows, queries); queries are held-out draws (the benchmark itself queries with stored rows)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    w = torch.randn(latent, dim, generator=g, device=dev) / math.sqrt(latent)
    centres = torch.randn(clusters, latent, generator=g, device=dev)

    def draw(count, ordered):
        which = torch.randint(0, clusters, (count,), generator=g, device=dev)
        z = centres[which] + spread * torch.randn(count, latent, generator=g, device=dev)
        if ordered:
            o1 = torch.argsort(z[:, 0])
            o2 = torch.argsort(which[o1], stable=True)
            z = z[o1[o2]]
        out = torch.empty(count, dim, device=dev, dtype=torch.float32)
        for s in range(0, count, chunk):
            c = min(chunk, count - s)
            x = z[s:s + c] @ w + noise * torch.randn(c, dim, generator=g, device=dev)
            out[s:s + c] = x / x.norm(dim=1, keepdim=True)
        return out

    return draw(n, True), draw(n_queries, False)


def gaussian_sphere(n: int, dim: int, n_queries: int, seed: int, device):
    """SURVEY.md 8(d) C2 as literally written: i.i.d. N(0,1), L2-normalised (worst case for any ANN)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.randn(n, dim, generator=g, device=dev)
    q = torch.randn(n_queries, dim, generator=g, device=dev)
    return x / x.norm(dim=1, keepdim=True), q / q.norm(dim=1, keepdim=True)


def clustered(n: int, dim: int, n_queries: int, seed: int, device, centres: int = 1024, sigma: float = 0.15, chunk: int = 1 << 18):
    """SURVEY.md 8(d) C2, the clustered variant: `centres` Gaussian centres drawn N(0, I) in the native `dim`-d space,
    rows = centre + sigma * N(0, I), L2-normalised like the i.i.d. corpus (so cosine and L2 rank identically);
    queries are held-out draws of the same mixture.  No low-dimensional latent structure: inside a cluster the rows are
    again i.i.d. Gaussian in `dim` dimensions."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    c = torch.randn(centres, dim, generator=g, device=dev)

    def draw(count):
        out = torch.empty(count, dim, device=dev, dtype=torch.float32)
        for s in range(0, count, chunk):
            m = min(chunk, count - s)
            which = torch.randint(0, centres, (m,), generator=g, device=dev)
            x = c[which] + sigma * torch.randn(m, dim, generator=g, device=dev)
            out[s:s + m] = x / x.norm(dim=1, keepdim=True)
        return out

    return draw(n), draw(n_queries)


def embedding_like_np(n: int, dim: int, n_queries: int, seed: int, latent: int = 16, clusters: int = 1024, spread: float = 0.6,
                      noise: float = 0.02):
    """numpy (PCG64) twin of `embedding_like`: the same distribution from a generator that gives identical rows on every
    machine with or without a GPU -- used where a CPU-side fixture and a GPU-side run must see the same corpus
    (bench.py `graph_equivalence`, tests/golden/make_graph_equivalence_ref.py)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w = (rng.standard_normal((latent, dim)) / math.sqrt(latent)).astype(np.float32)
    centres = rng.standard_normal((clusters, latent)).astype(np.float32)

    def draw(count):
        which = rng.integers(0, clusters, count)
        z = centres[which] + np.float32(spread) * rng.standard_normal((count, latent)).astype(np.float32)
        x = z @ w + np.float32(noise) * rng.standard_normal((count, dim)).astype(np.float32)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)

    return draw(n), draw(n_queries)


def clustered_np(n: int, dim: int, n_queries: int, seed: int, centres: int = 1024, sigma: float = 0.15):
    """numpy (PCG64) twin of `clustered` (SURVEY 8d's clustered variant): identical rows on every machine, for fixtures whose
    CPU side (the oracle's sequential insert_hnsw) and GPU side (the device builder) must see the same corpus."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.standard_normal((centres, dim)).astype(np.float32)

    def draw(count):
        which = rng.integers(0, centres, count)
        x = c[which] + np.float32(sigma) * rng.standard_normal((count, dim)).astype(np.float32)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)

    return draw(n), draw(n_queries)


def layer0_components(l0_offsets, l0_neighbors, node_ids=None, entry_point=None):
    """Connected components of a layer-0 graph given in hvx_index_import's CSR layout: count, sizes of the largest, and how many
    nodes the entry point's component holds (what a layer-0 search can ever reach)."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components
    off = np.asarray(l0_offsets, dtype=np.int64)
    n = off.size - 1
    nb = np.asarray(l0_neighbors, dtype=np.int64)
    if node_ids is not None:  # external -> internal
        ids = np.asarray(node_ids, dtype=np.int64)
        nb = np.searchsorted(ids, nb)
        if entry_point is not None:
            entry_point = int(np.searchsorted(ids, int(entry_point)))
    g = csr_matrix((np.ones(nb.size, np.int8), nb, off), shape=(n, n))
    nc, lab = connected_components(g, directed=False)
    sizes = np.sort(np.bincount(lab))[::-1]
    out = {"components": int(nc), "largest": [int(v) for v in sizes[:8]], "nodes_outside_the_largest": int(n - sizes[0])}
    if entry_point is not None:
        out["entry_component_nodes"] = int((lab == lab[int(entry_point)]).sum())
    return out


def load_fbin(path: str, expect_rows: int = None, expect_dim: int = None) -> np.ndarray:
    """Memory-map an fbin fixture the way the reference's million-row prefilter benchmark does
    (crates/db/tests/production_support/index_lifecycle_scale.rs:497-534, env HELIX_DBPEDIA_1M_FBIN): header
    `<u32 count><u32 dimension>` little-endian, then count x dimension little-endian f32; the file length must match the
    header exactly, the header must match the expected shape when one is given, and every value must be finite
    (`get` asserts that per row, :538-547).  Returns a read-only [count][dimension] float32 view (no copy)."""
    import os
    size = os.path.getsize(path)
    if size < 8:
        raise ValueError(f"{path}: shorter than the fbin header")
    hdr = np.fromfile(path, dtype="<u4", count=2)
    count, dim = int(hdr[0]), int(hdr[1])
    if expect_rows is not None and count != expect_rows:
        raise ValueError(f"{path}: header says {count} rows, expected {expect_rows}")
    if expect_dim is not None and dim != expect_dim:
        raise ValueError(f"{path}: header says dimension {dim}, expected {expect_dim}")
    if size != 8 + count * dim * 4:
        raise ValueError(f"{path}: {size} bytes, the header implies {8 + count * dim * 4}")
    rows = np.memmap(path, dtype="<f4", mode="r", offset=8, shape=(count, dim))
    step = max(1, (1 << 22) // max(dim, 1))
    for r0 in range(0, count, step):
        if not np.isfinite(rows[r0:r0 + step]).all():
            raise ValueError(f"{path}: non-finite value in rows [{r0}, {min(count, r0 + step)})")
    return rows


def write_fbin(path: str, rows: np.ndarray) -> None:
    """The same layout, for fixtures."""
    rows = np.ascontiguousarray(rows, dtype="<f4")
    with open(path, "wb") as f:
        np.array(rows.shape, dtype="<u4").tofile(f)
        rows.tofile(f)


def quantize_fp8_rows(x: np.ndarray) -> np.ndarray:
    """The values an fp8-e4m3fn-stored index holds (numpy twin of quantize_fp8_kernel, csrc/hvx_dtype.hip): per row
    scale = max|x| / 448, code = RNE(x / scale) to e4m3fn, value = fl32(scale * decode(code)).  Harness code: lets a
    checker score the stored values of an fp8 index on the host."""
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


def corpus(name: str, n: int, dim: int, n_queries: int, seed: int, device, **kw):
    """The benchmark corpora by name: 'gaussian' (SURVEY 8d as literally written), 'clustered' (its stated clustered
    variant), 'embedding' (low intrinsic dimension, what learned embeddings look like)."""
    if name == "gaussian":
        return gaussian_sphere(n, dim, n_queries, seed, device)
    if name == "clustered":
        return clustered(n, dim, n_queries, seed, device, **kw)
    if name == "embedding":
        return embedding_like(n, dim, n_queries, seed, device, **kw)
    if name.startswith("embedding") and name[9:].isdigit():  # 'embedding32' / 'embedding64': the same family at another latent dimension
        return embedding_like(n, dim, n_queries, seed, device, latent=int(name[9:]), **kw)
    if name == "topic_ordered":
        return topic_ordered(n, dim, n_queries, seed, device, **kw)
    raise ValueError(f"unknown corpus {name!r}")


def draw_levels(n: int, m: int, seed: int) -> np.ndarray:
    """select_layer_from_uniform (mod.rs:776-796): floor(-ln(U) * ml), U clamped to
    [MIN_POSITIVE, 1-eps], capped at 63; ml = 1/ln(max(m,2)) (mod.rs:705-708)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u = rng.random(n, dtype=np.float32)
    ml = np.float32(1.0) / np.log(np.float32(max(m, 2)))
    u = np.clip(u, np.finfo(np.float32).tiny, np.float32(1.0) - np.finfo(np.float32).eps)
    s = np.floor(-np.log(u).astype(np.float32) * ml)
    s = np.where(np.isfinite(s) & (s > 0), s, 0)
    return np.minimum(s, 63).astype(np.uint16)


# ---------------------------------------------------------------------------------------------
# exact kNN candidates
# ---------------------------------------------------------------------------------------------
def _mm_f32_out(a_lp, b_lp_t):
    """[r,d] x [d,c] low-precision GEMM with f32 output when the build supports it."""
    try:
        return torch.mm(a_lp, b_lp_t, out_dtype=torch.float32)
    except (TypeError, RuntimeError):
        return None


@torch.no_grad()
def knn_candidates(x: torch.Tensor, k: int, row_chunk: int = 2048, col_chunk: int = 1 << 18, fast: bool = True):
    """k nearest rows (squared L2, self excluded) for every row of x [n,d] f32.
    Candidate generation may run the GEMM in bf16 (f32 accumulate/output) with slack; the returned
    distances are re-scored in f32 and sorted ascending.  Returns (idx int64 [n,k], d2 f32 [n,k])."""
    n, d = x.shape
    k = min(k, n - 1)
    if k <= 0:
        return (torch.zeros(n, 0, dtype=torch.int64, device=x.device), torch.zeros(n, 0, device=x.device))
    on_gpu = x.is_cuda
    slack = 16 if (fast and on_gpu) else 0
    kk = min(k + 1 + slack, n)
    sq = (x * x).sum(1)
    x_lp = x.to(torch.bfloat16) if (fast and on_gpu) else None
    if x_lp is not None and _mm_f32_out(x_lp[:8], x_lp[:8].t()) is None:
        x_lp = None
    out_idx = torch.empty(n, k, dtype=torch.int64, device=x.device)
    out_d = torch.empty(n, k, dtype=torch.float32, device=x.device)
    for r0 in range(0, n, row_chunk):
        r1 = min(n, r0 + row_chunk)
        best_s = None
        best_i = None
        for c0 in range(0, n, col_chunk):
            c1 = min(n, c0 + col_chunk)
            if x_lp is not None:
                dots = torch.mm(x_lp[r0:r1], x_lp[c0:c1].t(), out_dtype=torch.float32)
            else:
                dots = x[r0:r1] @ x[c0:c1].t()
            score = dots.mul_(2.0).sub_(sq[c0:c1][None, :])  # maximise 2ab - |b|^2
            s, i = torch.topk(score, min(kk, c1 - c0), dim=1)
            i = i + c0
            if best_s is None:
                best_s, best_i = s, i
            else:
                cs = torch.cat([best_s, s], 1)
                ci = torch.cat([best_i, i], 1)
                s2, sel = torch.topk(cs, min(kk, cs.shape[1]), dim=1)
                best_s, best_i = s2, torch.gather(ci, 1, sel)
        # exact f32 re-score of the candidates, drop self, sort ascending by (d2, id)
        rows = torch.arange(r0, r1, device=x.device)
        cand = x[best_i]  # [r,kk,d]
        d2 = (cand - x[r0:r1, None, :]).square_().sum(-1)
        d2 = torch.where(best_i == rows[:, None], torch.full_like(d2, float("inf")), d2)
        # stable order by (distance, id): sort by id first, then stable-sort by distance
        id_order = torch.argsort(best_i, dim=1)
        d2 = torch.gather(d2, 1, id_order)
        bi = torch.gather(best_i, 1, id_order)
        order = torch.argsort(d2, dim=1, stable=True)[:, :k]
        out_idx[r0:r1] = torch.gather(bi, 1, order)
        out_d[r0:r1] = torch.gather(d2, 1, order)
    return out_idx, out_d


# ---------------------------------------------------------------------------------------------
# select_diverse, vectorised
# ---------------------------------------------------------------------------------------------
@torch.no_grad()
def select_diverse(x: torch.Tensor, idx: torch.Tensor, d2: torch.Tensor, m: int, chunk: int = 4096):
    """mod.rs:809-856 over candidate lists sorted closest-first.  Returns (order [n,K] int64 = the
    candidate positions ranked kept-first-then-backfill, kept [n,K] bool in that ranked order)."""
    n, kc = idx.shape
    rank_out = torch.empty(n, kc, dtype=torch.int64, device=x.device)
    kept_out = torch.empty(n, kc, dtype=torch.bool, device=x.device)
    pos = torch.arange(kc, device=x.device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        cand = x[idx[s:e]]  # [b,K,d]
        n2 = (cand * cand).sum(-1)
        g = n2[:, :, None] + n2[:, None, :] - 2.0 * torch.bmm(cand, cand.transpose(1, 2))
        dd = d2[s:e]
        valid = torch.isfinite(dd)
        kept = torch.zeros(e - s, kc, dtype=torch.bool, device=x.device)
        cnt = torch.zeros(e - s, dtype=torch.int64, device=x.device)
        for i in range(kc):
            viol = ((g[:, i, :] < dd[:, i:i + 1]) & kept).any(1)  # strict < rejects (mod.rs:832)
            ok = (~viol) & (cnt < m) & valid[:, i]
            kept[:, i] = ok
            cnt += ok.to(torch.int64)
        key = torch.where(kept, pos[None, :], pos[None, :] + kc)
        key = torch.where(valid, key, key + 2 * kc)
        order = torch.argsort(key, dim=1)
        rank_out[s:e] = order
        kept_out[s:e] = torch.gather(kept, 1, order)
    return rank_out, kept_out


# ---------------------------------------------------------------------------------------------
# one layer: candidates -> diverse forward lists -> + reverse edges -> capped canonical rows
# ---------------------------------------------------------------------------------------------
@torch.no_grad()
def build_layer(x: torch.Tensor, m_max: int, n_cand: int, fast: bool = True):
    """Rows of one layer over the nodes x [n,d] (local ids 0..n-1).
    Returns (offsets int64 [n+1], targets int64 [E]) with every row ascending."""
    n = x.shape[0]
    dev = x.device
    if n <= 1:
        return torch.zeros(n + 1, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int64, device=dev)
    idx, d2 = knn_candidates(x, min(n_cand, n - 1), fast=fast)
    order, kept = select_diverse(x, idx, d2, m_max)
    mm = min(m_max, idx.shape[1])
    sel = torch.gather(idx, 1, order[:, :mm])
    seld = torch.gather(d2, 1, order[:, :mm])
    selk = kept[:, :mm]
    src = torch.arange(n, device=dev)[:, None].expand(n, mm)
    ok = torch.isfinite(seld)
    f_src, f_dst, f_d, f_k = src[ok], sel[ok], seld[ok], selk[ok]
    # classes: 0 diverse forward, 1 reverse of a diverse edge, 2 backfill forward, 3 reverse of backfill
    e_src = torch.cat([f_src, f_dst])
    e_dst = torch.cat([f_dst, f_src])
    e_d = torch.cat([f_d, f_d])
    e_c = torch.cat([torch.where(f_k, 0, 2), torch.where(f_k, 1, 3)]).to(torch.int64)
    # dedupe (src,dst), keeping the best class
    pair = e_src * n + e_dst
    o = torch.argsort(pair * 4 + e_c)
    pair, e_src, e_dst, e_d, e_c = pair[o], e_src[o], e_dst[o], e_d[o], e_c[o]
    first = torch.ones_like(pair, dtype=torch.bool)
    first[1:] = pair[1:] != pair[:-1]
    e_src, e_dst, e_d, e_c = e_src[first], e_dst[first], e_d[first], e_c[first]
    # rank inside each source by (class, distance, id) and cap at the degree limit
    dbits = e_d.contiguous().view(torch.int32).to(torch.int64)  # d >= 0: bit pattern is monotone
    o = torch.argsort(e_dst, stable=True)
    e_src, e_dst, e_c, dbits = e_src[o], e_dst[o], e_c[o], dbits[o]
    o = torch.argsort((e_c << 32) | dbits, stable=True)
    e_src, e_dst = e_src[o], e_dst[o]
    o = torch.argsort(e_src, stable=True)
    e_src, e_dst = e_src[o], e_dst[o]
    counts = torch.bincount(e_src, minlength=n)
    starts = torch.cumsum(counts, 0) - counts
    rank = torch.arange(e_src.numel(), device=dev) - starts[e_src]
    keep = rank < m_max
    e_src, e_dst = e_src[keep], e_dst[keep]
    # canonical rows: ascending target id inside each source
    o = torch.argsort(e_src * n + e_dst)
    e_src, e_dst = e_src[o], e_dst[o]
    counts = torch.bincount(e_src, minlength=n)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(counts, 0)
    return offsets, e_dst


@torch.no_grad()
def build_hnsw_graph(x: torch.Tensor, m: int = 16, m0: int = 32, level_seed: int = 7, n_cand0: int = 64,
                     fast: bool = True, levels: np.ndarray | None = None):
    """Bulk-build an index image in the hvx_index_import layout (node ids = row numbers).
    Returns a dict of host numpy arrays (node_ids, l0_offsets, l0_neighbors, level, up_offsets,
    up_neighbors, entry_point, max_layer); vectors are not copied."""
    n = x.shape[0]
    lv = draw_levels(n, m, level_seed) if levels is None else np.asarray(levels, np.uint16)
    off0, tgt0 = build_layer(x, max(m0, 2 * m), n_cand0, fast=fast)
    out = dict(node_ids=np.arange(n, dtype=np.uint64), level=lv,
               l0_offsets=off0.cpu().numpy().astype(np.uint64), l0_neighbors=tgt0.cpu().numpy().astype(np.uint64))
    max_layer = int(lv.max()) if n else 0
    # upper rows, stored per node: layer 1..level[node]
    per_layer = []
    for layer in range(1, max_layer + 1):
        members = np.nonzero(lv >= layer)[0]
        mt = torch.from_numpy(members).to(x.device)
        off, tgt = build_layer(x[mt], m, 2 * m, fast=fast)
        per_layer.append((members, off.cpu().numpy(), members[tgt.cpu().numpy()] if tgt.numel() else np.zeros(0, np.int64)))
    up_nodes = np.nonzero(lv >= 1)[0]
    row_base = np.zeros(n + 1, np.int64)
    row_base[1:] = np.cumsum(lv.astype(np.int64))
    n_rows = int(row_base[-1])
    row_len = np.zeros(n_rows, np.int64)
    for li, (members, off, _) in enumerate(per_layer):
        row_len[row_base[members] + li] = off[1:] - off[:-1]
    up_offsets = np.zeros(n_rows + 1, np.int64)
    up_offsets[1:] = np.cumsum(row_len)
    up_neighbors = np.zeros(int(up_offsets[-1]), np.uint64)
    for li, (members, off, tgt) in enumerate(per_layer):
        if tgt.size == 0:
            continue
        rows = row_base[members] + li
        lens = off[1:] - off[:-1]
        dst = np.repeat(up_offsets[rows], lens) + (np.arange(tgt.size) - np.repeat(off[:-1], lens))
        up_neighbors[dst] = tgt.astype(np.uint64)
    out["up_offsets"] = up_offsets.astype(np.uint64)
    out["up_neighbors"] = up_neighbors
    if n:
        top = np.nonzero(lv == max_layer)[0]
        out["entry_point"], out["max_layer"] = int(top[0]), max_layer
    else:
        out["entry_point"], out["max_layer"] = None, 0
    del up_nodes
    return out
