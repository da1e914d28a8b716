"""Vector-id-range sharding of one index across the GPUs of a node.

One process per GPU.  Rank r owns the contiguous id range `plan_shards(n, world)[r]`, holds only
those rows and an HNSW graph built over them, and searches the WHOLE query batch on its shard.  The
only data-path exchange is the all-gather of the per-shard top-k (`b x k x (u64 id, f32 score)` +
`b` counts -- 120 KB per rank for b=1024, k=10), after which every rank merges the `world` sorted
lists by the reference's Candidate order (score asc, then id asc:
crates/db/src/search/vector/model.rs:55-61) on its own device (hvx_merge_topk_device).

`exchange_topk` is backend-agnostic (`nccl` = RCCL over xGMI on the GPU box, `gloo` in the CPU
tests); the merge itself has no CPU implementation in the product -- tests bring their own checker.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def plan_shards(n_rows: int, world: int) -> list[tuple[int, int]]:
    """Contiguous, near-equal [lo, hi) id ranges; the first `n_rows % world` shards get one extra row."""
    if world <= 0:
        raise ValueError("world must be positive")
    base, extra = divmod(n_rows, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def owner_of(node_id: int, n_rows: int, world: int) -> int:
    """Rank that owns `node_id` under plan_shards."""
    base, extra = divmod(n_rows, world)
    cut = extra * (base + 1)
    if node_id < cut:
        return node_id // (base + 1)
    return extra + (node_id - cut) // max(base, 1)


class TopkExchange:
    """Pre-allocated all-gather of per-shard top-k lists: [world][b][k] ids / scores, [world][b] counts."""

    def __init__(self, world: int, b: int, k: int, device, group=None):
        self.world, self.b, self.k, self.group = world, b, k, group
        self.ids = torch.zeros(world, b, k, dtype=torch.int64, device=device)      # u64 bit patterns
        self.scores = torch.zeros(world, b, k, dtype=torch.float32, device=device)
        self.counts = torch.zeros(world, b, dtype=torch.int32, device=device)

    def gather(self, ids: torch.Tensor, scores: torch.Tensor, counts: torch.Tensor):
        """ids [b,k] int64, scores [b,k] f32, counts [b] int32 of THIS rank -> the gathered tensors."""
        if self.world == 1:
            self.ids[0].copy_(ids); self.scores[0].copy_(scores); self.counts[0].copy_(counts)
        else:
            w, b, k = self.world, self.b, self.k  # concatenation along dim 0 == the [world][b][k] layout
            dist.all_gather_into_tensor(self.ids.view(w * b, k), ids.contiguous(), group=self.group)
            dist.all_gather_into_tensor(self.scores.view(w * b, k), scores.contiguous(), group=self.group)
            dist.all_gather_into_tensor(self.counts.view(w * b), counts.contiguous(), group=self.group)
        return self.ids, self.scores, self.counts


class ShardedSearcher:
    """Search a query batch over every shard and merge on the device (the N>1 step of bench.py)."""

    def __init__(self, index, world: int, b: int, k: int, device, group=None):
        self.ix, self.world, self.b, self.k = index, world, b, k
        self.ex = TopkExchange(world, b, k, device, group)
        self.m_ids = torch.zeros(b, k, dtype=torch.int64, device=device)
        self.m_scores = torch.zeros(b, k, dtype=torch.float32, device=device)
        self.m_counts = torch.zeros(b, dtype=torch.int32, device=device)

    def merge(self, ids, scores, counts):
        """All-gather this rank's lists and merge the `world` lists per query; returns merged tensors."""
        g_ids, g_sc, g_cnt = self.ex.gather(ids, scores, counts)
        self.ix.merge_topk_device(self.world, self.b, self.k, g_ids, g_sc, g_cnt, self.m_ids, self.m_scores, self.m_counts)
        return self.m_ids, self.m_scores, self.m_counts
