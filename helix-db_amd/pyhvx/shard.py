"""Vector-id-range sharding of one index across the GPUs of a node.

One process per GPU.  Rank r owns the contiguous id range `plan_shards(n, world)[r]`, holds only
those rows and an HNSW graph built over them, and searches the WHOLE query batch on its shard.  The
only data-path exchange is ONE all-gather of the per-shard top-k, packed (`b x k x (u64 id, f32 score)` +
`b` counts = 124 KB per rank for b=1024, k=10), after which every rank merges the `world` sorted
lists by the reference's Candidate order (score asc, then id asc:
crates/db/src/search/vector/model.rs:55-61) on its own device (hvx_merge_topk_packed_device).

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


def slice_candidates(sorted_unique_ids, lo: int, hi: int):
    """The candidate ids a shard holding node ids [lo, hi] keeps of a restricted search's id list ("the bitmap is
    sliced by the same id ranges"); mirrors hvx_shard_group_search_restricted_batch."""
    import numpy as np
    ids = np.asarray(sorted_unique_ids, dtype=np.uint64)
    return ids[np.searchsorted(ids, lo, side="left"): np.searchsorted(ids, hi, side="right")]


def merge_status(statuses):
    """Per-query status of a sharded step = element-wise maximum over the ranks' statuses [world][b]: a rejected query is rejected
    by every shard's validation (search.rs:1120-1125); mirrors merge_status_kernel (csrc/hvx_shard.hip)."""
    return torch.as_tensor(statuses).max(dim=0).values


def payload_bytes(b: int, k: int) -> int:
    """One rank's share of the packed exchange buffer (== hvx_topk_payload_bytes): ids [b][k] u64, scores [b][k] f32,
    counts [b] u32, padded to 8 bytes."""
    return (b * k * 12 + b * 4 + 7) & ~7


class TopkExchange:
    """ONE all-gather per step: every rank contributes one packed payload (ids, scores, counts) and receives
    [world][payload].  `ids` / `scores` / `counts` are views INTO the local payload, so a search that is handed them as
    its output buffers has nothing to copy before the exchange."""

    def __init__(self, world: int, b: int, k: int, device, group=None, stage_through_host: bool = False):
        self.world, self.b, self.k, self.group = world, b, k, group
        self.stage_through_host = stage_through_host  # a backend without device collectives (gloo): plumbing checks only
        self.payload = payload_bytes(b, k)
        self.send = torch.zeros(self.payload, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(world * self.payload, dtype=torch.uint8, device=device)
        self.ids, self.scores, self.counts = self.views(self.send, 0)

    def views(self, buf: torch.Tensor, rank: int):
        """(ids [b][k] int64, scores [b][k] f32, counts [b] int32) of one rank's payload inside `buf`."""
        b, k, o = self.b, self.k, rank * self.payload
        ids = buf[o: o + b * k * 8].view(torch.int64).view(b, k)
        scores = buf[o + b * k * 8: o + b * k * 12].view(torch.float32).view(b, k)
        counts = buf[o + b * k * 12: o + b * k * 12 + b * 4].view(torch.int32)
        return ids, scores, counts

    def gather(self, ids: torch.Tensor = None, scores: torch.Tensor = None, counts: torch.Tensor = None):
        """All-gather the local payload (after copying the given tensors into it, unless they ARE its views);
        returns the packed [world][payload] buffer."""
        for src, dst in ((ids, self.ids), (scores, self.scores), (counts, self.counts)):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        if self.world == 1:
            self.recv.copy_(self.send)
        elif self.stage_through_host and self.send.is_cuda:
            torch.cuda.current_stream(self.send.device).synchronize()
            parts = [torch.zeros(self.payload, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, self.send.cpu(), group=self.group)
            self.recv.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        return self.recv

    def gathered(self):
        """The received lists as dense tensors [world][b][k] / [world][b] (copies; for checkers and tests)."""
        parts = [self.views(self.recv, r) for r in range(self.world)]
        return (torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), torch.stack([p[2] for p in parts]))


class ShardedSearcher:
    """Search a query batch over every shard and merge on the device (the N>1 step of bench.py)."""

    def __init__(self, index, world: int, b: int, k: int, device, group=None, stage_through_host: bool = False):
        self.ix, self.world, self.b, self.k = index, world, b, k
        self.ex = TopkExchange(world, b, k, device, group, stage_through_host)
        self.m_ids = torch.zeros(b, k, dtype=torch.int64, device=device)
        self.m_scores = torch.zeros(b, k, dtype=torch.float32, device=device)
        self.m_counts = torch.zeros(b, dtype=torch.int32, device=device)
        # the search wrote the payload on the INDEX's stream and the merge kernel runs there: the exchange (torch collectives, staging
        # copies) is enqueued on the same stream, so a step needs no host synchronisation and no cross-stream event
        self.stream = torch.cuda.ExternalStream(index.stream(), device=device) if torch.device(device).type == "cuda" else None

    def outputs(self):
        """Output buffers for this rank's search: views into the exchange payload (no copy before the all-gather)."""
        return self.ex.ids, self.ex.scores, self.ex.counts

    def merge(self, ids=None, scores=None, counts=None):
        """One all-gather of this rank's lists, then the per-query merge of the `world` lists on the device."""
        if self.stream is None:
            packed = self.ex.gather(ids, scores, counts)
        else:
            with torch.cuda.stream(self.stream):
                packed = self.ex.gather(ids, scores, counts)
        self.ix.merge_topk_packed_device(self.world, self.b, self.k, packed, self.m_ids, self.m_scores, self.m_counts)
        return self.m_ids, self.m_scores, self.m_counts
