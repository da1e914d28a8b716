"""Device-index registry: when may a request use the GPU copy of an index?

Mirrors the reference's resident-cache rules (crates/db/src/search/vector/memory_registry.rs:38-45 identity,
read_index.rs:43-71 `ValidatedVectorReadIndex::managed`, memory_store.rs:611-613 `is_visible_to_snapshot`): an entry is
keyed by the COMPLETE generation identity, is hydrated by exactly one owner, becomes Ready with the database sequence
it was hydrated at, and is attached to a request only if

  * the request's visibility is comparable (a snapshot sequence is known),
  * an entry exists for exactly that identity and is Ready (not hydrating, retiring or closed),
  * its `visible_seq == snapshot_seq`, and
  * the request's distance is the generation's distance (MetricMismatch otherwise).

Every other case is a SAFE FALLBACK to the storage path (`attach` returns None): correctness never depends on the device
copy being fresh.  Read guards fence retirement: `retire` frees the device index only after the last guard is dropped.
Host logic only (no GPU needed); the index objects are whatever `hydrate` returns (a pyhvx.ValidatedVectorReadIndex).
"""
from __future__ import annotations

import threading
from dataclasses import dataclass
from typing import Callable, Dict, Optional

HYDRATING, READY, RETIRING, CLOSED = "hydrating", "ready", "retiring", "closed"


@dataclass(frozen=True)
class VectorCacheIdentity:  # memory_registry.rs:38-45
    scope: str
    index_id: int
    generation: int
    physical_index_id: int
    record_revision: int


class MetricMismatch(ValueError):
    """VectorGenerationValidationError::MetricMismatch (read_index.rs:49)."""


class _Entry:
    def __init__(self, metric: int):
        self.metric = metric
        self.state = HYDRATING
        self.index = None
        self.visible_seq = None
        self.guards = 0


class ReadGuard:
    """Keeps the entry's device index alive for the lifetime of one request (read_index.rs:33-35 `_cache_read_guard`)."""

    def __init__(self, registry: "DeviceIndexRegistry", identity: VectorCacheIdentity, entry: _Entry):
        self._registry, self._identity, self._entry = registry, identity, entry
        self.index = entry.index

    def release(self):
        if self._entry is not None:
            self._registry._release(self._identity, self._entry)
            self._entry = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()


class DeviceIndexRegistry:
    def __init__(self):
        self._lock = threading.Lock()
        self._entries: Dict[VectorCacheIdentity, _Entry] = {}

    def entry_for(self, identity: VectorCacheIdentity, metric: int) -> bool:
        """VectorCacheRegistry::entry_for: True iff the caller now OWNS the hydration of this identity."""
        with self._lock:
            if identity in self._entries and self._entries[identity].state != CLOSED:
                return False
            self._entries[identity] = _Entry(metric)
            return True

    def finish_hydration(self, identity: VectorCacheIdentity, index, visible_seq: int) -> bool:
        """The owner publishes the hydrated index with the sequence it contains rows up to; False if the entry was retired
        meanwhile (the index is then the caller's to free)."""
        with self._lock:
            e = self._entries.get(identity)
            if e is None or e.state != HYDRATING:
                return False
            e.index, e.visible_seq, e.state = index, int(visible_seq), READY
            return True

    def advance(self, identity: VectorCacheIdentity, visible_seq: int) -> bool:
        """Round 5: the owner applied a write batch to the resident copy (hvx_index_insert_batch on a growable image) and the copy now
        holds every row up to database sequence `visible_seq`: requests whose snapshot is that sequence attach again, older snapshots
        keep falling back to the storage path (the attach rule itself is unchanged: equality).  False when the entry is not Ready (being
        hydrated, retired or closed) or the sequence would move backwards -- the caller then re-hydrates instead."""
        with self._lock:
            e = self._entries.get(identity)
            if e is None or e.state != READY or e.visible_seq is None or int(visible_seq) < e.visible_seq:
                return False
            e.visible_seq = int(visible_seq)
            return True

    def attach(self, identity: VectorCacheIdentity, metric: int, snapshot_seq: Optional[int]) -> Optional[ReadGuard]:
        """read_index.rs:43-71.  `snapshot_seq=None` is VectorReadVisibility::Unavailable."""
        with self._lock:
            e = self._entries.get(identity)
            if e is not None and e.metric != metric:
                raise MetricMismatch("the request's distance is not the generation's distance")
            if snapshot_seq is None or e is None or e.state != READY or e.visible_seq != int(snapshot_seq):
                return None
            e.guards += 1
            return ReadGuard(self, identity, e)

    def retire(self, identity: VectorCacheIdentity, free: Callable = lambda ix: ix.close()):
        """A newer generation replaced this one: no new guards; the device memory goes once the last guard is dropped."""
        with self._lock:
            e = self._entries.get(identity)
            if e is None or e.state in (RETIRING, CLOSED):
                return
            e.state = RETIRING
            e._free = free
            done = e.guards == 0
        if done:
            self._close(identity, e)

    def _release(self, identity, e):
        with self._lock:
            e.guards -= 1
            done = e.state == RETIRING and e.guards == 0
        if done:
            self._close(identity, e)

    def _close(self, identity, e):
        ix, e.index, e.state = e.index, None, CLOSED
        if ix is not None:
            getattr(e, "_free", lambda i: i.close())(ix)

    def state(self, identity: VectorCacheIdentity) -> Optional[str]:
        with self._lock:
            e = self._entries.get(identity)
            return None if e is None else e.state
