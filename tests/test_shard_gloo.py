"""N > 1 path on CPU: id-range shard plan + all-gather of per-shard top-k (gloo, world_size 2 and 3),
merged by the Candidate-order checker and compared bit-for-bit with the oracle's exact scan over the
whole corpus.  The per-shard searches are done by the ORACLE here (there is no GPU in this container);
on the GPU box the same exchange feeds hvx_merge_topk_device (tests/test_gpu_parity.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import fixtures as fx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, n, dim, k, b):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "helix-db_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import orc
    from pyhvx import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(42)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    data[11] = data[n - 3]  # equal scores on two shards: the merge must break the tie by id
    q = rng.standard_normal((b, dim)).astype(np.float32)
    q[0] = data[11]
    lo, hi = shard.plan_shards(n, world)[rank]
    assert all(shard.owner_of(i, n, world) == rank for i in (lo, hi - 1))
    ids = np.arange(lo, hi, dtype=np.uint64)
    oix = orc.Index(dim, orc.L2SQ)
    m = hi - lo
    assert oix.seed(ids, data[lo:hi], np.zeros(m + 1, np.uint64), np.zeros(0, np.uint64), entry_point=lo) == orc.OK
    l_ids = np.zeros((b, k), np.int64); l_sc = np.zeros((b, k), np.float32); l_cnt = np.zeros(b, np.int32)
    for qi in range(b):
        rc, oid, osc = oix.flat(q[qi], k)
        assert rc == orc.OK
        l_ids[qi, :oid.size] = oid.astype(np.int64); l_sc[qi, :oid.size] = osc; l_cnt[qi] = oid.size
    ex = shard.TopkExchange(world, b, k, "cpu")
    assert ex.payload == shard.payload_bytes(b, k) and ex.payload % 8 == 0
    if rank % 2:   # one rank writes its results straight into the payload views (what the GPU search does) ...
        ex.ids.copy_(torch.from_numpy(l_ids)); ex.scores.copy_(torch.from_numpy(l_sc)); ex.counts.copy_(torch.from_numpy(l_cnt))
        packed = ex.gather()
    else:          # ... the other hands tensors over
        packed = ex.gather(torch.from_numpy(l_ids), torch.from_numpy(l_sc), torch.from_numpy(l_cnt))
    assert packed.numel() == world * ex.payload            # ONE all-gather carried ids, scores and counts
    g_ids, g_sc, g_cnt = ex.gathered()
    m_ids, m_sc, m_cnt = fx.merge_topk_reference(g_ids.numpy().view(np.uint64), g_sc.numpy(), g_cnt.numpy(), k)
    ok = True
    for qi in range(b):
        rc, tid, tsc = orc.flat_matrix(orc.L2SQ, data, q[qi], k)
        ok &= m_cnt[qi] == tid.size and m_ids[qi, :tid.size].tolist() == tid.tolist()
        ok &= m_sc[qi, :tid.size].view(np.uint32).tolist() == tsc.view(np.uint32).tolist()
    ok &= m_ids[0, :2].tolist() == sorted([11, n - 3])
    # every rank must hold the identical merged answer
    mine = torch.from_numpy(m_ids.view(np.int64).copy())
    ref = mine.clone()
    dist.broadcast(ref, src=0)
    ok &= bool((mine == ref).all())
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write("ok" if ok else "FAIL")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,k", [(2, 3001, 10), (3, 500, 10), (2, 12, 10)])
def test_sharded_exact_search_merges_to_the_global_answer(tmp_path, world, n, k):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), n, 24, k, 7), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def test_shard_plan_covers_ids_once():
    sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))
    from pyhvx import shard
    for n, w in [(10, 3), (1_000_000, 8), (7, 8), (0, 2)]:
        plan = shard.plan_shards(n, w)
        assert plan[0][0] == 0 and plan[-1][1] == n and all(a[1] == b_[0] for a, b_ in zip(plan, plan[1:]))
        assert max(h - l for l, h in plan) - min(h - l for l, h in plan) <= 1
        for i in range(0, n, max(1, n // 50)):
            r = shard.owner_of(i, n, w)
            assert plan[r][0] <= i < plan[r][1]


def _restricted_worker(rank, world, port, out_dir):
    """N > 1 restricted search on CPU: every rank holds the same candidate id list, keeps the ids of its own id range
    (shard.slice_candidates == hvx_shard_group_search_restricted_batch's slicing), answers over them (the oracle stands in for
    the shard's search), and the exchange carries ids, scores, counts AND the per-query status, merged by the element-wise
    maximum (shard.merge_status == merge_status_kernel)."""
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "helix-db_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import orc
    from pyhvx import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, dim, k, b = 2000, 16, 10, 6
    rng = np.random.default_rng(7)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((b, dim)).astype(np.float32)
    q[2, 1] = np.nan                                           # rejected by every shard's validation
    allowed = np.unique(rng.choice(np.arange(n + 50, dtype=np.uint64), 700))   # some ids beyond the corpus: no vector anywhere
    lo, hi = shard.plan_shards(n, world)[rank]
    mine = shard.slice_candidates(allowed, lo, hi - 1)
    ok = all(lo <= int(i) < hi for i in mine)
    parts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(parts, torch.tensor([mine.size]))
    ok &= sum(int(p) for p in parts) == int((allowed < n).sum())      # the slices partition the indexed candidates
    oix = orc.Index(dim, orc.L2SQ)
    m = hi - lo
    assert oix.seed(np.arange(lo, hi, dtype=np.uint64), data[lo:hi], np.zeros(m + 1, np.uint64), np.zeros(0, np.uint64), entry_point=lo) == orc.OK
    l_ids = np.zeros((b, k), np.int64); l_sc = np.zeros((b, k), np.float32); l_cnt = np.zeros(b, np.int32); l_st = np.zeros(b, np.int32)
    for qi in range(b):
        rc, bad = orc.validate(orc.L2SQ, q[qi], dim)
        if rc:
            l_st[qi] = rc
            continue
        if mine.size:
            rc, oid, osc = oix.flat(q[qi], k, allowed=mine)
            assert rc == orc.OK
            l_ids[qi, :oid.size] = oid.astype(np.int64); l_sc[qi, :oid.size] = osc; l_cnt[qi] = oid.size
    ex = shard.TopkExchange(world, b, k, "cpu")
    ex.gather(torch.from_numpy(l_ids), torch.from_numpy(l_sc), torch.from_numpy(l_cnt))
    sts = [torch.zeros(b, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(sts, torch.from_numpy(l_st))
    st = shard.merge_status(torch.stack(sts)).numpy()
    g_ids, g_sc, g_cnt = ex.gathered()
    m_ids, m_sc, m_cnt = fx.merge_topk_reference(g_ids.numpy().view(np.uint64), g_sc.numpy(), g_cnt.numpy(), k)
    m_cnt = np.where(st != 0, 0, m_cnt)
    full = orc.Index(dim, orc.L2SQ)
    assert full.seed(np.arange(n, dtype=np.uint64), data, np.zeros(n + 1, np.uint64), np.zeros(0, np.uint64), entry_point=0) == orc.OK
    for qi in range(b):
        if qi == 2:
            ok &= st[qi] == orc.ERR_NONFINITE and m_cnt[qi] == 0
            continue
        rc, tid, tsc = full.flat(q[qi], k, allowed=allowed)
        ok &= st[qi] == 0 and m_cnt[qi] == tid.size and m_ids[qi, :tid.size].tolist() == tid.tolist()
        ok &= m_sc[qi, :tid.size].view(np.uint32).tolist() == tsc.view(np.uint32).tolist()
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write("ok" if ok else "FAIL")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_restricted_search_slices_candidates_and_merges_status(tmp_path, world):
    port = _free_port()
    mp.spawn(_restricted_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"
