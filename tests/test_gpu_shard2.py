"""N = 2 ranks END TO END on the device: every rank builds its id-range shard with the device builder, searches the
whole query batch on it with the HIP kernels (exact scan and HNSW), exchanges the packed top-k payloads (ONE all-gather per step)
and merges on the device (hvx_merge_topk_packed_device).  Both ranks share GPU 0 -- RCCL refuses two ranks on one device, so the
collective is gloo with host staging; the RCCL transport itself is covered by the 1-rank group tests (test_gpu_parity.py) and the
exchange / merge logic by tests/test_shard_gloo.py.  Checked: the merged exact answer equals the oracle's exact scan over the WHOLE
corpus bit for bit (ids, score bits, ties across shards by id); the merged HNSW answer equals the Candidate-order merge of the two
shards' own device results and reaches the recall of a single index over all rows; both ranks hold the same answer."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "helix-db_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import fixtures as fx
    import orc
    import pyhvx as hv
    from pyhvx import shard
    msg = "ok"
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
        n, dim, b, k, ef = 6001, 128, 64, 10, 96
        rng = np.random.default_rng(4242)
        centres = rng.standard_normal((48, dim)).astype(np.float32)
        data = (centres[rng.integers(0, 48, n)] + 0.4 * rng.standard_normal((n, dim))).astype(np.float32)
        data[17] = data[n - 5]  # equal scores on two shards: the merge must break the tie by id
        q = (centres[rng.integers(0, 48, b)] + 0.4 * rng.standard_normal((b, dim))).astype(np.float32)
        q[0] = data[17]
        lo, hi = shard.plan_shards(n, world)[rank]
        ids = np.arange(lo, hi, dtype=np.uint64)
        lv = fx.draw_levels(hi - lo, 16, seed=100 + rank)
        ix, _ = hv.ValidatedVectorReadIndex.build(dim=dim, metric=hv.EUCLIDEAN, node_ids=ids, vectors=data[lo:hi], levels=lv, m=16, m0=32,
                                                  ef_construction=100, max_batch=256, batch_divisor=16, search_max_batch=b)
        dq = torch.from_numpy(q).to(dev)
        st = torch.zeros(b, dtype=torch.int32, device=dev)
        ss = shard.ShardedSearcher(ix, world, b, k, dev, None, stage_through_host=True)
        li, ls, lc = ss.outputs()
        # ---- exact scan: merged answer == the oracle's exact scan over all n rows ----
        ix.flat_search_batch_device(dq, k, li, ls, lc, st)
        merged = ss.merge(li, ls, lc)  # exchange + merge are enqueued on the index's stream, behind the search
        ix.sync()
        m_ids, m_sc, m_cnt = (t.cpu().numpy().copy() for t in merged)
        for qi in range(b):
            rc, tid, tsc = orc.flat_matrix(orc.L2SQ, data, q[qi], k)
            assert m_cnt[qi] == tid.size and m_ids[qi, :tid.size].astype(np.uint64).tolist() == tid.tolist(), \
                f"exact ids, query {qi}: {m_cnt[qi]} {m_ids[qi].tolist()} vs {tid.tolist()}"
            assert m_sc[qi, :tid.size].view(np.uint32).tolist() == tsc.view(np.uint32).tolist(), f"exact score bits, query {qi}"
        assert m_ids[0, :2].tolist() == sorted([17, n - 5])
        exact_ids = m_ids.copy()
        # ---- HNSW: merged answer == Candidate-order merge of the shards' own device results; recall vs the exact answer ----
        qs = torch.zeros(b, 4, dtype=torch.int32, device=dev)
        ix.search_batch_device(dq, k, ef, li, ls, lc, st, qs)
        merged = ss.merge(li, ls, lc)
        ix.sync()
        h_ids, h_sc, h_cnt = (t.cpu().numpy().copy() for t in merged)
        g_ids, g_sc, g_cnt = ss.ex.gathered()
        r_ids, r_sc, r_cnt = fx.merge_topk_reference(g_ids.cpu().numpy().view(np.uint64), g_sc.cpu().numpy(), g_cnt.cpu().numpy(), k)
        assert h_cnt.tolist() == r_cnt.tolist()
        for qi in range(b):
            c = int(h_cnt[qi])
            assert h_ids[qi, :c].astype(np.uint64).tolist() == r_ids[qi, :c].tolist(), f"hnsw merge ids, query {qi}"
            assert h_sc[qi, :c].view(np.uint32).tolist() == r_sc[qi, :c].view(np.uint32).tolist(), f"hnsw merge score bits, query {qi}"
        rec = fx.recall_at_k(h_ids, exact_ids)
        assert rec >= 0.95, rec
        # ---- both ranks hold the same merged answers ----
        mine = torch.from_numpy(np.concatenate([exact_ids.ravel(), h_ids.ravel()]).astype(np.int64))
        ref = mine.clone()
        dist.broadcast(ref, src=0)
        assert bool((mine == ref).all()), "ranks disagree"
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # the parent reads one line per rank
        import traceback
        msg = "FAIL " + "".join(traceback.format_exception(type(e), e, e.__traceback__))[-1500:]
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(msg)


def test_two_ranks_search_exchange_and_merge_on_the_device(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"
