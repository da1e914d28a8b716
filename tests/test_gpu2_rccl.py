"""Two (or, on the 8-GPU node, eight) REAL devices, RCCL inside the library (ncclAllGather with world > 1 had never executed anywhere -- RCCL refuses
two ranks on one device, so the 1-GPU boxes of this pool can only run the world-1 collective and the gloo twin).  On a box with >= 2
GPUs (`pytest -m gpu2`) two processes, one per device, each build their id-range
shard and run the one-call steps hvx_shard_group_search_batch_device / _flat_search_batch_device / _search_batch_params_device on TWO
execution lanes that share the rank's one communicator (hvx_shard_group_attach).  Checked on both ranks: the merged exact answer equals
the oracle's exact scan over the WHOLE corpus bit for bit; the merged HNSW answers equal the Candidate-order merge of the two shards'
own results; both lanes and both ranks agree.  Skipped (not failed) where fewer than two devices are visible."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu2  # (not `gpu`: the 1-GPU tier never selects it; `-m gpu2` on a multi-GPU box does)


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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    import fixtures as fx
    import orc
    import pyhvx as hv
    from pyhvx import shard
    msg = "ok"
    try:
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("gloo", rank=rank, world_size=world)   # host-side transport of the 128-byte unique id only
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
                                                  ef_construction=100, max_batch=256, batch_divisor=16, device=rank, search_max_batch=b)
        ix.set_simhash()
        uid = [hv.ShardGroup.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        lanes = [ix.fork(), ix.fork()]
        g0 = hv.ShardGroup(lanes[0], uid[0], rank, world, b, k)
        groups = [g0, g0.attach(lanes[1])]                         # ONE communicator per rank, two lanes
        assert hv.rccl_version() >= 2000
        dq = torch.from_numpy(q).to(dev)
        outs = [(torch.zeros(b, k, dtype=torch.int64, device=dev), torch.zeros(b, k, dtype=torch.float32, device=dev),
                 torch.zeros(b, dtype=torch.int32, device=dev), torch.zeros(b, dtype=torch.int32, device=dev)) for _ in groups]

        def both(fn):
            for rep in range(3):                                    # steps alternate over the lanes, same order on every rank
                for g, o in zip(groups, outs):
                    fn(g, o)
            for ln in lanes:
                ln.sync()
            a, c = outs
            assert bool((a[0] == c[0]).all()) and bool((a[1].view(torch.int32) == c[1].view(torch.int32)).all()) and bool((a[2] == c[2]).all())
            return a[0].cpu().numpy().astype(np.uint64), a[1].cpu().numpy(), a[2].cpu().numpy()

        # ---- exact scan over both shards == the oracle's exact scan over all n rows ----
        fi, fs, fc = both(lambda g, o: g.flat_search_batch_device(dq, k, o[0], o[1], o[2], o[3]))
        for qi in range(b):
            rc, tid, tsc = orc.flat_matrix(orc.L2SQ, data, q[qi], k)
            assert rc == orc.OK and fc[qi] == k
            assert fi[qi].tolist() == tid.tolist() and fs[qi].view(np.uint32).tolist() == tsc.view(np.uint32).tolist(), f"exact scan, query {qi}"
        # ---- strict HNSW and the production-default arm: merged == Candidate-order merge of the shards' own answers ----
        for name, step, local in (("strict", lambda g, o: g.search_batch_device(dq, k, ef, o[0], o[1], o[2], o[3]),
                                   lambda: ix.search_batch(q, hv.SearchParams(k).with_ef(ef))),
                                  ("default", lambda g, o: g.search_batch_params_device(dq, hv.SearchParams.new(k), o[0], o[1], o[2], o[3]),
                                   lambda: ix.search_batch(q, hv.SearchParams.new(k)))):
            mi, ms, mc = both(step)
            li, ls_, lc, _ = local()
            mine = [li, ls_.view(np.uint32), lc]
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            for qi in range(b):
                pool = sorted((int(g_[1][qi, j]), int(g_[0][qi, j])) for g_ in gathered for j in range(int(g_[2][qi])))[:k]   # (score bits, id): Candidate order
                assert [p_[1] for p_ in pool] == mi[qi, :mc[qi]].tolist(), f"{name}, query {qi}"
                assert [p_[0] for p_ in pool] == ms[qi, :mc[qi]].view(np.uint32).tolist()
        allres = [None] * world
        dist.all_gather_object(allres, fi.tolist())
        assert all(r_ == allres[0] for r_ in allres)                # every rank holds the same merged answer
        assert g0.wait(20000)                                       # hvx_shard_group_wait: the bounded form of the host's stream wait
        for g in groups:
            g.close()
        for ln in lanes:
            ln.close()
        ix.close()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        msg = f"rank {rank}: {type(e).__name__}: {e}\n{traceback.format_exc()}"
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(msg)


@pytest.mark.parametrize("want", [2, 8])
def test_ranks_on_real_devices_exchange_through_rccl_inside_the_library(tmp_path, want):
    """world = 2, and world = min(8, devices) on the 8-GPU node (round 6: the first real ncclAllGather at world > 2 should not be the bench)"""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two devices (RCCL refuses two ranks on one device): run with -m gpu2 on a multi-GPU box")
    world = min(want, torch.cuda.device_count())
    if want > 2 and world <= 2:
        pytest.skip("the wider group needs more than two devices")
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"
