import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))
import numpy as np, torch
import pyhvx as hv
from pyhvx import synth
n, dim = 400000, 1536
dev = torch.device("cuda", 0)
x, _ = synth.corpus("clustered", n, dim, 1, 20260923, dev, centres=1024, sigma=0.15)
ix = hv.ValidatedVectorReadIndex.managed(dim=dim, metric=hv.EUCLIDEAN, node_ids=np.arange(n, dtype=np.uint64), vectors=x, l0_offsets=np.zeros(n + 1, np.uint64),
                                         l0_neighbors=np.zeros(0, np.uint64), entry_point=0, device=0, max_batch=1024)
q = x[torch.randint(0, n, (32,), generator=torch.Generator().manual_seed(1)).to(dev)].cpu().numpy()
ix.set_option(hv.OPT_RESTRICTED_DIRECT, 1)
for size in (10000, 100000):
    allowed = np.arange(n // 2, n // 2 + size, dtype=np.uint64)
    for tail in (0, 1):
        ix.set_option(hv.OPT_FLAT_NO_TAIL, tail)
        for r in range(4):
            ix.search_restricted_batch(q, hv.SearchParams(10).with_ef(100), allowed)
