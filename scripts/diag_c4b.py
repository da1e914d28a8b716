import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "helix-db_amd")); sys.path.insert(0, ROOT)
import torch
import pyhvx as hv
from pyhvx import synth
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
args = bench.parse()
dev = torch.device("cuda", 0)
args.builder = "bulk"
for order in ("lanes_first", "truth_first", "lanes_first"):
    os.environ["DIAG_ORDER"] = order
    r, _ = bench.hnsw_leg(hv, synth, args, dev, "embedding", args.c4_rows, 768, 1024, 10, 128, dtype_name="bf16", steps=30)
    print(json.dumps({"variant": "hnsw_leg " + order + " HWQ=" + os.environ.get("GPU_MAX_HW_QUEUES", "default"), "ms_per_step": r["ms_per_step"], "frac": r["roofline"]["frac"], "each": r["roofline"]["kernel_ms_each"],
                      "dist": r["distance_computations_per_query"]}), flush=True)
