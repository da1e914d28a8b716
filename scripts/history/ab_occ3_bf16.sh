#!/bin/bash
# round 6: config #4 (bf16 rows, 1.25M x 768) with two / three queries per SIMD -- bench.py --occupancy N, only the headline + config4 legs
S=production,production_lanes,insert,batcher,datasets,iso_recall,latent,config3,config5,graph_equivalence,ef_sweep,vendor_gemm,traffic,peak
for o in 2 3 2 3; do
  timeout 300 python bench.py --occupancy $o --skip $S --cpu-seconds 0 --no-verify 2>/dev/null | tail -1 > /tmp/ab_occ_$o.json
  python - $o <<'PY'
import json, sys
d = json.load(open(f"/tmp/ab_occ_{sys.argv[1]}.json"))
print(json.dumps({"occupancy": int(sys.argv[1]), "headline_qps": d["value"], "headline_frac": d["roofline"]["frac"], "config4": d.get("config4")}))
PY
done
