#!/bin/bash
# usage: scripts/pmc_pass.sh <tag> <counters...>   (run on the GPU box; one rocprofv3 --pmc pass of bench.py)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o $tag -- python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-verify --no-production-default --graph-cache /tmp/g > /tmp/pmc_$tag.log 2>&1 || { tail -5 /tmp/pmc_$tag.log; exit 1; }
mkdir -p gpurun_out/pmc
f=$(ls /tmp/pmc_$tag/*counter_collection.csv | head -1)
head -1 "$f" > gpurun_out/pmc/${tag}_hnsw.csv
grep -E "hnsw_(wave|search)_kernel" "$f" >> gpurun_out/pmc/${tag}_hnsw.csv || true
grep '^{' /tmp/pmc_$tag.log | tail -1 > gpurun_out/pmc/${tag}_bench.json
python - "$tag" <<'PY'
import csv, sys, collections
tag = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/pmc/{tag}_hnsw.csv")))
acc = collections.defaultdict(list)
for r in rows:
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{tag}: {k:28s} dispatches {len(v):3d}  mean {sum(v)/len(v):.4g}")
PY
