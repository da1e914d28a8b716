#!/bin/bash
# usage (on the GPU box): scripts/pmc_tile.sh <tag> <bench_flat args...>
# rocprofv3 --pmc passes (own runs, kernel-trace only) of the exact-scan contraction kernel -> gpurun_out/pmc_tile_<tag>/
tag=$1; shift
out=gpurun_out/pmc_tile_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
pmc() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o $name -- python scripts/bench_flat.py $ARGS > /tmp/pmc_$name.log 2>&1
  f=$(ls /tmp/pmc_$name/*counter_collection.csv | head -1)
  (head -1 "$f"; grep -E "flat_tile256" "$f") > $out/pmc_$name.csv
  python - $out/pmc_$name.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"  {k}: n={len(v)} mean={sum(v)/len(v):.4g} max={max(v):.4g}")
PY
}
ARGS="$*"
pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc mem FETCH_SIZE TCC_EA0_RDREQ_sum
pmc sq GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM
