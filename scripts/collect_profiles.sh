#!/bin/bash
# usage (on the GPU box): scripts/collect_profiles.sh <tag>
# 1. un-profiled bench (with cpu_baseline + oracle parity)   -> gpurun_out/prof_<tag>/bench_line.json
# 2. rocprofv3 --kernel-trace --stats of the same command    -> kernel_stats_{full,hvx}.csv + bench_line_profiled.json
# 3. rocprofv3 --pmc passes (own runs, kernel-trace only)    -> pmc_*.csv, traffic.json (FETCH_SIZE x2, gfx950 correction)
set -e
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
BENCH="python bench.py --steps 20 --warmup 3 --graph-cache /tmp/g"
$BENCH 2> $out/bench.log | tail -1 > $out/bench_line.json
rm -rf /tmp/prof_$tag
# the headline kernel alone (no production-default leg: its strict ef=100 launches share the R=3 instantiation's name)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- $BENCH --cpu-seconds 0 --no-verify --no-production-default > /tmp/prof_$tag.log 2>&1
grep '^{' /tmp/prof_$tag.log | tail -1 > $out/bench_line_profiled.json
cp /tmp/prof_$tag/${tag}_kernel_stats.csv $out/kernel_stats_full.csv
(head -1 $out/kernel_stats_full.csv; grep "hvx::" $out/kernel_stats_full.csv) > $out/kernel_stats_hvx.csv
# the same with the production-default leg: the AD instantiation (..., false, true>) only runs SearchParams::new(k) launches
rm -rf /tmp/prof_${tag}p
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}p -o ${tag}p -- $BENCH --cpu-seconds 0 --no-verify > /tmp/prof_${tag}p.log 2>&1
grep '^{' /tmp/prof_${tag}p.log | tail -1 > $out/bench_line_profiled_production_default.json
(head -1 /tmp/prof_${tag}p/${tag}p_kernel_stats.csv; grep "hvx::" /tmp/prof_${tag}p/${tag}p_kernel_stats.csv) > $out/kernel_stats_hvx_production_default.csv
pmc() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o $name -- $BENCH --steps 5 --warmup 2 --cpu-seconds 0 --no-verify --no-production-default > /tmp/pmc_$name.log 2>&1
  f=$(ls /tmp/pmc_$name/*counter_collection.csv | head -1)
  (head -1 "$f"; grep -E "hnsw_(wave|search)_kernel" "$f") > $out/pmc_$name.csv
}
pmc mem FETCH_SIZE TCC_EA0_RDREQ_sum
pmc l2 TCC_HIT_sum TCC_MISS_sum
pmc sq GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
python - "$out" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
summary = {}
for name in ("mem", "l2", "sq"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/pmc_{name}.csv")):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        summary[k] = {"dispatches": len(v), "mean_per_launch": sum(v) / len(v)}
fetch_kb = summary["FETCH_SIZE"]["mean_per_launch"]
rd = summary["TCC_EA0_RDREQ_sum"]["mean_per_launch"]
traffic = {
    "kernel": "hnsw_wave_kernel",
    "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2),
    "how": "rocprofv3 --pmc FETCH_SIZE (own pass, --kernel-trace only), mean over the hnsw_wave_kernel dispatches of bench.py; "
           "FETCH_SIZE is KiB and on gfx950 reports 1/2 of a wide coalesced read (MI355X_MICROARCH.md, HBM section) => x1024 x2; "
           "cross-check: TCC_EA0_RDREQ_sum x 128 B",
    "fetch_size_kib_raw": fetch_kb,
    "tcc_ea0_rdreq_x128B": int(rd * 128),
    "counters": summary,
}
json.dump(traffic, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps({k: traffic[k] for k in ("hbm_bytes_per_launch", "tcc_ea0_rdreq_x128B")}))
PY
cat $out/kernel_stats_hvx.csv
