#!/bin/bash
# usage (on the GPU box): scripts/collect_profiles.sh <tag>
# 1. un-profiled bench (all legs, cpu_baseline + oracle parity) -> gpurun_out/prof_<tag>/bench_line.json
# 2. rocprofv3 --kernel-trace --stats of the headline leg        -> kernel_stats_{full,hvx}.csv, bench_line_profiled.json,
#    kernel_span.json (overlapped span of the timed hnsw_wave_kernel dispatches / their count, from the kernel trace)
# 3. the same with --lanes 1 --occupancy 1 (one lone batch at a time: AverageNs == roofline.lone_batch.kernel_ms)
# 4. rocprofv3 --pmc passes (own runs, kernel-trace only; counter collection serialises the dispatches)
#    -> pmc_*.csv, traffic.json (FETCH_SIZE x2, gfx950 correction)
set -e
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
EXTRA="--skip production,datasets,config3,config4,config5,graph_equivalence --cpu-seconds 0 --no-verify"
BENCH="python bench.py --steps 60 --warmup 6"
if [ -z "$SKIP_FULL" ]; then $BENCH 2> $out/bench.log | tail -1 > $out/bench_line.json; fi
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- $BENCH $EXTRA > /tmp/prof_$tag.log 2>&1
grep '^{' /tmp/prof_$tag.log | tail -1 > $out/bench_line_profiled.json
cp /tmp/prof_$tag/${tag}_kernel_stats.csv $out/kernel_stats_full.csv
(head -1 $out/kernel_stats_full.csv; grep "hvx::" $out/kernel_stats_full.csv) > $out/kernel_stats_hvx.csv
python - /tmp/prof_$tag/${tag}_kernel_trace.csv $out/kernel_span.json <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "hnsw_wave_kernel" in r["Kernel_Name"]]
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
out = {}
for name, v in by.items():
    v.sort()
    timed = v[-60:] if len(v) >= 60 else v       # the timed region = the last 60 dispatches of the two-per-SIMD build
    span = max(e for _, e in timed) - min(s for s, _ in timed)
    out[name] = {"dispatches": len(v), "timed_dispatches": len(timed), "average_ns_each": sum(e - s for s, e in timed) / len(timed),
                 "overlapped_span_ns": span, "span_per_dispatch_ns": span / len(timed)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
PY
rm -rf /tmp/prof_${tag}l
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}l -o ${tag}l -- $BENCH $EXTRA --lanes 1 --occupancy 1 > /tmp/prof_${tag}l.log 2>&1
grep '^{' /tmp/prof_${tag}l.log | tail -1 > $out/bench_line_profiled_lone_batch.json
(head -1 /tmp/prof_${tag}l/${tag}l_kernel_stats.csv; grep "hvx::" /tmp/prof_${tag}l/${tag}l_kernel_stats.csv) > $out/kernel_stats_hvx_lone_batch.csv
pmc() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o $name -- python bench.py --steps 6 --warmup 3 $EXTRA > /tmp/pmc_$name.log 2>&1
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
        if ", 2, false>(" not in r["Kernel_Name"]:   # the headline: two-queries-per-SIMD build, not the BUILD / re-run / lone-batch instantiations
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        summary[k] = {"dispatches": len(v), "mean_per_launch": sum(v) / len(v)}
fetch_kb = summary["FETCH_SIZE"]["mean_per_launch"]
rd = summary["TCC_EA0_RDREQ_sum"]["mean_per_launch"]
traffic = {
    "kernel": "hnsw_wave_kernel",
    "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2),
    "how": "rocprofv3 --pmc FETCH_SIZE (own pass, --kernel-trace only), mean over the hnsw_wave_kernel dispatches of bench.py "
           "(counter collection serialises the dispatches: per-launch traffic is unaffected, overlap is not visible here); "
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
