#!/bin/bash
# usage (on the GPU box): scripts/collect_profiles_r03.sh <tag>
# round-3 kernels: the restricted walk, the small-batch exact scan and its selection / re-rank tail
#  1. rocprofv3 --kernel-trace --stats of the config #3 leg of bench.py (both strategies) -> c3_kernel_stats_hvx.csv, c3_bench_line.json
#  2. the same of scripts/bench_restricted_scan.py (exact strategy by candidate-set size)   -> rscan_kernel_stats_hvx.csv, rscan.log
#  3. --pmc FETCH_SIZE / TCC_EA0_RDREQ pass of (2) -> rscan_pmc_mem.csv + rscan_traffic.json (HBM bytes per launch of flat_smallb_kernel)
set -e
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
C3="python bench.py --skip production,datasets,config4,config5,graph_equivalence --cpu-seconds 0 --no-verify --steps 10 --warmup 2"
rm -rf /tmp/c3_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3_$tag -o c3 -- $C3 > /tmp/c3_$tag.log 2>&1 || true
grep '^{' /tmp/c3_$tag.log | tail -1 > $out/c3_bench_line.json
(head -1 /tmp/c3_$tag/c3_kernel_stats.csv; grep "hvx::\|restricted_walk\|set_bits" /tmp/c3_$tag/c3_kernel_stats.csv) > $out/c3_kernel_stats_hvx.csv
RS="python scripts/bench_restricted_scan.py 400000 1536 32"
rm -rf /tmp/rs_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rs_$tag -o rs -- $RS > $out/rscan.log 2>&1 || true
(head -1 /tmp/rs_$tag/rs_kernel_stats.csv; grep "hvx::" /tmp/rs_$tag/rs_kernel_stats.csv) > $out/rscan_kernel_stats_hvx.csv
rm -rf /tmp/rsp_$tag
rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/rsp_$tag -o rsp -- $RS > /tmp/rsp_$tag.log 2>&1 || true
f=$(ls /tmp/rsp_$tag/*counter_collection.csv | head -1)
(head -1 "$f"; grep -E "flat_smallb_kernel|flat_select_radix|rerank_bf16" "$f") > $out/rscan_pmc_mem.csv
python - "$out" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{out}/rscan_pmc_mem.csv")):
    name = r["Kernel_Name"].split("(")[0]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for name, cs in acc.items():
    res[name] = {k: {"dispatches": len(v), "max_per_launch": max(v), "mean_per_launch": sum(v) / len(v)} for k, v in cs.items()}
    if "FETCH_SIZE" in cs:
        res[name]["hbm_bytes_largest_launch_fetch_x1024_x2"] = int(max(cs["FETCH_SIZE"]) * 1024 * 2)
    if "TCC_EA0_RDREQ_sum" in cs:
        res[name]["hbm_bytes_largest_launch_rdreq_x128"] = int(max(cs["TCC_EA0_RDREQ_sum"]) * 128)
res["note"] = ("largest launch of flat_smallb_kernel<2,1,8,8,true> = the 100 000-candidate scan: algorithmic bytes 100 000 x 1536 x 4 = 614.4 MB "
               "(two depth halves of 307.2 MB each); FETCH_SIZE is KiB and on gfx950 reports 1/2 of a wide coalesced read (MI355X_MICROARCH.md)")
json.dump(res, open(f"{out}/rscan_traffic.json", "w"), indent=1)
print(json.dumps(res)[:1500])
PY
cat $out/c3_kernel_stats_hvx.csv | cut -c1-180
