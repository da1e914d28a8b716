#!/bin/bash
# usage (on the GPU box): scripts/collect_profiles_r06.sh <tag>      -> gpurun_out/prof_<tag>/
#  1. the driver's command, un-profiled                                  -> bench_line.json (+ bench.err)
#  2. rocprofv3 --kernel-trace --stats of the SAME command (every leg)   -> kernel_stats_full.csv, kernel_stats_hvx.csv (every kernel of
#     the library: hvx:: and the anonymous-namespace kernels -- restricted_walk_kernel, flat_smallb_kernel, build_link_wg_kernel,
#     hnsw_pair_kernel, audit kernels ...; round 3's filter lost the walk), bench_line_profiled.json, kernel_span.json
#  3. rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum (own pass, --kernel-trace only) of the headline leg -> pmc_mem.csv, traffic.json
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
if [ -z "$SKIP_PLAIN" ]; then timeout 900 python bench.py 2> $out/bench.err | tail -1 > $out/bench_line.json; cp bench_full.json $out/bench_full.json; fi
if [ -z "$SKIP_TRACE" ]; then
rm -rf /tmp/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python bench.py --deadline 600 > /tmp/prof_$tag.log 2>&1 || true
grep '^{' /tmp/prof_$tag.log | tail -1 > $out/bench_line_profiled.json
st=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
if [ -n "$st" ]; then cp "$st" $out/kernel_stats_full.csv; (head -1 $out/kernel_stats_full.csv; grep -E "hvx::|anonymous namespace" $out/kernel_stats_full.csv) > $out/kernel_stats_hvx.csv; fi
tr=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
if [ -n "$tr" ]; then python - "$tr" $out/kernel_span.json <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "hnsw_wave_kernel<1u, 3, 24, false, false, false, true, 2, false>" in r["Kernel_Name"]]
v = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
out = {}
if v:
    timed = v[5:25] if len(v) >= 25 else v   # the 20 timed dispatches of the headline behind its 5 warm-ups (the driver's command)
    span = max(e for _, e in timed) - min(s for s, _ in timed)
    out = {"kernel": "hnsw_wave_kernel<L2, R=3, NK=24, f32, strict, two per SIMD>", "dispatches": len(v), "timed_dispatches": len(timed),
           "average_ns_each": sum(e - s for s, e in timed) / len(timed), "overlapped_span_ns": span, "span_per_dispatch_ns": span / len(timed)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
PY
fi
fi
rm -rf /tmp/pmc_$tag
EXTRA="--skip production,production_lanes,insert,batcher,datasets,config3,config4,config5,graph_equivalence,ef_sweep,vendor_gemm,peak --cpu-seconds 0 --no-verify"
timeout 600 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/pmc_$tag -o pmc -- python bench.py --steps 6 --warmup 3 $EXTRA > /tmp/pmc_$tag.log 2>&1 || true
f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then
(head -1 "$f"; grep -E "hnsw_(wave|pair)_kernel" "$f") > $out/pmc_mem.csv
python - "$out" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{out}/pmc_mem.csv")):
    name = r["Kernel_Name"]
    key = "headline" if "hnsw_wave_kernel<1u, 3, 24, false, false, false, true, 2, false>" in name else "pair" if "hnsw_pair_kernel<1u, 3, 24, false, 3>" in name else None
    if key:
        acc[key][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
res = {}
for key, cs in acc.items():
    # the headline leg runs first: its 3 warm-up + 6 timed launches are the first nine dispatches of the instantiation (later legs --
    # shard searches, host timing -- launch the same build over other query batches)
    res[key] = {}
    for k, v in cs.items():
        vals = [x for _, x in sorted(v)][:9] if key == "headline" else [x for _, x in v]
        res[key][k] = {"dispatches": len(vals), "mean_per_launch": sum(vals) / len(vals), "min": min(vals), "max": max(vals)}
h = res.get("headline", {})
traffic = {"kernel": "hnsw_wave_kernel (headline instantiation: L2, 192-entry beam, dim 768, f32 rows, two queries per SIMD)",
           "hbm_bytes_per_launch": int(h["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2) if "FETCH_SIZE" in h else None,
           "tcc_ea0_rdreq_x128B": int(h["TCC_EA0_RDREQ_sum"]["mean_per_launch"] * 128) if "TCC_EA0_RDREQ_sum" in h else None,
           "pair_kernel_hbm_bytes_per_launch": int(res["pair"]["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2) if "pair" in res and "FETCH_SIZE" in res["pair"] else None,
           "how": "rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum (own pass, --kernel-trace only), mean over the headline kernel's dispatches of bench.py "
                  "(counter collection serialises the dispatches: per-launch traffic is unaffected, overlap is not visible here); FETCH_SIZE is KiB and on "
                  "gfx950 reports 1/2 of a wide coalesced read (MI355X_MICROARCH.md, HBM section) => x1024 x2; cross-check: TCC_EA0_RDREQ_sum x 128 B",
           "counters": res}
json.dump(traffic, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps({k: traffic[k] for k in ("hbm_bytes_per_launch", "tcc_ea0_rdreq_x128B", "pair_kernel_hbm_bytes_per_launch")}))
PY
fi
# 4. (round 6) the restricted scans: kernel trace + PMC pass of scripts/bench_restricted_direct.py -> restricted_kernel_stats.csv, restricted_traffic.json
rm -rf /tmp/rs_$tag /tmp/rsp_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rs_$tag -o rs -- python scripts/bench_restricted_direct.py 400000 1536 > /tmp/rs_$tag.log 2>&1 || true
st=$(find /tmp/rs_$tag -name '*kernel_stats.csv' | head -1)
if [ -n "$st" ]; then (head -1 "$st"; grep -E "restricted_direct|exact_tail|smallq|smallb|stage_|validate_vectors|bitmap_|bfs_level|expand_collect" "$st") > $out/restricted_kernel_stats.csv; fi
timeout 400 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/rsp_$tag -o pmc -- python scripts/bench_restricted_direct.py 400000 1536 > /tmp/rsp_$tag.log 2>&1 || true
f=$(find /tmp/rsp_$tag -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then
python - "$f" $out/restricted_traffic.json <<'PY'
import csv, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    key = "restricted_direct_kernel" if "restricted_direct_kernel" in n else "flat_exact_tail_kernel" if "exact_tail" in n else "flat_smallq_kernel" if "smallq" in n else None
    if key:
        acc[key + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {"dispatches": len(next(iter(v.values()))), "hbm_bytes_per_launch_fetch_size_x2048": int(sum(v.get("FETCH_SIZE", [0])) / max(len(v.get("FETCH_SIZE", [1])), 1) * 2048),
           "tcc_ea0_rdreq_x128B": int(sum(v.get("TCC_EA0_RDREQ_sum", [0])) / max(len(v.get("TCC_EA0_RDREQ_sum", [1])), 1) * 128)} for k, v in acc.items()}
json.dump({"how": "rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum of scripts/bench_restricted_direct.py 400000 1536 (own pass); per kernel and grid size "
                  "(grid = threads: slices x 256 by query tiles); FETCH_SIZE KiB x 2 on gfx950", "kernels": res}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(res)[:1500])
PY
fi
# 5. (round 6) writes: deletes and one-node inserts / upserts at 200 000 x 768 under the kernel trace -> delete_bench.json, delete_kernel_stats.csv,
#    seq_insert.json, seq_insert_kernel_stats.csv
rm -rf /tmp/dl_$tag /tmp/si_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dl_$tag -o d -- python scripts/delete_bench.py 200000 768 100 > /tmp/dl_$tag.log 2>&1 || true
grep '^{' /tmp/dl_$tag.log | tail -1 > $out/delete_bench.json
st=$(find /tmp/dl_$tag -name '*kernel_stats.csv' | head -1)
if [ -n "$st" ]; then (head -1 "$st"; grep -E "delete_" "$st") > $out/delete_kernel_stats.csv; fi
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/si_$tag -o s -- python scripts/seq_insert_bench.py 200000 768 208 > /tmp/si_$tag.log 2>&1 || true
grep '^{' /tmp/si_$tag.log | tail -1 > $out/seq_insert.json
st=$(find /tmp/si_$tag -name '*kernel_stats.csv' | head -1)
if [ -n "$st" ]; then (head -1 "$st"; grep -E "build_|delete_|hnsw_wave_kernel<1u, 6, 24" "$st") > $out/seq_insert_kernel_stats.csv; fi
timeout 120 python scripts/delete_bench.py 200000 768 100 2>/dev/null | grep '^{' | tail -1 > $out/delete_bench_unprofiled.json
timeout 120 python scripts/seq_insert_bench.py 200000 768 208 2>/dev/null | grep '^{' | tail -1 > $out/seq_insert_unprofiled.json
[ -f $out/kernel_stats_hvx.csv ] && cut -c1-170 $out/kernel_stats_hvx.csv | head -40
