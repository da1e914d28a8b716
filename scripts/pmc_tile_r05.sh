#!/bin/bash
# usage (on the GPU box): scripts/pmc_tile_r05.sh  -> gpurun_out/pmc_tile_r05/{timeline_*.txt,summary.txt}
# where do the exact-scan tile kernels' cycles go?  (1) the kernel timeline of ONE scan (rocprofv3 --kernel-trace),
# f32 rows 1024 x 1M x 768 and fp8 rows 4096 x 2M x 1536; (2) counter passes (own runs, --kernel-trace only) that split the 50 % of
# wave-cycles spent waiting: any wait vs instruction-issue waits, vector-memory issue cycles, VALU instructions, L1 pending stalls, L2 hit rate.
out=gpurun_out/pmc_tile_r05
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
: > $out/summary.txt
run_shape() { # name rows dim batch dtype
  rm -rf /tmp/tl_$1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$1 -o t -- python scripts/bench_flat.py --rows $2 --dim $3 --batch $4 --dtype $5 --steps 2 > /tmp/tl_$1.log 2>&1
  tr=$(find /tmp/tl_$1 -name '*kernel_trace.csv' | head -1)
  python - "$tr" > $out/timeline_$1.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the LAST scan: from the last split_queries* kernel onwards
last = max(i for i, r in enumerate(rows) if "split_queries" in r["Kernel_Name"])
first = last
while first > 0 and "validate_vectors_kernel" not in rows[first]["Kernel_Name"]:
    first -= 1
t0 = int(rows[first]["Start_Timestamp"])
tot = {}
for r in rows[first:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "at::native" in n or "elementwise" in n:
        break
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:10.1f} us  +{(e-s)/1e3:9.1f} us  {n[:90]}  grid {r.get('Grid_Size','?')}")
    tot[n] = tot.get(n, 0) + (e - s)
    end = e
print(f"scan span {end/1e3:.1f} us; per kernel: " + "; ".join(f"{k[:50]} {v/1e3:.1f}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])))
PY
  tail -1 $out/timeline_$1.txt
}
run_shape f32_c2 1000000 768 1024 f32
run_shape fp8_c5 2000000 1536 4096 fp8
for shape in "bf16 1000000 768 1024" "fp8 2000000 1536 4096"; do
  set -- $shape; dt=$1; rows=$2; dim=$3; b=$4
  for grp in "wait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "l1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    set -- $grp; name=$1; shift
    rm -rf /tmp/pmc5_${dt}_$name
    timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc5_${dt}_$name -o p -- python scripts/bench_flat.py --rows $rows --dim $dim --batch $b --dtype $dt --steps 1 > /tmp/pmc5_${dt}_$name.log 2>&1
    f=$(find /tmp/pmc5_${dt}_$name -name '*counter_collection.csv' | head -1)
    if [ -z "$f" ]; then echo "$dt [$name]: no counter file ($(tail -2 /tmp/pmc5_${dt}_$name.log | tr '\n' ' ' | cut -c1-200))" >> $out/summary.txt; continue; fi
    python - "$f" $dt $name >> $out/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "flat_tile" in n:
        acc[n.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kern, cs in acc.items():
    print(f"{sys.argv[2]} [{sys.argv[3]}] {kern}")
    for k, v in cs.items():
        print(f"    {k}: dispatches={len(v)} sum={sum(v):.5g} largest={max(v):.5g}")
PY
  done
done
cat $out/summary.txt
