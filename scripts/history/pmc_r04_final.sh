#!/bin/bash
# usage (on the GPU box): scripts/pmc_r04_final.sh -> gpurun_out/pmc_r04_final/summary.txt
# (1) HBM traffic of the ring build of the small-batch scan (FETCH_SIZE, own pass) against its algorithmic bytes;
# (2) matrix-core busy / LDS wait of the fp8 tile kernel, MX-scaled build (0) against the bf16-widening build (4).
out=gpurun_out/pmc_r04_final
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
: > $out/summary.txt
rm -rf /tmp/pmc_sq
timeout 200 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/pmc_sq -o p -- python scripts/bench_restricted_scan.py 400000 1536 32 > /tmp/pmc_sq.log 2>&1
f=$(find /tmp/pmc_sq -name '*counter_collection.csv' | head -1)
python - "$f" >> $out/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "flat_smallq_kernel" in n or "flat_smallb_kernel" in n:
        acc[n.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kern, cs in acc.items():
    fs = cs.get("FETCH_SIZE", [])
    if fs:
        big = max(fs)
        print(f"{kern}: largest launch (100 000 candidates x 1536 f32 = 614.4 MB algorithmic) FETCH_SIZE {big:.0f} KiB x 1024 x 2 = {big * 2048 / 1e6:.1f} MB "
              f"({big * 2048 / 614.4e6:.3f} x); TCC_EA0_RDREQ x 128 B = {max(cs.get('TCC_EA0_RDREQ_sum', [0])) * 128 / 1e6:.1f} MB")
PY
for tb in 0 4; do
  rm -rf /tmp/pmc_mx_$tb
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc_mx_$tb -o p -- python scripts/bench_flat.py --rows 2000000 --dim 1536 --batch 4096 --dtype fp8 --steps 1 --tile-builds $tb > /tmp/pmc_mx_$tb.log 2>&1
  f=$(find /tmp/pmc_mx_$tb -name '*counter_collection.csv' | head -1)
  python - "$f" $tb >> $out/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "flat_tile2" in n:
        acc[n.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kern, cs in acc.items():
    i = max(range(len(cs["GRBM_GUI_ACTIVE"])), key=lambda j: cs["GRBM_GUI_ACTIVE"][j])  # the largest launch
    g = {k: v[i] for k, v in cs.items() if len(v) > i}
    print(f"tile build {sys.argv[2]}: {kern}, largest launch: " + ", ".join(f"{k} {v:.4g}" for k, v in sorted(g.items())))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "SQ_BUSY_CYCLES" in g:
        print(f"    MFMA busy / SQ busy = {g['SQ_VALU_MFMA_BUSY_CYCLES'] / g['SQ_BUSY_CYCLES']:.2f}; wait-LDS / wave cycles = {g.get('SQ_WAIT_INST_LDS', 0) / g['SQ_WAVE_CYCLES']:.3f}; "
              f"wait-any / wave cycles = {g['SQ_WAIT_INST_ANY'] / g['SQ_WAVE_CYCLES']:.3f}; active / wave cycles = {g['SQ_ACTIVE_INST_ANY'] / g['SQ_WAVE_CYCLES']:.3f}")
PY
done
cat $out/summary.txt
