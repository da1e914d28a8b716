#!/bin/bash
# usage (on the GPU box): scripts/pmc_tile_r04.sh  -> gpurun_out/pmc_tile_r04/summary.txt
# rocprofv3 --pmc passes (own runs, --kernel-trace only) of the exact-scan tile kernels, build 0 (two 256-thread workgroups per CU) and
# build 2 (512 threads, role-split): matrix-core busy, wait / issue split, LDS conflicts -- 1024 x 1M x 768 bf16.
out=gpurun_out/pmc_tile_r04
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
: > $out/summary.txt
for tb in 0 2; do
  for grp in "sq GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM"; do
    set -- $grp; name=$1; shift
    rm -rf /tmp/pmct_${tb}_$name
    timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmct_${tb}_$name -o p -- python scripts/bench_flat.py --rows 1000000 --dim 768 --batch 1024 --dtype bf16 --steps 2 --tile-builds $tb > /tmp/pmct_${tb}_$name.log 2>&1
    f=$(find /tmp/pmct_${tb}_$name -name '*counter_collection.csv' | head -1)
    python - "$f" $tb $name >> $out/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "flat_tile" in n:
        acc[n.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kern, cs in acc.items():
    print(f"build {sys.argv[2]} [{sys.argv[3]}] {kern}")
    for k, v in cs.items():
        print(f"    {k}: dispatches={len(v)} mean={sum(v)/len(v):.5g} max={max(v):.5g}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs:
        import statistics
        ratios = [a / b for a, b in zip(cs["SQ_VALU_MFMA_BUSY_CYCLES"], cs["SQ_BUSY_CYCLES"]) if b]
        print(f"    => MFMA busy / SQ busy: mean {sum(ratios)/len(ratios):.3f}, largest launch {cs['SQ_VALU_MFMA_BUSY_CYCLES'][-1] / cs['SQ_BUSY_CYCLES'][-1]:.3f}")
PY
  done
done
cat $out/summary.txt
