#!/bin/bash
# round-3 GPU call M: tile kernel A/B (default vs 128 x 128-per-wavefront builds, plain and with interleaved copies), MFMA issue rates,
# headline + config #4 after the lane-order change.  Everything under timeouts; outputs in gpurun_out/r03m/.
out=gpurun_out/r03m
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export HVX_LIB_PATH=$GRAFT_REPO_ROOT/helix-db_amd/libhelix_vec_gfx950_tuning.so
timeout 60 scripts/_bin/mfma_rate_probe > $out/mfma_rate.log 2>&1; echo "probe rc=$?"; cat $out/mfma_rate.log
timeout 240 rocprofv3 --kernel-trace --stats -d $out/prof_bf16 -o t -- python scripts/tile_ab.py --dtype bf16 --builds 0,4,6 > $out/ab_bf16.log 2>&1; echo "bf16 rc=$?"; grep '^{' $out/ab_bf16.log
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_fp8 -o t -- python scripts/tile_ab.py --dtype fp8 --rows 2000000 --dim 1536 --batch 4096 --steps 2 --builds 0,5,7 > $out/ab_fp8.log 2>&1; echo "fp8 rc=$?"; grep '^{' $out/ab_fp8.log
for d in bf16 fp8; do f=$(find $out/prof_$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep -i "tile" $f | cut -c1-200; done
unset HVX_LIB_PATH
timeout 400 python bench.py --steps 20 --warmup 5 --skip production,datasets,iso_recall,config3,config5,graph_equivalence --cpu-seconds 0 > $out/bench_c4.json 2> $out/bench_c4.err; echo "bench rc=$?"
python - <<'P'
import json
try:
    d=json.loads(open('gpurun_out/r03m/bench_c4.json').read().strip().splitlines()[-1])
    print('headline', d['value'], d['ms_per_step'], d['roofline'].get('frac'))
    c=d.get('config4_bf16',{}); print('c4', {k:c.get(k) for k in ('qps','ms_per_step','recall_at_10')}, c.get('roofline'))
except Exception as e: print('parse failed', e)
P
