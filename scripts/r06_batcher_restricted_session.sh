#!/bin/bash
# round 6: the batching operator for the prefiltered branch -- 256 concurrent callers x 1 000-id candidate sets
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06e_batcher_restricted.log
: > $OUT
export BATCHER_RESTRICTED=1000
for nb in "" "8x128" "4x256"; do
  if [ -n "$nb" ]; then export BATCHER_NB=$nb; else unset BATCHER_NB; fi
  echo "=== BATCHER_RESTRICTED=1000 BATCHER_NB=$nb (256 callers)" >> $OUT
  timeout 300 python scripts/bench_batcher.py --rows 300000 --dim 768 --threads 256 --per-thread 400 --modes strict --lanes 2,4 --no-direct --dir /tmp/hvx_batcher_r >> $OUT 2>&1
done
unset BATCHER_NB
echo "=== one call per query (no batching), 64 callers" >> $OUT
timeout 200 scripts/_bin/bench_batcher /tmp/hvx_batcher_r 64 100 strict 2 >> $OUT 2>&1
grep -v "^/opt/amdgpu" $OUT | tail -40
