#!/bin/bash
# usage (on the GPU box): scripts/gpu_tile4_round.sh <tag>
# First thing for the opt-in 128 x 128-per-wavefront exact-scan kernel (HVX_FLAT_TILE_BUILD=4: bf16 / f32 shadow, 5: fp8 too):
# parity tests under a short timeout (a wrong barrier / wait shows as a hang), then the A/B against the default build.
tag=$1
out=gpurun_out/tile4_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for b in 4 5; do
  HVX_FLAT_TILE_BUILD=$b timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -k "through_the_256_tile_kernel or restricted_scan_through" > $out/pytest_build$b.log 2>&1
  echo "build $b parity: rc=$? $(tail -1 $out/pytest_build$b.log)"
done
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env HVX_FLAT_DEBUG=1 "${envs[@]}" timeout 300 python scripts/bench_flat.py "$@" > $out/$name.log 2>&1
  echo "== $name: $(grep '^{' $out/$name.log | tail -1 | cut -c1-220)"; }
FP8="--rows 2000000 --dim 1536 --batch 4096 --dtype fp8 --steps 3"
B16="--rows 1000000 --dim 768 --batch 1024 --dtype bf16 --steps 5"
F32="--rows 1000000 --dim 768 --batch 1024 --dtype f32 --steps 5"
run bf16_default -- $B16
run bf16_build4 HVX_FLAT_TILE_BUILD=4 -- $B16
run f32_build4 HVX_FLAT_TILE_BUILD=4 -- $F32
run fp8_default -- $FP8
run fp8_build5 HVX_FLAT_TILE_BUILD=5 -- $FP8
