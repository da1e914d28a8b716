#!/bin/bash
# usage (on the GPU box): scripts/gpu_tile_round.sh <tag> [tests]
# exact-scan contraction kernels: tests, A/B of the 128 x 128 vs 256 x 256 (PIPE on / off) kernels on the config #5 / #4 /
# f32 shapes, rocprofv3 kernel stats of the tile build -> gpurun_out/tile_<tag>/
tag=$1
out=gpurun_out/tile_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
if [ "$2" = "tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
  tail -5 $out/pytest.log
elif [ "$2" = "tiletests" ]; then
  timeout 600 python -m pytest tests -m gpu -x -q -k "tile or exact_scan or restricted" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
  tail -5 $out/pytest.log
fi
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env HVX_FLAT_DEBUG=1 "${envs[@]}" timeout 600 python scripts/bench_flat.py "$@" > $out/$name.log 2>&1
  echo "== $name: $(grep '^{' $out/$name.log | tail -1 | cut -c1-400)"; grep "hvx flat" $out/$name.log | tail -2
}
FP8="--rows 2000000 --dim 1536 --batch 4096 --dtype fp8 --steps 3"
B16="--rows 1000000 --dim 768 --batch 1024 --dtype bf16 --steps 5"
F32="--rows 1000000 --dim 768 --batch 1024 --dtype f32 --steps 5"
if [ -z "$ONLY_PROFILE" ]; then
run fp8_default -- $FP8
run fp8_chunk16k HVX_FLAT_CHUNK=16384 -- $FP8
run bf16_default -- $B16
run bf16_chunk32k HVX_FLAT_CHUNK=32768 -- $B16
run bf16_chunk16k HVX_FLAT_CHUNK=16384 -- $B16
run bf16_chunk8k HVX_FLAT_CHUNK=8192 -- $B16
run f32_chunk16k HVX_FLAT_CHUNK=16384 -- $F32
fi
for leg in fp8 bf16 f32; do
  case $leg in fp8) A="$FP8";; bf16) A="$B16";; f32) A="$F32";; esac
  rm -rf /tmp/prof_$leg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$leg -o $leg -- python scripts/bench_flat.py $A > /tmp/prof_$leg.log 2>&1
  (head -1 /tmp/prof_$leg/${leg}_kernel_stats.csv; grep "hvx::" /tmp/prof_$leg/${leg}_kernel_stats.csv) > $out/kernel_stats_$leg.csv
  echo "== kernel stats $leg"; cut -d, -f1-4 $out/kernel_stats_$leg.csv | cut -c1-160 | head -12
done
