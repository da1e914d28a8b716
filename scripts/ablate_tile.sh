#!/bin/bash
# usage (on the GPU box): scripts/ablate_tile.sh <tag> <bench_flat args...>
# kernel time of the two-buffer 256 x 256 build with parts switched off (HVX_FLAT_TILE_ABLATE: 1 no operand copies, 2 no MFMAs,
# 4 no epilogue) -> which phase bounds the tile.  Results of ablated runs are wrong by construction; only the timing is read.
tag=$1; shift
out=gpurun_out/ablate_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for ab in 0 4 5 6 7; do
  rm -rf /tmp/ab_$ab
  HVX_FLAT_TILE_BUILD=0 HVX_FLAT_TILE_ABLATE=$ab timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$ab -o ab -- python scripts/bench_flat.py "$@" > /tmp/ab_$ab.log 2>&1
  echo "ablate=$ab: $(grep flat_tile256_kernel /tmp/ab_$ab/ab_kernel_stats.csv | cut -d, -f2-7 | tr -d '"')"
  grep flat_tile256_kernel /tmp/ab_$ab/ab_kernel_stats.csv > $out/ablate_$ab.csv
done
