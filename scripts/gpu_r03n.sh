#!/bin/bash
# round-3 GPU call N: workgroup-per-link build kernel: parity tests, then the 1M x 768 build A/B (link_mode 0 = new, 1 = old)
out=gpurun_out/r03n
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 400 python -m pytest tests/test_gpu_build.py -x -q > $out/pytest_build.log 2>&1; echo "pytest rc=$? $(tail -1 $out/pytest_build.log)"
for m in 0 1; do timeout 200 python scripts/bench_build.py 1000000 2048 32 embedding $m > $out/build_mode$m.log 2>&1; echo "mode $m rc=$? $(grep '^{' $out/build_mode$m.log | tail -1)"; done
