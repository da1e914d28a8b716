mkdir -p gpurun_out/r05j
(time python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "non_strict or uncached or batcher or float_kernels") > gpurun_out/r05j/pytest.log 2>&1; tail -3 gpurun_out/r05j/pytest.log
AB_QUICK=1 AB_ONLY="l2 cosine default" python scripts/ab_ad_lanes.py 24 > gpurun_out/r05j/ab_spec.log 2>&1
HVX_LIB_PATH=$PWD/helix-db_amd/libhelix_vec_gfx950_tuning.so AB_QUICK=1 AB_ONLY="l2 cosine default" python scripts/ab_ad_lanes.py 24 > gpurun_out/r05j/ab_nospec.log 2>&1
echo spec; grep "^{" gpurun_out/r05j/ab_spec.log | cut -c1-210; echo nospec; grep "^{" gpurun_out/r05j/ab_nospec.log | cut -c1-210
python scripts/bench_batcher.py --no-direct --modes default --lanes 4 --per-thread 300 > gpurun_out/r05j/batcher_default_occ1.log 2>&1; tail -2 gpurun_out/r05j/batcher_default_occ1.log | cut -c1-400
