mkdir -p gpurun_out/r05k
(time python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16 or pair or wide_beams or two_queries") > gpurun_out/r05k/pytest.log 2>&1; tail -3 gpurun_out/r05k/pytest.log
AB_ROWS=400000 AB_DIM=1536 AB_DTYPE=bf16 AB_ONLY="cosine strict default" python scripts/ab_ad_lanes.py 24 > gpurun_out/r05k/ab_new.log 2>&1
HVX_LIB_PATH=$PWD/helix-db_amd/libhelix_vec_gfx950_tuning.so AB_ROWS=400000 AB_DIM=1536 AB_DTYPE=bf16 AB_ONLY="cosine strict" python scripts/ab_ad_lanes.py 24 > gpurun_out/r05k/ab_old.log 2>&1
echo new; grep "^{" gpurun_out/r05k/ab_new.log | cut -c1-210; echo old; grep "^{" gpurun_out/r05k/ab_old.log | cut -c1-210; tail -3 gpurun_out/r05k/ab_new.log | cut -c1-300
