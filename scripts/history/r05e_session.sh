mkdir -p gpurun_out/r05e
HVX_LIB_PATH=$PWD/helix-db_amd/libhelix_vec_gfx950_tuning.so HVX_WAVE_PROF=1 AB_QUICK=1 AB_ONLY="l2 cosine default strict" python scripts/ab_ad_lanes.py 6 > gpurun_out/r05e/prof.log 2>&1
grep "hvx prof" gpurun_out/r05e/prof.log | sort | uniq -c | sort -rn | head -20
python scripts/bench_batcher.py --no-direct --modes strict --lanes 3 --per-thread 300 > gpurun_out/r05e/batcher_blocking.log 2>&1
E=scripts/_bin/bench_batcher; D=/tmp/hvx_batcher
for nb in 8x128 4x256 16x64 8x256; do BATCHER_WAIT=200 BATCHER_NB=$nb $E $D 1024 300 strict 3 nodirect > gpurun_out/r05e/nb_$nb.log 2>&1; done
BATCHER_WAIT=200 BATCHER_OCC=2 $E $D 1024 300 default 4 nodirect > gpurun_out/r05e/default_occ2_l4.log 2>&1
BATCHER_WAIT=200 BATCHER_OCC=1 $E $D 1024 300 default 4 nodirect > gpurun_out/r05e/default_occ1_l4.log 2>&1
BATCHER_WAIT=200 BATCHER_OCC=2 $E $D 2048 200 default 4 nodirect > gpurun_out/r05e/default_occ2_l4_2048.log 2>&1
BATCHER_WAIT=200 BATCHER_OCC=2 BATCHER_NB=8x256 $E $D 1024 300 default 4 nodirect > gpurun_out/r05e/default_nb_8x256.log 2>&1
tail -n 4 gpurun_out/r05e/batcher_blocking.log gpurun_out/r05e/nb_*.log gpurun_out/r05e/default_*.log | cut -c1-420
