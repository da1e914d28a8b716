#!/bin/bash
# round 6: (1) the headline leg at dim 384 and 3072 (shapes outside the unrolled wave-kernel builds: VERDICT r5 missing #5),
#          (2) the N = 8 plumbing on ONE GPU (gloo transport, host staging): headline leg + the config-5 leg
cd "$(dirname "$0")/.."
SK=production,production_lanes,batcher,insert,datasets,iso_recall,latent,config3,config4,config5,graph_equivalence,vendor_gemm
for spec in "384 1000000" "3072 250000" "2048 400000"; do
  set -- $spec
  timeout 400 python bench.py --dim $1 --rows $2 --skip $SK --full-record /tmp/full_$1.json > gpurun_out/r06h_dim$1.log 2> gpurun_out/r06h_dim$1.err
  tail -1 gpurun_out/r06h_dim$1.log > gpurun_out/r06h_dim$1.json
done
export HVX_BENCH_SHARED_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 6 --warmup 2 \
    --rows 100000 --skip $SK,ef_sweep,peak --cpu-seconds 0 --full-record /tmp/full_8.json > gpurun_out/r06h_eight_rank_shared_gpu.log 2> gpurun_out/r06h_eight_rank_shared_gpu.err
tail -1 gpurun_out/r06h_eight_rank_shared_gpu.log > gpurun_out/r06h_eight_rank_shared_gpu.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 8 --steps 3 --warmup 1 \
    --leg config5 --c5-rows 200000 --full-record /tmp/full_8c5.json > gpurun_out/r06h_eight_rank_shared_gpu_config5.log 2> gpurun_out/r06h_eight_rank_shared_gpu_config5.err
tail -1 gpurun_out/r06h_eight_rank_shared_gpu_config5.log > gpurun_out/r06h_eight_rank_shared_gpu_config5.json
for f in gpurun_out/r06h_*.json; do echo "== $f"; cut -c1-900 $f; done
tail -3 gpurun_out/r06h_eight_rank_shared_gpu.err gpurun_out/r06h_eight_rank_shared_gpu_config5.err | cut -c1-300
