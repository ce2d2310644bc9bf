set -o pipefail
mkdir -p gpurun_out/r5z
for gc in f32 bf16; do
  for oop in 0 1; do
    SEI_EXCHANGE_OUT_OF_PLACE=$oop SEI_FORCE_EXCHANGE=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600+oop)) python bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-profile-gemms --grad-comm $gc --grad-comm-mode rs_ag > gpurun_out/r5z/dist1_${gc}_${oop}.log 2>&1
    echo "$gc out-of-place=$oop: $(grep -o '"value": [0-9.]*, "unit": "images/s"[^}]*"ms_per_step": [0-9.]*' gpurun_out/r5z/dist1_${gc}_${oop}.log | tail -1)"
  done
done
