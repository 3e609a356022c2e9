# A/B of the bench chain with and without torch + RCCL alive in the process (hardware-queue sharing), one box
nproc
run() { "$@" > gpurun_out/ab.log 2>&1; grep -o '"value": [0-9.]*\|"chain_ms": [0-9.]*' gpurun_out/ab.log | tr '\n' ' '; grep "host timing" gpurun_out/ab.log | cut -c1-260; echo; }
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3000 --warmup 200 --cpu-steps 0 --profile-steps 0"
export ICP_HOST_TIMING=1
echo "== torchrun default"; run $T
echo "== plain"; run python bench.py --cpu-steps 0 --profile-steps 0
echo "== torchrun GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 run $T
echo "== torchrun default"; run $T
echo "== plain"; run python bench.py --cpu-steps 0 --profile-steps 0
