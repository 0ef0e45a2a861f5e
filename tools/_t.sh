set -e
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 300 --warmup 50 --no-cpu-baseline | tail -1 | cut -c1-400
echo ---- 2 ranks, gloo, one device
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 300 --warmup 50 --backend gloo --single-device | tail -1 | cut -c1-600
