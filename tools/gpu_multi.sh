# two ranks sharing GPU 0 (gloo): exercises the real multi-process path of bench.py end to end on a 1-GPU box
export SS_BENCH_DEVICE=0 OMP_WAIT_POLICY=passive
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --model base.en --batch 4 --dist-backend gloo --no-cpu-baseline 2>&1 | grep -E "^\{|Error|error" | cut -c1-900
timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --model base.en --batch 4 --no-cpu-baseline 2>/dev/null | cut -c1-500
