export SS_BENCH_DEVICE=0 OMP_WAIT_POLICY=passive
for n in 2 3; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 3 --warmup 1 --dist-backend gloo --no-cpu-baseline 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ranks sharing one GPU:', d['n_gpus'], 'xRT', d['value'], 'ms/step', d['ms_per_step'], d['phase_ms'])"
done
