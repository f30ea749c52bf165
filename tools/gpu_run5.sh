mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
for i in 1 2; do timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['phase_ms'], d['phase_roofline']['decode_step_ms'], d['roofline']['achieved'])"; done | tee gpurun_out/bench2.log
