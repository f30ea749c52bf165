mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
timeout 900 python -m pytest tests -q -m gpu --durations=10 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench1.log 2> gpurun_out/bench1.err; echo "bench rc=$?"
tail -5 gpurun_out/bench1.err; cat gpurun_out/bench1.log
