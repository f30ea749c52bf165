mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench2.log 2> gpurun_out/bench2.err; echo "bench rc=$?"
tail -3 gpurun_out/bench2.err; cat gpurun_out/bench2.log
