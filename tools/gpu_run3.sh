mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_f16.log 2> gpurun_out/bench_f16.err; echo "bench rc=$?"; cat gpurun_out/bench_f16.log
timeout 900 python bench.py --steps 3 --warmup 1 --dtype bf16 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>/dev/null; cat gpurun_out/bench_bf16.log | cut -c1-400
timeout 900 python bench.py --steps 2 --warmup 1 --batch 16 --no-cpu-baseline > gpurun_out/bench_b16.log 2>/dev/null; cat gpurun_out/bench_b16.log | cut -c1-400
timeout 900 python bench.py --steps 2 --warmup 1 --fixed-steps 0 --no-cpu-baseline > gpurun_out/bench_modeN.log 2>/dev/null; cat gpurun_out/bench_modeN.log | cut -c1-600
bash tools/gpu_prof.sh r01h > gpurun_out/prof_r01h.txt 2>&1; head -24 gpurun_out/prof_r01h.txt | cut -c1-170
