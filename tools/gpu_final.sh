# full measurement pass: GPU tests, bench variants, rocprof kernel stats, PMC traffic.  usage: bash tools/gpu_final.sh <tag>
TAG=${1:-r01_final}
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_${TAG}_f16.json 2> gpurun_out/bench_f16.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/bench_${TAG}_f16.json
timeout 900 python bench.py --steps 3 --warmup 1 --dtype bf16 --no-cpu-baseline > gpurun_out/bench_${TAG}_bf16.json 2>/dev/null; cut -c1-200 gpurun_out/bench_${TAG}_bf16.json
timeout 900 python bench.py --steps 2 --warmup 1 --batch 16 --no-cpu-baseline > gpurun_out/bench_${TAG}_f16_batch16.json 2>/dev/null; cut -c1-200 gpurun_out/bench_${TAG}_f16_batch16.json
timeout 900 python bench.py --steps 2 --warmup 1 --fixed-steps 0 --no-cpu-baseline > gpurun_out/bench_${TAG}_f16_modeN.json 2>/dev/null; cut -c1-200 gpurun_out/bench_${TAG}_f16_modeN.json
bash tools/gpu_prof.sh ${TAG} > gpurun_out/prof_${TAG}.txt 2>&1; tail -3 gpurun_out/prof_${TAG}.txt | cut -c1-200
bash tools/gpu_pmc.sh > gpurun_out/pmc_${TAG}.txt 2>&1; tail -20 gpurun_out/pmc_${TAG}.txt | cut -c1-200
