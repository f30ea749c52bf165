# round 2, run G: GPU suite (continuous batching, quantised files, whisper.h surface), default bench, PMC traffic of the decoder pass.   usage: bash tools/gpu_r02_g.sh <tag>
TAG=${1:-r02_g}
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
rm -f gpurun_out/parity_report.txt
( time timeout 1800 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -25 gpurun_out/pytest_gpu_$TAG.log
timeout 900 python bench.py > gpurun_out/bench_${TAG}_default.json 2> gpurun_out/bench_${TAG}_default.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/bench_${TAG}_default.json
bash tools/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.txt 2>&1; tail -32 gpurun_out/pmc_$TAG.txt | cut -c1-220
